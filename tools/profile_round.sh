#!/bin/bash
# Round evidence, run on the GPU box:  tools/profile_round.sh r02 [cfgN]
#   1. PMC pass      -> HBM bytes per launch per kernel (tools/pmc_traffic.sh) -> profiles/traffic.json  (metric config only)
#   2. bench.py      -> the JSON line (roofline.traffic filled from 1.)
#   3. rocprofv3 --kernel-trace --stats of the same bench command -> kernel_stats.csv
# Everything lands in gpurun_out/profiles_<tag>/ ; copy that directory's files into profiles/.
set -e
TAG=${1:-r02}
CFG=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT profiles
if [ -z "$CFG" ]; then
    NAME=$TAG; ARGS="--steps 200 --warmup 20"
    TAG=$TAG tools/pmc_traffic.sh > $OUT/${TAG}_pmc_traffic_head.txt 2>&1
    cp gpurun_out/traffic.json profiles/traffic.json
    cp gpurun_out/traffic.json $OUT/traffic.json
    cp gpurun_out/pmc_traffic.txt $OUT/${TAG}_pmc_traffic.txt
else
    NAME=${TAG}_$CFG; ARGS="--config $CFG --steps 20 --warmup 3 --no-cpu-baseline"
fi
python3 bench.py $ARGS > $OUT/${NAME}_bench.json 2> $OUT/${NAME}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -o $NAME -- python3 bench.py $ARGS --no-cpu-baseline --elbo-epochs 0 > $OUT/${NAME}_bench_under_rocprof.json 2> /dev/null
cp $(ls gpurun_out/prof_$NAME/*/*kernel_stats.csv gpurun_out/prof_$NAME/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${NAME}_kernel_stats.csv
python3 tools/showbench.py $OUT/${NAME}_bench.json > $OUT/.show.txt 2>/dev/null; sed -n 1,8p $OUT/.show.txt | cut -c1-300; rm -f $OUT/.show.txt
head -14 $OUT/${NAME}_kernel_stats.csv | cut -c1-200
