#!/bin/bash
# Round evidence, run on the GPU box:  tools/profile_round.sh r01
#   1. PMC pass      -> HBM bytes per launch per kernel (tools/pmc_traffic.sh) -> profiles/traffic.json
#   2. bench.py      -> the JSON line (roofline.traffic filled from 1.)
#   3. rocprofv3 --kernel-trace --stats of the same bench command -> kernel_stats.csv
# Everything lands in gpurun_out/profiles_<tag>/ ; copy that directory's files into profiles/.
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT profiles
tools/pmc_traffic.sh > $OUT/${TAG}_pmc_traffic_head.txt 2>&1
cp gpurun_out/traffic.json profiles/traffic.json
cp gpurun_out/traffic.json $OUT/traffic.json
cp gpurun_out/r01_pmc_traffic.txt $OUT/${TAG}_pmc_traffic.txt
python3 bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp $(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv gpurun_out/prof_$TAG/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
python3 tools/showbench.py $OUT/${TAG}_bench.json > $OUT/.show.txt 2>/dev/null; sed -n 1,8p $OUT/.show.txt; rm -f $OUT/.show.txt
head -12 $OUT/${TAG}_kernel_stats.csv
