#!/bin/bash
# Round-5 evidence in GPU calls of <= 15 minutes:  tools/round5_evidence.sh <step>      (outputs: gpurun_out/profiles_r05/, copy into profiles/)
#   a: cfg2 (the metric): PMC traffic -> traffic.json, bench line, rocprofv3 --kernel-trace --stats of the same command, SQ counters;
#      the relaxed (Gumbel-Softmax) step: bench line + kernel-trace stats
#   c: cfg3 / cfg4: bench line + kernel-trace stats + PMC traffic          b: cfg5: bench line + kernel-trace stats
# Each rocprofv3 pass has the program straight after `--`; --pmc passes carry --kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r05
O=gpurun_out/profiles_$T
mkdir -p $O
case "$1" in
a)  tools/profile_round.sh $T > $O/log_a.txt 2>&1
    tools/pmc_sq.sh $T >> $O/log_a.txt 2>&1; cp gpurun_out/${T}_pmc_sq.txt $O/
    python3 bench.py --mode relaxed --steps 200 --warmup 20 --no-cpu-baseline --elbo-epochs 0 > $O/${T}_relaxed_bench.json 2> $O/${T}_relaxed_bench.err
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_relaxed -o ${T}_relaxed -- python3 bench.py --mode relaxed --steps 200 --warmup 20 --no-cpu-baseline --elbo-epochs 0 > /dev/null 2>&1
    cp $(ls gpurun_out/prof_${T}_relaxed/*/*kernel_stats.csv gpurun_out/prof_${T}_relaxed/*kernel_stats.csv 2>/dev/null | head -1) $O/${T}_relaxed_kernel_stats.csv ;;
c)  for c in cfg3 cfg4; do
      tools/profile_round.sh $T $c > $O/log_c_$c.txt 2>&1
      CFG=$c PMC_STEPS=10 tools/pmc_traffic.sh >> $O/log_c_$c.txt 2>&1; cp gpurun_out/traffic_$c.json $O/; cp gpurun_out/pmc_traffic_$c.txt $O/${T}_${c}_pmc_traffic.txt
    done ;;
b)  tools/profile_round.sh $T cfg5 > $O/log_b.txt 2>&1 ;;
esac
ls $O
