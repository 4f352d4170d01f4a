#!/bin/bash
# Round-3 counter evidence in one GPU call:  tools/round3_evidence.sh <step>
#   step a: cfg2 bench line + rocprofv3 --kernel-trace of the same command + PMC traffic + SQ counters
#   step b: cfg5 SQ counters + PMC traffic; in-kernel clock of the macro-tile K loop (abl6 build)
#   step c: cfg3 / cfg4 PMC traffic + SQ counters
# Each rocprofv3 pass has the program straight after `--`; --pmc passes carry --kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r03
O=gpurun_out/profiles_$T
mkdir -p $O
case "$1" in
a)  tools/profile_round.sh $T > $O/log_a.txt 2>&1
    tools/pmc_sq.sh $T >> $O/log_a.txt 2>&1; cp gpurun_out/${T}_pmc_sq.txt $O/ ;;
b)  CFG=cfg5 PMC_STEPS=6 tools/pmc_sq.sh $T > $O/log_b.txt 2>&1; cp gpurun_out/${T}_cfg5_pmc_sq.txt $O/
    CFG=cfg5 PMC_STEPS=6 tools/pmc_traffic.sh >> $O/log_b.txt 2>&1; cp gpurun_out/traffic_cfg5.json $O/; cp gpurun_out/pmc_traffic_cfg5.txt $O/${T}_cfg5_pmc_traffic.txt
    DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl6.so python3 tools/clock256.py > $O/${T}_clock256.txt 2>> $O/log_b.txt
    tools/profile_round.sh $T cfg5 >> $O/log_b.txt 2>&1 ;;
c)  for c in cfg3 cfg4; do
      CFG=$c PMC_STEPS=10 tools/pmc_traffic.sh > $O/log_c_$c.txt 2>&1; cp gpurun_out/traffic_$c.json $O/; cp gpurun_out/pmc_traffic_$c.txt $O/${T}_${c}_pmc_traffic.txt
      CFG=$c PMC_STEPS=10 tools/pmc_sq.sh $T >> $O/log_c_$c.txt 2>&1; cp gpurun_out/${T}_${c}_pmc_sq.txt $O/
      tools/profile_round.sh $T $c >> $O/log_c_$c.txt 2>&1
    done ;;
esac
ls -la $O
