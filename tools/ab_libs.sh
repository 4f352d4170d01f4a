#!/bin/bash
# GPU box: alternate bench runs (graph replay, no profiling) between library builds under
# deep-mixture-vae_amd/build/libdmvae_hip_<name>.so :   tools/ab_libs.sh base new [rounds]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for n in $A $B; do
    DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_$n.so python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --profile-steps 0 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'])"
  done
done
