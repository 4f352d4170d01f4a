#!/bin/bash
# GPU box: per-kernel event timings (eager profile pass of bench.py) for two library builds:  tools/ab_kernels.sh base new
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_$n.so python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 10 2>/dev/null \
    | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$n ms/step %.4f | ' % d['ms_per_step'] + '  '.join('%s x%.0f %.1f' % (k['kernel'].replace('gemm_bf16_','').replace('kernel',''), k['launches_per_step'], k['avg_us']) for k in d['kernels']))"
done
