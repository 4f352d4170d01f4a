#!/bin/bash
# GPU box: per-kernel event timing of the whole step with each ablated GEMM library
cd $GRAFT_REPO_ROOT
for n in 0 "$@"; do
  if [ $n = 0 ]; then unset DMVAE_HIP_LIB; else export DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl$n.so; fi
  python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('abl$n ms/step %.4f | '%d['ms_per_step'] + '  '.join('%s %.1f'%(k['kernel'].replace('gemm_bf16_','').replace('kernel',''),k['avg_us']) for k in d['kernels'] if 'gemm' in k['kernel']))"
done
