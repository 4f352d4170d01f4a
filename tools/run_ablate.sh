#!/bin/bash
# GPU box: time product + ablated libraries (build with tools/ablate.sh N... first): run_ablate.sh "0 0" 4 5
cd $GRAFT_REPO_ROOT
t="$1"; shift
python3 tools/gemm_ablate.py $t 2>&1 | grep -v amdgpu.ids
for n in "$@"; do
  DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl$n.so python3 tools/gemm_ablate.py $t 2>&1 | grep -v amdgpu.ids
done
