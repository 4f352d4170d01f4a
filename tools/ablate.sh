#!/bin/bash
# Builds measurement variants of the kernels (DMVAE_ABLATE=1..11: csrc/measure.h lists them) into
# deep-mixture-vae_amd/build/libdmvae_hip_abl<N>.so (git-ignored; select with DMVAE_HIP_LIB).
# Results of an ablated library are WRONG by construction: timing only.
set -e
cd "$(dirname "$0")/../deep-mixture-vae_amd"
python3 build.py > /dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ../include -DDMVAE_ABLATE=$n -c csrc/gemm_bf16.hip -o build/gemm_bf16_abl$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I ../include -DDMVAE_ABLATE=$n -c csrc/latent.hip -o build/latent_abl$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ../include -DDMVAE_ABLATE=$n -c csrc/gemm_bf16_256.hip -o build/gemm_bf16_256_abl$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ../include -DDMVAE_ABLATE=$n -c csrc/heads_dx.hip -o build/heads_dx_abl$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ../include -DDMVAE_ABLATE=$n -c csrc/heads_latent.hip -o build/heads_latent_abl$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ../include -DDMVAE_ABLATE=$n -c csrc/api.hip -o build/api_abl$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libdmvae_hip_abl$n.so build/gemm_bf16_abl$n.o build/latent_abl$n.o \
      build/gemm_bf16_256_abl$n.o build/gemm_f32.o build/latent_mfma.o build/latent_vade.o build/elementwise.o build/conv.o build/heads_dx_abl$n.o build/heads_latent_abl$n.o build/strip_fwd2.o build/api_abl$n.o
  echo build/libdmvae_hip_abl$n.so
done
