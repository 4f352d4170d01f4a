#!/bin/bash
# A second build of the library from the WORKING TREE with extra compile flags on some sources, for same-box A/B runs of compile-time choices:
#   tools/build_flag_variant.sh <name> "<extra hipcc flags>" <source.hip> [<source.hip> ...]
#   -> deep-mixture-vae_amd/build/variants/<name>.so   (travels with gpurun; git-ignored; select with DMVAE_HIP_LIB or tools/ab_lib.sh <name>)
# The named sources are recompiled with the flags; every other object is the in-tree build's.
set -e
NAME=${1:?variant name}; EXTRA=${2?extra flags}; shift 2
cd "$(dirname "$0")/../deep-mixture-vae_amd"
python3 build.py > /dev/null
mkdir -p build/variants build/var_$NAME
OBJS=""
for b in $(python3 -c "import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do
  use=build/$b.o
  for s in "$@"; do if [ "$(basename $s .hip)" = "$b" ]; then use=build/var_$NAME/$b.o; fi; done
  OBJS="$OBJS $use"
done
for s in "$@"; do
  b=$(basename $s .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA -Rpass-analysis=kernel-resource-usage -c csrc/$b.hip -o build/var_$NAME/$b.o 2> build/var_$NAME/$b.log &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$NAME.so $OBJS
echo "built deep-mixture-vae_amd/build/variants/$NAME.so ($EXTRA: $*)"
