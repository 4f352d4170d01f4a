#!/bin/bash
# A second build of the library from another revision, for same-box A/B runs:
#   tools/build_variant.sh <git-rev> <name>   ->  deep-mixture-vae_amd/build/variants/<name>.so   (travels with gpurun; git-ignored)
# then on the GPU box:  tools/ab_lib.sh <name> [bench.py args]
set -e
REV=${1:?git revision}; NAME=${2:?variant name}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/dmvae_variant.XXXXXX)
git -C "$ROOT" archive "$REV" deep-mixture-vae_amd/csrc deep-mixture-vae_amd/build.py include | tar -x -C "$TMP"
mkdir -p "$ROOT/deep-mixture-vae_amd/build/variants" "$TMP/deep-mixture-vae_amd/dmvae_hip"
(cd "$TMP/deep-mixture-vae_amd" && python3 -c "import build; build.build(verbose=False)")
cp "$TMP/deep-mixture-vae_amd/dmvae_hip/libdmvae_hip.so" "$ROOT/deep-mixture-vae_amd/build/variants/$NAME.so"
rm -rf "$TMP"
echo "built deep-mixture-vae_amd/build/variants/$NAME.so from $REV"
