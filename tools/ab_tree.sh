#!/bin/bash
# Same-box A/B of two TREES (library + host code): alternates  python <other tree>/bench.py  and  python bench.py, 3 rounds.
#   tools/ab_tree.sh <path of the other tree, relative to the repo root> [bench.py args...]
# The other tree is made on the build machine (e.g. git archive <rev> | tar -x -C deep-mixture-vae_amd/build/r3tree, then its build.py): build/ is
# git-ignored and travels with gpurun.
OTHER=${1:?path of the other tree}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
for r in 1 2 3; do
  for w in other this; do
    if [ $w = other ]; then T="$ROOT/$OTHER"; else T="$ROOT"; fi
    (cd "$T" && python3 bench.py "$@" --no-cpu-baseline --elbo-epochs 0 > "$ROOT/gpurun_out/abtree_${w}_$r.json") || exit 1
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('%-6s %.4f ms/step  %.0f %s' % (sys.argv[2], d['ms_per_step'], d['value'], d['unit']), flush=True)" "$ROOT/gpurun_out/abtree_${w}_$r.json" $w
  done
done
