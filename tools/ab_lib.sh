#!/bin/bash
# Same-box A/B of two library builds (box-to-box spread is ~1-2 %, more than most kernel changes):
#   tools/ab_lib.sh <variant> [bench.py args...]      e.g.  tools/ab_lib.sh old --config cfg5 --steps 20 --warmup 3
# alternates  DMVAE_HIP_LIB=build/variants/<variant>.so  and the in-tree library, 3 rounds, prints ms/step of every run.
VAR=${1:?variant name}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
for r in 1 2 3; do
  for w in variant tree; do
    if [ $w = variant ]; then export DMVAE_HIP_LIB="$ROOT/deep-mixture-vae_amd/build/variants/$VAR.so"; else unset DMVAE_HIP_LIB; fi
    python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --elbo-epochs 0 > "$ROOT/gpurun_out/ab_${w}_$r.json" || exit 1
    python3 - "$ROOT/gpurun_out/ab_${w}_$r.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ks = {k["kernel"]: k["ms_per_step"] for k in d.get("kernels", [])}
top = sorted(ks.items(), key=lambda kv: -kv[1])[:4]
print("%-8s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], "  ".join("%s %.3f" % (k[:34], v) for k, v in top)), flush=True)
PY
  done
done
