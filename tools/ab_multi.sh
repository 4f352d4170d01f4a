#!/bin/bash
# Same-box comparison of SEVERAL library builds (tools/build_variant.sh / tools/build_flag_variant.sh), processes alternating:
#   tools/ab_multi.sh <rounds> "<variant> <variant> ..." [bench.py args...]      ("tree" = the in-tree library)
# prints ms/step and the four longest kernels of every run; the order is reversed every other round.
R=${1:?rounds}; VARS=${2:?variants}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
for r in $(seq 1 $R); do
  L="$VARS"; if [ $((r % 2)) = 0 ]; then L=$(echo $VARS | tr ' ' '\n' | tac | tr '\n' ' '); fi
  for w in $L; do
    if [ $w = tree ]; then unset DMVAE_HIP_LIB; else export DMVAE_HIP_LIB="$ROOT/deep-mixture-vae_amd/build/variants/$w.so"; fi
    python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --elbo-epochs 0 > "$ROOT/gpurun_out/abm_${w}_$r.json" || exit 1
    python3 - "$ROOT/gpurun_out/abm_${w}_$r.json" $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ks = {k["kernel"]: k["avg_us"] for k in d.get("kernels", [])}
top = sorted(ks.items(), key=lambda kv: -kv[1])[:3]
print("%-12s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], "  ".join("%s %.1f us" % (k[:40], v) for k, v in top)), flush=True)
PY
  done
done
