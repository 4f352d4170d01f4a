#!/bin/bash
# bench.py --gpus 2 over gloo on ONE GPU box (weak and strong form): a rehearsal of the multi-rank orchestration -- rank spawn, settle loop,
# timed regions, per-rank min / max, exchange diagnostics, ONE JSON line on stdout.  Not a scaling number (both ranks share the card).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 3 --no-cpu-baseline --elbo-epochs 0 > gpurun_out/dp_gloo2_weak.json 2> gpurun_out/dp_gloo2_weak.err; echo "weak rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --config cfg4-strong --steps 5 --warmup 2 --no-cpu-baseline --elbo-epochs 0 --rows 65536 > gpurun_out/dp_gloo2_strong.json 2> gpurun_out/dp_gloo2_strong.err; echo "strong rc=$?"
python3 - <<'PY'
import json
for f in ("weak", "strong"):
    t = open("gpurun_out/dp_gloo2_%s.json" % f).read().strip().splitlines()
    print(f, "lines on stdout:", len(t))
    if not t:
        continue
    d = json.loads(t[-1])
    print("  ", d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], d["config"]["collective_backend"], "|", d["config"]["update"][:90])
    print("   settle", d.get("settle_ms_per_step"), "| rank min/max", d.get("rank_ms_per_step_min_max"), "| exchange", {k: v for k, v in (d.get("exchange") or {}).items() if k != "what"})
PY
grep -h "Error" gpurun_out/dp_gloo2_*.err | grep -v "^\[W" | head -4
exit 0
