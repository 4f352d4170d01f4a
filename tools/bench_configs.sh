#!/bin/bash
# GPU box: one bench line per BASELINE.json config that fits one GPU (cfg1 plumbing size, cfg2 = the metric, cfg3, cfg4 and cfg5
# per-GPU shards) + cfg2 in fp32 parity mode -> gpurun_out/configs.jsonl
cd $GRAFT_REPO_ROOT
out=gpurun_out/configs.jsonl; : > $out
run() { python3 bench.py "$@" --no-cpu-baseline --elbo-epochs 0 --profile-steps 3 2>/dev/null >> $out; }
run --config cfg1 --steps 200 --warmup 20
run --config cfg2 --steps 200 --warmup 20
run --config cfg2 --steps 50 --warmup 5 --dtype fp32
run --config cfg3 --steps 100 --warmup 10
run --config cfg4 --steps 100 --warmup 10
run --config cfg5 --steps 20 --warmup 3
python3 - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); r = d["roofline"]; c = d["config"]
    print("B=%-6d %-4s %12.0f img/s  %8.4f ms/step  mfma %.3f  kernels/step %5.1f | dominant %s %.1f us %s frac %.3f" % (c["per_gpu_batch"], d["dtype"], d["value"], d["ms_per_step"], d["step_mfma_frac_of_peak"], d.get("kernels_per_step", 0), r["kernel"], r["avg_launch_us"], r["bound"], r["frac"]))
PY
