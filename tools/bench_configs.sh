#!/bin/bash
# GPU box: one bench line per BASELINE.json config that fits one GPU (cfg1 plumbing size, cfg2, cfg3, cfg4 per-GPU shard) + cfg2 in fp32
cd $GRAFT_REPO_ROOT
out=gpurun_out/configs.jsonl; : > $out
run() { python3 bench.py "$@" --no-cpu-baseline --profile-steps 3 2>/dev/null >> $out; }
run --batch 100 --latent_dim 10 --n_clusters 10 --steps 200 --warmup 20
run --batch 4096 --latent_dim 64 --n_clusters 10 --steps 200 --warmup 20
run --batch 4096 --latent_dim 64 --n_clusters 10 --steps 50 --warmup 5 --dtype fp32
run --batch 16384 --latent_dim 128 --n_clusters 10 --steps 100 --warmup 10
run --batch 8192 --latent_dim 256 --n_clusters 50 --steps 100 --warmup 10
python3 - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); r = d["roofline"]; c = d["config"]
    print("B=%-6d %-4s %12.0f img/s  %8.4f ms/step  mfma %.3f | dominant %s %.1f us %s frac %.3f" % (c["per_gpu_batch"], d["dtype"], d["value"], d["ms_per_step"], d["step_mfma_frac_of_peak"], r["kernel"], r["avg_launch_us"], r["bound"], r["frac"]))
PY
