"""bench.py as the driver launches it for N > 1 -- rehearsed on ONE GPU with two gloo ranks (RCCL needs one GPU per rank; gloo lets the ranks
share the card).  It exercises what no single-process test reaches: the rank spawn, capture_step under the sharded exchange, the settle loop's
and the timed regions' collectives, the exchange diagnostics, the fp32-master sync in front of the per-kernel profiling leg, and the contract
itself: exit code 0 and exactly ONE JSON line on stdout.  (Round 4: the stale-master guard of the engine broke this path in two places -- in
capture_step and in the profiling leg -- and nothing but a rehearsal by hand noticed.)  Timings of such a run mean nothing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form", ["weak", "strong"])
def test_bench_two_ranks_over_gloo_prints_one_json_line(form):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--stable-warmup", "2",
           "--profile-steps", "2", "--no-cpu-baseline", "--elbo-epochs", "0"]
    if form == "strong":
        cmd += ["--global-batch", "1024", "--rows", "4096"]           # 512 rows per rank
    else:
        cmd += ["--batch", "512", "--rows", "4096"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DMVAE_DP_FORCE"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == form and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["unit"] == "images/sec" and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["config"]["per_gpu_batch"] == 512 and d["config"]["global_batch"] == 1024 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["collective_backend"].startswith("gloo") and d["config"]["rccl_ranks"] == 0          # a rehearsal says so
    assert "reduce-scatter" in d["config"]["update"] and "one collective after the backward pass" in d["config"]["update"]
    assert len(d["rank_ms_per_step_min_max"]) == 3 and all(a <= b for a, b in d["rank_ms_per_step_min_max"])
    ex = d["exchange"]
    assert ex["exposed_us"] > 0 and ex["step_us"] >= ex["exposed_us"] and ex["exposed_us_min_over_ranks"] <= ex["exposed_us_max_over_ranks"]
    assert ex["self_check"] == "ok"              # the replicas' checksums after the first sharded step agreed (parallel.ShardedExchange.self_check)
    assert d["roofline"]["frac"] <= 1.0 and d["kernels_per_step"] >= 10            # the profiling leg ran (behind sync_master)
