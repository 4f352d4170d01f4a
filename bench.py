#!/usr/bin/env python3
"""DMVAE training-step benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): images/sec (train), MNIST-shaped K=10 z_dim=64
batch=4096 per GPU, bf16 MFMA GEMMs, synthetic 28x28 inputs resident in HBM.
One "step" = batch assembly + forward + loss + backward + (gradient all-reduce)
+ Adam, i.e. one session.run([loss, train_step]) of code/base_models.py:126-129.

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline"     : the dominant kernel family, measured with HIP events around
                   every launch of a few eager steps (same process, same shapes)
  "cpu_baseline" : the oracle (CPU restatement of the reference step, float32,
                   NumPy/OpenBLAS on the box's host cores) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (cfg2: 4096)")
    ap.add_argument("--lr", type=float, default=0.002, help="Adam learning rate (train.py default 0.002; the 4096-wide cfg5 "
                    "stack overflows exp(log_var) after one step at that rate in fp32 and bf16 alike: use 1e-4 there)")
    ap.add_argument("--input_dim", type=int, default=784)
    ap.add_argument("--enc_layers", default="500,500")
    ap.add_argument("--head_dim", type=int, default=2000)
    ap.add_argument("--dec_layers", default="2000,500,500")
    ap.add_argument("--cnn", action="store_true", help="the checked-in CNN encoder trunk (base_models.py:176-216) instead of the "
                    "MLP branch the metric names; --enc_layers is then its one dense layer (500)")
    ap.add_argument("--latent_dim", type=int, default=64)
    ap.add_argument("--n_clusters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--rows", type=int, default=65536, help="synthetic dataset rows resident in HBM")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--deterministic", action="store_true", help="no float atomics (split-K off)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dp-dry-run", default="", choices=["", "overlap", "single"],
                    help="N = 1 only: issue the data-parallel launch sequence (staged backward + stand-alone Adam, or whole "
                         "backward + Adam) with a no-op exchange, to price the N > 1 path's compute")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    return ap.parse_args()


def flops_per_image(I, D, K, enc=(500, 500), head=2000, dec=(2000, 500, 500), cnn=False):
    """train FLOP/img = 2*(3M - I*enc0), M = MACs of one forward.  cnn: the six 3x3
    convolutions of base_models.py:181-201 (pixels * 9*cin * cout MACs each) in front of fc 2048 -> enc[0];
    the first layer (conv0 / enc0) has no input gradient."""
    m, prev, first = 0, I, I * enc[0]
    if cnn:
        convs = ((784, 1, 32), (784, 32, 32), (196, 32, 64), (196, 64, 64), (49, 64, 128), (49, 128, 128))
        m += sum(px * 9 * ci * co for px, ci, co in convs)
        prev, first = 2048, 784 * 9 * 32
    for h in enc:
        m += prev * h
        prev = h
    m += 2 * prev * head + 2 * head * D + head * K
    prev = D
    for h in dec:
        m += prev * h
        prev = h
    m += prev * I
    return 2 * (3 * m - first)


def cpu_baseline(args, seconds):
    """The oracle timed on the host cores: float32 NumPy restatement of the same
    step (reported baseline, not the optimisation target; the TensorFlow-1.x
    reference itself cannot run in this image)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import dmvae_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    out = {}
    for B in (args.batch, 100):
        cfg = O.Config(args.input_dim, args.latent_dim, args.n_clusters, tuple(int(v) for v in args.enc_layers.split(",")),
                       args.head_dim, tuple(int(v) for v in args.dec_layers.split(",")))
        p = O.init_params(cfg, 0, np.float32)
        m, v = O.adam_tf_init(p)
        X = O.synthetic_images(B, args.input_dim, seed=1)
        rng = np.random.RandomState(0)
        eps = rng.randn(B, args.latent_dim).astype(np.float32)
        budget = seconds * (0.75 if B == args.batch else 0.25)
        t, n, t0 = 1, 0, None
        start = time.perf_counter()
        while True:
            O.train_step(p, m, v, t, cfg, X, eps)
            t += 1
            if t0 is None:
                t0 = time.perf_counter()      # first step = warm-up
                continue
            n += 1
            if time.perf_counter() - start > budget or n >= 200:
                break
        dt = time.perf_counter() - t0
        out[B] = (B * n / dt, n)
    return {"value": round(out[args.batch][0], 1), "unit": "images/sec", "cores": int(cores), "kind": "port",
            "sample": "oracle/dmvae_oracle.py float32 NumPy step, batch %d, %d timed steps after 1 warm-up" % (args.batch, out[args.batch][1]),
            "batch100_images_per_sec": round(out[100][0], 1)}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: libraries that write to file descriptor 1 themselves
    # (RCCL prints a version banner there at communicator creation) are pointed at stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DMVAE_DP_FORCE=1 on a one-GPU box: a one-rank RCCL communicator, so the N > 1 launch sequence
    # runs with the real (identity) collectives -- a rehearsal, never the reported N = 1 number
    force_dp = world == 1 and os.environ.get("DMVAE_DP_FORCE") == "1"
    if force_dp:
        for k, v in (("MASTER_PORT", "29513"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":     # rehearsal of the N > 1 orchestration on a one-GPU box: ranks share the card
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import dmvae_hip
    from dmvae_hip import StepEngine, GradExchange, prof_enable, prof_collect

    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    I, D, K, B = args.input_dim, args.latent_dim, args.n_clusters, args.batch
    enc = tuple(int(v) for v in args.enc_layers.split(","))
    dec = tuple(int(v) for v in args.dec_layers.split(","))
    eng = StepEngine(I, D, K, enc_layers=enc, head_dim=args.head_dim, dec_layers=dec, dtype=args.dtype, max_batch=B,
                     seed=1234 + rank, deterministic=args.deterministic, cnn=args.cnn)
    eng.init_parameters(0)
    eng.write_state(lr=args.lr)
    ex = GradExchange()
    ex.broadcast_(eng.param)
    eng.refresh_shadow()

    # synthetic MNIST-like rows resident in HBM: x = u * 1[v < 0.19]
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    data = torch.rand((args.rows, I), device=dev, generator=gen)
    data = data * (torch.rand((args.rows, I), device=dev, generator=gen) < 0.19)
    perm = torch.randperm(args.rows, device=dev, generator=gen).to(torch.int32)
    bpe = args.rows // B
    eng.reset_epoch(bpe, kl_ratio=1.0)
    sync = ex if ex.enabled else None
    if args.dp_dry_run and world == 1:
        # the data-parallel launch sequence without a communicator: what the N > 1 path costs in compute alone
        class _NoComm:
            overlap = args.dp_dry_run == "overlap"
            def __call__(self, g): return g
            def start(self, g): return None
            def wait(self, h): pass
        sync = _NoComm()

    if args.no_graph:
        def step():
            eng.train_step(data, perm, use_state_cursor=True, grad_sync=sync, grad_scale=ex.grad_scale)
    else:
        step = eng.capture_step(data, perm, grad_sync=sync, grad_scale=ex.grad_scale)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    st = eng.read_state()

    # ---- per-kernel HIP-event timing: eager launches of the same step, event pair around each launch
    rows = []
    if args.profile_steps > 0:
        torch.cuda.synchronize()
        prof_enable(True)
        import ctypes
        for _ in range(args.profile_steps):
            # keep the stream busy ~2 ms so the host enqueues the whole step before the GPU starts it:
            # the event brackets then hold kernel time only (no host launch latency inside them)
            dmvae_hip.lib.dmvae_debug_spin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 2000)
            eng.train_step(data, perm, use_state_cursor=True, grad_sync=None, grad_scale=1.0)
        torch.cuda.synchronize()
        rows = prof_collect()
        prof_enable(False)

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    ms_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    fpi = flops_per_image(I, D, K, enc, args.head_dim, dec, args.cnn)
    is_cfg2 = (I, D, K, B, enc, args.head_dim, dec, args.dtype, args.cnn) == (784, 64, 10, 4096, (500, 500), 2000, (2000, 500, 500), "bf16", False)
    arch = ("cnn(32,32,p,64,64,p,128,128,p)-" if args.cnn else "") + "%d-%s-(%d|%d)-z%d/K%d-%s-%d" % (I, "-".join(map(str, enc)), args.head_dim, args.head_dim, D, K, "-".join(map(str, dec)), I)
    out = {
        "metric": "images/sec (train), MNIST K=10 z=64 batch=4096/GPU bf16" if is_cfg2
                  else "images/sec (train), DMVAE %s batch=%d/GPU %s" % (arch, B, args.dtype),
        "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%sDMVAE MLP %s, one ELBO training step (gather+fwd+loss+bwd+Adam), synthetic rows resident in HBM"
                               % ("configs[1]: " if is_cfg2 else "", arch),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                   "hip_graph": (not args.no_graph) and sync is None, "float_atomics": bool(args.cnn),
                   "update": "stand-alone adam" if args.cnn and sync is None else "adam fused into the dW launch" if sync is None else "bucketed all-reduce overlapped with backward, then adam"},
        "step_flops_algorithmic": fpi * B,
        "step_mfma_frac_of_peak": round(fpi * B / (ms_step * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
        "last_loss": round(float(st.last_loss), 4),
    }
    if rows:
        ps = float(args.profile_steps)
        table = []
        for r in rows:
            ms = r["total_ms"] / ps
            table.append({"kernel": r["name"], "launches_per_step": r["launches"] / ps, "ms_per_step": round(ms, 4),
                          "avg_us": round(1e3 * r["total_ms"] / max(1, r["launches"]), 3),
                          "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2) if r["total_ms"] > 0 else 0.0,
                          "gbs": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1) if r["total_ms"] > 0 else 0.0})
        dom = max(rows, key=lambda r: r["total_ms"])
        # which roof bounds the dominant kernel: its algorithmic intensity against the ridge point
        # (dense MFMA peak / HBM peak = 312 flop/B for bf16).  The fused dW + Adam launch moves
        # ~300 MB for 41 GFLOP = 137 flop/B: HBM side of the ridge.
        peak_fl = 157.3 if args.dtype == "fp32" else PEAK_BF16_TFLOPS
        ai = dom["flops"] / max(dom["bytes"], 1.0)
        is_gemm = dom["name"].startswith("gemm") and ai >= peak_fl * 1e12 / (PEAK_HBM_GBS * 1e9)
        ach = (dom["flops"] if is_gemm else dom["bytes"]) / (dom["total_ms"] * 1e-3) / (1e12 if is_gemm else 1e9)
        peak = peak_fl if is_gemm else PEAK_HBM_GBS
        traffic = None      # HBM bytes per launch from PMC counters: a separate rocprofv3 --pmc run (tools/pmc_traffic.sh)
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom["name"], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"kernel": dom["name"], "bound": "mfma" if is_gemm else "hbm", "achieved": round(ach, 2),
                           "peak": peak, "unit": "TFLOP/s" if is_gemm else "GB/s", "frac": round(ach / peak, 4),
                           "traffic": traffic,
                           "launches_per_step": dom["launches"] / ps,
                           "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 3),
                           "algorithmic_per_launch": (dom["flops"] if is_gemm else dom["bytes"]) / dom["launches"],
                           "flops_per_launch": dom["flops"] / dom["launches"], "bytes_per_launch": dom["bytes"] / dom["launches"],
                           "flop_per_byte": round(ai, 1),
                           "mfma_frac": round(dom["flops"] / (dom["total_ms"] * 1e-3) / (peak_fl * 1e12), 4),
                           "hbm_frac": round(dom["bytes"] / (dom["total_ms"] * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                           "method": "hipEvent pair around every launch, %d eager steps after the timed region" % args.profile_steps}
        out["kernels"] = table
        out["kernel_ms_per_step_sum"] = round(sum(t["ms_per_step"] for t in table), 4)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
