#!/usr/bin/env python3
"""DMVAE training-step benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): images/sec (train), MNIST-shaped K=10 z_dim=64
batch=4096 per GPU, bf16 MFMA GEMMs, synthetic 28x28 inputs resident in HBM.
One "step" = batch assembly + forward + loss + backward + (gradient all-reduce)
+ Adam, i.e. one session.run([loss, train_step]) of code/base_models.py:126-129.

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline"     : the dominant kernel family; durations are each dispatch's own
                   begin -> end timestamps (event pair bound to the dispatch by
                   hipExtLaunchKernelGGL: what rocprofv3 --kernel-trace reports) over a
                   few eager steps of the same process and shapes
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


# BASELINE.json configs as presets (per-GPU sizes; cfg4 / cfg5 name a global batch of 65 536 over 8 GPUs = 8192 per GPU).
PRESETS = {
    "cfg1": dict(batch=100, latent_dim=10, n_clusters=10),
    "cfg2": dict(batch=4096, latent_dim=64, n_clusters=10),
    "cfg3": dict(batch=16384, latent_dim=128, n_clusters=10),
    "cfg4": dict(batch=8192, latent_dim=256, n_clusters=50),
    # the north star's scaling statement, ">= 4x images/sec 1 -> 8 GPUs at batch 65 536": the SAME global batch at every N
    "cfg4-strong": dict(global_batch=65536, latent_dim=256, n_clusters=50),
    "cfg5": dict(batch=8192, latent_dim=512, n_clusters=256, input_dim=4096, enc_layers="4096,4096,4096,4096", head_dim=4096,
                 dec_layers="4096,4096,4096,4096", lr=1e-4),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3, help="the timed region (exactly --steps steps between barriers) is run this many "
                    "times back to back; the line reports the MEDIAN region (SURVEY 8d), all regions in repeat_ms_per_step")
    ap.add_argument("--config", default="", choices=[""] + sorted(PRESETS), help="a BASELINE.json config as a preset of the shape flags "
                    "below (per-GPU batch; cfg4 = the north star's scaling point, 8192 per GPU, D=256, K=50); default = cfg2, the metric")
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (cfg2: 4096)")
    ap.add_argument("--global-batch", dest="global_batch", type=int, default=0, help="STRONG scaling: this many images per step over all "
                    "--gpus ranks (per-GPU batch = global / N, which must be a whole number); the line then says scaling: strong")
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
    ap.add_argument("--mode", default="exact", choices=["exact", "relaxed"], help="exact: the checked-in graph, KL as the expectation over q(c|x) "
                    "(priors.py:130-145, cluster_sample False); relaxed: the Gumbel-Softmax sample zeta = softmax((logits + g) / tau) "
                    "(priors.py:170-181) feeding the cluster_sample branch of the KL (priors.py:118-128), Gumbel noise from the device Philox stream")
    ap.add_argument("--temperature", type=float, default=1.0, help="tau of --mode relaxed")
    ap.add_argument("--stable-warmup", dest="stable_warmup", type=int, default=8, help="after --warmup steps, run untimed 20-step regions until two "
                    "consecutive ones agree within 1 %% (the clock ramp: the first region after 5-20 steps reads 3-4 %% slow), at most this many; 0 = off")
    ap.add_argument("--rows", type=int, default=65536, help="synthetic dataset rows resident in HBM")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--deterministic", action="store_true", help="no float atomics (split-K off)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed oracle steps at the bench batch (after 3 warm-up steps)")
    ap.add_argument("--cpu-seconds", type=float, default=45.0, help="upper bound on the oracle's timed loop (larger configs stop early; the sample says so)")
    ap.add_argument("--elbo-epochs", type=int, default=50, help="N = 1, the metric's config only: the metric's second half, ELBO@50ep -- this many "
                    "epochs of bf16 training against the fp32 parity engine on identical batches and noise (0 = skip)")
    ap.add_argument("--dp-dry-run", default="", choices=["", "overlap", "single", "sharded"],
                    help="N = 1 only: issue the data-parallel launch sequence (staged backward + stand-alone Adam, or whole "
                         "backward + Adam) with a no-op exchange, to price the N > 1 path's compute")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    args = ap.parse_args()
    if args.config:
        for k, v in PRESETS[args.config].items():
            setattr(args, k, v)
    return args


def rocprof_name(name):
    """the library reports a grouped launch as gemm_bf16_grouped_mixed_tiles<Ll, Ee> (it dispatches 128x128 / 128x64 /
    64x64 tiles per problem); rocprofv3 prints the instantiation, whose template arguments are the smallest tile and
    the short-K variant flag (knob 7, off)"""
    import re
    m = re.match(r"gemm_bf16_grouped_mixed_tiles<L(\d+), E(\d+)>", name)
    return "gemm_bf16_grouped_kernel<64, 64, %s, %s, 4, 4, false>" % (m.group(1), m.group(2)) if m else name


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


def cpu_baseline(args):
    """The oracle timed on the host cores: float32 NumPy restatement of the same step (a reported baseline, not
    the optimisation target).  The TensorFlow-1.x reference itself cannot run in this image (TensorFlow is not
    installed); SURVEY 8d names a torch-CPU restatement -- the oracle IS the restatement this repo pins against the
    reference's golden vectors, it is NumPy on OpenBLAS with every host thread (the same sgemm torch-CPU would
    call), so it is the one timed, labelled "port".  3 warm-up + --cpu-steps (20) timed steps at the bench batch,
    then the reference's own batch size 100 (train.py:215-216)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import dmvae_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    out = {}
    for B, want, budget in ((args.batch, args.cpu_steps, args.cpu_seconds), (100, 50, 4.0)):
        cfg = O.Config(args.input_dim, args.latent_dim, args.n_clusters, tuple(int(v) for v in args.enc_layers.split(",")),
                       args.head_dim, tuple(int(v) for v in args.dec_layers.split(",")), cnn=args.cnn)
        p = O.init_params(cfg, 0, np.float32)
        m, v = O.adam_tf_init(p)
        X = O.synthetic_images(B, args.input_dim, seed=1)
        rng = np.random.RandomState(0)
        eps = rng.randn(B, args.latent_dim).astype(np.float32)
        t = 1
        for _ in range(3):                  # warm-up (BLAS thread pool, page faults of the 4 parameter-sized arrays)
            O.train_step(p, m, v, t, cfg, X, eps, lr=args.lr)
            t += 1
        n, t0 = 0, time.perf_counter()
        while n < want and (n < 2 or time.perf_counter() - t0 < budget):
            O.train_step(p, m, v, t, cfg, X, eps, lr=args.lr)
            t += 1
            n += 1
        dt = time.perf_counter() - t0
        out[B] = (B * n / dt, n, dt)
    v, n, dt = out[args.batch]
    return {"value": round(v, 1), "unit": "images/sec", "cores": int(cores), "kind": "port",
            "sample": "oracle/dmvae_oracle.py float32 NumPy/OpenBLAS step, batch %d, %d timed steps (%.1f s) after 3 warm-up steps%s"
                      % (args.batch, n, dt, "" if n >= args.cpu_steps else " (stopped by --cpu-seconds)"),
            "timed_steps": n, "batch100_images_per_sec": round(out[100][0], 1)}


def elbo_leg(args, data, epochs, StepEngine):
    """ELBO@50ep, the metric's second half: `epochs` epochs over the resident synthetic rows (no MNIST files in this
    image), bf16 against the fp32 parity engine on IDENTICAL batches (per-epoch device permutations, seed 1) and the
    IDENTICAL noise stream (device Philox: a draw depends on (seed, step, row, column), not on the arithmetic type).
    ELBO = -(epoch-mean loss) in nats/image as VAE.train_op returns it (base_models.py:130)."""
    rows, B = data.shape[0], args.batch
    bpe = rows // B
    pg = torch.Generator(device=data.device)
    pg.manual_seed(1)
    perms = [torch.randperm(rows, device=data.device, generator=pg).to(torch.int32) for _ in range(epochs)]
    enc = tuple(int(v) for v in args.enc_layers.split(","))
    dec = tuple(int(v) for v in args.dec_layers.split(","))
    final, secs = {}, {}
    for dtype in ("bf16", "fp32"):
        eng = StepEngine(args.input_dim, args.latent_dim, args.n_clusters, enc_layers=enc, head_dim=args.head_dim, dec_layers=dec,
                         dtype=dtype, max_batch=B, seed=1234)
        eng.init_parameters(0)
        eng.write_state(lr=args.lr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(epochs):
            eng.reset_epoch(bpe, kl_ratio=1.0)
            for _ in range(bpe):
                eng.train_step(data, perms[ep], use_state_cursor=True)
        torch.cuda.synchronize()
        secs[dtype] = time.perf_counter() - t0
        final[dtype] = float(eng.read_state().epoch_loss)
        del eng
    rel = abs(final["bf16"] - final["fp32"]) / abs(final["fp32"])
    return {"epochs": epochs, "elbo_bf16": round(-final["bf16"], 4), "elbo_fp32_parity": round(-final["fp32"], 4),
            "rel_diff": float("%.3e" % rel), "tolerance": 1e-3, "within_tolerance": bool(rel <= 1e-3),
            "unit": "nats/image", "data": "synthetic 28x28 stand-in, %d rows, batch %d (no MNIST files in the image)" % (rows, B),
            "reference": "unpinned: TensorFlow is not installed, the reference publishes no ELBO value (BASELINE.md)",
            "wall_seconds": {k: round(v, 2) for k, v in secs.items()}}


def respawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT torchrun: start the N ranks as fresh child processes through
    torch.distributed.run -- before this process has touched the GPU -- relay their output and exit with their code.
    (It used to run one rank and print n_gpus: 1.)"""
    import socket
    import subprocess
    have = torch.cuda.device_count()        # counting devices does not initialise the GPU
    if have < args.gpus and args.backend != "gloo":      # (a gloo rehearsal lets the ranks share one card)
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: libraries that write to file descriptor 1 themselves
    # (RCCL prints a version banner there at communicator creation) are pointed at stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(json_fd, 1)             # the children inherit the real stdout; rank 0 prints the line
        respawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (world == 1 and os.environ.get("DMVAE_DP_FORCE") == "1"):
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
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

    import dmvae_hip
    from dmvae_hip import StepEngine, make_exchange, prof_enable, prof_collect

    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    for kv in os.environ.get("DMVAE_KNOBS", "").split(","):      # measurement only: "knob=value,..." (include/dmvae_hip_debug.h)
        if kv:
            dmvae_hip._lib.check(dmvae_hip.lib.dmvae_debug_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1])))
    if args.global_batch:      # strong scaling: one global batch cut over the ranks
        if args.global_batch % world:
            raise SystemExit("bench.py: --global-batch %d is not a multiple of %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
        args.rows = max(args.rows, args.batch)
    I, D, K, B = args.input_dim, args.latent_dim, args.n_clusters, args.batch
    enc = tuple(int(v) for v in args.enc_layers.split(","))
    if args.cnn:
        enc = enc[-1:]            # the CNN trunk ends in ONE dense layer, fc 2048 -> 500 (base_models.py:202)
        args.enc_layers = str(enc[0])
    dec = tuple(int(v) for v in args.dec_layers.split(","))
    eng = StepEngine(I, D, K, enc_layers=enc, head_dim=args.head_dim, dec_layers=dec, dtype=args.dtype, max_batch=B,
                     seed=1234 + rank, deterministic=args.deterministic, cnn=args.cnn, mode=args.mode, temperature=args.temperature)
    eng.init_parameters(0)
    eng.write_state(lr=args.lr)
    ex = make_exchange(4 * eng.param.numel())
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
    # untimed: the chip's clock (and, N > 1, RCCL's channels) settle over the first few hundred microseconds of back-to-back steps --
    # region 1 of the round-3 driver line read 3.9 % slow behind 5 warm-up steps.  Run 20-step regions until two in a row agree
    # within 1 % (every rank takes the same decision: the slowest rank's clock decides), then time.
    settle = []
    if args.stable_warmup > 0:
        prev = None
        for _ in range(args.stable_warmup):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            cur = time.perf_counter() - t0
            if world > 1:
                import torch.distributed as dist
                t = torch.tensor([cur], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                cur = t.item()
            settle.append(cur / 20)
            if prev is not None and abs(cur - prev) <= 0.01 * prev:
                break
            prev = cur
    regions, rank_regions = [], []
    for _ in range(max(1, args.repeats)):       # each region: EXACTLY --steps steps between barrier + synchronize
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0             # this rank's own finish (before the closing barrier)
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the slowest rank's clock
            el = t.item()
            lo = torch.tensor([mine], dtype=torch.float64, device=dev)
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            rank_regions.append((lo.item(), hi.item()))
        regions.append(el)
    elapsed = sorted(regions)[len(regions) // 2]               # the median region
    st = eng.read_state()

    # N > 1 (or a forced one-rank communicator): what of the exchange is EXPOSED -- the time from the end of the backward pass's last
    # kernel to the end of the step's last kernel or collective (event pair on the compute stream around everything
    # _step_with_exchange issues behind the backward pass), averaged over a few eager steps; and the step with a no-op exchange.
    exch = None
    if sync is not None and getattr(sync, "enabled", False):
        exch = eng.measure_exchange(data, perm, sync, ex.grad_scale, steps=10)
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([exch["exposed_us"], exch["step_us"]], dtype=torch.float64, device=dev)
            tmax, tmin = t.clone(), t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            exch["exposed_us_max_over_ranks"], exch["exposed_us_min_over_ranks"] = round(tmax[0].item(), 1), round(tmin[0].item(), 1)
            exch["step_us_max_over_ranks"], exch["step_us_min_over_ranks"] = round(tmax[1].item(), 1), round(tmin[1].item(), 1)

    # ---- per-kernel HIP-event timing: eager launches of the same step, event pair around each launch
    rows = []
    if args.profile_steps > 0:
        torch.cuda.synchronize()
        if sync is not None and getattr(sync, "enabled", False):
            # the eager steps below are FUSED single-process steps (Adam over the whole arena): after sharded data-parallel steps a rank's
            # fp32 master is current on its own slice only, and the engine refuses to update stale weights -- bring it up to date first
            # (a collective: every rank runs this leg)
            eng.sync_master(ex)
            # ... and its Adam moments likewise (sync_master gathers weights only): the profiled steps run on a fresh optimizer
            # (timing does not care; the engine refuses a replicated update on slice-only moments)
            eng.reset_optimizer()
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
    is_cfg2 = (I, D, K, B, enc, args.head_dim, dec, args.dtype, args.cnn, args.mode) == (784, 64, 10, 4096, (500, 500), 2000, (2000, 500, 500), "bf16", False, "exact")
    arch = ("cnn(32,32,p,64,64,p,128,128,p)-" if args.cnn else "") + "%d-%s-(%d|%d)-z%d/K%d-%s-%d" % (I, "-".join(map(str, enc)), args.head_dim, args.head_dim, D, K, "-".join(map(str, dec)), I)
    out = {
        "metric": "images/sec (train), MNIST K=10 z=64 batch=4096/GPU bf16" if is_cfg2
                  else "images/sec (train), DMVAE %s batch=%d/GPU %s%s" % (arch, B, args.dtype, "" if args.mode == "exact" else " relaxed (Gumbel-Softmax)"),
        "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%sDMVAE MLP %s, one ELBO training step (gather+fwd+loss+bwd+Adam), synthetic rows resident in HBM"
                               % ("configs[1]: " if is_cfg2 else "", arch),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                   # ranks of the RCCL communicator the gradient exchange ran on (0: single process, no communicator;
                   # a gloo rehearsal on one card also says 0 and names its backend)
                   "rccl_ranks": (world if (world > 1 or force_dp) and args.backend == "nccl" else 0),
                   "collective_backend": (("rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo (rehearsal)") if (world > 1 or force_dp) else None),
                   # what has run BEFORE this line was printed: gloo at world 2 / 4 / 8 (CPU), two processes on the real kernels over gloo,
                   # a one-rank RCCL communicator (identity collectives) -- never a multi-rank RCCL job (gpurun gives one GPU)
                   "multi_rank_rccl_verified_before_this_run": False,
                   "hip_graph": (not args.no_graph) and sync is None, "float_atomics": False,
                   "update": "stand-alone adam" if args.cnn and sync is None else "adam fused into the dW launch" if sync is None else
                             ("%s, %s" % ("weights: reduce-scatter -> adam on the owned 1/world slice -> all-gather of the %s; biases + prior tables: all-reduce -> replicated adam"
                                          % ("bf16 shadow" if args.dtype == "bf16" else "fp32 weights") if getattr(sync, "sharded", False) else "all-reduce -> replicated adam",
                                          "three buckets overlapped with the backward pass" if getattr(sync, "overlap", False) else "one collective after the backward pass"))},
        "step_flops_algorithmic": fpi * B,
        "step_mfma_frac_of_peak": round(fpi * B / (ms_step * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
        "last_loss": round(float(st.last_loss), 4),
    }
    out["repeat_ms_per_step"] = [round(1e3 * r / args.steps, 4) for r in regions]
    out["settle_ms_per_step"] = [round(1e3 * r, 4) for r in settle]          # the untimed 20-step regions run until two agreed within 1 %
    out["config"]["mode"] = ("exact (expectation over q(c|x), priors.py:130-145)" if args.mode == "exact" else
                             "relaxed (Gumbel-Softmax sample, tau = %g, device Philox Gumbel noise; priors.py:118-128,170-181)" % args.temperature)
    if rank_regions:      # per-rank finish of each timed region before the closing barrier: min / max over ranks, ms per step
        out["rank_ms_per_step_min_max"] = [[round(1e3 * a / args.steps, 4), round(1e3 * b / args.steps, 4)] for a, b in rank_regions]
    if exch is not None:
        # the first sharded step of a multi-rank job compares the replicas (checksums of the gathered weights and of the replicated
        # tail, all-reduce MIN == MAX: parallel.ShardedExchange.self_check); a failed check raises on every rank before this line
        exch["self_check"] = getattr(eng, "_exchange_check", None) or ("not run (%s)" % ("all-reduce form: every rank applies the same sum" if not getattr(sync, "sharded", False) else "one rank"))
        out["exchange"] = exch
    out["timing"] = "median of %d regions of %d steps, each bracketed by barrier + synchronize" % (len(regions), args.steps)
    if rows:
        ps = float(args.profile_steps)
        # Per-kernel durations: each dispatch's OWN begin -> end timestamps (the event pair hipExtLaunchKernelGGL binds to
        # the dispatch: the completion signal's start / end, the figure rocprofv3 --kernel-trace prints), taken on eager
        # launches after the timed region.  No scaling of any kind (r02 scaled event BRACKETS by one factor so that they
        # summed to the graph-replayed step; a bracket's excess over the kernel is additive -- its dispatch and markers --
        # so that took time from the long kernels and gave it to the short ones: VERDICT r2 weak #3 / ADVICE r2).  The
        # bracket is kept as avg_us_event.  Were the runtime to return no dispatch timestamps the bracket itself would
        # be used, unscaled (the conservative figure), and `method` would say so.
        have_k = all(r["kernel_launches"] > 0 and r["kernel_ms"] > 0 for r in rows)
        dur = (lambda r: r["kernel_ms"]) if have_k else (lambda r: r["total_ms"])
        ev_sum = sum(r["total_ms"] for r in rows) / ps
        table = []
        for r in rows:
            tot = dur(r)
            table.append({"kernel": r["name"], "launches_per_step": r["launches"] / ps, "ms_per_step": round(tot / ps, 4),
                          "avg_us": round(1e3 * tot / max(1, r["launches"]), 3),
                          "avg_us_event": round(1e3 * r["total_ms"] / max(1, r["launches"]), 3),
                          "dispatches_per_launch": round(r["kernel_launches"] / max(1, r["launches"]), 2),
                          "tflops": round(r["flops"] / (tot * 1e-3) / 1e12, 2) if tot > 0 else 0.0,
                          "gbs": round(r["bytes"] / (tot * 1e-3) / 1e9, 1) if tot > 0 else 0.0})
        dom = max(rows, key=dur)
        dom_ms = dur(dom)
        # which roof bounds the dominant kernel: its algorithmic intensity against the ridge point
        # (dense MFMA peak / HBM peak = 312 flop/B for bf16).  The fused dW + Adam launch moves
        # ~300 MB for 41 GFLOP = 137 flop/B: HBM side of the ridge.
        peak_fl = 157.3 if args.dtype == "fp32" else PEAK_BF16_TFLOPS
        ai = dom["flops"] / max(dom["bytes"], 1.0)
        is_gemm = dom["name"].startswith("gemm") and ai >= peak_fl * 1e12 / (PEAK_HBM_GBS * 1e9)
        ach = (dom["flops"] if is_gemm else dom["bytes"]) / (dom_ms * 1e-3) / (1e12 if is_gemm else 1e9)
        peak = peak_fl if is_gemm else PEAK_HBM_GBS
        # HBM bytes per launch: NOT measured in this run -- PMC counters need their own rocprofv3 --pmc passes
        # (tools/pmc_traffic.sh); the figure for this kernel is read from the committed file named in traffic_source
        traffic, traffic_source = None, None
        rocname = rocprof_name(dom["name"])
        tname = "traffic.json" if is_cfg2 else ("traffic_%s.json" % args.config if args.config else None)
        tp = os.path.join(ROOT, "profiles", tname) if tname else None
        if tp and os.path.exists(tp) and not args.cnn:      # (a PMC pass is of ONE config: other configs share kernel names, not traffic)
            try:
                tj = json.load(open(tp))
                hit = tj.get(dom["name"]) or tj.get(rocname)
                if hit is None:      # tolerate a template argument list that grew since the PMC pass was keyed
                    stem = rocname.rstrip(">")
                    hit = next((v for k, v in tj.items() if k.startswith(stem)), {})
                traffic = hit.get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = ("committed PMC pass profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                      "FETCH_SIZE x 2 per the gfx950 correction; not collected in this run" % tname)
            except Exception:
                traffic = None
        out["roofline"] = {"kernel": dom["name"], "rocprof_kernel": rocname,
                           "bound": "mfma" if is_gemm else "hbm", "achieved": round(ach, 2),
                           "peak": peak, "unit": "TFLOP/s" if is_gemm else "GB/s", "frac": round(ach / peak, 4),
                           "traffic": traffic, "traffic_source": traffic_source,
                           "launches_per_step": dom["launches"] / ps,
                           "avg_launch_us": round(1e3 * dom_ms / dom["launches"], 3),
                           "avg_launch_us_event": round(1e3 * dom["total_ms"] / dom["launches"], 3),
                           "algorithmic_per_launch": (dom["flops"] if is_gemm else dom["bytes"]) / dom["launches"],
                           "flops_per_launch": dom["flops"] / dom["launches"], "bytes_per_launch": dom["bytes"] / dom["launches"],
                           "flop_per_byte": round(ai, 1),
                           "mfma_frac": round(dom["flops"] / (dom_ms * 1e-3) / (peak_fl * 1e12), 4),
                           "hbm_frac": round(dom["bytes"] / (dom_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                           "method": ("dispatch timestamps: an event pair bound to every kernel dispatch (hipExtLaunchKernelGGL) of %d eager steps after the "
                                      "timed region; unscaled; the durations rocprofv3 --kernel-trace reports" if have_k else
                                      "hipEvent BRACKET around every launch of %d eager steps after the timed region (the runtime returned no dispatch "
                                      "timestamps); unscaled: holds the dispatch besides the kernel") % args.profile_steps}
        assert out["roofline"]["frac"] <= 1.0, "roofline.frac %.4f > 1: the kernel's algorithmic bytes / flops are over-counted" % out["roofline"]["frac"]
        out["kernels"] = table
        out["kernels_per_step"] = round(sum(t["launches_per_step"] for t in table), 2)
        out["kernel_ms_per_step_sum"] = round(sum(t["ms_per_step"] for t in table), 4)
        out["kernel_ms_per_step_sum_event"] = round(ev_sum, 4)
    if world == 1 and is_cfg2 and args.elbo_epochs > 0 and not args.dp_dry_run and not force_dp:
        del eng, step
        out["elbo50" if args.elbo_epochs == 50 else "elbo"] = elbo_leg(args, data, args.elbo_epochs, StepEngine)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
