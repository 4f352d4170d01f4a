"""A/B a tuning knob on the step of a BASELINE.json config: interleaved rounds of HIP-graph replays on one box.
python tools/knob_step.py <cfg1..cfg5>[:batch=N] <knob | pf> <values...>   (knobs: include/dmvae_hip_debug.h, dmvae_debug_set_knob; pf: the pipelined capture, 0 / 1)
One engine; one captured graph per LISTED value, all on the same buffers; a value may be listed more than once.  CONTROL: list the values
as  a b b a : the spread between the two graphs of the same value is the harness's own spread, and a difference between values means
something only beyond it."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, ROOT)
from dmvae_hip import StepEngine, _lib as L
import bench
cfg = dict(bench.PRESETS[sys.argv[1].split(":")[0]])
for kv in sys.argv[1].split(":")[1:]: cfg[kv.split("=")[0]] = int(kv.split("=")[1])          # e.g. cfg2:batch=1024
which = -1 if sys.argv[2] == "pf" else int(sys.argv[2]); values = [int(v) for v in sys.argv[3:]]      # "pf": capture_step(pipelined = value)
torch.cuda.set_device(0)
tup = lambda t: tuple(int(x) for x in t.split(","))
B, I = cfg["batch"], cfg.get("input_dim", 784)
rows = 4 * B
data = torch.rand((rows, I), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(rows, device="cuda").to(torch.int32)
# ONE engine, one graph per listed value captured on the SAME buffers (the knob is read at enqueue time, so each graph keeps the kernels
# it was captured with).  Round 4: with one engine per value, engines of the SAME value differed by 0.9-1.9 % (later-created engines ran
# slower: other workspace addresses) -- as much as the effects under test.  Parameters evolve as the graphs replay; timing does not care.
e = StepEngine(I, cfg["latent_dim"], cfg["n_clusters"], enc_layers=tup(cfg.get("enc_layers", "500,500")), head_dim=cfg.get("head_dim", 2000),
               dec_layers=tup(cfg.get("dec_layers", "2000,500,500")), dtype="bf16", max_batch=B)
e.init_parameters(0); e.write_state(lr=cfg.get("lr", 0.002)); e.reset_epoch(4)
graphs = []
for v in values:
    if which >= 0: L.check(L.lib.dmvae_debug_set_knob(which, v))
    if which < 0: L.check(L.lib.dmvae_debug_set_knob(17, v - 10 if v >= 10 else 1))   # pf 1x: pipelined with knob 17 = x (0: the gather as a launch of its own, ...)
    if which < 0 and v == 3:                       # pf 3 (control): two PLAIN graphs replayed in turn -- what alternating graph executables costs by itself
        pair = [e.capture_step(data, perm, pipelined=False) for _ in range(2)]
        e._graph_keep = getattr(e, "_graph_keep", []) + [pair]
        state = [0]
        def alt(pair=pair, state=state):
            pair[state[0]](); state[0] ^= 1
        graphs.append(alt)
        continue
    graphs.append(e.capture_step(data, perm, pipelined=bool(v)) if which < 0 else e.capture_step(data, perm))
    e._graph_keep = getattr(e, "_graph_keep", []) + [e._graph]          # capture_step keeps only the latest graph alive
res = {i: [] for i in range(len(values))}
steps = 200 if B <= 4096 else 30
for rnd in range(9):
    order = list(range(len(values)))
    if rnd % 2: order.reverse()
    for i in order:
        rp = graphs[i]
        for _ in range(5): rp()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): rp()
        torch.cuda.synchronize(); res[i].append((time.perf_counter() - t0) / steps * 1e3)
assert torch.isfinite(e.param).all()
for i, v in enumerate(values):
    r = sorted(res[i]); print("%s knob %d = %3d (graph %d) : ms/step median %.4f  min %.4f" % (sys.argv[1], which, v, i, r[len(r) // 2], r[0]), flush=True)
by = {}
for i, v in enumerate(values): by.setdefault(v, []).append(sorted(res[i])[len(res[i]) // 2])
if any(len(m) > 1 for m in by.values()):
    print("%s: same-value graphs differ by up to %.2f %% (the harness's own spread); value means: %s" % (sys.argv[1],
          100 * max((max(m) - min(m)) / min(m) for m in by.values() if len(m) > 1), {v: round(sum(m) / len(m), 4) for v, m in by.items()}), flush=True)
