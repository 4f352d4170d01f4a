"""Does a HIP-graph launch boundary cost more than a kernel boundary?  The cfg2 step captured as ONE step per graph (what capture_step does) against TWO
and FOUR steps per graph: ms per step over 400 steps, interleaved rounds."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, ROOT)
from dmvae_hip import StepEngine
torch.cuda.set_device(0)
B, I = 4096, 784
data = torch.rand((4 * B, I), device="cuda"); perm = torch.randperm(4 * B, device="cuda").to(torch.int32)
e = StepEngine(I, 64, 10, dtype="bf16", max_batch=B); e.init_parameters(0); e.write_state(lr=0.002); e.reset_epoch(4)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): e.train_step(data, perm, use_state_cursor=True)
side.synchronize()
graphs = {}
for n in (1, 2, 4):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n): e.train_step(data, perm, None, None, None, 0, True)
    graphs[n] = g
res = {n: [] for n in graphs}
for rnd in range(7):
    for n, g in (graphs.items() if rnd % 2 == 0 else reversed(list(graphs.items()))):
        for _ in range(8 // n): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(400 // n): g.replay()
        torch.cuda.synchronize(); res[n].append((time.perf_counter() - t0) / 400 * 1e3)
for n, v in res.items(): print("%d step(s) per graph: ms/step median %.4f  min %.4f" % (n, sorted(v)[len(v) // 2], min(v)))
