"""A/B a tuning knob on the CNN step: tools/cnn_knob.py <knob> <values...>  (ms/step by HIP-graph replay)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import StepEngine, _lib as L
which = int(sys.argv[1]); values = [int(v) for v in sys.argv[2:]]
torch.cuda.set_device(0)
B = int(os.environ.get("AB_B", 4096))
data = torch.rand((4 * B, 784), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(4 * B, device="cuda").to(torch.int32)
engs, res = {}, {v: [] for v in values}
for v in values:
    L.check(L.lib.dmvae_debug_set_knob(which, v))
    e = StepEngine(784, 64, 10, enc_layers=(500,), dtype="bf16", max_batch=B, cnn=True)
    e.init_parameters(0); e.reset_epoch(4)
    engs[v] = (e, e.capture_step(data, perm))
for rnd in range(3):
    for v in values:
        rp = engs[v][1]
        for _ in range(3): rp()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): rp()
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 30 * 1e3)
for v in values:
    r = sorted(res[v]); print("knob %d = %2d : ms/step median %.4f  min %.4f" % (which, v, r[len(r) // 2], r[0]))
