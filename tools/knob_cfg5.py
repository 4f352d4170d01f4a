"""A/B a tuning knob on the cfg5 step (4096-wide stack, B = 8192): interleaved rounds of HIP-graph replays.
python tools/knob_cfg5.py <knob> <values...>"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import StepEngine, _lib as L
which = int(sys.argv[1]); values = [int(v) for v in sys.argv[2:]]
torch.cuda.set_device(0)
B, I = 8192, 4096
data = torch.rand((4 * B, I), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(4 * B, device="cuda").to(torch.int32)
res = {v: [] for v in values}
engs = {}
for v in values:
    L.check(L.lib.dmvae_debug_set_knob(which, v))
    e = StepEngine(I, 512, 256, enc_layers=(4096,) * 4, head_dim=4096, dec_layers=(4096,) * 4, dtype="bf16", max_batch=B)
    e.init_parameters(0); e.write_state(lr=1e-4); e.reset_epoch(4)
    engs[v] = (e, e.capture_step(data, perm))        # the knob is baked into the captured graph
for rnd in range(4):
    for v in values:
        rp = engs[v][1]
        for _ in range(3): rp()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): rp()
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 15 * 1e3)
for v in values:
    r = sorted(res[v]); print("knob %d = %3d : ms/step median %.4f  min %.4f" % (which, v, r[len(r) // 2], r[0]))
