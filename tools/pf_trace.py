"""Replays the pipelined cfg2 step 300 times with knob 17 = argv[1] (run under rocprofv3 --kernel-trace --stats: tools/prefetch_trace.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, ROOT)
from dmvae_hip import StepEngine, _lib as L
torch.cuda.set_device(0)
B, I = 4096, 784
data = torch.rand((4 * B, I), device="cuda"); perm = torch.randperm(4 * B, device="cuda").to(torch.int32)
e = StepEngine(I, 64, 10, dtype="bf16", max_batch=B); e.init_parameters(0); e.write_state(lr=0.002); e.reset_epoch(4)
k = int(sys.argv[1])
for kv in os.environ.get("DMVAE_KNOBS", "").split(","):          # e.g. DMVAE_KNOBS=18=2
    if kv: L.check(L.lib.dmvae_debug_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1])))
if k >= 0: L.check(L.lib.dmvae_debug_set_knob(17, k))
rp = e.capture_step(data, perm, pipelined=k >= 0)
for _ in range(300): rp()
torch.cuda.synchronize()
