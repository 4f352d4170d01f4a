"""Timeline of the merged weight-gradient + Adam grid of the cfg5 step (needs the DMVAE_ABLATE=6 build: tools/ablate.sh 6):
per workgroup K-loop and epilogue spans, and how many workgroups sit in their epilogue at the same time.
DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl6.so python3 tools/anatomy256.py"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import StepEngine, _lib as L
torch.cuda.set_device(0)
for kv in os.environ.get("DMVAE_KNOBS", "").split(","):
    if kv: L.check(L.lib.dmvae_debug_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1])))
B, I = 8192, 4096
data = torch.rand((2 * B, I), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(2 * B, device="cuda").to(torch.int32)
e = StepEngine(I, 512, 256, enc_layers=(4096,) * 4, head_dim=4096, dec_layers=(4096,) * 4, dtype="bf16", max_batch=B)
e.init_parameters(0); e.write_state(lr=1e-4); e.reset_epoch(2)
for _ in range(3): e.train_step(data, perm, use_state_cursor=True)
torch.cuda.synchronize()
p = C.c_void_p(); L.check(L.lib.dmvae_debug_anatomy256(C.byref(p)))
buf = torch.empty(4096 * 8, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(buf.data_ptr()), p, C.c_size_t(4096 * 64), 3)
s = buf.cpu().numpy().reshape(4096, 8)
s = s[s[:, 2] != 0]                      # the LAST macro-tile launch of the step = the merged dW + Adam grid (its bias workgroups write nothing)
t0 = s[:, 0].min()
ent, kend, eend = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0
q = lambda v: "min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f" % tuple(np.percentile(v, [0, 25, 50, 75, 100]))
print("workgroups %d, span %.1f us" % (len(s), eend.max()))
print("K loop   us:", q(kend - ent))
print("epilogue us:", q(eend - kend))
order = np.argsort(ent)
for lo in range(0, len(s), 256):
    sel = order[lo:lo + 256]
    print("  workgroups %4d..%4d by start: start %7.1f..%7.1f  K loop median %6.1f  epilogue median %5.1f (min %5.1f max %5.1f)" %
          (lo, lo + len(sel) - 1, ent[sel].min(), ent[sel].max(), np.median((kend - ent)[sel]), np.median((eend - kend)[sel]), (eend - kend)[sel].min(), (eend - kend)[sel].max()))
# concurrency of epilogues over time
ts = np.arange(0, eend.max(), 5.0)
conc = np.array([((kend <= t) & (eend > t)).sum() for t in ts])
print("workgroups in their epilogue at once: mean %.1f  p50 %d  p90 %d  max %d  (share of time with none: %.2f)" % (conc.mean(), np.median(conc), np.percentile(conc, 90), conc.max(), (conc == 0).mean()))
