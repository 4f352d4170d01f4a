"""Placement and timeline of the grouped dW launch's workgroups (needs the DMVAE_ABLATE=6 build):
DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl6.so python3 tools/stamps.py"""
import ctypes as C, os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import StepEngine, _lib as L
torch.cuda.set_device(0)
data = torch.rand((65536, 784), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(65536, device="cuda").to(torch.int32)
e = StepEngine(784, 64, 10, dtype="bf16", max_batch=4096); e.init_parameters(0); e.reset_epoch(16)
for _ in range(5): e.train_step(data, perm, use_state_cursor=True)
torch.cuda.synchronize()
p = C.c_void_p()
L.check(L.lib.dmvae_debug_stamps(C.byref(p)))
buf = torch.empty(2048 * 4, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(buf.data_ptr()), p, C.c_size_t(2048 * 32), 3)
st = buf.cpu().view(2048, 4).numpy()
rows = [(i, int(r[0]), int(r[1]), int(r[2]) >> 32 & 0xffffffff, int(r[2]) & 0xf, int(r[3]) >> 8 & 0xff, int(r[3]) & 0xff)
        for i, r in enumerate(st) if r[1] != 0 and (int(r[3]) >> 8 & 0xff) == 2]
kloop = {i: (int(r[3]) >> 16) / 100.0 for i, r in enumerate(st) if r[1] != 0}
t_min = min(r[1] for r in rows); t_max = max(r[2] for r in rows)
print("dW workgroups %d, span %.2f us" % (len(rows), (t_max - t_min) / 100.0))
cus = collections.defaultdict(list)
for i, t0, t1, hw, xcc, lay, kind in rows:
    cu = (xcc, hw >> 13 & 7, hw >> 12 & 1, hw >> 8 & 15)          # xcc, se, sh, cu
    cus[cu].append((kind, (t0 - t_min) / 100.0, (t1 - t_min) / 100.0, i))
print("distinct CUs used: %d" % len(cus))
hist = collections.Counter()
for cu, l in cus.items():
    hist[tuple(sorted(k for k, _, _, _ in l))] += 1
for k, v in sorted(hist.items()): print("  CU holds tile kinds %s : %d CUs" % (list(k), v))
ends = sorted(max(x[2] for x in l) for l in cus.values())
print("CU finish time us: min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f" % (ends[0], ends[len(ends) // 4], ends[len(ends) // 2], ends[3 * len(ends) // 4], ends[-1]))
for kind in (0, 1, 2):
    d = sorted(t1 - t0 for i, t0, t1, hw, xcc, lay, k in rows if k == kind)
    if d:
        kl = sorted(kloop[i] for i, t0, t1, hw, xcc, lay, k in rows if k == kind)
        ep = sorted((t1 - t0) / 100.0 - kloop[i] for i, t0, t1, hw, xcc, lay, k in rows if k == kind)
        print("kind %d: n %d  duration us min %.1f median %.1f max %.1f | K loop median %.1f max %.1f | epilogue median %.1f max %.1f"
              % (kind, len(d), d[0] / 100.0, d[len(d) // 2] / 100.0, d[-1] / 100.0, kl[len(kl) // 2], kl[-1], ep[len(ep) // 2], ep[-1]))
late = sorted(rows, key=lambda r: -r[2])[:8]
print("last finishers:", [(r[0], r[6], round((r[1] - t_min) / 100.0, 1), round((r[2] - t_min) / 100.0, 1)) for r in late])
# the stragglers: which XCD / CU, their K loop and epilogue, and what shared the CU with them
cu_of = {}
for cu, l in cus.items():
    for k, a, b, i in l: cu_of[i] = cu
print("id    kind xcc  K loop  epilogue  end    | co-resident on the CU (kind, start, end)")
for r in sorted(rows, key=lambda r: -r[2])[:24]:
    i = r[0]; cu = cu_of[i]
    others = [(k, round(a, 1), round(b, 1)) for k, a, b, j in cus[cu] if j != i]
    print("%-5d %-4d %-3d  %6.1f  %6.1f   %6.1f | %s" % (i, r[6], r[4], kloop[i], (r[2] - r[1]) / 100.0 - kloop[i], (r[2] - t_min) / 100.0, others))
byx = collections.defaultdict(list)
for r in rows:
    if r[6] == 0: byx[r[4]].append((r[2] - t_min) / 100.0)
print("kind-0 end time by XCD (median / max):", {x: (round(sorted(v)[len(v) // 2], 1), round(max(v), 1)) for x, v in sorted(byx.items())})
