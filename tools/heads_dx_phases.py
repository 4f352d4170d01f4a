"""Where a workgroup of csrc/heads_dx.hip spends its time (measurement build 10: tools/ablate.sh 10, then
    DMVAE_HIP_LIB=deep-mixture-vae_amd/build/libdmvae_hip_abl10.so python tools/heads_dx_phases.py [cfg2 cfg3]).
Every workgroup keeps shader-clock sums per phase in registers and writes them when it ends (measure.h).  Printed: medians over the
workgroups of each problem, in us at the clock the launch ran at (cycles / 100 MHz ticks of the same workgroup)."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L

SHAPES = {"cfg2": (4096, 128, 64), "cfg3": (16384, 256, 64)}
H = 2048
torch.cuda.set_device(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tab = C.c_void_p()
L.check(L.lib.dmvae_debug_stamps(C.byref(tab)))
for name in (sys.argv[1:] or ["cfg2", "cfg3"]):
    B, D2, Kp = SHAPES[name]
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    dmv = torch.randn(B, D2, device="cuda", generator=g).bfloat16(); dlg = torch.randn(B, Kp, device="cuda", generator=g).bfloat16()
    Wmv = torch.randn(H, D2, device="cuda", generator=g).bfloat16(); Wl = torch.randn(H, Kp, device="cuda", generator=g).bfloat16()
    hzc = torch.relu(torch.randn(B, 2 * H, device="cuda", generator=g)).bfloat16()
    out = torch.zeros(B, 2 * H, device="cuda", dtype=torch.bfloat16)
    probs = (L.GemmProblem * 2)()
    for i, (dY, Kd, W) in enumerate(((dmv, D2, Wmv), (dlg, Kp, Wl))):
        p = probs[i]; p.M, p.N, p.K = B, H, Kd
        p.A, p.lda, p.B, p.ldb = dY.data_ptr(), Kd, W.data_ptr(), Kd
        p.epi.kind = L.EPI_RELU_MASK; p.epi.out = out.data_ptr() + i * H * 2; p.epi.ldo = 2 * H
        p.epi.aux0 = hzc.data_ptr() + i * H * 2; p.epi.ld0 = 2 * H
    hip = C.CDLL("libamdhip64.so")
    for _ in range(20): L.check(L.lib.dmvae_gemm_grouped(st(), 1, L.GEMM_DX, probs, 2))       # warm clocks
    torch.cuda.synchronize()
    assert hip.hipMemset(tab, 0, 1024 * 16 * 8) == 0
    L.check(L.lib.dmvae_gemm_grouped(st(), 1, L.GEMM_DX, probs, 2)); torch.cuda.synchronize()
    buf = torch.empty(1024 * 16, dtype=torch.int64, device="cuda")
    assert hip.hipMemcpy(C.c_void_p(buf.data_ptr()), tab, 1024 * 16 * 8, 3) == 0        # 3 = device to device
    t = buf.cpu().numpy().astype(np.int64).reshape(1024, 16)
    t = t[t[:, 13] > 0]
    cyc_per_us = np.median((t[:, 4] - t[:, 0]) / np.maximum(1, (t[:, 11] - t[:, 10])) * 100.0)
    t0 = t[:, 10].min()
    print("%s: %d workgroups, shader clock %.0f MHz (median); launch span %.2f us (first entry -> last store acknowledged)"
          % (name, len(t), cyc_per_us, (t[:, 11].max() - t0) / 100.0))
    nsl = (H // (64 if D2 == 256 else 128), H // 128)          # column slices of the two problems; workgroups: chunks x slices each, first problem first
    n1 = len(t) * nsl[0] // (nsl[0] + nsl[1])
    for half, u in (("first problem (K = %d, %d-column slices)" % (D2, 64 if D2 == 256 else 128), t[:n1]), ("second problem (K = %d, 128-column slices)" % Kp, t[n1:])):
        f = lambda a: np.median(a) / cyc_per_us
        steps = np.median(u[:, 13])
        walk = u[:, 5] + u[:, 6] + u[:, 7] + u[:, 8] + u[:, 9]
        print("  %s, %d workgroups x %d steps, us (median over workgroups): entry %.2f after the first | W slice landed +%.2f | W fragments read, gates of tile 0 parked +%.2f |"
              " per step: barrier %.2f + fragments / MFMA / park %.2f + epilogue, stores issued %.2f + wait for the next gates, park them %.2f = %.2f | store drain %.2f | workgroup total %.2f"
              % (half, len(u), steps, np.median(u[:, 10] - t0) / 100.0, f(u[:, 1] - u[:, 0]), f(u[:, 2] - u[:, 1]), f(u[:, 5] + u[:, 6]) / steps, f(u[:, 7]) / steps,
                 f(u[:, 8]) / steps, f(u[:, 9]) / steps, f(walk) / steps, f(u[:, 4] - u[:, 3]), f(u[:, 4] - u[:, 0])))
