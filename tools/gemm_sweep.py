"""Sweeps tile / supertile height / ring depth for representative GEMM shapes of the step."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
B = 4096
shapes = [("F3 zc", 0, B, 4096, 512), ("F7 dec1", 0, B, 512, 2048), ("F1 enc0", 0, B, 512, 832), ("X7 d_enc1", 1, B, 512, 4096),
          ("X3 d_dec0", 1, B, 2048, 512), ("W7 zc", 2, 512, 4096, B), ("W3 dec1", 2, 2048, 512, B), ("W2 dec2", 2, 512, 512, B)]
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(lay, M, N, K, A, lda, Bm, ldb, e, split=1):
    for _ in range(3): L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), split))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), split)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / 20 * 1e3
for name, lay, M, N, K in shapes:
    if lay == 0: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = K, N
    elif lay == 1: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(N, K, device="cuda").bfloat16(); lda, ldb = K, K
    else: A = torch.randn(K, M, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = M, N
    out = torch.zeros(M, N, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_STORE_F32; e.out = out.data_ptr(); e.ldo = N
    print("%-10s %dx%dx%d  (ideal @2.5PF %.1f us)" % (name, M, N, K, 2.0 * M * N * K / 2.5e9))
    for (bm, bn, stg) in [(128, 128, 0), (128, 128, 1), (128, 64, 0), (128, 64, 1), (64, 128, 1), (64, 64, 0), (64, 64, 1)]:
        L.check(L.lib.dmvae_debug_set_tile(bm, bn)); L.check(L.lib.dmvae_debug_set_knob(1, stg))
        line = "   %3dx%3d s%d:" % (bm, bn, stg)
        for gm in (1, 8):
            L.check(L.lib.dmvae_debug_set_knob(0, gm))
            line += " g%-2d %6.1f" % (gm, timeit(lay, M, N, K, A, lda, Bm, ldb, e))
        if lay == 2:
            ea = L.Epilogue(); ea.kind = L.EPI_ATOMIC_F32; ea.out = out.data_ptr(); ea.ldo = N
            L.check(L.lib.dmvae_debug_set_knob(0, 8))
            for sp in (2, 4, 8): line += "  sk%d %6.1f" % (sp, timeit(lay, M, N, K, A, lda, Bm, ldb, ea, sp))
        print(line, flush=True)
