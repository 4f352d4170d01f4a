"""Sweeps tile / waves-per-workgroup for representative GEMM shapes of the step."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
B = 4096
shapes = [("F3 zc", 0, B, 4096, 512), ("F7 dec1", 0, B, 512, 2048), ("F1 enc0", 0, B, 512, 832), ("F2 enc1", 0, B, 512, 512),
          ("X7 d_enc1", 1, B, 512, 4096), ("X3 d_dec0", 1, B, 2048, 512), ("X5 dhz", 1, B, 2048, 128), ("X1 d_dec2", 1, B, 512, 832),
          ("W7 zc", 2, 512, 4096, B), ("W3 dec1", 2, 2048, 512, B)]
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
    line = "%-10s %5dx%5dx%5d ideal %4.1f |" % (name, M, N, K, 2.0 * M * N * K / 2.5e9)
    for (bm, bn, nw8) in [(128, 128, 0), (128, 128, 1), (128, 64, 0), (128, 64, 1), (64, 128, 0), (64, 64, 0)]:
        L.check(L.lib.dmvae_debug_set_tile(bm, bn)); L.check(L.lib.dmvae_debug_set_knob(1, nw8))
        line += " %dx%d%s %5.1f" % (bm, bn, "w8" if nw8 else "  ", timeit(lay, M, N, K, A, lda, Bm, ldb, e))
    print(line, flush=True)
