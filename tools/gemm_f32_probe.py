"""f32 MFMA GEMM (parity-mode kernel) on the shapes the MFMA form of the latent contractions would use.
python tools/gemm_f32_probe.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
shapes = [("cfg5 W.T1", 0, 8192, 1024, 256, 1), ("cfg5 X1.T2^T", 1, 8192, 256, 1024, 1), ("cfg5 W^T.X1", 2, 256, 1088, 8192, 1),
          ("cfg5 W^T.X1 s8", 2, 256, 1088, 8192, 8),
          ("cfg4 W.T1", 0, 8192, 512, 64, 1), ("cfg4 X1.T2^T", 1, 8192, 64, 512, 1), ("cfg4 W^T.X1", 2, 64, 576, 8192, 1), ("cfg4 W^T.X1 s16", 2, 64, 576, 8192, 16)]
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, lay, M, N, K, split in shapes:
    if lay == 0: A = torch.randn(M, K, device="cuda"); Bm = torch.randn(K, N, device="cuda"); lda, ldb = K, N
    elif lay == 1: A = torch.randn(M, K, device="cuda"); Bm = torch.randn(N, K, device="cuda"); lda, ldb = K, K
    else: A = torch.randn(K, M, device="cuda"); Bm = torch.randn(K, N, device="cuda"); lda, ldb = M, N
    out = torch.zeros(M, N, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_ATOMIC_F32 if split > 1 else L.EPI_STORE_F32; e.out = out.data_ptr(); e.ldo = N
    for _ in range(3): L.check(L.lib.dmvae_gemm(st, 0, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), split))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): L.lib.dmvae_gemm(st, 0, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), split)
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    print("%-18s %5dx%5dx%5d split %2d  %8.1f us  %6.1f TF" % (name, M, N, K, split, us, 2.0 * M * N * K / us / 1e6), flush=True)
