"""Runs one bf16 GEMM shape repeatedly (for rocprofv3 counter passes): layout M N K [bm bn] [iters]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
lay, M, N, K = (int(v) for v in sys.argv[1:5])
bm, bn = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 0)
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
if lay == 0: A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = K, N
elif lay == 1: A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16(); lda, ldb = K, K
else: A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = M, N
out = torch.zeros(M, N, device="cuda")
e = L.Epilogue(); e.kind = L.EPI_STORE_F32; e.out = out.data_ptr(); e.ldo = N
L.check(L.lib.dmvae_debug_set_tile(bm, bn))
for _ in range(iters):
    L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(B), ldb, C.byref(e), 1))
torch.cuda.synchronize()
