"""Does a power-of-two leading dimension (8 KiB row stride) cost the 256x256 kernel anything?  Same GEMM with ld = K and ld = K + 64.
python tools/ld_probe.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 8192, 4096, 4096
for lay, name in ((0, "fwd"), (1, "dX"), (2, "dW")):
    for pad in (0, 64, 192):
        if lay == 0: ra, ca, rb, cb = M, K, K, N
        elif lay == 1: ra, ca, rb, cb = M, K, N, K
        else: ra, ca, rb, cb = K, M, K, N
        A = torch.randn(ra, ca + pad, device="cuda").bfloat16(); B = torch.randn(rb, cb + pad, device="cuda").bfloat16()
        out = torch.zeros(M, N + pad, device="cuda", dtype=torch.bfloat16 if lay != 2 else torch.float32)
        Y = torch.ones(M, N + pad, device="cuda", dtype=torch.bfloat16); bias = torch.zeros(N, device="cuda")
        e = L.Epilogue(); e.kind = (L.EPI_BIAS_RELU, L.EPI_RELU_MASK, L.EPI_STORE_F32)[lay]
        e.out, e.ldo, e.bias, e.aux0, e.ld0 = out.data_ptr(), N + pad, bias.data_ptr(), Y.data_ptr(), N + pad
        for _ in range(3): L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), ca + pad, L.ptr(B), cb + pad, C.byref(e), 1))
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20): L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), ca + pad, L.ptr(B), cb + pad, C.byref(e), 1)
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / 20 * 1e3
        print("%-4s ld = dim + %3d : %7.1f us  %7.1f TF" % (name, pad, us, 2.0 * M * N * K / us / 1e6), flush=True)
