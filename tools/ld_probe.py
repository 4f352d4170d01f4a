"""Does a power-of-two leading dimension (8 KiB row stride) cost the 256x256 kernel anything?  Same GEMM with ld = K and ld = K + 64.
python tools/ld_probe.py [step]      step: operands shaped like the training step's (post-ReLU activations, small weights,
ReLU-gated gradients) instead of dense N(0, 1) -- the chip's clock depends on the operand bits"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 8192, 4096, 4096
for lay, name in ((0, "fwd"), (1, "dX"), (2, "dW")):
    for pad, padb in ((0, 0), (192, 192), (192, 0), (0, 192), (0, 0), (192, 192)):
        if lay == 0: ra, ca, rb, cb = M, K, K, N
        elif lay == 1: ra, ca, rb, cb = M, K, N, K
        else: ra, ca, rb, cb = K, M, K, N
        A = torch.randn(ra, ca + pad, device="cuda"); B = torch.randn(rb, cb + padb, device="cuda")
        if len(sys.argv) > 1 and sys.argv[1] == "step":
            act = lambda t: torch.relu(t)
            grad = lambda t: 1e-4 * t * (torch.rand_like(t) < 0.5)
            if lay == 0: A, B = act(A), 0.02 * B
            elif lay == 1: A, B = grad(A), 0.02 * B
            else: A, B = act(A), grad(B)
        A = A.bfloat16(); B = B.bfloat16()
        out = torch.zeros(M, N + pad, device="cuda", dtype=torch.bfloat16 if lay != 2 else torch.float32)
        Y = torch.ones(M, N + pad, device="cuda", dtype=torch.bfloat16); bias = torch.zeros(N, device="cuda")
        e = L.Epilogue(); e.kind = (L.EPI_BIAS_RELU, L.EPI_RELU_MASK, L.EPI_STORE_F32)[lay]
        e.out, e.ldo, e.bias, e.aux0, e.ld0 = out.data_ptr(), N + pad, bias.data_ptr(), Y.data_ptr(), N + pad
        for _ in range(60): L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), ca + pad, L.ptr(B), cb + padb, C.byref(e), 1))
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(40): L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), ca + pad, L.ptr(B), cb + padb, C.byref(e), 1)
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / 40 * 1e3
        print("%-4s ldA = dim + %3d  ldB = dim + %3d : %7.1f us  %7.1f TF" % (name, pad, padb, us, 2.0 * M * N * K / us / 1e6), flush=True)
