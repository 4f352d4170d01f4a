"""A/B of the 256x256 macro-tile kernel (knob 6 = 2) against the 64..128-wide tiles (knob 6 = 0) on the large GEMM
shapes of the configs, random bf16 operands, through the C ABI.   python tools/gemm256_bench.py [B]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [  # (name, layout, M, N, K, epi)
    ("cfg5 fwd 4096->4096", 0, B, 4096, 4096, L.EPI_BIAS_RELU),
    ("cfg5 fwd zc 4096->8192", 0, B, 8192, 4096, L.EPI_BIAS_RELU),
    ("cfg5 dX 4096<-4096", 1, B, 4096, 4096, L.EPI_RELU_MASK),
    ("cfg5 dX 4096<-8192", 1, B, 4096, 8192, L.EPI_RELU_MASK),
    ("cfg5 dW 4096x4096", 2, 4096, 4096, B, L.EPI_STORE_F32),
    ("cfg5 dW 4096x8192", 2, 4096, 8192, B, L.EPI_STORE_F32),
    ("cfg2 fwd zc 512->4096", 0, B, 4096, 512, L.EPI_BIAS_RELU),
    ("cfg2 dX 512<-4096", 1, B, 512, 4096, L.EPI_RELU_MASK),
    ("cfg2 fwd dec0 64->2048", 0, B, 2048, 64, L.EPI_BIAS_RELU),
    ("cfg2 fwd dec1 2048->512", 0, B, 512, 2048, L.EPI_BIAS_RELU),
    ("cfg2 dW 512x4096", 2, 512, 4096, B, L.EPI_STORE_F32),
    ("cfg2 dW 2048x512", 2, 2048, 512, B, L.EPI_STORE_F32),
]
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, lay, M, N, K, epi in shapes:
    if lay == 0: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = K, N
    elif lay == 1: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(N, K, device="cuda").bfloat16(); lda, ldb = K, K
    else: A = torch.randn(K, M, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = M, N
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); outf = torch.zeros(M, N, device="cuda")
    bias = torch.zeros(N, device="cuda"); Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    e = L.Epilogue(); e.kind = epi
    e.out = (outf if epi == L.EPI_STORE_F32 else outb).data_ptr(); e.ldo = N
    e.bias = bias.data_ptr(); e.aux0 = Y.data_ptr(); e.ld0 = N
    line = "%-26s %5dx%5dx%5d " % (name, M, N, K)
    for policy in (0, 2):
        if policy == 2 and (M % 256 or N % 256):
            line += "   256: n/a"
            continue
        L.check(L.lib.dmvae_debug_set_knob(6, policy))
        for _ in range(3): L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), 1))
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        n = 20
        t0.record()
        for _ in range(n): L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), 1)
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / n * 1e3
        line += "  %s %8.1f us %7.1f TF" % ("small:" if policy == 0 else "256:", us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
L.lib.dmvae_debug_set_knob(6, 1)
