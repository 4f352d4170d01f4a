"""Times every GEMM of one cfg2 training step (bf16) per tile choice, through the C ABI."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [  # (name, layout, M, N, K, epi)
    ("F1 x->enc0", 0, B, 512, 832, L.EPI_BIAS_RELU), ("F2 enc1", 0, B, 512, 512, L.EPI_BIAS_RELU),
    ("F3 zc", 0, B, 4096, 512, L.EPI_BIAS_RELU), ("F4 mv", 0, B, 128, 2048, L.EPI_BIAS_F32),
    ("F5 logits", 0, B, 64, 2048, L.EPI_BIAS_F32), ("F6 dec0", 0, B, 2048, 64, L.EPI_BIAS_RELU),
    ("F7 dec1", 0, B, 512, 2048, L.EPI_BIAS_RELU), ("F9 out", 0, B, 832, 512, L.EPI_BIAS_RELU),
    ("X1 d_dec2", 1, B, 512, 832, L.EPI_RELU_MASK), ("X3 d_dec0", 1, B, 2048, 512, L.EPI_RELU_MASK),
    ("X4 dZ", 1, B, 64, 2048, L.EPI_STORE_F32), ("X5 dhz", 1, B, 2048, 128, L.EPI_RELU_MASK),
    ("X6 dhc", 1, B, 2048, 64, L.EPI_RELU_MASK), ("X7 d_enc1", 1, B, 512, 4096, L.EPI_RELU_MASK),
    ("X8 d_enc0", 1, B, 512, 512, L.EPI_RELU_MASK),
    ("W1 out", 2, 512, 832, B, L.EPI_STORE_F32), ("W2 dec2", 2, 512, 512, B, L.EPI_STORE_F32),
    ("W3 dec1", 2, 2048, 512, B, L.EPI_STORE_F32), ("W4 dec0", 2, 64, 2048, B, L.EPI_STORE_F32),
    ("W5 mv", 2, 2048, 128, B, L.EPI_STORE_F32), ("W6 logits", 2, 2048, 64, B, L.EPI_STORE_F32),
    ("W7 zc", 2, 512, 4096, B, L.EPI_STORE_F32), ("W9 enc0", 2, 832, 512, B, L.EPI_STORE_F32),
]
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = {}
for name, lay, M, N, K, epi in shapes:
    if lay == 0: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = K, N
    elif lay == 1: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(N, K, device="cuda").bfloat16(); lda, ldb = K, K
    else: A = torch.randn(K, M, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = M, N
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); outf = torch.zeros(M, N, device="cuda")
    bias = torch.zeros(N, device="cuda"); Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    e = L.Epilogue(); e.kind = epi
    e.out = (outf if epi in (L.EPI_STORE_F32, L.EPI_BIAS_F32, L.EPI_ATOMIC_F32) else outb).data_ptr(); e.ldo = N
    e.bias = bias.data_ptr(); e.aux0 = Y.data_ptr(); e.ld0 = N
    line = "%-12s %5dx%5dx%5d " % (name, M, N, K)
    best = None
    for (bm, bn) in [(0, 0), (128, 128), (128, 64), (64, 128), (64, 64)]:
        if bm and (M % bm or N % bn): line += "      -  "; continue
        L.check(L.lib.dmvae_debug_set_tile(bm, bn))
        for split in ([1] if lay != 2 or bm == 0 else [1, 4]):
            ee = e
            if split > 1:
                ee = L.Epilogue(); C.memmove(C.byref(ee), C.byref(e), C.sizeof(e)); ee.kind = L.EPI_ATOMIC_F32
            for _ in range(3): L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(ee), split))
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20): L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(ee), split)
            t1.record(); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) / 20 * 1e3
            line += "%s%6.1f " % ("s4:" if split > 1 else ("a:" if bm == 0 else ""), us)
            if bm == 0: tot["auto"] = tot.get("auto", 0) + us
            elif best is None or us < best: best = us
    tot["best"] = tot.get("best", 0) + (best or 0)
    print(line + "| TF(auto->best) %.0f" % (2.0 * M * N * K / best / 1e6), flush=True)
L.lib.dmvae_debug_set_tile(0, 0)
print("sum us: auto %.1f  best-per-shape %.1f   (columns: auto, 128x128, 128x64, 64x128, 64x64; s4 = split-K 4 atomics)" % (tot["auto"], tot["best"]))
