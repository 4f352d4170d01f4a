"""Times the step's GEMMs (auto tile) with the library DMVAE_HIP_LIB selects; one line per shape.
Used with tools/ablate.sh variants to split a GEMM's time into MFMA / LDS-read / global-load parts."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
B = 4096
tile = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 0)
shapes = [("F2 enc1", 0, B, 512, 512, L.EPI_BIAS_RELU), ("F3 hz", 0, B, 2048, 512, L.EPI_BIAS_RELU),
          ("F7 dec1", 0, B, 512, 2048, L.EPI_BIAS_RELU), ("X3 d_dec0", 1, B, 2048, 512, L.EPI_RELU_MASK),
          ("X7 d_enc1", 1, B, 512, 2048, L.EPI_RELU_MASK), ("W3 dec1", 2, 2048, 512, B, L.EPI_STORE_F32),
          ("W2 dec2", 2, 512, 512, B, L.EPI_STORE_F32)]
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.check(L.lib.dmvae_debug_set_tile(*tile))
for kv in filter(None, os.environ.get("KNOBS", "").split(",")):      # e.g. KNOBS=3=1,1=0
    L.check(L.lib.dmvae_debug_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1])))
out = []
for name, lay, M, N, K, epi in shapes:
    if lay == 0: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = K, N
    elif lay == 1: A = torch.randn(M, K, device="cuda").bfloat16(); Bm = torch.randn(N, K, device="cuda").bfloat16(); lda, ldb = K, K
    else: A = torch.randn(K, M, device="cuda").bfloat16(); Bm = torch.randn(K, N, device="cuda").bfloat16(); lda, ldb = M, N
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); outf = torch.zeros(M, N, device="cuda")
    bias = torch.zeros(N, device="cuda"); Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    e = L.Epilogue(); e.kind = epi
    e.out = (outf if epi == L.EPI_STORE_F32 else outb).data_ptr(); e.ldo = N
    e.bias = bias.data_ptr(); e.aux0 = Y.data_ptr(); e.ld0 = N
    for _ in range(5): L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), 1))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50): L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), 1)
    t1.record(); torch.cuda.synchronize()
    out.append("%s %.1f" % (name.split()[0], t0.elapsed_time(t1) / 50 * 1e3))
print("%-28s tile %s | " % (os.path.basename(os.environ.get("DMVAE_HIP_LIB", "product")), tile) + "  ".join(out), flush=True)
