"""Can a streaming Adam kernel run BESIDE the 256x256 macro-tile GEMM (one 128-KiB workgroup per CU)?  The GEMM on one
stream, dmvae_adam_tf on another; wall time of both against each alone.  python tools/overlap_probe.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa, pb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
M, N, K = 8192, 4096, 4096
NP = 4096 * 4096
prm = [torch.randn(NP, device="cuda") * 0.02 for _ in range(4)]
prm[3].abs_()
pbf = torch.zeros(NP, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(N, device="cuda")
REP = 20

def gemm(lay, A, B, e, lda, ldb):
    L.check(L.lib.dmvae_gemm(pa, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(B), ldb, C.byref(e), 1))

def adam():
    L.check(L.lib.dmvae_adam_tf(pb, NP, L.ptr(prm[0]), L.ptr(prm[1]), L.ptr(prm[2]), L.ptr(prm[3]), L.ptr(pbf), 1e-4, 0.9, 0.999, 1e-8, 1.0, 0, 1, None))

for lay, name in ((1, "dX"), (0, "fwd"), (2, "dW")):
    if lay == 0: ra, ca, rb, cb = M, K, K, N
    elif lay == 1: ra, ca, rb, cb = M, K, N, K
    else: ra, ca, rb, cb = K, M, K, N
    A = torch.relu(torch.randn(ra, ca, device="cuda")).bfloat16(); B = (0.02 * torch.randn(rb, cb, device="cuda")).bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if lay != 2 else torch.float32)
    Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    e = L.Epilogue(); e.kind = (L.EPI_BIAS_RELU, L.EPI_RELU_MASK, L.EPI_STORE_F32)[lay]
    e.out, e.ldo, e.bias, e.aux0, e.ld0 = out.data_ptr(), N, bias.data_ptr(), Y.data_ptr(), N
    for _ in range(40): gemm(lay, A, B, e, ca, cb)
    torch.cuda.synchronize()
    res = {}
    for mode in ("gemm", "adam", "both"):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record(torch.cuda.default_stream())
        sa.wait_stream(torch.cuda.default_stream()); sb.wait_stream(torch.cuda.default_stream())
        for _ in range(REP):
            if mode != "adam": gemm(lay, A, B, e, ca, cb)
            if mode != "gemm": adam()
        torch.cuda.default_stream().wait_stream(sa); torch.cuda.default_stream().wait_stream(sb)
        t1.record(torch.cuda.default_stream())
        torch.cuda.synchronize()
        res[mode] = t0.elapsed_time(t1) / REP * 1e3
    print("%-4s per pair: gemm alone %6.1f us, adam alone %6.1f us, both streams %6.1f us (serial sum %6.1f)" %
          (name, res["gemm"], res["adam"], res["both"], res["gemm"] + res["adam"]), flush=True)
