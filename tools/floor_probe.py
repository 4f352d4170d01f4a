"""Fixed cost of a small GEMM launch: the same output shape at K = 64 ... 2048 (back-to-back launches in one stream,
HIP-graph replayed so the host is out of the picture).  Intercept = launch + prologue + epilogue, slope = K loop.
python tools/floor_probe.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
s = torch.cuda.Stream()
st = C.c_void_p(s.cuda_stream)
M = 4096
if os.environ.get("FLOOR_TILE"):            # e.g. FLOOR_TILE=64,64: force the tile shape (dmvae_debug_set_tile)
    bm, bn = (int(v) for v in os.environ["FLOOR_TILE"].split(","))
    L.check(L.lib.dmvae_debug_set_tile(bm, bn))
for lay, name, epi in ((0, "fwd bias+relu", L.EPI_BIAS_RELU), (1, "dX relu-gate", L.EPI_RELU_MASK)):
    for N in (512, 2048):
        for K in ((64, 2048) if os.environ.get("FLOOR_SHORT") else (64, 256, 512, 1024, 2048)):
            A = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
            B = (0.02 * torch.randn(K, N, device="cuda") if lay == 0 else 0.02 * torch.randn(N, K, device="cuda")).bfloat16()
            out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
            bias = torch.zeros(N, device="cuda")
            e = L.Epilogue(); e.kind = epi
            e.out, e.ldo, e.bias, e.aux0, e.ld0 = out.data_ptr(), N, bias.data_ptr(), Y.data_ptr(), N
            def run():
                L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), K, L.ptr(B), N if lay == 0 else K, C.byref(e), 1))
            with torch.cuda.stream(s):
                for _ in range(3): run()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(20): run()
                g.replay(); torch.cuda.synchronize()
                t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
                t0.record(s)
                for _ in range(10): g.replay()
                t1.record(s); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) / 200 * 1e3
            print("%-14s M %d N %4d K %4d : %6.2f us  %6.1f TF" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
