"""The dZ GEMM (dX of the first decoder layer: M = B rows, N = the padded latent width, K = 2000 -> 2048) is 64..256 workgroups over a
32-tile K chain.  How much shorter is the chain in K slices?  Times the shape with the plain f32 store (1 slice) and with 2 / 4 / 8 K
slices (timed as the same launch shape, without any reduction logic): python tools/dz_probe.py [B] [N]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = 2048
torch.cuda.set_device(0)
side = torch.cuda.Stream()
dY = torch.randn(B, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
out = torch.zeros(B, N, device="cuda")
def timeit(fn, n=20, reps=5):
    with torch.cuda.stream(side):
        st = C.c_void_p(side.cuda_stream)
        for _ in range(2): fn(st)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(n): fn(st)
        ts = []
        for _ in range(reps):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(side); gr.replay(); t1.record(side); side.synchronize()
            ts.append(t0.elapsed_time(t1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
# (only the DW layout has the atomic epilogue: S slices are timed as the same launch shape -- S x the rows, 1 / S of the K, plain f32 store)
for split in (1, 2, 4, 8):
    Ms, Ks = B * split, K // split
    dYs = torch.randn(Ms, Ks, device="cuda").bfloat16(); Ws = torch.randn(N, Ks, device="cuda").bfloat16(); outs = torch.zeros(Ms, N, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_STORE_F32; e.out = outs.data_ptr(); e.ldo = N
    f = lambda st: L.check(L.lib.dmvae_gemm(st, 1, L.GEMM_DX, Ms, N, Ks, L.ptr(dYs), Ks, L.ptr(Ws), Ks, C.byref(e), 1))
    print("B=%d N=%d K=%d  as %d K slice(s) (%d workgroup rows x K=%d): %6.1f us" % (B, N, K, split, Ms // 64, Ks, timeit(f)), flush=True)
