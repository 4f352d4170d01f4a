"""F4 / F5 (sibling head layers) separately vs as one grouped launch."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 4096
hzc = torch.randn(B, 4096, device="cuda").bfloat16()
Wmv = torch.randn(2048, 128, device="cuda").bfloat16(); Wl = torch.randn(2048, 64, device="cuda").bfloat16()
mv = torch.zeros(B, 128, device="cuda"); lg = torch.zeros(B, 64, device="cuda")
bias = torch.zeros(256, device="cuda")
def prob(A, lda, W, N, out):
    p = L.GemmProblem(); p.M, p.N, p.K = B, N, 2048
    p.A, p.lda, p.B, p.ldb = A, lda, W.data_ptr(), N
    p.epi.kind = L.EPI_BIAS_F32; p.epi.out = out.data_ptr(); p.epi.ldo = N; p.epi.bias = bias.data_ptr()
    return p
p4 = prob(hzc.data_ptr(), 4096, Wmv, 128, mv); p5 = prob(hzc.data_ptr() + 2048 * 2, 4096, Wl, 64, lg)
def timeit(fn, n=30):
    for _ in range(3): fn()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3
def single(p): return lambda: L.check(L.lib.dmvae_gemm(st, 1, 0, p.M, p.N, p.K, p.A, p.lda, p.B, p.ldb, C.byref(p.epi), 1))
def group(ps):
    arr = (L.GemmProblem * len(ps))(*ps)
    return lambda: L.check(L.lib.dmvae_gemm_grouped(st, 1, 0, arr, len(ps)))
print("F4 alone %.1f  F5 alone %.1f  F4 then F5 %.1f" % (timeit(single(p4)), timeit(single(p5)), timeit(lambda: (single(p4)(), single(p5)()))))
print("group[F4] %.1f  group[F5] %.1f  group[F4,F5] %.1f  group[F5,F4] %.1f" % (timeit(group([p4])), timeit(group([p5])), timeit(group([p4, p5])), timeit(group([p5, p4]))))
for gm in (1, 2, 64):
    L.lib.dmvae_debug_set_knob(0, gm)
    print("group_m=%d: group[F4,F5] %.1f  F4 alone %.1f" % (gm, timeit(group([p4, p5])), timeit(single(p4))))
