"""The grouped forward launch of the two head layers ([mean | log_var] = hz . Wmv, logits = hc . Wl; K = 2048) alone, per tile policy of the
grouped planner (knob 2: 0 all 64x64, 1 planned, 2 largest dividing tile):  python tools/heads_fwd_probe.py [cfg2|cfg3|cfg4]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
SHAPES = {"cfg2": (4096, 128, 64), "cfg3": (16384, 256, 64), "cfg4": (8192, 512, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B, D2, Kp = SHAPES[name]
H = 2048
torch.cuda.set_device(0)
side = torch.cuda.Stream()
hzc = torch.randn(B, 2 * H, device="cuda").bfloat16()
Wmv = torch.randn(H, D2, device="cuda").bfloat16(); Wl = torch.randn(H, Kp, device="cuda").bfloat16()
mv = torch.zeros(B, D2, device="cuda"); lg = torch.zeros(B, Kp, device="cuda"); bias = torch.zeros(1024, device="cuda")
def prob(A, W, N, out):
    p = L.GemmProblem(); p.M, p.N, p.K = B, N, H
    p.A, p.lda, p.B, p.ldb = A, 2 * H, W.data_ptr(), N
    p.epi.kind = L.EPI_BIAS_F32; p.epi.out = out.data_ptr(); p.epi.ldo = N; p.epi.bias = bias.data_ptr()
    return p
pz, pc = prob(hzc.data_ptr(), Wmv, D2, mv), prob(hzc.data_ptr() + H * 2, Wl, Kp, lg)
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
def group(ps):
    arr = (L.GemmProblem * len(ps))(*ps)
    return lambda st: L.check(L.lib.dmvae_gemm_grouped(st, 1, L.GEMM_FWD, arr, len(ps)))
def single(p):
    return lambda st: L.check(L.lib.dmvae_gemm(st, 1, L.GEMM_FWD, p.M, p.N, p.K, p.A, p.lda, p.B, p.ldb, C.byref(p.epi), 1))
print("%s: B=%d  N = %d | %d  K = %d   A = %.0f MB" % (name, B, D2, Kp, H, B * 2 * H * 2 / 1e6))
for mode in (1, 0, 2):
    L.check(L.lib.dmvae_debug_set_knob(2, mode))
    print("grouped, knob 2 = %d : %6.1f us" % (mode, timeit(group([pz, pc]))), flush=True)
L.check(L.lib.dmvae_debug_set_knob(2, 1))
print("z alone %6.1f us   c alone %6.1f us" % (timeit(single(pz)), timeit(single(pc))), flush=True)
for t in ((128, 64), (64, 64), (128, 128)):
    L.check(L.lib.dmvae_debug_set_tile(*t))
    print("tile %dx%d: z alone %6.1f us   c alone %6.1f us" % (t[0], t[1], timeit(single(pz)), timeit(single(pc))), flush=True)
