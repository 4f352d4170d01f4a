"""Does a TF-Adam stream fit in the SHADOW of the 256x256 macro-tile weight-gradient GEMM?
The dW GEMM of one 4096 x 4096 layer at 8192 rows (STORE_F32: the gradient dumped, no fused update) on one stream; on another the Adam update
of a layer's 16.8 M parameters -- the plain kernel (adam_tf_kernel, 2048 workgroups), and the SHADOW form (csrc/elementwise.hip
adam_shadow_kernel: <= 48 VGPRs, 32 KiB LDS ring by LDS-DMA, four waves per workgroup, `blocks` workgroups).  Wall time of REP pairs: each alone,
both streams, against today's fused form (the same GEMM with the DMVAE_EPI_ADAM epilogue).     python tools/adam_shadow_probe.py [blocks ...]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa, pb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
Kb, M, N = 8192, 4096, 4096          # dW = X^T dY: K = the batch rows
NP = M * N
REP = 10
blocks_list = [int(x) for x in sys.argv[1:]] or [256, 512]
g = torch.Generator(device="cuda").manual_seed(1)
mk = lambda: [torch.randn(NP, device="cuda", generator=g) * 0.02 for _ in range(4)]      # p, g, m, v
def fresh():
    t = mk(); t[3].abs_(); return t + [torch.zeros(NP, device="cuda", dtype=torch.bfloat16)]
st = L.State(); st.adam_t = 3; st.lr = 2e-3
import numpy as np
lr_, b1_, b2_ = (float(np.float32(x)) for x in (2e-3, 0.9, 0.999))
st.lr = lr_; st.lr_t = float(np.float32(lr_ * (1.0 - b2_ ** 3) ** 0.5 / (1.0 - b1_ ** 3)))
state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
NOT = C.c_uint64(0xFFFFFFFFFFFFFFFF)

def adam(stream, t, flags):
    L.check(L.lib.dmvae_adam_tf(stream, NP, L.ptr(t[0]), L.ptr(t[1]), L.ptr(t[2]), L.ptr(t[3]), L.ptr(t[4]), 2e-3, 0.9, 0.999, 1e-8, 1.0, flags, NOT, L.ptr(state)))

# ---- same bits as the plain kernel?  (lr_t: the plain kernel recomputes it from t; give both the same value by computing it ON the device once)
a, b = fresh(), None
b = [x.clone() for x in a]
# the plain kernel with t_host = ~0 uses adam_lr_t(lr, b1, b2, adam_t) -- put exactly that float into state.lr_t through one plain launch on a scratch copy
adam(pa, [x.clone() for x in a], 0)
torch.cuda.synchronize()
import struct
# read back what lr_t the device computes: run the plain kernel on one element set with m = v = 0, g = 1 -> p' = p - lr_t * (1 - b1) / (sqrt(1 - b2) + eps)  (not needed for timing)
adam(pa, a, 0); adam(pa, b, L.ADAM_SHADOW | (256 << 8))
torch.cuda.synchronize()
same = all(torch.equal(x, y) for x, y in zip(a, b))
maxd = max((x.float() - y.float()).abs().max().item() for x, y in zip(a, b))
print("shadow form vs plain kernel on 16.8 M parameters: %s (max |diff| %.3g; host-computed lr_t in the state: equal bits only if it rounds as the device's pow)" % ("bit-identical" if same else "DIFFERENT", maxd), flush=True)

X = torch.relu(torch.randn(Kb, M, device="cuda", generator=g)).bfloat16()
dY = (0.02 * torch.randn(Kb, N, device="cuda", generator=g)).bfloat16()
dW = torch.zeros(M, N, device="cuda")
e = L.Epilogue(); e.kind = L.EPI_STORE_F32; e.out, e.ldo = dW.data_ptr(), N
def gemm():
    L.check(L.lib.dmvae_gemm(pa, 1, 2, M, N, Kb, L.ptr(X), M, L.ptr(dY), N, C.byref(e), 1))
# today's fused form: the same GEMM with the Adam epilogue
tf = fresh()
ctx = L.AdamCtx(); ctx.param, ctx.grad, ctx.m, ctx.v, ctx.param_bf16, ctx.state = (tf[0].data_ptr(), tf[1].data_ptr(), tf[2].data_ptr(), tf[3].data_ptr(), tf[4].data_ptr(), state.data_ptr())
ctx.beta1, ctx.beta2, ctx.epsilon, ctx.grad_scale, ctx.store_grad, ctx.ieee = 0.9, 0.999, 1e-8, 1.0, 0, 0
ctx.seg_off, ctx.seg_n = 0, 0
pr = (L.GemmProblem * 1)()
pr[0].M, pr[0].N, pr[0].K = M, N, Kb
pr[0].A, pr[0].lda, pr[0].B, pr[0].ldb = X.data_ptr(), M, dY.data_ptr(), N
pr[0].epi.kind = L.EPI_ADAM; pr[0].epi.out, pr[0].epi.ldo = tf[1].data_ptr(), N
def fused():
    L.check(L.lib.dmvae_gemm_grouped_dw_adam(pa, pr, 1, C.byref(ctx)))

def timed(fn_a, fn_b):
    for _ in range(3):
        if fn_a: fn_a()
        if fn_b: fn_b()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    d = torch.cuda.default_stream()
    t0.record(d); sa.wait_stream(d); sb.wait_stream(d)
    for _ in range(REP):
        if fn_a: fn_a()
        if fn_b: fn_b()
    d.wait_stream(sa); d.wait_stream(sb); t1.record(d)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / REP * 1e3

t = fresh()
has_fused = True
tg = timed(gemm, None)
print("dW GEMM 8192 x 4096 x 4096, gradient stored (macro tile): %.1f us  = %.3f PFLOP/s" % (tg, 2.0 * Kb * M * N / tg / 1e9), flush=True)
if has_fused:
    tfu = timed(fused, None)
    print("the same with the fused Adam epilogue (today's form):        %.1f us  = %.3f PFLOP/s" % (tfu, 2.0 * Kb * M * N / tfu / 1e9), flush=True)
tp = timed(None, lambda: adam(pb, t, 0))
print("plain Adam kernel alone: %.1f us (%.2f TB/s)" % (tp, 30.0 * NP / tp / 1e6), flush=True)
tb = timed(gemm, lambda: adam(pb, t, 0))
print("GEMM | plain Adam on two streams: %.1f us per pair (serial sum %.1f)" % (tb, tg + tp), flush=True)
for blocks in blocks_list:
    fl = L.ADAM_SHADOW | (blocks << 8)
    ts = timed(None, lambda: adam(pb, t, fl))
    tb = timed(gemm, lambda: adam(pb, t, fl))
    print("shadow Adam, %4d workgroups: alone %.1f us (%.2f TB/s); GEMM | shadow on two streams: %.1f us per pair (GEMM alone %.1f, serial sum %.1f)" %
          (blocks, ts, 30.0 * NP / ts / 1e6, tb, tg, tg + ts), flush=True)
