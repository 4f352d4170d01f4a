"""The cfg5-shaped weight-gradient phase two ways, same box, same buffers:
  fused   ONE merged macro-tile grid over NP problems of 4096 x 4096 at 8192 rows with the TF-Adam update in the tile epilogue (today's form);
  shadow  the same problems cut into chunks (e.g. 4,3,2,1): each chunk ONE merged grid that only stores the gradient (STORE_F32), and behind its
          event, on a second stream, the update of that chunk's parameters by the low-footprint Adam kernel (adam_shadow_kernel) -- running in the
          shadow of the NEXT chunk's grid; only the last chunk's update is exposed.
python tools/shadow_pipeline_probe.py [chunks, e.g. 4,3,2,1] [shadow workgroups]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
chunks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,3,2,1").split(",")]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NPR = sum(chunks)
Kb, M, N = 8192, 4096, 4096
PE = M * N
REP = 5
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa, pb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(2)
X = [torch.relu(torch.randn(Kb, M, device="cuda", generator=g)).bfloat16() for _ in range(NPR)]
dY = [(0.02 * torch.randn(Kb, N, device="cuda", generator=g)).bfloat16() for _ in range(NPR)]
n_el = NPR * PE
param = torch.randn(n_el, device="cuda", generator=g) * 0.02
arena = {k: torch.zeros(n_el, device="cuda") for k in ("grad", "m", "v")}
shadow = torch.zeros(n_el, device="cuda", dtype=torch.bfloat16)
lr_, b1_, b2_ = (float(np.float32(x)) for x in (2e-3, 0.9, 0.999))
st = L.State(); st.adam_t = 1; st.lr = lr_; st.lr_t = float(np.float32(lr_ * (1.0 - b2_) ** 0.5 / (1.0 - b1_)))
state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
NOT = C.c_uint64(0xFFFFFFFFFFFFFFFF)

def problems(lo, hi, kind, gbuf):
    pr = (L.GemmProblem * (hi - lo))()
    for j, i in enumerate(range(lo, hi)):
        pr[j].M, pr[j].N, pr[j].K = M, N, Kb
        pr[j].A, pr[j].lda, pr[j].B, pr[j].ldb = X[i].data_ptr(), M, dY[i].data_ptr(), N
        pr[j].epi.kind = kind; pr[j].epi.out, pr[j].epi.ldo = gbuf.data_ptr() + 4 * i * PE, N
    return pr

def ctx_of(p, m, v, s, gbuf):
    c = L.AdamCtx()
    c.param, c.grad, c.m, c.v, c.param_bf16, c.state = p.data_ptr(), gbuf.data_ptr(), m.data_ptr(), v.data_ptr(), s.data_ptr(), state.data_ptr()
    c.beta1, c.beta2, c.epsilon, c.grad_scale, c.store_grad, c.ieee, c.seg_off, c.seg_n = 0.9, 0.999, 1e-8, 1.0, 0, 0, 0, 0
    return c

def run_fused(p, m, v, s, gbuf):
    pr = problems(0, NPR, L.EPI_ADAM, gbuf); c = ctx_of(p, m, v, s, gbuf)
    L.check(L.lib.dmvae_gemm_grouped_dw_adam(pa, pr, NPR, C.byref(c)))

def run_shadow(p, m, v, s, gbuf, plain=False):
    lo = 0
    for n in chunks:
        pr = problems(lo, lo + n, L.EPI_STORE_F32, gbuf)
        L.check(L.lib.dmvae_gemm_grouped_dw(pa, 1, pr, n))
        ev = torch.cuda.Event(); ev.record(sa); sb.wait_event(ev)
        o, cnt = lo * PE, n * PE
        fl = 0 if plain else (L.ADAM_SHADOW | (blocks << 8))
        L.check(L.lib.dmvae_adam_tf(pb, cnt, p.data_ptr() + 4 * o, gbuf.data_ptr() + 4 * o, m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, s.data_ptr() + 2 * o,
                                    lr_, 0.9, 0.999, 1e-8, 1.0, fl, NOT, L.ptr(state)))
        lo += n
    sa.wait_stream(sb)

# same bits?
a = [param.clone(), arena["m"].clone(), arena["v"].clone(), shadow.clone(), torch.zeros(n_el, device="cuda")]
b = [param.clone(), arena["m"].clone(), arena["v"].clone(), shadow.clone(), torch.zeros(n_el, device="cuda")]
with torch.cuda.stream(sa):
    run_fused(*a); run_shadow(*b)
torch.cuda.synchronize()
print("fused grid vs chunked grids + shadow Adam, %d problems: %s" % (NPR, "bit-identical p / m / v / shadow" if all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) else "DIFFERENT"), flush=True)

def timed(fn):
    d = torch.cuda.default_stream()
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(d); sa.wait_stream(d)
    for _ in range(REP): fn()
    d.wait_stream(sa); d.wait_stream(sb); t1.record(d)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / REP
fl = 2.0 * Kb * M * N * NPR
for rnd in range(2):
    tf = timed(lambda: run_fused(*a))
    ts = timed(lambda: run_shadow(*b))
    tp = timed(lambda: run_shadow(*b, plain=True))
    print("round %d: fused grid %.3f ms (%.3f PFLOP/s) | chunks %s + shadow Adam (%d wg) %.3f ms (%.3f PFLOP/s) | chunks + plain Adam kernel %.3f ms" %
          (rnd, tf, fl / tf / 1e12, chunks, blocks, ts, fl / ts / 1e12, tp), flush=True)
only = (L.GemmProblem * 1)()
def gemm_only():
    lo = 0
    for n in chunks:
        pr = problems(lo, lo + n, L.EPI_STORE_F32, b[4]); L.check(L.lib.dmvae_gemm_grouped_dw(pa, 1, pr, n)); lo += n
tg = timed(gemm_only)
print("the chunked grids alone (gradient stored, no update): %.3f ms (%.3f PFLOP/s)" % (tg, fl / tg / 1e12), flush=True)
