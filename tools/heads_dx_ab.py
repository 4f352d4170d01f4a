"""The dX of the two head layers alone, grouped tiles (knob 13 = 0) against the streaming kernel (csrc/heads_dx.hip, knob 13 = 1):
interleaved rounds in ONE process (graph replay of 20 launches per sample), per BASELINE config shape, beside a plain copy of the same
bytes (torch) as the rate a streaming kernel reaches on this box.

    python tools/heads_dx_ab.py [cfg2 cfg3 cfg4]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L

SHAPES = {"cfg2": (4096, 128, 64), "cfg3": (16384, 256, 64), "cfg4": (8192, 512, 64)}
H = 2048
torch.cuda.set_device(0)
side = torch.cuda.Stream()


def graph_of(fn, n=20):
    with torch.cuda.stream(side):
        st = C.c_void_p(side.cuda_stream)
        for _ in range(2): fn(st)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(n): fn(st)
    return gr


def sample(gr, n=20):
    with torch.cuda.stream(side):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(side); gr.replay(); t1.record(side); side.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


for name in (sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]):
    B, D2, Kp = SHAPES[name]
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    dmv = torch.randn(B, D2, device="cuda", generator=g).bfloat16()
    dlg = torch.randn(B, Kp, device="cuda", generator=g).bfloat16()
    Wmv = torch.randn(H, D2, device="cuda", generator=g).bfloat16()
    Wl = torch.randn(H, Kp, device="cuda", generator=g).bfloat16()
    hzc = torch.relu(torch.randn(B, 2 * H, device="cuda", generator=g)).bfloat16()
    out = torch.zeros(B, 2 * H, device="cuda", dtype=torch.bfloat16)
    probs = (L.GemmProblem * 2)()
    for i, (dY, Kd, W) in enumerate(((dmv, D2, Wmv), (dlg, Kp, Wl))):
        p = probs[i]; p.M, p.N, p.K = B, H, Kd
        p.A, p.lda, p.B, p.ldb = dY.data_ptr(), Kd, W.data_ptr(), Kd
        p.epi.kind = L.EPI_RELU_MASK; p.epi.out = out.data_ptr() + i * H * 2; p.epi.ldo = 2 * H
        p.epi.aux0 = hzc.data_ptr() + i * H * 2; p.epi.ld0 = 2 * H
    launch = lambda st: L.check(L.lib.dmvae_gemm_grouped(st, 1, L.GEMM_DX, probs, 2))
    graphs = {}
    for knob in (0, 1):           # the knob is read at enqueue time: each graph keeps the kernel it was captured with
        L.check(L.lib.dmvae_debug_set_knob(13, knob))
        graphs[knob] = graph_of(launch)
    L.check(L.lib.dmvae_debug_set_knob(13, 1))
    graphs["copy"] = graph_of(lambda st: out.copy_(hzc))       # the same bytes read and written once each (torch.where(..., out=) ran at 1.9 TB/s: not a ceiling)
    ts = {k: [] for k in graphs}
    for _ in range(3):
        for k in graphs: sample(graphs[k])
    for _ in range(9):
        for k in graphs: ts[k].append(sample(graphs[k]))
    mb = 2 * 2 * B * 2 * H / 1e6           # mask read + output written, both problems
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print("%s  B=%d K=(%d,%d): grouped tiles %.2f us (%.2f TB/s) | streaming %.2f us (%.2f TB/s) | plain copy of the same bytes %.2f us (%.2f TB/s)   [min: %.2f / %.2f / %.2f]"
          % (name, B, D2, Kp, med[0], mb / med[0], med[1], mb / med[1], med["copy"], mb / med["copy"], min(ts[0]), min(ts[1]), min(ts["copy"])))
