"""The output layer's forward GEMM at cfg2 (4096 x 832 x 512) per tiling: 128 x 64 tiles (416 workgroups: two per CU on 160 CUs), against
128 x 128 tiles on a 896-wide problem (224 workgroups, one per CU) and on the first 768 columns alone (192 workgroups).  BIAS_RELU epilogue
(the reconstruction epilogue comes on top in either tiling).  Graph replay of 20 launches per sample, interleaved."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
side = torch.cuda.Stream()
M, K = 4096, 512
A = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
cases = {}
for name, N, tile in (("832 cols, 128x64", 832, (128, 64)), ("896 cols, 128x128", 896, (128, 128)), ("768 cols, 128x128", 768, (128, 128)), ("832 cols, 64x64", 832, (64, 64))):
    Bm = (0.02 * torch.randn(K, N, device="cuda")).bfloat16(); out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); bias = torch.zeros(N, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU; e.out, e.ldo, e.bias = out.data_ptr(), N, bias.data_ptr()
    cases[name] = (N, tile, Bm, out, bias, e)
def launch(name, st):
    N, tile, Bm, out, bias, e = cases[name]
    L.check(L.lib.dmvae_gemm(st, 1, 0, M, N, K, L.ptr(A), K, L.ptr(Bm), N, C.byref(e), 1))
graphs = {}
with torch.cuda.stream(side):
    st = C.c_void_p(side.cuda_stream)
    for name in cases:
        L.check(L.lib.dmvae_debug_set_tile(*cases[name][1]))
        for _ in range(2): launch(name, st)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20): launch(name, st)
        graphs[name] = g
    L.check(L.lib.dmvae_debug_set_tile(0, 0))
def sample(g):
    with torch.cuda.stream(side):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(side); g.replay(); t1.record(side); side.synchronize()
    return t0.elapsed_time(t1) / 20 * 1e3
ts = {k: [] for k in graphs}
for r in range(12):
    for k in graphs:
        v = sample(graphs[k])
        if r >= 3: ts[k].append(v)
for k, v in ts.items(): print("%-20s %6.2f us median  %6.2f min" % (k, sorted(v)[len(v) // 2], min(v)))
