"""What the reconstruction epilogue of the output layer's GEMM costs at cfg2 (4096 x 832 x 512, 128 x 64 tiles): the plain BIAS_RELU
epilogue against BIAS_RECON with its f32 targets (a) contiguous, binary cross-entropy, (b) contiguous, squared error (no exp / log),
(c) fetched through the permutation from the dataset (the step path).  Graph replay of 20 launches per sample, interleaved."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
side = torch.cuda.Stream()
M, N, K, I, ROWS = 4096, 832, 512, 784, 60000
A = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
Bm = (0.02 * torch.randn(K, N, device="cuda")).bfloat16(); bias = torch.zeros(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
xt = torch.rand(M, N, device="cuda"); data = torch.rand(ROWS, I, device="cuda"); perm = torch.randperm(ROWS, device="cuda").int()
parts = torch.zeros(L.lib.dmvae_gemm_partials(1, M, N), device="cuda")
def epi(kind):
    e = L.Epilogue(); e.out, e.ldo, e.bias = out.data_ptr(), N, bias.data_ptr()
    if kind == "relu": e.kind = L.EPI_BIAS_RELU; return e
    e.kind = L.EPI_BIAS_RECON; e.m_valid, e.n_valid, e.scale, e.partials = M, I, 1.0 / M, parts.data_ptr()
    if kind == "bce": e.recon_kind = 0; e.aux0, e.ld0 = xt.data_ptr(), N
    if kind == "mse": e.recon_kind = 1; e.aux0, e.ld0 = xt.data_ptr(), N
    if kind == "bce, targets through the permutation": e.recon_kind = 0x100; e.aux0, e.ld0, e.aux1, e.ld1, e.ld2 = data.data_ptr(), I, perm.data_ptr(), 0, ROWS
    return e
cases = {k: epi(k) for k in ("relu", "bce", "mse")}      # (the permutation form is plan-internal: dmvae_gemm refuses it; in the step that launch takes 16.1 us)
def launch(k, st): L.check(L.lib.dmvae_gemm(st, 1, 0, M, N, K, L.ptr(A), K, L.ptr(Bm), N, C.byref(cases[k]), 1))
graphs = {}
with torch.cuda.stream(side):
    st = C.c_void_p(side.cuda_stream)
    for k in cases:
        for _ in range(2): launch(k, st)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20): launch(k, st)
        graphs[k] = g
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
for k, v in ts.items(): print("%-42s %6.2f us median  %6.2f min" % (k, sorted(v)[len(v) // 2], min(v)))
