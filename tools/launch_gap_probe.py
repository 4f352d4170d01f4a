"""What a small GEMM launch costs inside a replayed HIP graph, and why it costs more inside the training step (8.7 us for the 4096 x 512 x 512
forward layer in the step's per-kernel table) than back to back on hot data.  Graphs of 192 dependent launches each:
  empty        near-empty kernel (dmvae_debug_spin 0 us): the floor of a dependent launch
  same         one GEMM, same operands every launch (code, kernel arguments, weights and activations hot)
  chain        the GEMM reads what the previous launch wrote (ping-pong A <-> out, square 512-wide layer): the step's data dependence
  rotate       16 operand sets in turn (136 MB: cold in the L2s, warm in the Infinity Cache)
  two kernels  forward GEMM and dX GEMM (another instantiation) alternating, hot operands: instruction-cache / argument effects
  chain+two    both"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
side = torch.cuda.Stream()
M, N, K = 4096, 512, 512
NS = 16
A = [torch.relu(torch.randn(M, K, device="cuda")).bfloat16() for _ in range(NS)]
W = [(0.02 * torch.randn(K, N, device="cuda")).bfloat16() for _ in range(NS)]
O = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
bias = torch.zeros(N, device="cuda"); gate = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
def fwd(st, a, w, o):
    e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU; e.out, e.ldo, e.bias = o.data_ptr(), N, bias.data_ptr()
    L.check(L.lib.dmvae_gemm(st, 1, 0, M, N, K, L.ptr(a), K, L.ptr(w), N, C.byref(e), 1))
def dx(st, a, w, o):
    e = L.Epilogue(); e.kind = L.EPI_RELU_MASK; e.out, e.ldo, e.aux0, e.ld0 = o.data_ptr(), N, gate.data_ptr(), N
    L.check(L.lib.dmvae_gemm(st, 1, 1, M, N, K, L.ptr(a), K, L.ptr(w), K, C.byref(e), 1))
cases = {
    "empty": lambda st, i: L.check(L.lib.dmvae_debug_spin(st, 0)),
    "same": lambda st, i: fwd(st, A[0], W[0], O[0]),
    "chain": lambda st, i: fwd(st, (A[0], O[0])[i & 1], W[0], (O[0], A[0])[i & 1]),
    "rotate": lambda st, i: fwd(st, A[i % NS], W[i % NS], O[i % NS]),
    "two kernels": lambda st, i: (fwd, dx)[i & 1](st, A[0], W[0], O[0]),
    "chain+two": lambda st, i: (fwd, dx)[i & 1](st, (A[0], O[0])[i & 1], W[0], (O[0], A[0])[i & 1]),
    "chain+two+rotating weights": lambda st, i: (fwd, dx)[i & 1](st, (A[0], O[0])[i & 1], W[i % NS], (O[0], A[0])[i & 1]),
}
NL = 192
graphs = {}
with torch.cuda.stream(side):
    st = C.c_void_p(side.cuda_stream)
    for name, fn in cases.items():
        for i in range(4): fn(st, i)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(NL): fn(st, i)
        graphs[name] = g
ts = {k: [] for k in graphs}
for r in range(9):
    for k, g in graphs.items():
        with torch.cuda.stream(side):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(side); g.replay(); t1.record(side); side.synchronize()
        if r >= 3: ts[k].append(t0.elapsed_time(t1) / NL * 1e3)
for k, v in ts.items(): print("%-28s %.2f us per launch (median of 6; min %.2f)" % (k, sorted(v)[3], min(v)))
