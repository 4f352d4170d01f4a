"""two consecutive narrow dense layers (enc0 -> enc1: 784 (832) -> 512 -> 512, bias + ReLU) as ONE row-strip launch (csrc/strip_fwd2.hip) against the
two tiled launches it would replace (dmvae_gemm, DMVAE_EPI_BIAS_RELU): same bits?  and the time of each form as interleaved HIP-graph replays on shared buffers.
python tools/strip2_probe.py [rows ...]        (default 4096 8192 16384)"""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
K0, N = 832, 512
g = torch.Generator(device="cuda").manual_seed(5)
W0 = (torch.randn(K0, N, device="cuda", generator=g) * 0.05).bfloat16(); W1 = (torch.randn(N, N, device="cuda", generator=g) * 0.06).bfloat16()
b0 = torch.randn(N, device="cuda", generator=g) * 0.1; b1 = torch.randn(N, device="cuda", generator=g) * 0.1
side = torch.cuda.Stream()
ps = C.c_void_p(side.cuda_stream)

def two(X, Y1, Y2, B):
    e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU; e.out, e.ldo, e.bias = Y1.data_ptr(), N, b0.data_ptr()
    L.check(L.lib.dmvae_gemm(ps, 1, 0, B, N, K0, L.ptr(X), K0, L.ptr(W0), N, C.byref(e), 1))
    e2 = L.Epilogue(); e2.kind = L.EPI_BIAS_RELU; e2.out, e2.ldo, e2.bias = Y2.data_ptr(), N, b1.data_ptr()
    L.check(L.lib.dmvae_gemm(ps, 1, 0, B, N, N, L.ptr(Y1), N, L.ptr(W1), N, C.byref(e2), 1))

def strip(X, Y1, Y2, B):
    L.check(L.lib.dmvae_debug_strip_fwd2(ps, B, K0, L.ptr(X), K0, L.ptr(W0), N, L.ptr(b0), L.ptr(W1), N, L.ptr(b1), L.ptr(Y1), N, L.ptr(Y2), N))

for B in [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384]:
    X = torch.relu(torch.randn(B, K0, device="cuda", generator=g)).bfloat16()
    X[:, 784:] = 0
    outs = {}
    for name, fn in (("two", two), ("strip", strip)):
        Y1 = torch.full((B, N), 7.0, device="cuda", dtype=torch.bfloat16); Y2 = torch.full((B, N), 7.0, device="cuda", dtype=torch.bfloat16)
        with torch.cuda.stream(side): fn(X, Y1, Y2, B)
        torch.cuda.synchronize()
        outs[name] = (Y1, Y2)
    ref1 = torch.relu(X.double() @ W0.double() + b0.double())
    err = (outs["strip"][0].double() - ref1).abs().max().item()
    same = torch.equal(outs["two"][0], outs["strip"][0]) and torch.equal(outs["two"][1], outs["strip"][1])
    print("%5d rows: strip vs two launches: %s (Y1 max |err| vs float64 %.3g, differing Y1 %d / Y2 %d)" % (B, "bit-identical" if same else "DIFFERENT", err,
          int((outs["two"][0] != outs["strip"][0]).sum()), int((outs["two"][1] != outs["strip"][1]).sum())), flush=True)
    # timing: one graph of 20 launches (pairs) per form, on the same buffers, interleaved
    graphs = {}
    Y1, Y2 = outs["two"]
    for name, fn in (("two", two), ("strip", strip)):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20): fn(X, Y1, Y2, B)
        graphs[name] = gr
    res = {k: [] for k in graphs}
    for rnd in range(7):
        for name in (("two", "strip") if rnd % 2 == 0 else ("strip", "two")):
            graphs[name].replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): graphs[name].replay()
            torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 200 * 1e6)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print("%5d rows: two tiled launches %.2f us | one row-strip launch %.2f us  (%+.1f %%)" % (B, med["two"], med["strip"], 100 * (med["strip"] / med["two"] - 1)), flush=True)
