"""XCD-sliced layer chain (dmvae_debug_chain) against the same layers as separate launches: L square bias + ReLU layers
[4096 x N] x [N x N], both forms HIP-graph replayed; results compared bit for bit.  python tools/chain_probe.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
s = torch.cuda.Stream()
st = C.c_void_p(s.cuda_stream)
M = 4096
sync = torch.zeros(1024, dtype=torch.int32, device="cuda")
err = torch.zeros(4, dtype=torch.int32, device="cuda")

def timed(fn, reps):
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(s)
        for _ in range(10): g.replay()
        t1.record(s); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (10 * reps) * 1e3

for N in (512, 1024):
    for NL in (2, 4, 8):
        torch.manual_seed(N + NL)
        x0 = torch.relu(torch.randn(M, N, device="cuda")).bfloat16()
        Ws = [(torch.randn(N, N, device="cuda") * (2.0 / N) ** 0.5).bfloat16() for _ in range(NL)]
        bs = [0.01 * torch.randn(N, device="cuda") for _ in range(NL)]
        a = [x0.clone(), torch.zeros_like(x0)]
        Wp = (C.c_void_p * NL)(*[w.data_ptr() for w in Ws]); bp = (C.c_void_p * NL)(*[b.data_ptr() for b in bs])

        def separate():
            for l in range(NL):
                e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU
                e.out, e.ldo, e.bias = a[(l + 1) % 2].data_ptr(), N, bs[l].data_ptr()
                L.check(L.lib.dmvae_gemm(st, 1, 0, M, N, N, L.ptr(a[l % 2]), N, L.ptr(Ws[l]), N, C.byref(e), 1))
        with torch.cuda.stream(s):
            a[0].copy_(x0); separate(); torch.cuda.synchronize()
        ref = a[NL % 2].clone()
        line = "N %4d layers %d : separate %6.2f us/layer" % (N, NL, timed(separate, 5) / NL)
        for variant in (0, 1):
            def chain():
                L.check(L.lib.dmvae_debug_chain(st, variant, NL, M, N, L.ptr(a[0]), L.ptr(a[1]), Wp, bp, L.ptr(sync), L.ptr(err)))
            bad = 0
            for trial in range(3):          # fresh inputs each trial: a stale L1 line would show
                with torch.cuda.stream(s):
                    a[0].copy_(x0 * (1.0 + 0.25 * trial)); a[1].zero_(); separate(); torch.cuda.synchronize(); want = a[NL % 2].clone()
                    a[0].copy_(x0 * (1.0 + 0.25 * trial)); a[1].zero_(); chain(); torch.cuda.synchronize()
                bad += int((a[NL % 2] != want).sum().item())
            line += " | chain v%d %6.2f us/layer  mismatches %d err %d" % (variant, timed(chain, 5) / NL, bad, int(err[0].item()))
        print(line, flush=True)
