"""The clock the chip holds INSIDE the K loop of the 256x256 macro-tile kernel (MI355X_MICROARCH.md, DVFS give-back item 6):
per workgroup (s_memtime at K-loop end - s_memtime at entry) / (s_memrealtime ... ) x 100 MHz, median over workgroups, stamped
after >= 2 s of back-to-back launches on operands shaped like the step's.  Needs the diagnostic build (tools/ablate.sh 6):
    DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl6.so python3 tools/clock256.py [B]
Prints, per layout, TFLOP/s by wall time, the held clock, and the K loop's MFMA rate against 2.5 PF AND against the peak at the
held clock (256 CUs x 4 SIMDs x 1024 flop/cycle: 16x16x32 bf16 = 16384 flop per 16 cycles)."""
import ctypes as C, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
hip = C.cdll.LoadLibrary("libamdhip64.so")
p = C.c_void_p()
L.check(L.lib.dmvae_debug_anatomy256(C.byref(p)), "dmvae_debug_anatomy256 (needs the DMVAE_ABLATE=6 build)")
L.check(L.lib.dmvae_debug_set_knob(6, 2))
shapes = [("fwd 4096->4096", 0, B, 4096, 4096, L.EPI_BIAS_RELU), ("dX 4096<-4096", 1, B, 4096, 4096, L.EPI_RELU_MASK),
          ("dW 4096x4096", 2, 4096, 4096, B, L.EPI_STORE_F32)]
out = []
for name, lay, M, N, K, epi in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    act = lambda r, c: (torch.randn(r, c, device="cuda", generator=g).clamp_(min=0) * 0.5).bfloat16()      # post-ReLU activations
    wgt = lambda r, c: (torch.randn(r, c, device="cuda", generator=g) * 0.02).bfloat16()
    if lay == 0: A, Bm, lda, ldb = act(M, K), wgt(K, N), K, N
    elif lay == 1: A, Bm, lda, ldb = wgt(M, K) * 0.1, wgt(N, K), K, K
    else: A, Bm, lda, ldb = act(K, M), wgt(K, N) * 0.1, M, N
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); outf = torch.zeros(M, N, device="cuda")
    bias = torch.zeros(N, device="cuda"); Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    e = L.Epilogue(); e.kind = epi
    e.out = (outf if epi == L.EPI_STORE_F32 else outb).data_ptr(); e.ldo = N
    e.bias = bias.data_ptr(); e.aux0 = Y.data_ptr(); e.ld0 = N
    call = lambda: L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), lda, L.ptr(Bm), ldb, C.byref(e), 1)
    L.check(call())
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.2:          # >= 2 s of back-to-back launches: the clock has settled
        for _ in range(50): call()
        torch.cuda.synchronize(); n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    buf = torch.empty(4096 * 8, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    hip.hipMemcpy(C.c_void_p(buf.data_ptr()), p, C.c_size_t(4096 * 64), 3)
    s = buf.cpu().numpy().reshape(4096, 8)[: (M // 256) * (N // 256)]
    rt = (s[:, 1] - s[:, 0]).astype(np.float64)              # 100 MHz ticks in the K loop
    cyc = (s[:, 5] - s[:, 4]).astype(np.float64)
    ok = rt > 0
    ghz = np.median(cyc[ok] / rt[ok]) * 0.1
    kloop_us = np.median(rt[ok]) / 100.0
    flop_tile = 2.0 * 256 * 256 * K
    tf_loop = flop_tile / (kloop_us * 1e-6) * 256 / 1e12      # all 256 CUs in their K loops
    peak_clk = 256 * 4 * 1024 * ghz * 1e9 / 1e12
    line = ("%-16s %5dx%5dx%5d  launch %7.1f us = %6.1f TF | K loop %6.1f us/tile = %6.1f TF chip-wide = %.3f of 2.5 PF | in-kernel clock %.3f GHz "
            "(p10 %.3f p90 %.3f) -> peak at that clock %6.1f TF, K loop = %.3f of it" %
            (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, kloop_us, tf_loop, tf_loop / 2500.0, ghz,
             np.percentile(cyc[ok] / rt[ok], 10) * 0.1, np.percentile(cyc[ok] / rt[ok], 90) * 0.1, peak_clk, tf_loop / peak_clk))
    print(line, flush=True)
