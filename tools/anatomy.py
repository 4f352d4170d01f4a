"""Where the time of ONE small bf16 GEMM launch goes (needs the DMVAE_ABLATE=6 build: tools/ablate.sh 6):
per-workgroup stamps {entry, first K tile landed, K loop done, epilogue issued, stores acknowledged}.
DMVAE_HIP_LIB=$PWD/deep-mixture-vae_amd/build/libdmvae_hip_abl6.so python3 tools/anatomy.py [M N K layout]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(4096, 512, 512, 0), (4096, 512, 832, 0), (4096, 4096, 512, 0), (4096, 512, 512, 1), (4096, 512, 2048, 1), (4096, 64, 2048, 1)]
if len(sys.argv) == 5: shapes = [tuple(int(v) for v in sys.argv[1:5])]
for kv in os.environ.get("DMVAE_KNOBS", "").split(","):          # e.g. DMVAE_KNOBS=3=1,1=0
    if kv: L.check(L.lib.dmvae_debug_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1])))
p = C.c_void_p(); L.check(L.lib.dmvae_debug_anatomy(C.byref(p)))
hip = C.cdll.LoadLibrary("libamdhip64.so")
for M, N, K, lay in shapes:
    A = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
    B = (0.02 * torch.randn(K, N, device="cuda") if lay == 0 else 0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); Y = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda"); junk = torch.empty(64 << 20, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU if lay == 0 else L.EPI_RELU_MASK
    e.out, e.ldo, e.bias, e.aux0, e.ld0 = out.data_ptr(), N, bias.data_ptr(), Y.data_ptr(), N
    for mode in os.environ.get("ANATOMY_MODES", "hot").split(","):      # hot | mall (48 MB written between launches: out of the L2s, still in the Infinity Cache) | cold (256 MB)
        for _ in range(3):
            if mode == "cold": junk.fill_(1.0)          # evict L2 / Infinity Cache between launches
            if mode == "mall": junk[:12 << 20].fill_(1.0)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            L.check(L.lib.dmvae_gemm(st, 1, lay, M, N, K, L.ptr(A), K, L.ptr(B), N if lay == 0 else K, C.byref(e), 1))
            t1.record(); torch.cuda.synchronize()
        buf = torch.empty(2048 * 8, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(buf.data_ptr()), p, C.c_size_t(2048 * 64), 3)
        s = buf.cpu().numpy().reshape(2048, 8)
        s = s[s[:, 4] != 0]
        t_first = s[:, 0].min()
        us = lambda col: (s[:, col] - t_first) / 100.0
        seg = lambda a, b: (s[:, b] - s[:, a]) / 100.0
        q = lambda v: "%5.2f/%5.2f/%5.2f" % (np.min(v), np.median(v), np.max(v))
        print("%s M %d N %d K %d lay %d  WGs %d  event %.2f us | entry %s | first tile +%s | K loop +%s | epilogue +%s | store ack +%s | last end %.2f"
              % (mode, M, N, K, lay, len(s), t0.elapsed_time(t1) * 1e3, q(us(0)), q(seg(0, 1)), q(seg(1, 2)), q(seg(2, 3)), q(seg(3, 4)), us(4).max()), flush=True)
        hip.hipMemset(p, 0, C.c_size_t(2048 * 64))
