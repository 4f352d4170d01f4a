"""Phase stamps of block 0 of the fused heads + latent launch (csrc/heads_latent.hip) at the metric's shape -- needs the abl7 measurement build
(tools/ablate.sh 7; DMVAE_HIP_LIB=deep-mixture-vae_amd/build/libdmvae_hip_abl7.so).  Stamps (100 MHz): 0 entry, 1 first-row / table loads requested,
13 K loop done, 12 f32 tile parked + written (mid() returned), 2 prior tables staged + per-row prologue (softmax, KL_C) done,
3 phase 1a, 4 phase 1b, 5 phase 2, 6 end.      python tools/hl_phases.py [rows]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D, K, Hp, Dp, Kp = 64, 10, 2048, 64, 64
g = torch.Generator(device="cuda").manual_seed(3)
hzc = torch.relu(torch.randn(B, 2 * Hp, device="cuda", generator=g)).bfloat16()
Wmv = (torch.randn(Hp, 2 * Dp, device="cuda", generator=g) * 0.03).bfloat16(); Wlg = (torch.randn(Hp, Kp, device="cuda", generator=g) * 0.03).bfloat16()
bmv, blg = torch.zeros(2 * Dp, device="cuda"), torch.zeros(Kp, device="cuda")
mv, lg = torch.zeros(B, 2 * Dp, device="cuda"), torch.zeros(B, Kp, device="cuda")
Z = torch.zeros(B, Dp, dtype=torch.bfloat16, device="cuda"); Zf = torch.zeros(B, Dp, device="cuda"); w = torch.zeros(B, Kp, device="cuda")
gmu, glv, clv = (torch.zeros(B, Dp, device="cuda") for _ in range(3)); dlg = torch.zeros(B, Kp, dtype=torch.bfloat16, device="cuda")
nblk = B // 16
dpri = torch.zeros(nblk, 2 * K * D, device="cuda"); lp = torch.zeros(2 * nblk + 64, device="cuda")
pm, plv = torch.randn(K, D, device="cuda", generator=g), torch.zeros(K, D, device="cuda")
a = L.LatentArgs(); a.B, a.B_pad, a.D, a.K, a.mode, a.act_dtype = B, B, D, K, 0, 1
a.kl_ratio, a.temperature, a.inv_B, a.seed, a.noise_step = 1.0, 1.0, 1.0 / B, 1, 1
a.mean, a.ld_mean, a.log_var, a.ld_log_var, a.logits, a.ld_logits = mv.data_ptr(), 2 * Dp, mv.data_ptr() + 4 * Dp, 2 * Dp, lg.data_ptr(), Kp
a.prior_means, a.prior_log_vars = pm.data_ptr(), plv.data_ptr()
a.Z_act, a.ld_Z, a.Z_f32, a.ld_Zf, a.weights, a.ld_w = Z.data_ptr(), Dp, Zf.data_ptr(), Dp, w.data_ptr(), Kp
a.gmu, a.glv, a.clv, a.ld_g, a.dlogits_act, a.ld_dl = gmu.data_ptr(), glv.data_ptr(), clv.data_ptr(), Dp, dlg.data_ptr(), Kp
a.dprior_partials, a.loss_partials = dpri.data_ptr(), lp.data_ptr()
h = L.HeadsArgs(); h.hz, h.lda, h.Hp, h.Dp, h.Kp = hzc.data_ptr(), 2 * Hp, Hp, Dp, Kp
h.W_mv, h.ld_mv, h.W_lg, h.ld_lg, h.b_mv, h.b_lg = Wmv.data_ptr(), 2 * Dp, Wlg.data_ptr(), Kp, bmv.data_ptr(), blg.data_ptr()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rows = []
for it in range(30):
    L.check(L.lib.dmvae_heads_latent_fwd(s, C.byref(h), C.byref(a)))
    torch.cuda.synchronize()
    st = lp[2 * nblk:].view(torch.int64).cpu().numpy()[:16].astype(np.int64)
    if it >= 10: rows.append(st.copy())
st = np.median(np.array(rows), axis=0)
order = [(0, "entry"), (1, "ring slots + stage's loads requested"), (13, "K loop done"), (12, "tile parked + written"), (2, "tables staged, row prologue (softmax, KL_C)"),
         (3, "phase 1a"), (4, "phase 1b"), (5, "phase 2"), (6, "end")]
prev = st[0]
for i, name in order:
    print("%-34s +%6.2f us  (at %6.2f)" % (name, (st[i] - prev) / 100.0, (st[i] - st[0]) / 100.0)); prev = st[i]
