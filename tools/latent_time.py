"""Times dmvae_latent_fwd alone (HIP events, 50 launches) for several batch sizes / noise sources."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib
from dmvae_hip._lib import lib, check
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run(B, D, K, host_eps, f32z, mode=0):
    ldD, ldK = (D + 63) // 64 * 64, (K + 63) // 64 * 64
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    md, lvd, lgd = torch.randn(B, 2 * ldD, device=dev), None, torch.randn(B, ldK, device=dev)
    eps = torch.randn(B, D, device=dev)
    pm, plv = torch.randn(K, D, device=dev), z(K, D)
    Zb = torch.zeros(B, ldD, dtype=torch.bfloat16, device=dev); Zf = z(B, ldD); w = z(B, ldK)
    gmu, glv, clv = z(B, ldD), z(B, ldD), z(B, ldD); dlg = torch.zeros(B, ldK, dtype=torch.bfloat16, device=dev)
    nblk = lib.dmvae_latent_nblocks(B, D, K)
    dpri, lp = z(nblk, 2 * K * D), z(nblk + 32, 2)
    a = _lib.LatentArgs()
    a.B, a.B_pad, a.D, a.K, a.mode, a.act_dtype = B, B, D, K, mode, _lib.BF16
    a.kl_ratio, a.temperature, a.inv_B, a.seed = 1.0, 0.5, 1.0 / B, 7
    a.mean, a.ld_mean = md.data_ptr(), 2 * ldD
    a.log_var, a.ld_log_var = md.data_ptr() + 4 * ldD, 2 * ldD
    a.logits, a.ld_logits = lgd.data_ptr(), ldK
    if host_eps: a.eps, a.ld_eps = eps.data_ptr(), D
    a.prior_means, a.prior_log_vars = pm.data_ptr(), plv.data_ptr()
    a.Z_act, a.ld_Z = Zb.data_ptr(), ldD
    if f32z: a.Z_f32, a.ld_Zf = Zf.data_ptr(), ldD
    a.weights, a.ld_w = w.data_ptr(), ldK
    a.gmu, a.glv, a.clv, a.ld_g = gmu.data_ptr(), glv.data_ptr(), clv.data_ptr(), ldD
    a.dlogits_act, a.ld_dl = dlg.data_ptr(), ldK
    a.dprior_partials, a.loss_partials = dpri.data_ptr(), lp.data_ptr()
    for _ in range(5): check(lib.dmvae_latent_fwd(st, C.byref(a)))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50): lib.dmvae_latent_fwd(st, C.byref(a))
    t1.record(); torch.cuda.synchronize()
    print("B=%5d D=%3d K=%2d host_eps=%d f32z=%d blocks=%d : %.2f us" % (B, D, K, host_eps, f32z, nblk, t0.elapsed_time(t1) / 50 * 1e3), flush=True)
    if "abl7" in os.environ.get("DMVAE_HIP_LIB", ""):
        t = lp.view(-1)[2 * nblk: 2 * nblk + 24].view(torch.int64).cpu().numpy()
        print("     phase 1a of wave 0 (noise | columns: exp, z, 3 stores | k loop | gradient stores | wait for the other waves):", [round((t[j] - t[i]) / 100.0, 2) for i, j in ((2, 7), (7, 8), (8, 9), (9, 10), (10, 3))])
        print("     block 0 phases us (early loads | prologue | phase 1a | 1b | 2 | finalize):", [round((t[i + 1] - t[i]) / 100.0, 2) for i in range(6)])
for B in (64, 256, 1024, 4096, 16384):
    run(B, 64, 10, 0, 1)
run(4096, 64, 10, 1, 1); run(4096, 64, 10, 1, 0); run(4096, 16, 10, 0, 1); run(4096, 64, 1, 0, 1); run(8192, 256, 50, 0, 1)
