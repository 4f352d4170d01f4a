"""GPU parity tests of the fused launch  heads forward + latent stage  (csrc/heads_latent.hip, dmvae_heads_latent_fwd):
bit-identical to the two launches it replaces (dmvae_gemm_grouped with DMVAE_EPI_BIAS_F32, then dmvae_latent_fwd), and green against the
reference's own priors.py outputs (tests/golden/priors_golden.npz) THROUGH the fused entry."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import dmvae_hip
    from dmvae_hip import _lib
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return _lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


def run_heads_latent(L, hzc, Wmv, Wlg, bmv, blg, eps, gumbel, pm, plv, mode, tau, r, B, D, K, fused, seed=1234, step=7, kslices=1):
    """hzc: torch bf16 [B_pad][2 Hp]; Wmv bf16 [Hp][2 Dp]; Wlg bf16 [Hp][Kp]; returns every output of the stage as torch tensors"""
    B_pad, Hp = hzc.shape[0], hzc.shape[1] // 2
    Dp, Kp = Wmv.shape[1] // 2, Wlg.shape[1]
    mv = torch.full((B_pad, 2 * Dp), 7.0, device="cuda")
    lg = torch.full((B_pad, Kp), 7.0, device="cuda")
    Z = torch.full((B_pad, Dp), 9.0, dtype=torch.bfloat16, device="cuda")
    Zf = torch.zeros((B_pad, Dp), device="cuda")
    w = torch.zeros((B_pad, Kp), device="cuda")
    gmu, glv, clv = (torch.full((B_pad, Dp), 9.0, device="cuda") for _ in range(3))
    dlg = torch.full((B_pad, Kp), 9.0, dtype=torch.bfloat16, device="cuda")
    nblk = B_pad // 16
    assert L.lib.dmvae_latent_nblocks(B_pad, D, K) == nblk          # the unfused kernel's geometry at these sizes: 16 rows per block too
    dpri = torch.zeros((nblk, 2 * K * D), device="cuda")
    lp = torch.zeros((nblk, 2), device="cuda")
    a = L.LatentArgs()
    a.B, a.B_pad, a.D, a.K, a.mode, a.act_dtype = B, B_pad, D, K, mode, 1
    a.kl_ratio, a.temperature, a.inv_B, a.seed, a.noise_step = r, tau, 1.0 / B, seed, step
    a.mean, a.ld_mean = mv.data_ptr(), 2 * Dp
    a.log_var, a.ld_log_var = mv.data_ptr() + 4 * Dp, 2 * Dp
    a.logits, a.ld_logits = lg.data_ptr(), Kp
    keep = []
    if eps is not None:
        e = dev(eps); keep.append(e); a.eps, a.ld_eps = e.data_ptr(), D
    if gumbel is not None:
        gq = dev(gumbel); keep.append(gq); a.gumbel, a.ld_gumbel = gq.data_ptr(), K
    pmd, plvd = dev(pm), dev(plv)
    a.prior_means, a.prior_log_vars = pmd.data_ptr(), plvd.data_ptr()
    a.Z_act, a.ld_Z = Z.data_ptr(), Dp
    a.Z_f32, a.ld_Zf = Zf.data_ptr(), Dp
    a.weights, a.ld_w = w.data_ptr(), Kp
    a.gmu, a.glv, a.clv, a.ld_g = gmu.data_ptr(), glv.data_ptr(), clv.data_ptr(), Dp
    a.dlogits_act, a.ld_dl = dlg.data_ptr(), Kp
    a.dprior_partials, a.loss_partials = dpri.data_ptr(), lp.data_ptr()
    if fused:
        assert L.lib.dmvae_heads_latent_ok(B_pad, D, K, Dp, Kp, Hp, mode) == 1
        h = L.HeadsArgs()
        h.hz, h.lda, h.Hp, h.Dp, h.Kp = hzc.data_ptr(), 2 * Hp, Hp, Dp, Kp
        h.W_mv, h.ld_mv, h.W_lg, h.ld_lg = Wmv.data_ptr(), 2 * Dp, Wlg.data_ptr(), Kp
        h.b_mv, h.b_lg = bmv.data_ptr(), blg.data_ptr()
        if kslices > 1:
            nfl = L.lib.dmvae_heads_latent_kslice_floats(B_pad, Dp, kslices)
            ksws = torch.full((nfl,), float("nan"), device="cuda")                      # stale slabs must not matter
            ticks = torch.zeros(B_pad // 16, dtype=torch.int32, device="cuda")
            h.kslices, h.kslice_ws, h.kslice_ws_floats, h.kslice_tick = kslices, ksws.data_ptr(), nfl, ticks.data_ptr()
        L.check(L.lib.dmvae_heads_latent_fwd(stream(), C.byref(h), C.byref(a)), "dmvae_heads_latent_fwd")
    else:
        pr = (L.GemmProblem * 2)()
        pr[0].M, pr[0].N, pr[0].K = B_pad, 2 * Dp, Hp
        pr[0].A, pr[0].lda, pr[0].B, pr[0].ldb = hzc.data_ptr(), 2 * Hp, Wmv.data_ptr(), 2 * Dp
        pr[0].epi.kind, pr[0].epi.out, pr[0].epi.ldo, pr[0].epi.bias = L.EPI_BIAS_F32, mv.data_ptr(), 2 * Dp, bmv.data_ptr()
        pr[1].M, pr[1].N, pr[1].K = B_pad, Kp, Hp
        pr[1].A, pr[1].lda, pr[1].B, pr[1].ldb = hzc.data_ptr() + 2 * Hp, 2 * Hp, Wlg.data_ptr(), Kp
        pr[1].epi.kind, pr[1].epi.out, pr[1].epi.ldo, pr[1].epi.bias = L.EPI_BIAS_F32, lg.data_ptr(), Kp, blg.data_ptr()
        L.check(L.lib.dmvae_gemm_grouped(stream(), 1, 0, pr, 2), "dmvae_gemm_grouped")
        L.check(L.lib.dmvae_latent_fwd(stream(), C.byref(a)), "dmvae_latent_fwd")
    torch.cuda.synchronize()
    if fused and kslices > 1:
        assert not ticks.any().item(), "the tickets must be back at zero"
    return dict(mv=mv, logits=lg, Z=Z, Zf=Zf, w=w, gmu=gmu, glv=glv, clv=clv, dlogits=dlg, dpri=dpri, lp=lp)


CASES = [  # B, D, K, Hp, mode, noise
    (100, 10, 10, 2048, 0, "device"),        # cfg1's geometry: Dp = 64, DC = 16, a ragged last block
    (4096, 64, 10, 2048, 0, "device"),       # cfg2, the metric's shape: 256 workgroups
    (4096, 64, 10, 2048, 1, "device"),       # ... its Gumbel-Softmax step
    (256, 64, 10, 512, 0, "caller"),         # caller-supplied eps (parity-mode noise)
    (256, 20, 33, 320, 1, "caller"),         # three logits column tiles, DC = 32, caller eps + Gumbel noise
    (1000, 128, 10, 1024, 0, "device"),      # Dp = 128 (cfg3's latent geometry at one round of the chip): DC = 128, three ring slots
    (200, 100, 16, 192, 1, "device"),        # Dp = 128, D < Dp, fewer K tiles than ring slots would need at Dp = 64
    (48, 32, 64, 128, 0, "device"),          # K = 64: all four logits tiles; nk = 2 < NSTAGE
    (16, 3, 5, 64, 0, "caller"),             # one block, one K tile
]


@pytest.mark.parametrize("B,D,K,Hp,mode,noise", CASES)
def test_heads_latent_fused_equals_the_two_launches(hip, B, D, K, Hp, mode, noise):
    """every output of the fused launch -- the heads' f32 results, Z (bf16 and f32), weights, KL gradients, reparam coefficient, dlogits, the
    per-block prior-table partials and loss partials -- equals, bit for bit, what dmvae_gemm_grouped (BIAS_F32) + dmvae_latent_fwd leave"""
    L = hip
    rng = np.random.RandomState(B + 7 * D + 13 * K + mode)
    B_pad = (B + 63) // 64 * 64
    Dp, Kp = (D + 63) // 64 * 64, 64
    hzc = np.maximum(rng.randn(B_pad, 2 * Hp), 0.0) * 0.5
    hzc[B:] = 0.0
    Wmv = np.zeros((Hp, 2 * Dp)); Wmv[:, :D] = rng.randn(Hp, D) * 0.05; Wmv[:, Dp:Dp + D] = rng.randn(Hp, D) * 0.03
    Wlg = np.zeros((Hp, Kp)); Wlg[:, :K] = rng.randn(Hp, K) * 0.06
    bmv = np.zeros(2 * Dp); bmv[:D] = rng.randn(D) * 0.1; bmv[Dp:Dp + D] = rng.randn(D) * 0.1 - 0.3
    blg = np.zeros(Kp); blg[:K] = rng.randn(K) * 0.2
    pm, plv = rng.randn(K, D), rng.randn(K, D) * 0.4
    eps = rng.randn(B, D) if noise == "caller" else None
    gum = -np.log(-np.log(rng.rand(B, K))) if (noise == "caller" and mode == 1) else None
    t = [dev(hzc, torch.bfloat16), dev(Wmv, torch.bfloat16), dev(Wlg, torch.bfloat16), dev(bmv), dev(blg)]
    two = run_heads_latent(L, *t, eps, gum, pm, plv, mode, 0.7, 0.8, B, D, K, fused=False)
    one = run_heads_latent(L, *t, eps, gum, pm, plv, mode, 0.7, 0.8, B, D, K, fused=True)
    assert torch.isfinite(two["lp"]).all() and two["lp"][:, 0].abs().sum().item() > 0
    # the heads' own results against float64 on the bf16-rounded operands (so that the comparison below is not two equal mistakes)
    ref = t[0][:, :Hp].double() @ t[1].double() + t[3].double()
    np.testing.assert_allclose(one["mv"].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    refl = t[0][:, Hp:].double() @ t[2].double() + t[4].double()
    np.testing.assert_allclose(one["logits"].double().cpu().numpy(), refl.cpu().numpy(), rtol=1e-4, atol=1e-4)
    for k in two:
        assert torch.equal(two[k], one[k]), "%s differs (%d elements)" % (k, int((two[k] != one[k]).sum().item()))
    again = run_heads_latent(L, *t, eps, gum, pm, plv, mode, 0.7, 0.8, B, D, K, fused=True)
    for k in one:
        assert torch.equal(again[k], one[k]), "%s: not reproducible" % k


@pytest.mark.parametrize("B,D,K,Hp,mode,S", [(100, 10, 10, 2048, 0, 8), (256, 64, 10, 2048, 1, 4), (1000, 128, 10, 1024, 0, 2), (16, 3, 5, 128, 0, 2)])
def test_heads_latent_in_k_slices(hip, B, D, K, Hp, mode, S):
    """K slices of the fused launch (dmvae_heads_args::kslices; small batches: few 16-row blocks, each a long K chain): S workgroups per block, the last to
    arrive adds the partial tiles in ascending order, then biases and the latent stage.  Against the unsliced launch: the heads' outputs agree to f32
    summation order (1e-5), everything downstream to the tolerances below; two sliced runs are bit-identical; the tickets end at zero."""
    L = hip
    rng = np.random.RandomState(3 * B + D + K + S)
    B_pad = (B + 63) // 64 * 64
    Dp, Kp = (D + 63) // 64 * 64, 64
    hzc = np.maximum(rng.randn(B_pad, 2 * Hp), 0.0) * 0.5
    hzc[B:] = 0.0
    Wmv = np.zeros((Hp, 2 * Dp)); Wmv[:, :D] = rng.randn(Hp, D) * 0.05; Wmv[:, Dp:Dp + D] = rng.randn(Hp, D) * 0.03
    Wlg = np.zeros((Hp, Kp)); Wlg[:, :K] = rng.randn(Hp, K) * 0.06
    bmv = np.zeros(2 * Dp); bmv[:D] = rng.randn(D) * 0.1; bmv[Dp:Dp + D] = rng.randn(D) * 0.1 - 0.3
    blg = np.zeros(Kp); blg[:K] = rng.randn(K) * 0.2
    pm, plv = rng.randn(K, D), rng.randn(K, D) * 0.4
    t = [dev(hzc, torch.bfloat16), dev(Wmv, torch.bfloat16), dev(Wlg, torch.bfloat16), dev(bmv), dev(blg)]
    one = run_heads_latent(L, *t, None, None, pm, plv, mode, 0.7, 0.8, B, D, K, fused=True)
    sl = run_heads_latent(L, *t, None, None, pm, plv, mode, 0.7, 0.8, B, D, K, fused=True, kslices=S)
    sl2 = run_heads_latent(L, *t, None, None, pm, plv, mode, 0.7, 0.8, B, D, K, fused=True, kslices=S)
    for k in sl:
        assert torch.equal(sl[k], sl2[k]), "%s: two sliced runs differ" % k
    np.testing.assert_allclose(sl["mv"].cpu().numpy(), one["mv"].cpu().numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(sl["logits"].cpu().numpy(), one["logits"].cpu().numpy(), rtol=1e-5, atol=2e-5)
    for k, tol in (("Zf", 1e-4), ("w", 1e-4), ("gmu", 1e-3), ("glv", 1e-3), ("clv", 1e-4)):
        a, b = one[k].cpu().numpy(), sl[k].cpu().numpy()
        assert np.abs(a - b).max() <= tol * max(1e-6, np.abs(a).max()), k
    assert sl["lp"].double().sum(0).cpu().numpy() == pytest.approx(one["lp"].double().sum(0).cpu().numpy(), rel=1e-5)


def _three_bf16_pieces(x):
    """x (float32) = p0 + p1 + p2 exactly, each piece representable in bf16 (8 significant bits each, truncation)"""
    x = np.asarray(x, np.float32)
    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    p0 = trunc(x.copy()); r = x - p0
    p1 = trunc(r.copy()); p2 = r - p1
    assert np.array_equal(trunc(p2.copy()), p2) and np.array_equal((p0 + p1) + p2, x)
    return p0, p1, p2


@pytest.fixture(scope="module")
def golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "priors_golden.npz"))


def test_heads_latent_against_reference_golden_vectors(hip, golden):
    """The reference's own priors.py outputs through the FUSED entry.  The golden vectors give mean / log_var / logits as inputs; here they are
    made the exact result of the head GEMMs: row b of the hidden layers is the indicator of columns 3 b .. 3 b + 2, and rows 3 b .. 3 b + 2 of the
    head kernels hold the three bf16 pieces of row b's values (their f32 sum is the value, in any order of addition) -- so the launch's
    [mean | log_var] and logits must equal the golden inputs bit for bit, and everything downstream must meet the tolerances of
    test_latent_fwd_against_reference_golden_vectors."""
    L = hip
    for ci in range(int(golden["n_cases"])):
        pre = "c%d_s0_" % ci
        gv = lambda k: golden[pre + k]
        B, D, K = (int(x) for x in gv("shape"))
        B_pad, Dp, Kp = (B + 15) // 16 * 16, 64, 64
        Hp = (3 * B + 127) // 128 * 128
        hzc = np.zeros((B_pad, 2 * Hp), np.float32)
        Wmv, Wlg = np.zeros((Hp, 2 * Dp), np.float32), np.zeros((Hp, Kp), np.float32)
        for i, (pmean, plvar, plog) in enumerate(zip(_three_bf16_pieces(gv("mean")), _three_bf16_pieces(gv("log_var")), _three_bf16_pieces(gv("logits")))):
            for b in range(B):
                hzc[b, 3 * b + i] = 1.0; hzc[b, Hp + 3 * b + i] = 1.0
                Wmv[3 * b + i, :D] = pmean[b]; Wmv[3 * b + i, Dp:Dp + D] = plvar[b]
                Wlg[3 * b + i, :K] = plog[b]
        t = [dev(hzc, torch.bfloat16), dev(Wmv, torch.bfloat16), dev(Wlg, torch.bfloat16), dev(np.zeros(2 * Dp)), dev(np.zeros(Kp))]
        for mode, tau, ti, S in ((0, 1.0, None, 1), (1, 1.0, 0, 1), (1, 0.5, 1, 1), (0, 1.0, None, 2), (1, 0.5, 1, 4 if Hp % 256 == 0 else 2)):
            # (S > 1: in K slices -- the three pieces of a value may meet in different slices; their f32 sum is the value in any order, so the heads' outputs
            #  are still the golden inputs bit for bit)
            o = run_heads_latent(L, *t, gv("eps"), gv("gumbel").reshape(B, K), gv("prior_means"), gv("prior_log_vars"), mode, tau, 1.0, B, D, K, fused=True, kslices=S)
            mv, lg = o["mv"].cpu().numpy(), o["logits"].cpu().numpy()
            np.testing.assert_array_equal(mv[:B, :D], gv("mean").astype(np.float32))
            np.testing.assert_array_equal(mv[:B, Dp:Dp + D], gv("log_var").astype(np.float32))
            np.testing.assert_array_equal(lg[:B, :K], gv("logits").astype(np.float32))
            klz, klc = o["lp"][:, 0].double().sum().item() / B, o["lp"][:, 1].double().sum().item() / B
            np.testing.assert_allclose(o["Zf"].cpu().numpy()[:B, :D], gv("Z"), rtol=3e-6, atol=3e-6)
            assert klc == pytest.approx(float(gv("kl_c")), rel=3e-5, abs=1e-6)
            if mode == 0:
                assert klz == pytest.approx(float(gv("kl_z_exact")), rel=3e-5)
                np.testing.assert_allclose(o["w"].cpu().numpy()[:B, :K], gv("w"), rtol=1e-5, atol=1e-7)
            else:
                np.testing.assert_allclose(o["w"].cpu().numpy()[:B, :K], gv("zeta_t%d" % ti).reshape(B, K), rtol=2e-5, atol=1e-7)
                assert klz == pytest.approx(float(gv("kl_z_relaxed_t%d" % ti)), rel=5e-5)


def test_heads_latent_refuses_what_it_cannot_take(hip):
    L = hip
    assert L.lib.dmvae_heads_latent_ok(8192, 64, 10, 64, 64, 2048, 0) == 0       # more than one round of the chip
    assert L.lib.dmvae_heads_latent_ok(4096, 256, 50, 256, 64, 2048, 0) == 0     # Dp = 256 / the MFMA latent form
    assert L.lib.dmvae_heads_latent_ok(4096, 64, 100, 64, 128, 2048, 0) == 0     # Kp = 128
    assert L.lib.dmvae_heads_latent_ok(4096, 64, 10, 64, 64, 2000, 0) == 0       # Hp not a multiple of 64
    assert L.lib.dmvae_heads_latent_ok(4096, 64, 10, 64, 64, 2048, 2) == 0       # VaDE's latent stage
    assert L.lib.dmvae_heads_latent_ok(4096, 64, 10, 64, 64, 2048, 1) == 1


def test_step_with_and_without_the_fused_heads_latent_launch(hip):
    """the whole bf16 training step with knob 19 = 1 (fused, default) and 0 (two launches): same loss, parameters, moments after three steps
    (K slices off -- knob 21 = 0: at 512 rows the fused launch would run in slices, whose f32 summation order is another one)"""
    L = hip
    import dmvae_oracle as O
    from dmvae_hip import StepEngine
    kw = dict(input_dim=784, latent_dim=64, n_classes=10)
    B = 512
    X = torch.as_tensor(O.synthetic_images(2 * B, 784, seed=4)).cuda()
    perm = torch.randperm(2 * B, device="cuda").to(torch.int32)
    out = []
    try:
        L.check(L.lib.dmvae_debug_set_knob(21, 0))
        for knob in (1, 0):
            L.check(L.lib.dmvae_debug_set_knob(19, knob))
            e = StepEngine(dtype="bf16", max_batch=B, seed=3, **kw)
            e.init_parameters(0); e.reset_epoch(2)
            for _ in range(3):
                e.train_step(X, perm, use_state_cursor=True)
            torch.cuda.synchronize()
            st = e.read_state()
            out.append((st.last_loss, st.epoch_loss, e.param.clone(), e.m.clone(), e.v.clone(), e.view("mean").clone(), e.view("logits").clone()))
    finally:
        L.check(L.lib.dmvae_debug_set_knob(19, 1))
        L.check(L.lib.dmvae_debug_set_knob(21, 1))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    for a, b in zip(out[0][2:], out[1][2:]):
        assert torch.equal(a, b)
