"""GPU parity tests, kernel level: every HIP kernel is called through the C ABI
(ctypes) and compared with the oracle / float64 math on the same inputs."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import dmvae_oracle as O


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


def gemm(L, dtype, layout, M, N, K, A, lda, B, ldb, epi, split=1):
    L.check(L.lib.dmvae_gemm(stream(), dtype, layout, M, N, K, L.ptr(A), lda, L.ptr(B), ldb, C.byref(epi), split), "dmvae_gemm")
    torch.cuda.synchronize()


def operands(layout, M, N, K, rng, integer):
    """returns (A_logical [M,K], B_logical [K,N], A_mem, B_mem) as float64 numpy"""
    if integer:
        A = rng.randint(-3, 4, size=(M, K)).astype(np.float64)
        B = rng.randint(-3, 4, size=(K, N)).astype(np.float64)
    else:
        A = rng.randn(M, K)
        B = rng.randn(K, N)
    if layout == 0:
        return A, B, A, B
    if layout == 1:
        return A, B, A, B.T.copy()
    return A, B, A.T.copy(), B


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 192, 128), (64, 64, 256), (384, 128, 192), (128, 320, 64)])
def test_gemm_layouts_exact_integers(hip, dtype, layout, shape):
    """small integers are exact in bf16 and in f32 accumulation: any fragment /
    transposed-read / C-layout mistake shows as a hard mismatch (asymmetric data)."""
    L = hip
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K + layout)
    A, B, Am, Bm = operands(layout, M, N, K, rng, True)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    Ad, Bd = dev(Am, tdt), dev(Bm, tdt)
    out = torch.full((M, N), 777.0, dtype=torch.float32, device="cuda")
    e = L.Epilogue()
    e.kind = L.EPI_STORE_F32
    e.out, e.ldo = L.ptr(out).value, N
    gemm(L, dtype, layout, M, N, K, Ad, Am.shape[1], Bd, Bm.shape[1], e)
    np.testing.assert_array_equal(out.cpu().numpy().astype(np.float64), A @ B)


@pytest.mark.parametrize("dtype", [1, 0])
def test_gemm_large_tile_paths_and_random_data(hip, dtype):
    """shapes that select the 128x128 / 128x64 / 64x128 tiles; random data within
    bf16 / f32 rounding of the float64 product of the rounded operands."""
    L = hip
    rng = np.random.RandomState(0)
    for layout, (M, N, K) in [(0, (2048, 4096, 128)), (0, (4096, 832, 64)), (1, (4096, 512, 128)),
                              (2, (832, 512, 1024)), (2, (512, 4096, 512)), (1, (1024, 2048, 64))]:
        A, B, Am, Bm = operands(layout, M, N, K, rng, False)
        tdt = torch.bfloat16 if dtype == 1 else torch.float32
        Ad, Bd = dev(Am, tdt), dev(Bm, tdt)
        Ar = Ad.float().cpu().numpy().astype(np.float64)
        Br = Bd.float().cpu().numpy().astype(np.float64)
        Al = Ar if layout != 2 else Ar.T
        Bl = Br if layout != 1 else Br.T
        ref = Al @ Bl
        out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        e = L.Epilogue()
        e.kind = L.EPI_STORE_F32
        e.out, e.ldo = L.ptr(out).value, N
        gemm(L, dtype, layout, M, N, K, Ad, Am.shape[1], Bd, Bm.shape[1], e)
        err = np.abs(out.cpu().numpy() - ref).max()
        assert err < 2e-3 * math.sqrt(K), (layout, M, N, K, err)


@pytest.mark.parametrize("dtype", [1, 0])
def test_gemm_split_k_atomic(hip, dtype):
    L = hip
    M, N, K = 128, 192, 1024
    rng = np.random.RandomState(5)
    A, B, Am, Bm = operands(2, M, N, K, rng, True)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    e = L.Epilogue()
    e.kind = L.EPI_ATOMIC_F32
    e.out, e.ldo = L.ptr(out).value, N
    gemm(L, dtype, 2, M, N, K, dev(Am, tdt), M, dev(Bm, tdt), N, e, split=4)
    np.testing.assert_array_equal(out.cpu().numpy().astype(np.float64), A @ B)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("shape", [(64, 64, 128), (512, 832, 1024), (2048, 128, 256), (128, 192, 4096)])
def test_gemm_dw_fused_bias_gradient(hip, dtype, shape):
    """DW layout with epi.out2: db[n] = sum_k dY[k][n] from the ones-operand MFMA (exact on integers)."""
    L = hip
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A, B, Am, Bm = operands(2, M, N, K, rng, True)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    Ad, Bd = dev(Am, tdt), dev(Bm, tdt)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    db = torch.full((N,), 5.0, dtype=torch.float32, device="cuda")
    e = L.Epilogue()
    e.kind = L.EPI_STORE_F32
    e.out, e.ldo, e.out2 = L.ptr(out).value, N, L.ptr(db).value
    gemm(L, dtype, 2, M, N, K, Ad, M, Bd, N, e)
    np.testing.assert_array_equal(out.cpu().numpy().astype(np.float64), A @ B)
    np.testing.assert_array_equal(db.cpu().numpy().astype(np.float64), B.sum(0))


@pytest.mark.parametrize("dtype", [1, 0])
def test_gemm_grouped_dw_matches_individual_products(hip, dtype):
    """nine dW problems of mixed shapes in ONE launch (exact on integers), each with its fused db."""
    L = hip
    K = 512
    shapes = [(512, 832), (512, 512), (2048, 512), (64, 2048), (2048, 128), (2048, 64), (512, 4096), (512, 512), (832, 512)]
    rng = np.random.RandomState(9)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    probs = (L.GemmProblem * len(shapes))()
    keep, refs = [], []
    for i, (M, N) in enumerate(shapes):
        A, B, Am, Bm = operands(2, M, N, K, rng, True)
        Ad, Bd = dev(Am, tdt), dev(Bm, tdt)
        out = torch.full((M, N), 3.0, dtype=torch.float32, device="cuda")
        db = torch.full((N,), 3.0, dtype=torch.float32, device="cuda")
        keep.append((Ad, Bd, out, db))
        refs.append((A @ B, B.sum(0)))
        p = probs[i]
        p.M, p.N, p.K = M, N, K
        p.A, p.lda, p.B, p.ldb = L.ptr(Ad).value, M, L.ptr(Bd).value, N
        p.epi.kind = L.EPI_STORE_F32
        p.epi.out, p.epi.ldo, p.epi.out2 = L.ptr(out).value, N, L.ptr(db).value
    L.check(L.lib.dmvae_gemm_grouped_dw(stream(), dtype, probs, len(shapes)), "dmvae_gemm_grouped_dw")
    torch.cuda.synchronize()
    for (Ad, Bd, out, db), (ref, dbref) in zip(keep, refs):
        np.testing.assert_array_equal(out.cpu().numpy().astype(np.float64), ref)
        np.testing.assert_array_equal(db.cpu().numpy().astype(np.float64), dbref)


def _act(dtype):
    return torch.bfloat16 if dtype == 1 else torch.float32


@pytest.mark.parametrize("dtype", [1, 0])
def test_epilogues(hip, dtype):
    L = hip
    rng = np.random.RandomState(11)
    M, N, K = 256, 192, 128
    tdt = _act(dtype)
    tol = 2e-2 if dtype == 1 else 1e-5
    A, B, Am, Bm = operands(0, M, N, K, rng, False)
    A *= 0.2
    Ad, Bd = dev(A, tdt), dev(B, tdt)
    ref = Ad.float().cpu().numpy().astype(np.float64) @ Bd.float().cpu().numpy().astype(np.float64)
    bias = rng.randn(N)
    bd = dev(bias)
    # BIAS_RELU
    out = torch.zeros((M, N), dtype=tdt, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_BIAS_RELU; e.out, e.ldo = L.ptr(out).value, N; e.bias = L.ptr(bd).value
    gemm(L, dtype, 0, M, N, K, Ad, K, Bd, N, e)
    np.testing.assert_allclose(out.float().cpu().numpy(), np.maximum(ref + bias, 0), atol=tol * 3, rtol=tol)
    # BIAS_F32
    outf = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_BIAS_F32; e.out, e.ldo = L.ptr(outf).value, N; e.bias = L.ptr(bd).value
    gemm(L, dtype, 0, M, N, K, Ad, K, Bd, N, e)
    np.testing.assert_allclose(outf.cpu().numpy(), ref + bias, atol=1e-3 if dtype == 1 else 2e-5)
    # BIAS_SIGMOID
    e = L.Epilogue(); e.kind = L.EPI_BIAS_SIGMOID; e.out, e.ldo = L.ptr(outf).value, N; e.bias = L.ptr(bd).value
    gemm(L, dtype, 0, M, N, K, Ad, K, Bd, N, e)
    np.testing.assert_allclose(outf.cpu().numpy(), 1 / (1 + np.exp(-(ref + bias))), atol=1e-3 if dtype == 1 else 2e-6)
    # BIAS_RECON binary + real, with masked rows / cols and the logits copy
    mv, nv = 250, 185
    x = rng.rand(M, N)
    xd = dev(x)
    for kind in (0, 1):
        dl = torch.full((M, N), 5.0, dtype=tdt, device="cuda")
        lg = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        npart = L.lib.dmvae_gemm_partials(dtype, M, N)
        parts = torch.zeros(npart, dtype=torch.float32, device="cuda")
        e = L.Epilogue(); e.kind = L.EPI_BIAS_RECON; e.out, e.ldo = L.ptr(dl).value, N; e.bias = L.ptr(bd).value
        e.out2, e.ldo2 = L.ptr(lg).value, N
        e.aux0, e.ld0 = L.ptr(xd).value, N
        e.m_valid, e.n_valid, e.recon_kind, e.scale = mv, nv, kind, 1.0 / mv
        e.partials = L.ptr(parts).value
        gemm(L, dtype, 0, M, N, K, Ad, K, Bd, N, e)
        l = (ref + bias)
        lgpu = lg.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(lgpu, l, atol=1e-3 if dtype == 1 else 2e-5)
        mask = np.zeros((M, N)); mask[:mv, :nv] = 1
        if kind == 0:
            per = np.maximum(lgpu, 0) - lgpu * x + np.log1p(np.exp(-np.abs(lgpu)))
            dref = (1 / (1 + np.exp(-lgpu)) - x) / mv
        else:
            per = 0.5 * (lgpu - x) ** 2
            dref = (lgpu - x) / mv
        assert parts.sum().item() == pytest.approx((per * mask).sum(), rel=2e-5)
        np.testing.assert_allclose(dl.float().cpu().numpy(), dref * mask, atol=(3e-5 if dtype == 1 else 1e-7), rtol=1e-2 if dtype == 1 else 1e-5)
    # RELU_MASK (DX layout)
    Mx, Nx, Kx = 128, 256, 192
    A2, B2, Am2, Bm2 = operands(1, Mx, Nx, Kx, rng, False)
    A2d, B2d = dev(Am2 * 0.2, tdt), dev(Bm2, tdt)
    ref2 = A2d.float().cpu().numpy().astype(np.float64) @ B2d.float().cpu().numpy().astype(np.float64).T
    Y = rng.randn(Mx, Nx); Y[Y < 0] = 0
    Yd = dev(Y, tdt)
    out2 = torch.zeros((Mx, Nx), dtype=tdt, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_RELU_MASK; e.out, e.ldo = L.ptr(out2).value, Nx; e.aux0, e.ld0 = L.ptr(Yd).value, Nx
    gemm(L, dtype, 1, Mx, Nx, Kx, A2d, Kx, B2d, Kx, e)
    np.testing.assert_allclose(out2.float().cpu().numpy(), ref2 * (Y > 0), atol=tol * 3, rtol=tol)
    # LATENT (DX layout): out[:, :D] = dZ + gmu ; out[:, D:2D] = dZ*clv + glv
    D = 64
    A3, B3, Am3, Bm3 = operands(1, Mx, D, Kx, rng, False)
    A3d, B3d = dev(Am3 * 0.2, tdt), dev(Bm3, tdt)
    dZ = A3d.float().cpu().numpy().astype(np.float64) @ B3d.float().cpu().numpy().astype(np.float64).T
    gmu, glv, clv = rng.randn(Mx, D), rng.randn(Mx, D), rng.randn(Mx, D)
    g1, g2, g3 = dev(gmu), dev(glv), dev(clv)
    out3 = torch.zeros((Mx, 2 * D), dtype=tdt, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_LATENT; e.out, e.ldo, e.d_off = L.ptr(out3).value, 2 * D, D
    e.aux0, e.ld0, e.aux1, e.ld1, e.aux2, e.ld2 = L.ptr(g1).value, D, L.ptr(g2).value, D, L.ptr(g3).value, D
    gemm(L, dtype, 1, Mx, D, Kx, A3d, Kx, B3d, Kx, e)
    o = out3.float().cpu().numpy()
    np.testing.assert_allclose(o[:, :D], dZ + gmu, atol=tol * 3, rtol=tol)
    np.testing.assert_allclose(o[:, D:], dZ * clv + glv, atol=tol * 3, rtol=tol)


@pytest.mark.parametrize("M,K,form", [(256, 1024, "16-row tiles"), (4096, 2048, "16-row tiles, one round of 256"), (8192, 1024, "32-row tiles"),
                                      (16384, 1024, "the general 64 x 64 tiles (thin ones would not fit one round)")])
def test_dz_gemm_on_thin_tiles(hip, M, K, form):
    """The dZ GEMM (DX layout, LATENT epilogue, 64 columns, K >= 1024) runs on the thinnest of 16- / 32-row tiles that fits one round of the
    chip (gemm_bf16_thin_rows): against the float64 product of the bf16 operands, and bit-identical to the general 64 x 64 tiles (knob 18 = 0):
    the same MFMA chain per output element, k ascending, whatever the tile height."""
    L = hip
    D = 64
    g = torch.Generator(device="cuda"); g.manual_seed(M + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.2).bfloat16()
    W = (torch.randn(D, K, device="cuda", generator=g) * 0.2).bfloat16()
    gmu, glv, clv = (torch.randn(M, D, device="cuda", generator=g) for _ in range(3))
    outs = []
    for knob in (0, 2):
        L.check(L.lib.dmvae_debug_set_knob(18, knob))
        try:
            out = torch.full((M + 64, 2 * D), 7.0, device="cuda", dtype=torch.bfloat16)
            e = L.Epilogue(); e.kind = L.EPI_LATENT; e.out, e.ldo, e.d_off = out.data_ptr(), 2 * D, D
            e.aux0, e.ld0, e.aux1, e.ld1, e.aux2, e.ld2 = gmu.data_ptr(), D, glv.data_ptr(), D, clv.data_ptr(), D
            L.check(L.lib.dmvae_gemm(stream(), L.BF16, 1, M, D, K, L.ptr(A), K, L.ptr(W), K, C.byref(e), 1))
            torch.cuda.synchronize()
            outs.append(out)
        finally:
            L.check(L.lib.dmvae_debug_set_knob(18, 2))
    assert torch.equal(outs[0], outs[1]), form
    assert bool((outs[1][M:] == 7.0).all()), "stray store below the output"
    dZ = A.double() @ W.double().t()
    o = outs[1][:M].double()
    scale = float(dZ.abs().max())
    assert float((o[:, :D] - (dZ + gmu.double())).abs().max()) <= 2e-2 * max(1.0, scale)
    assert float((o[:, D:] - (dZ * clv.double() + glv.double())).abs().max()) <= 2e-2 * max(1.0, scale) * max(1.0, float(clv.abs().max()))


def test_gemm_rejects_unaligned_shapes(hip):
    L = hip
    a = torch.zeros(64 * 64, dtype=torch.float32, device="cuda")
    e = L.Epilogue(); e.kind = L.EPI_STORE_F32; e.out, e.ldo = L.ptr(a).value, 64
    rc = L.lib.dmvae_gemm(stream(), 0, 0, 60, 64, 64, L.ptr(a), 64, L.ptr(a), 64, C.byref(e), 1)
    assert rc == -1 and b"multiples of 64" in L.lib.dmvae_last_error()
    with pytest.raises(L.DmvaeError):
        L.check(L.lib.dmvae_gemm(stream(), 0, 0, 64, 64, 64, L.ptr(a), 64, L.ptr(a), 64, C.byref(e), 2))


# ------------------------------------------------------------------ latent kernel
def run_latent(L, mean, log_var, logits, eps, gumbel, pm, plv, mode, tau, kl_ratio, act_dtype, B_pad=None, ldpad=0, mfma=False):
    B, D = mean.shape
    K = logits.shape[1]
    B_pad = B_pad or ((B + 63) // 64 * 64)
    ldD, ldK = D + ldpad, K + ldpad

    def padrows(a, ld):
        out = np.zeros((B_pad, ld), np.float32)
        out[:a.shape[0], :a.shape[1]] = a
        return dev(out)
    t = dict(mean=padrows(mean, ldD), log_var=padrows(log_var, ldD), logits=padrows(logits, ldK))
    epsd = dev(eps) if eps is not None else None
    gd = dev(gumbel) if gumbel is not None else None
    pmd, plvd = dev(pm), dev(plv)
    adt = torch.bfloat16 if act_dtype == 1 else torch.float32
    ldZ = (D + 63) // 64 * 64
    ldl = (K + 63) // 64 * 64
    Z = torch.full((B_pad, ldZ), 9.0, dtype=adt, device="cuda")
    Zf = torch.zeros((B_pad, D), dtype=torch.float32, device="cuda")
    w = torch.zeros((B_pad, K), dtype=torch.float32, device="cuda")
    gmu = torch.full((B_pad, ldZ), 9.0, dtype=torch.float32, device="cuda")
    glv = torch.full((B_pad, ldZ), 9.0, dtype=torch.float32, device="cuda")
    clv = torch.full((B_pad, ldZ), 9.0, dtype=torch.float32, device="cuda")
    dlg = torch.full((B_pad, ldl), 9.0, dtype=adt, device="cuda")
    nblk = L.lib.dmvae_latent_nblocks(B_pad, D, K)
    dpri = torch.zeros((nblk, 2 * K * D), dtype=torch.float32, device="cuda")
    lp = torch.zeros((nblk, 2), dtype=torch.float32, device="cuda")
    a = L.LatentArgs()
    a.B, a.B_pad, a.D, a.K, a.mode, a.act_dtype = B, B_pad, D, K, mode, act_dtype
    a.kl_ratio, a.temperature, a.inv_B, a.seed, a.noise_step = kl_ratio, tau, 1.0 / B, 1234, 7
    a.mean, a.ld_mean = L.ptr(t["mean"]).value, ldD
    a.log_var, a.ld_log_var = L.ptr(t["log_var"]).value, ldD
    a.logits, a.ld_logits = L.ptr(t["logits"]).value, ldK
    if epsd is not None:
        a.eps, a.ld_eps = L.ptr(epsd).value, D
    if gd is not None:
        a.gumbel, a.ld_gumbel = L.ptr(gd).value, K
    a.prior_means, a.prior_log_vars = L.ptr(pmd).value, L.ptr(plvd).value
    a.Z_act, a.ld_Z = L.ptr(Z).value, ldZ
    a.Z_f32, a.ld_Zf = L.ptr(Zf).value, D
    a.weights, a.ld_w = L.ptr(w).value, K
    a.gmu, a.glv, a.clv, a.ld_g = L.ptr(gmu).value, L.ptr(glv).value, L.ptr(clv).value, ldZ
    a.dlogits_act, a.ld_dl = L.ptr(dlg).value, ldl
    a.dprior_partials, a.loss_partials = L.ptr(dpri).value, L.ptr(lp).value
    if mfma:          # the MFMA form of the contractions (csrc/latent_mfma.hip): the caller brings the scratch
        nb = L.lib.dmvae_latent_ws_bytes(B_pad, D, K, mode)
        assert nb > 0, "the MFMA form does not apply to this shape / mode"
        ws = torch.full((nb // 4,), float("nan"), dtype=torch.float32, device="cuda")      # stale scratch must not matter
        a.mfma_ws, a.mfma_ws_bytes = L.ptr(ws).value, nb
    L.check(L.lib.dmvae_latent_fwd(stream(), C.byref(a)), "dmvae_latent_fwd")
    torch.cuda.synchronize()
    dp = dpri.double().sum(0).cpu().numpy()
    return dict(Z=Z.float().cpu().numpy(), Zf=Zf.cpu().numpy(), w=w.cpu().numpy(), gmu=gmu.cpu().numpy(),
                glv=glv.cpu().numpy(), clv=clv.cpu().numpy(), dlogits=dlg.float().cpu().numpy(),
                dpm=dp[:K * D].reshape(K, D), dplv=dp[K * D:].reshape(K, D),
                klz=lp[:, 0].double().sum().item() / B, klc=lp[:, 1].double().sum().item() / B)


def oracle_latent(mean, log_var, logits, eps, gumbel, pm, plv, mode, tau, r):
    K = logits.shape[1]
    cfg = O.Config(4, mean.shape[1], K)
    a = dict(mean=mean, logvar=log_var, eps=eps, q=O.softmax(logits), kl_ratio=r, temperature=tau,
             mode="exact" if mode == 0 else "relaxed")
    a["w"] = a["q"] if mode == 0 else O.gumbel_softmax(logits, gumbel, tau)
    p = dict(prior_means=pm, prior_log_vars=plv)
    dmean, dlv, dlogits, dpm, dplv = O.latent_backward(cfg, a, p, np.zeros_like(mean))
    klz = (O.kl_mixture_exact if mode == 0 else O.kl_mixture_relaxed)(mean, log_var, a["w"], pm, plv)
    return dict(Z=O.gaussian_reparam(mean, log_var, eps), w=a["w"], gmu=dmean, glv=dlv,
                clv=eps * 0.5 * np.exp(log_var / 2), dlogits=dlogits, dpm=dpm, dplv=dplv, klz=klz,
                klc=O.kl_categorical(logits, K))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(4, 3, 5), (100, 10, 10), (200, 64, 10), (130, 300, 50), (70, 96, 130),
                                   (128, 256, 50), (64, 512, 256)])
def test_latent_fwd_matches_oracle(hip, mode, shape):
    """covers D < 16, D-chunking (D=300 -> chunks), K > 64, ragged B, and the prior-table
    shapes of BASELINE.json configs[3] (K=50, D=256) and configs[4] (K=256, D=512)."""
    L = hip
    B, D, K = shape
    rng = np.random.RandomState(B + D + K + mode)
    mean, lv = rng.randn(B, D) * 1.2, rng.randn(B, D) * 0.5 - 0.2
    logits = rng.randn(B, K) * 1.5
    eps = rng.randn(B, D)
    gum = O.sample_gumbel((B, K), rng)
    pm, plv = rng.randn(K, D), rng.randn(K, D) * 0.4
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    mean, lv, logits, eps, gum, pm, plv = map(f32, (mean, lv, logits, eps, gum, pm, plv))
    g = run_latent(L, mean, lv, logits, eps, gum, pm, plv, mode, 0.7, 0.6, 0, ldpad=4)
    o = oracle_latent(mean, lv, logits, eps, gum, pm, plv, mode, 0.7, 0.6)
    assert g["klz"] == pytest.approx(o["klz"], rel=2e-5, abs=1e-5)
    assert g["klc"] == pytest.approx(o["klc"], rel=2e-5, abs=1e-6)
    np.testing.assert_allclose(g["Zf"][:B], o["Z"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(g["Z"][:B, :D], o["Z"], rtol=2e-6, atol=2e-6)
    assert not g["Z"][:, D:].any() and not g["Z"][B:].any()
    np.testing.assert_allclose(g["w"][:B], o["w"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(g["clv"][:B, :D], o["clv"], rtol=1e-5, atol=1e-6)
    sc = 1.0 / B
    np.testing.assert_allclose(g["gmu"][:B, :D], o["gmu"], rtol=2e-4, atol=2e-5 * sc)
    np.testing.assert_allclose(g["glv"][:B, :D], o["glv"], rtol=2e-4, atol=2e-5 * sc)
    np.testing.assert_allclose(g["dlogits"][:B, :K], o["dlogits"], rtol=5e-4, atol=5e-5 * sc)
    assert not g["dlogits"][:, K:].any() and not g["dlogits"][B:].any() and not g["gmu"][B:, :D].any()
    np.testing.assert_allclose(g["dpm"], o["dpm"], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(g["dplv"], o["dplv"], rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(130, 300, 50), (70, 96, 130), (128, 256, 50), (64, 512, 256), (1000, 64, 64), (333, 30, 200)])
@pytest.mark.parametrize("noise", ["caller", "device"])
def test_latent_mfma_form_matches_oracle(hip, shape, noise):
    """K * D >= 4096, exact mode: the contractions as three f32 MFMA GEMMs between two row kernels.  Same oracle, same
    tolerances as the one-kernel form (the expanded squares are computed in exact f32); with device noise the drawn
    epsilon is recovered from Z and fed to the oracle."""
    L = hip
    B, D, K = shape
    rng = np.random.RandomState(B + D + K)
    mean, lv = rng.randn(B, D) * 1.2, rng.randn(B, D) * 0.5 - 0.2
    logits = rng.randn(B, K) * 1.5
    eps = rng.randn(B, D)
    pm, plv = rng.randn(K, D), rng.randn(K, D) * 0.4
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    mean, lv, logits, eps, pm, plv = map(f32, (mean, lv, logits, eps, pm, plv))
    g = run_latent(L, mean, lv, logits, eps if noise == "caller" else None, None, pm, plv, 0, 1.0, 0.6, 0, ldpad=4, mfma=True)
    if noise == "device":
        eps = (g["Zf"][:B].astype(np.float64) - mean) / np.exp(lv / 2)
        assert abs(eps.mean()) < 0.05 and abs(eps.std() - 1.0) < 0.05
        g2 = run_latent(L, mean, lv, logits, None, None, pm, plv, 0, 1.0, 0.6, 0, ldpad=4, mfma=True)
        np.testing.assert_array_equal(g["Zf"], g2["Zf"])           # same (seed, step) -> same draw
    o = oracle_latent(mean, lv, logits, eps, None, pm, plv, 0, 1.0, 0.6)
    assert g["klz"] == pytest.approx(o["klz"], rel=2e-5, abs=1e-5)
    assert g["klc"] == pytest.approx(o["klc"], rel=2e-5, abs=1e-6)
    if noise == "caller":
        np.testing.assert_allclose(g["Zf"][:B], o["Z"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(g["Z"][:B, :D], o["Z"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(g["clv"][:B, :D], o["clv"], rtol=1e-5, atol=1e-6)
    assert not g["Z"][:, D:].any() and not g["Z"][B:].any()
    np.testing.assert_allclose(g["w"][:B], o["w"], rtol=1e-5, atol=1e-7)
    sc = 1.0 / B
    np.testing.assert_allclose(g["gmu"][:B, :D], o["gmu"], rtol=2e-4, atol=2e-5 * sc)
    np.testing.assert_allclose(g["glv"][:B, :D], o["glv"], rtol=2e-4, atol=2e-5 * sc)
    np.testing.assert_allclose(g["dlogits"][:B, :K], o["dlogits"], rtol=5e-4, atol=5e-5 * sc)
    assert not g["dlogits"][:, K:].any() and not g["dlogits"][B:].any() and not g["gmu"][B:, :D].any()
    np.testing.assert_allclose(g["dpm"], o["dpm"], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(g["dplv"], o["dplv"], rtol=2e-4, atol=1e-5)
    # and against the one-kernel form on the same inputs
    if noise == "caller":
        c = run_latent(L, mean, lv, logits, eps, None, pm, plv, 0, 1.0, 0.6, 0, ldpad=4)
        assert g["klz"] == pytest.approx(c["klz"], rel=2e-5)
        np.testing.assert_allclose(g["dlogits"][:B, :K], c["dlogits"][:B, :K], rtol=1e-3, atol=1e-4 * sc)


def test_latent_fwd_against_reference_golden_vectors(hip, golden):
    """the reference's own priors.py outputs (tests/golden/priors_golden.npz)."""
    L = hip
    for ci in range(int(golden["n_cases"])):
        pre = "c%d_s0_" % ci
        g = lambda k: golden[pre + k]
        B, D, K = g("shape")
        out = run_latent(L, g("mean"), g("log_var"), g("logits"), g("eps"), g("gumbel").reshape(B, K),
                         g("prior_means"), g("prior_log_vars"), 0, 1.0, 1.0, 0)
        assert out["klz"] == pytest.approx(float(g("kl_z_exact")), rel=3e-5)
        assert out["klc"] == pytest.approx(float(g("kl_c")), rel=3e-5, abs=1e-6)
        np.testing.assert_allclose(out["Zf"][:B], g("Z"), rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(out["w"][:B], g("w"), rtol=1e-5, atol=1e-7)
        for ti, tau in enumerate((1.0, 0.5)):
            o2 = run_latent(L, g("mean"), g("log_var"), g("logits"), g("eps"), g("gumbel").reshape(B, K),
                            g("prior_means"), g("prior_log_vars"), 1, tau, 1.0, 0)
            np.testing.assert_allclose(o2["w"][:B], g("zeta_t%d" % ti).reshape(B, K), rtol=2e-5, atol=1e-7)
            assert o2["klz"] == pytest.approx(float(g("kl_z_relaxed_t%d" % ti)), rel=5e-5)


def test_latent_bf16_outputs_and_device_noise(hip):
    L = hip
    rng = np.random.RandomState(3)
    B, D, K = 256, 64, 10
    mean, lv, logits = rng.randn(B, D), rng.randn(B, D) * 0.3, rng.randn(B, K)
    pm, plv = rng.randn(K, D), np.zeros((K, D))
    g1 = run_latent(L, mean, lv, logits, None, None, pm, plv, 0, 1.0, 1.0, 1)
    g2 = run_latent(L, mean, lv, logits, None, None, pm, plv, 0, 1.0, 1.0, 1)
    np.testing.assert_array_equal(g1["Z"], g2["Z"])            # same (seed, step) -> same noise
    epsd = (g1["Zf"][:B] - mean) / np.exp(lv / 2)              # recover the Philox normals
    assert abs(epsd.mean()) < 0.02 and abs(epsd.std() - 1.0) < 0.02
    # four draws share one Philox block (two Box-Muller pairs): still i.i.d. N(0,1)
    z0 = run_latent(L, np.zeros((2048, 64)), np.zeros((2048, 64)), rng.randn(2048, K), None, None, pm, plv, 0, 1.0, 1.0, 0)["Zf"][:2048]
    assert abs(z0.mean()) < 0.01 and abs(z0.std() - 1.0) < 0.01
    assert abs((z0 ** 3).mean()) < 0.03 and abs((z0 ** 4).mean() - 3.0) < 0.1
    assert len(np.unique(z0.round(6))) > 0.95 * z0.size > 0                     # no block reused across rows / lanes
    cc = np.corrcoef(z0.T)                                                        # columns (incl. the 4 of one lane: d, d+16, d+32, d+48)
    assert np.abs(cc - np.eye(64)).max() < 0.12
    assert abs(np.corrcoef(z0[:-1].ravel(), z0[1:].ravel())[0, 1]) < 0.01         # consecutive rows
    np.testing.assert_allclose(g1["Z"][:B, :D], g1["Zf"][:B], rtol=1e-2, atol=1e-2)   # bf16 copy of the f32 Z


# ------------------------------------------------------------------ HBM-bound kernels
@pytest.mark.parametrize("act", [0, 1])
def test_recon_standalone(hip, act):
    L = hip
    rng = np.random.RandomState(1)
    B, Bp, I, Ip = 100, 128, 784, 832
    lg = np.zeros((Bp, Ip)); lg[:B, :I] = rng.randn(B, I) * 3
    x = np.zeros((Bp, Ip)); x[:B, :I] = rng.rand(B, I)
    lgd, xd, zd = dev(lg), dev(x), torch.zeros((Bp, Ip), device="cuda")
    for kind in (0, 1):
        nb = L.lib.dmvae_recon_nblocks(Bp, Ip)
        parts = torch.zeros(nb, dtype=torch.float32, device="cuda")
        dl = torch.full((Bp, Ip), 3.0, dtype=_act(act), device="cuda")
        L.check(L.lib.dmvae_recon_fwd_bwd(stream(), act, kind, B, Bp, I, Ip, L.ptr(lgd), Ip, L.ptr(xd), Ip,
                                          1.0 / B, L.ptr(dl), Ip, L.ptr(parts)))
        torch.cuda.synchronize()
        cfg = O.Config(I, 2, 2, input_type="binary" if kind == 0 else "real")
        l32, x32 = lg.astype(np.float32).astype(np.float64)[:B, :I], x.astype(np.float32).astype(np.float64)[:B, :I]
        assert parts.double().sum().item() / B == pytest.approx(O.recon_loss(cfg, x32, l32), rel=1e-5)
        dref = ((1 / (1 + np.exp(-l32)) - x32) if kind == 0 else (l32 - x32)) / B
        got = dl.float().cpu().numpy()
        np.testing.assert_allclose(got[:B, :I], dref, rtol=1e-2 if act else 1e-5, atol=1e-8)
        assert not got[B:].any() and not got[:, I:].any()
    # closed form: logits == 0 -> I*ln2 for any x
    parts = torch.zeros(L.lib.dmvae_recon_nblocks(Bp, Ip), dtype=torch.float32, device="cuda")
    dl = torch.zeros((Bp, Ip), dtype=_act(act), device="cuda")
    L.check(L.lib.dmvae_recon_fwd_bwd(stream(), act, 0, B, Bp, I, Ip, L.ptr(zd), Ip,
                                      L.ptr(xd), Ip, 1.0 / B, L.ptr(dl), Ip, L.ptr(parts)))
    torch.cuda.synchronize()
    assert parts.double().sum().item() / B == pytest.approx(784 * math.log(2), rel=1e-6)


def test_colsum(hip):
    L = hip
    rng = np.random.RandomState(2)
    for (M, N, dt) in [(100, 70, 0), (4096, 832, 1), (513, 300, 0), (8192, 64, 1)]:
        a = rng.randn(M, N)
        ad = dev(a, _act(dt))
        out = torch.zeros(N, dtype=torch.float32, device="cuda")
        L.check(L.lib.dmvae_colsum(stream(), dt, L.ptr(ad), N, M, N, L.ptr(out)))
        torch.cuda.synchronize()
        ref = ad.double().sum(0).cpu().numpy()
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-4)


# bf16 mode (hardware sqrt / reciprocal, 1 ulp each) | bf16 mode with DMVAE_ADAM_IEEE (dmvae_config.adam_ieee) | fp32 parity mode (IEEE)
@pytest.mark.parametrize("shadow,ieee", [(True, False), (True, True), (False, False)])
def test_adam_tf_matches_oracle_over_steps(hip, shadow, ieee):
    L = hip
    flags = L.ADAM_ZERO_GRAD | (L.ADAM_IEEE if ieee else 0)
    rng = np.random.RandomState(4)
    n = 4096 + 64
    p = {"a": rng.randn(n).astype(np.float32).astype(np.float64)}
    m, v = O.adam_tf_init(p)
    pd, md, vd = dev(p["a"]), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.zeros(n, dtype=torch.bfloat16, device="cuda") if shadow else None
    p32, m32, v32 = pd.clone(), md.clone(), vd.clone()          # the fp32 mode beside it: IEEE in bf16 mode must give ITS bits
    for t in range(1, 6):
        g = (rng.randn(n) * 10 ** rng.uniform(-4, 1)).astype(np.float32).astype(np.float64)
        gd = dev(g * 2.0)    # grad_scale 0.5 undoes the factor 2 (the 1/world path)
        O.adam_tf(p, {"a": g}, m, v, t, lr=0.002)
        L.check(L.lib.dmvae_adam_tf(stream(), n, L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), L.ptr(pb) if shadow else None, 0.002, 0.9, 0.999,
                                    1e-8, 0.5, flags, t, None))
        if ieee:
            g32 = dev(g * 2.0)
            L.check(L.lib.dmvae_adam_tf(stream(), n, L.ptr(p32), L.ptr(g32), L.ptr(m32), L.ptr(v32), None, 0.002, 0.9, 0.999, 1e-8, 0.5, 0, t, None))
            assert torch.equal(pd, p32) and torch.equal(md, m32) and torch.equal(vd, v32)
        torch.cuda.synchronize()
        np.testing.assert_allclose(pd.cpu().numpy(), p["a"], rtol=3e-6, atol=3e-7)
        np.testing.assert_allclose(md.cpu().numpy(), m["a"], rtol=1e-5, atol=2e-7)
        # (1 - beta2) is evaluated in float32, as TF's kernel does: 1.3e-5 relative to the double oracle
        np.testing.assert_allclose(vd.cpu().numpy(), v["a"], rtol=3e-5, atol=1e-12)
        assert not gd.any().item()                                   # zero_grad
        if shadow: np.testing.assert_array_equal(pb.float().cpu().numpy(), pd.to(torch.bfloat16).float().cpu().numpy())
    # step 1 from zero state ~ -lr*sign(g)
    p1, g1 = dev(np.zeros(8)), dev(np.array([1, -1, 2, -3, 1e-2, -1e-2, 5, -5.0]))
    z1, z2 = torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda")
    L.check(L.lib.dmvae_adam_tf(stream(), 8, L.ptr(p1), L.ptr(g1), L.ptr(z1), L.ptr(z2), None, 0.002, 0.9, 0.999, 1e-8,
                                1.0, 0, 1, None))
    np.testing.assert_allclose(p1.cpu().numpy(), -0.002 * np.sign(g1.cpu().numpy()), rtol=1e-4)


@pytest.mark.parametrize("shadow,ieee,blocks", [(True, False, 0), (True, True, 64), (False, False, 7)])
def test_adam_shadow_form_equals_the_plain_kernel(hip, shadow, ieee, blocks):
    """DMVAE_ADAM_SHADOW (csrc/elementwise.hip adam_shadow_kernel: <= 48 VGPRs, loads in flight in a wave-private LDS ring filled by LDS-DMA) gives the
    bits of adam_tf_kernel -- both arithmetic modes, a size that is not a multiple of its 256-element chunks (the remainder runs on the plain
    kernel), few and many workgroups, zero_grad -- over three steps driven by the device state; without the state it is refused."""
    L = hip
    rng = np.random.RandomState(11)
    n = 256 * 1237 + 132
    lr, b1, b2 = (float(np.float32(x)) for x in (0.002, 0.9, 0.999))
    base = [dev(rng.randn(n)), dev(np.zeros(n)), dev(np.abs(rng.randn(n)) * 1e-3)]
    a = [t.clone() for t in base]; b = [t.clone() for t in base]
    pa = torch.zeros(n, dtype=torch.bfloat16, device="cuda") if shadow else None
    pb = torch.zeros(n, dtype=torch.bfloat16, device="cuda") if shadow else None
    fl = L.ADAM_ZERO_GRAD | (L.ADAM_IEEE if ieee else 0)
    NOT = C.c_uint64(2 ** 64 - 1)
    for t in range(1, 4):
        st = L.State(); st.adam_t, st.lr = t, lr
        st.lr_t = float(np.float32(lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)))          # as common.h adam_lr_t rounds it
        state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
        g = dev(rng.randn(n) * 10 ** rng.uniform(-3, 1))
        ga, gb = g.clone(), g.clone()
        L.check(L.lib.dmvae_adam_tf(stream(), n, L.ptr(a[0]), L.ptr(ga), L.ptr(a[1]), L.ptr(a[2]), L.ptr(pa) if shadow else None, lr, b1, b2, 1e-8, 0.5, fl, NOT, L.ptr(state)))
        L.check(L.lib.dmvae_adam_tf(stream(), n, L.ptr(b[0]), L.ptr(gb), L.ptr(b[1]), L.ptr(b[2]), L.ptr(pb) if shadow else None, lr, b1, b2, 1e-8, 0.5,
                                    fl | L.ADAM_SHADOW | (blocks << 8), NOT, L.ptr(state)))
        torch.cuda.synchronize()
        for x, y, what in zip(a, b, "pmv"):
            assert torch.equal(x, y), (what, t, int((x != y).sum().item()))
        assert not ga.any().item() and not gb.any().item()
        if shadow: assert torch.equal(pa, pb) and torch.equal(pb, b[0].to(torch.bfloat16))
    assert L.lib.dmvae_adam_tf(stream(), n, L.ptr(b[0]), L.ptr(gb), L.ptr(b[1]), L.ptr(b[2]), None, lr, b1, b2, 1e-8, 1.0, L.ADAM_SHADOW, 1, None) != 0      # needs the state


def test_gather_rows_follows_dataset_order(hip):
    L = hip
    rng = np.random.RandomState(6)
    N, dim, B, Bp, ld = 23, 10, 5, 128, 64
    data = rng.rand(N, dim).astype(np.float32)
    perm = rng.permutation(N).astype(np.int32)
    dd, pd = dev(data), torch.as_tensor(perm).cuda()
    for act in (0, 1):
        for first, nv in ((0, 5), (20, 3)):
            oa = torch.full((Bp, ld), 7.0, dtype=_act(act), device="cuda")
            of = torch.full((Bp, ld), 7.0, dtype=torch.float32, device="cuda")
            L.check(L.lib.dmvae_gather_rows(stream(), act, L.ptr(dd), N, dim, L.ptr(pd), first, B, nv, Bp,
                                            L.ptr(oa), ld, L.ptr(of), ld, None))
            torch.cuda.synchronize()
            exp = np.zeros((Bp, ld), np.float32)
            exp[:nv, :dim] = data[perm[first:first + nv]]
            np.testing.assert_array_equal(of.cpu().numpy(), exp)
            np.testing.assert_array_equal(oa.float().cpu().numpy(), torch.as_tensor(exp).to(_act(act)).float().numpy())


def test_philox_noise_moments_and_cast(hip):
    L = hip
    n = 1 << 20
    a = torch.zeros(n, device="cuda")
    L.check(L.lib.dmvae_philox_normal(stream(), L.ptr(a), n, 42, 3, 0))
    x = a.double().cpu().numpy()
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1) < 5e-3 and abs((x ** 3).mean()) < 2e-2
    assert abs((x ** 4).mean() - 3) < 5e-2
    b = torch.zeros(n, device="cuda")
    L.check(L.lib.dmvae_philox_normal(stream(), L.ptr(b), n, 42, 4, 0))
    assert abs(np.corrcoef(x, b.double().cpu().numpy())[0, 1]) < 5e-3
    L.check(L.lib.dmvae_philox_gumbel(stream(), L.ptr(b), n, 42, 3, 1))
    gm = b.double().cpu().numpy()
    assert abs(gm.mean() - 0.5772) < 5e-3 and abs(gm.var() - math.pi ** 2 / 6) < 2e-2   # Gumbel(0,1)
    c = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib.dmvae_cast_f32_to_bf16(stream(), L.ptr(a), L.ptr(c), n))
    assert torch.equal(c, a.to(torch.bfloat16))
    d = torch.zeros(n, device="cuda")
    L.check(L.lib.dmvae_cast_bf16_to_f32(stream(), L.ptr(c), L.ptr(d), n))
    assert torch.equal(d, c.float())


# ---------------------------------------------------------------- streaming dX of the two head layers (csrc/heads_dx.hip, round 4)
def _heads_dx_case(L, M, Ks, N, seed, reps=1):
    """the grouped DX / RELU_MASK launch of len(Ks) sibling problems writing side-by-side column ranges of one [M, len(Ks) * N] output
    (the layout of d[hz | hc], code/base_models.py:229-248), on the grouped tiles (knob 13 = 0) and on the streaming kernel (1)"""
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    P = len(Ks)
    dY = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for K in Ks]
    W = [(torch.randn(N, K, device="cuda", generator=g) * 0.25).bfloat16() for K in Ks]       # [in = N][out = K]: contraction-contiguous for dX
    gate = torch.relu(torch.randn(M, P * N, device="cuda", generator=g)).bfloat16()           # ~half of the units closed
    SENT = 7.0
    outs = []
    for knob in (0, 1):
        L.check(L.lib.dmvae_debug_set_knob(13, knob))
        try:
            res = []
            for _ in range(reps):
                out = torch.full((M + 64, P * N + 64), SENT, device="cuda", dtype=torch.bfloat16)    # a guard band of rows and columns around the output
                probs = (L.GemmProblem * P)()
                for i in range(P):
                    p = probs[i]
                    p.M, p.N, p.K = M, N, Ks[i]
                    p.A, p.lda, p.B, p.ldb = dY[i].data_ptr(), Ks[i], W[i].data_ptr(), Ks[i]
                    p.epi.kind = L.EPI_RELU_MASK
                    p.epi.out = out.data_ptr() + i * N * 2; p.epi.ldo = P * N + 64
                    p.epi.aux0 = gate.data_ptr() + i * N * 2; p.epi.ld0 = P * N
                L.check(L.lib.dmvae_gemm_grouped(stream(), L.BF16, L.GEMM_DX, probs, P))
                torch.cuda.synchronize()
                res.append(out)
            outs.append(res)
        finally:
            L.check(L.lib.dmvae_debug_set_knob(13, 1))
    for r in range(reps):
        a, b = outs[0][r], outs[1][r]
        assert bool((b[M:] == SENT).all()) and bool((b[:, P * N:] == SENT).all()), "stray store outside the output"
        assert torch.equal(a, b), (M, Ks, N, r, int((a != b).sum().item()))
    # and against the definition (fp32 accumulate over bf16 products, gate, one rounding to bf16)
    b = outs[1][0]
    for i in range(P):
        ref = (dY[i].float() @ W[i].float().t()) * (gate[:, i * N:(i + 1) * N].float() > 0)
        got = b[:M, i * N:(i + 1) * N].float()
        assert torch.allclose(got, ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 8), (M, Ks[i])
        assert bool(((got == 0) | (gate[:, i * N:(i + 1) * N].float() > 0)).all())      # closed units are exact zeros


# M chosen so that the walk of a workgroup has 1, 2, 3, 4, 5, 6 and 11 steps (every branch of the step dispatcher: single step, first + last,
# the two-steps-per-iteration loop once and twice, both tails), and cases whose last row chunk is ragged (a workgroup with a single
# step inside a multi-step launch); K = 64 / 128 / 256 are the kernel's three bodies, K = 512 stays on the grouped tiles
@pytest.mark.parametrize("M,Ks,N", [(64, (128, 64), 2048), (128, (128, 64), 2048), (3072, (128, 64), 2048), (4096, (128, 64), 2048),
                                    (5120, (128, 64), 2048), (6144, (128, 64), 2048), (7168, (256, 64), 2048), (3392, (128, 64), 2048), (192, (64,), 128),
                                    (1024, (256,), 256), (2048, (512, 64), 2048), (640, (128, 128), 384)])
def test_heads_dx_stream_equals_the_grouped_kernel(hip, M, Ks, N):
    """csrc/heads_dx.hip against gemm_bf16_grouped_kernel on the same operands: bit-identical (the same MFMA chain per output
    element, k ascending), nothing written outside the output, closed ReLU units exact zeros."""
    _heads_dx_case(hip, M, Ks, N, seed=M + sum(Ks))


def test_heads_dx_stream_repeated_launches_stay_identical(hip):
    """a synchronisation fault (a slot reused before every wave has read it, a gate consumed before it landed) shows as rare
    mismatches that come and go: the same launch 25 times at the metric's shape, every result compared with the grouped kernel's"""
    _heads_dx_case(hip, 4096, (128, 64), 2048, seed=99, reps=25)
