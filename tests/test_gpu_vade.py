"""GPU parity tests of VaDE (code/base_models.py:435-670) on the HIP path: the latent stage
(dmvae_latent_fwd mode 2, csrc/latent_vade.hip) and get_cluster_probs against the oracle and the reference-generated
golden vectors, the whole step (dmvae_config.model = DMVAE_MODEL_VADE) against the float64 oracle, and the class surface.
Tolerances as for the DMVAE step (tests/test_gpu_step.py): fp32 loss |delta| <= 1e-3, gradients <= 1e-4 of the tensor's
max; bf16 loss <= 2e-3 relative."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import dmvae_oracle as O


def make(kw, dtype, B, seed=0):
    from dmvae_hip import StepEngine
    eng = StepEngine(dtype=dtype, max_batch=B, deterministic=True, model="vade", head_dim=64, **kw)
    eng.init_parameters(seed)
    return eng


def ocfg(kw):
    return O.VadeConfig(kw["input_dim"], kw["latent_dim"], kw["n_classes"], kw["enc_layers"], kw["dec_layers"], cnn=kw.get("cnn", False))


REF = dict(input_dim=784, latent_dim=10, n_classes=10, enc_layers=(2000, 500, 500), dec_layers=(500, 500, 2000))
SMALL = dict(input_dim=40, latent_dim=6, n_classes=5, enc_layers=(70, 50), dec_layers=(50, 30, 60))
# VaDE(cnn=True), base_models.py:456-488: conv / pool stack -> ("fc", 2048 -> 128) -> mean / log_var
CNN = dict(input_dim=784, latent_dim=10, n_classes=10, enc_layers=(128,), dec_layers=(500, 500, 2000), cnn=True)


@pytest.mark.parametrize("shape", [(100, 10, 10), (37, 6, 5), (200, 64, 20), (70, 33, 3)])
def test_vade_latent_stage_matches_oracle(shape):
    from dmvae_hip import latent_eval, lib, _lib
    B, D, K = shape
    rng = np.random.RandomState(B + D + K)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    mean, lv, eps = f32(rng.randn(B, D) * 1.2), f32(rng.randn(B, D) * 0.5 - 0.2), f32(rng.randn(B, D))
    pm, plv = f32(rng.randn(K, D)), f32(rng.randn(K, D) * 0.4)
    cfg = O.VadeConfig(4, D, K, (4,), (4,))
    a = dict(mean=mean, logvar=lv, eps=eps, Z=O.gaussian_reparam(mean, lv, eps), kl_ratio=0.6)
    a["w"] = O.cluster_probs(a["Z"], pm, plv)
    p = dict(prior_means=pm, prior_log_vars=plv)
    _, _, dpm, dplv, gmu2, glv2 = O.vade_latent_backward(cfg, a, p, np.zeros_like(mean))
    klz = O.kl_mixture_exact(mean, lv, a["w"], pm, plv)
    klc = np.mean(np.sum(a["w"] * (np.log(a["w"] + 1e-20) + np.log(K)), axis=1))
    # through the C ABI with every output requested
    Bp, ldD = (B + 63) // 64 * 64, (D + 63) // 64 * 64
    dev = lambda x, ld: torch.nn.functional.pad(torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)), (0, ld - x.shape[1], 0, Bp - x.shape[0])).cuda().contiguous()
    md, lvd = dev(mean, ldD), dev(lv, ldD)
    epsd = torch.as_tensor(eps.astype(np.float32)).cuda()
    pmd, plvd = torch.as_tensor(pm.astype(np.float32)).cuda(), torch.as_tensor(plv.astype(np.float32)).cuda()
    Z = torch.full((Bp, ldD), 9.0, device="cuda"); w = torch.zeros((Bp, K), device="cuda")
    gmu, glv, clv = (torch.zeros((Bp, ldD), device="cuda") for _ in range(3))
    nblk = lib.dmvae_latent_nblocks_vade(Bp)
    dpri, lp = torch.zeros((nblk, 2 * K * D), device="cuda"), torch.zeros((nblk, 2), device="cuda")
    la = _lib.LatentArgs()
    la.B, la.B_pad, la.D, la.K, la.mode, la.act_dtype = B, Bp, D, K, 2, _lib.F32
    la.kl_ratio, la.temperature, la.inv_B = 0.6, 1.0, 1.0 / B
    la.mean, la.ld_mean, la.log_var, la.ld_log_var = md.data_ptr(), ldD, lvd.data_ptr(), ldD
    la.eps, la.ld_eps = epsd.data_ptr(), D
    la.prior_means, la.prior_log_vars = pmd.data_ptr(), plvd.data_ptr()
    la.Z_act, la.ld_Z, la.weights, la.ld_w = Z.data_ptr(), ldD, w.data_ptr(), K
    la.gmu, la.glv, la.clv, la.ld_g = gmu.data_ptr(), glv.data_ptr(), clv.data_ptr(), ldD
    la.dprior_partials, la.loss_partials = dpri.data_ptr(), lp.data_ptr()
    _lib.check(lib.dmvae_latent_fwd(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(la)), "dmvae_latent_fwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(Z[:B, :D].cpu().numpy(), a["Z"], rtol=2e-6, atol=2e-6)
    assert not Z[:, D:].any() and not Z[B:].any()
    np.testing.assert_allclose(w[:B].cpu().numpy(), a["w"], rtol=2e-4, atol=1e-7)
    assert lp[:, 0].double().sum().item() / B == pytest.approx(klz, rel=3e-5, abs=1e-5)
    assert lp[:, 1].double().sum().item() / B == pytest.approx(klc, rel=3e-5, abs=1e-6)
    sc = 1.0 / B
    np.testing.assert_allclose(gmu[:B, :D].cpu().numpy(), gmu2, rtol=5e-4, atol=5e-5 * sc)
    np.testing.assert_allclose(glv[:B, :D].cpu().numpy(), glv2, rtol=5e-4, atol=5e-5 * sc)
    dp = dpri.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(dp[:K * D].reshape(K, D), dpm, rtol=5e-4, atol=2e-5)
    np.testing.assert_allclose(dp[K * D:].reshape(K, D), dplv, rtol=5e-4, atol=2e-5)
    # latent_eval's "vade" mode: the same responsibilities
    le = latent_eval(mean, lv, np.zeros((B, K)), pm, plv, eps=eps, mode="vade", kl_ratio=0.6)
    np.testing.assert_allclose(le["weights"], a["w"], rtol=2e-4, atol=1e-7)


def test_get_cluster_probs_against_reference_golden_vectors(golden):
    """priors.NormalMixtureFactorial.get_cluster_probs on the HIP kernel vs the reference's own (priors.py:91-102)"""
    import priors
    for ci in range(int(golden["n_cases"])):
        g = lambda k: golden["c%d_s0_%s" % (ci, k)]
        B, D, K = g("shape")
        mix = priors.NormalMixtureFactorial("representation", int(D), int(K))
        mix.means, mix.log_vars = g("prior_means"), g("prior_log_vars")
        cp = mix.get_cluster_probs(g("Z"))
        np.testing.assert_allclose(cp, g("cluster_probs"), rtol=3e-4, atol=1e-7)
        np.testing.assert_allclose(cp.sum(1), 1.0, atol=1e-5)


@pytest.mark.parametrize("kw,B", [(SMALL, 37), (REF, 100), (CNN, 24)])
def test_vade_fp32_step_matches_oracle(kw, B):
    eng = make(kw, "fp32", B)
    cfg = ocfg(kw)
    rng = np.random.RandomState(1)
    p = {k: v.astype(np.float64) for k, v in eng.get_parameters().items()}
    assert set(p) == set(O.init_params(cfg, 0))
    for k, v in O.init_params(cfg, 0).items():       # the engine's init is the oracle's stream (xavier FullyConnected biases)
        np.testing.assert_array_equal(p[k], v.astype(np.float32).astype(np.float64), err_msg=k)
    p["prior_log_vars"] = (rng.randn(*p["prior_log_vars"].shape) * 0.3).astype(np.float32).astype(np.float64)
    eng.set_parameters(p)
    X = (rng.rand(B, cfg.input_dim) * (rng.rand(B, cfg.input_dim) < 0.3)).astype(np.float32)
    eps = rng.randn(B, cfg.latent_dim).astype(np.float32)
    eng.write_state(kl_ratio=0.8, lr=0.002)
    eng.load_batch(torch.as_tensor(X).cuda(), None, 0, B)
    eng.forward_backward(B, torch.as_tensor(eps).cuda())
    torch.cuda.synchronize()
    a = O.vade_forward(p, cfg, X.astype(np.float64), eps.astype(np.float64), 0.8)
    masks = {k: (v > 0).cpu().numpy() for k, v in eng.hidden_activations(B).items()}
    if cfg.cnn:          # conv activations come back [B, H, W, C] like the oracle's
        assert {k for k in masks if k.startswith("conv")} == {"conv%d" % i for i in range(6)}
    assert {k for k in masks if not k.startswith("conv")} == {"enc%d" % i for i in range(len(cfg.enc_layers))} | {"dec%d" % i for i in range(len(cfg.dec_layers))}
    flips = sum(int((masks[k] != (a[k] > 0)).sum()) for k in masks)
    assert flips <= 1e-4 * sum(mk.size for mk in masks.values()), flips
    g = O.vade_backward(p, cfg, a, masks)
    st = eng.read_state()
    assert abs(st.last_loss - a["loss"]) <= 1e-3, (st.last_loss, a["loss"])
    assert abs(st.last_klz - a["kl_z"]) <= 1e-4 * max(1.0, abs(a["kl_z"]))
    assert abs(st.last_klc - a["kl_c"]) <= 1e-5
    np.testing.assert_allclose(eng.view("mean", B).cpu().numpy(), a["mean"], atol=2e-5)
    np.testing.assert_allclose(eng.view("weights", B).cpu().numpy(), a["w"], atol=2e-5)
    gg = eng.get_gradients()
    assert set(gg) == set(g)
    for k in g:
        scale = np.abs(g[k]).max() + 1e-12
        assert np.abs(gg[k] - g[k]).max() <= (2e-4 if cfg.cnn else 1e-4) * scale, (k, np.abs(gg[k] - g[k]).max(), scale)      # (cnn: the CNN step's tolerance, tests/test_gpu_cnn.py)
    # three Adam steps
    m, v = O.adam_tf_init(p)
    O.adam_tf(p, g, m, v, 1, 0.002)
    eng.update(1.0)
    for t in (2, 3):
        eng.forward_backward(B, torch.as_tensor(eps).cuda())
        eng.update(1.0)
        a2, _ = O.vade_train_step(p, m, v, t, cfg, X.astype(np.float64), eps.astype(np.float64), 0.8, 0.002)
        torch.cuda.synchronize()
        assert abs(eng.read_state().last_loss - a2["loss"]) <= 2e-3
    pg = eng.get_parameters()
    for k in p:
        assert np.percentile(np.abs(pg[k] - p[k]), 99.0) <= 2e-4, k


def test_vade_bf16_step_fused_update_and_graph_replay():
    kw, B, N = REF, 256, 1024
    cfg = ocfg(kw)
    rng = np.random.RandomState(2)
    X = O.synthetic_images(B, 784, seed=3)
    eps = rng.randn(B, 10).astype(np.float32)
    engs = [make(kw, "bf16", B, seed=4) for _ in range(2)]
    p = {k: v.astype(np.float64) for k, v in engs[0].get_parameters().items()}
    a = O.vade_forward(p, cfg, X.astype(np.float64), eps.astype(np.float64))
    for fused, eng in zip((False, True), engs):
        eng.load_batch(torch.as_tensor(X).cuda(), None, 0, B)
        if fused:
            eng.forward_backward_update(B, torch.as_tensor(eps).cuda())
        else:
            eng.forward_backward(B, torch.as_tensor(eps).cuda())
            eng.update(1.0)
    torch.cuda.synchronize()
    st = engs[0].read_state()
    assert abs(st.last_loss - a["loss"]) <= 2e-3 * abs(a["loss"]), (st.last_loss, a["loss"])
    for name in ("param", "m", "v", "param_bf16"):
        assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name
    # graph replay == eager with device noise
    data = torch.as_tensor(O.synthetic_images(N, 784, seed=1)).cuda()
    perm = torch.randperm(N, device="cuda").to(torch.int32)
    out = []
    for use_graph in (False, True):
        eng = make(kw, "bf16", B, seed=4)
        eng.reset_epoch(N // B, kl_ratio=1.0)
        step = eng.capture_step(data, perm) if use_graph else (lambda: eng.train_step(data, perm, use_state_cursor=True))
        for _ in range(N // B):
            step()
        torch.cuda.synchronize()
        out.append((eng.read_state().epoch_loss, eng.param.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
    assert np.isfinite(out[0][0]) and 100 < out[0][0] < 700


def test_vade_class_surface_trains_and_scores(tmp_path):
    import base_models
    from includes.utils import Dataset
    np.random.seed(0)
    rng = np.random.RandomState(0)
    N, K = 600, 4
    cls = rng.randint(0, K, N)
    X = np.clip(O.synthetic_images(N, 784, seed=2) * 0.3 + (np.arange(784)[None, :] % K == cls[:, None]) * 0.7, 0, 1).astype(np.float32)
    data = Dataset((X, cls), batch_size=200)
    model = base_models.VaDE("vade_t", "binary", 784, 8, K, activation="relu", initializer="xavier", batch_size=200, dtype="fp32",
                             enc_layers=(256, 64, 64), dec_layers=(64, 64, 256)).build_graph()
    assert set(model.latent_variables) == {"Z", "C"} and model.latent_variables["C"][1] is None        # C has no noise placeholder
    assert list(model.sample_reparametrization_variables(5)) == [model.epsilon]
    model.path = str(tmp_path / "vade")
    model.define_train_step(0.002, 100)
    model.define_pretrain_step(0.0005)
    model.pretrain(None, data, 2, 5)                 # recon-only Adam, then the GMM initialisation of the prior tables
    losses = [model.train_op(None, data, 1.0) for _ in range(4)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]
    acc = model.get_accuracy(None, data, k=3)
    assert 0.0 <= acc <= 1.0
    mean, log_var = model.encode(X[:50])
    assert mean.shape == (50, 8) and log_var.shape == (50, 8)
    cp = model.cluster_probabilities(X[:50], np.zeros((50, 8), np.float32))
    np.testing.assert_allclose(cp, O.cluster_probs(mean.astype(np.float64), model.latent_variables["Z"][0].means.astype(np.float64),
                                                   model.latent_variables["Z"][0].log_vars.astype(np.float64)), rtol=1e-3, atol=1e-6)
    rec = model.reconstruct(X[:50])
    assert rec.shape == (50, 784) and 0.0 <= rec.min() and rec.max() <= 1.0


def test_vade_cnn_class_surface_and_bf16_step(tmp_path):
    """VaDE(cnn=True) through the class surface (VERDICT r2 missing #3): the reference's spec list drives the plan, the bf16 step
    stays close to the float64 oracle and trains, get_accuracy runs."""
    import base_models
    from includes.utils import Dataset
    np.random.seed(0)
    model = base_models.VaDE("vade_c", "binary", 784, 10, 4, activation="relu", initializer="xavier", cnn=True, batch_size=128,
                             dtype="bf16").build_graph()
    assert model.enc_layers == (128,) and model.cnn
    kinds = [type(l).__name__ for l in model.encoder_network.layers]
    assert kinds == ["Convolution", "Convolution", "MaxPooling"] * 3 + ["FullyConnected"]
    assert (model.encoder_network.layers[-1].input_dim, model.encoder_network.layers[-1].output_dim) == (2048, 128)
    eng = model.engine
    assert eng.tensors["W_enc0"][1:3] == (2048, 128) and eng.tensors["W_mean"][1:3] == (128, 10) and "W_zh" not in eng.tensors
    B = 128
    X = O.synthetic_images(B, 784, seed=6)
    eps = np.random.RandomState(1).randn(B, 10).astype(np.float32)
    cfg = O.VadeConfig(784, 10, 4, cnn=True)
    p = {k: v.astype(np.float64) for k, v in eng.get_parameters().items()}
    assert set(p) == set(O.init_params(cfg, 0))
    a = O.vade_forward(p, cfg, X.astype(np.float64), eps.astype(np.float64))
    eng.load_batch(torch.as_tensor(X).cuda(), None, 0, B)
    eng.forward_backward(B, torch.as_tensor(eps).cuda())
    torch.cuda.synchronize()
    st = eng.read_state()
    assert abs(st.last_loss - a["loss"]) <= 5e-3 * abs(a["loss"]), (st.last_loss, a["loss"])       # bf16 conv trunk: the CNN step's bound (test_gpu_cnn.py)
    g = O.vade_backward(p, cfg, a)
    gg = eng.get_gradients()
    for k in ("W_conv0", "W_conv5", "W_enc0", "W_mean", "W_dec0", "W_out", "prior_means"):
        err = np.linalg.norm(gg[k] - g[k]) / (np.linalg.norm(g[k]) + 1e-30)
        assert err <= 0.15, (k, err)
    cls = np.random.RandomState(2).randint(0, 4, 512)
    Xs = np.clip(O.synthetic_images(512, 784, seed=2) * 0.3 + (np.arange(784)[None, :] % 4 == cls[:, None]) * 0.7, 0, 1).astype(np.float32)
    data = Dataset((Xs, cls), batch_size=128)
    model.path = str(tmp_path / "vade_c")
    model.define_train_step(0.0005, 100)
    losses = [model.train_op(None, data, 1.0) for _ in range(4)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0], losses
    assert 0.0 <= model.get_accuracy(None, data, k=2) <= 1.0
