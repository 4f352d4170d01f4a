"""Closed-form known answers and gradient checks of the oracle's
hand-derived backward against torch autograd float64."""
import math

import numpy as np
import pytest
import torch

import dmvae_oracle as O


def small_cfg(**kw):
    d = dict(input_dim=32, latent_dim=4, n_classes=3, enc_layers=(12, 10),
             head_dim=14, dec_layers=(14, 10, 9))
    d.update(kw)
    return O.Config(**d)


def test_param_count_matches_survey():
    # SURVEY 8d quotes P = 4.373 M (cfg1), 4.698 M (cfg2), 5.955 M (cfg4)
    assert abs(O.Config(784, 10, 10).n_params() - 4.373e6) < 2e3
    assert abs(O.Config(784, 64, 10).n_params() - 4.698e6) < 2e3
    assert abs(O.Config(784, 256, 50).n_params() - 5.955e6) < 2e3


def test_init_statistics():
    cfg = O.Config(784, 10, 10)
    p = O.init_params(cfg, 0)
    lim = math.sqrt(6.0 / (784 + 500))
    assert np.abs(p["W_enc0"]).max() <= lim and np.abs(p["W_enc0"]).max() > 0.99 * lim
    assert not p["b_enc0"].any() and not p["b_out"].any()
    # FullyConnected bias (1,out) is xavier too: U(+-sqrt(6/(1+out)))
    lb = math.sqrt(6.0 / (1 + 2000))
    assert 0.9 * lb < np.abs(p["b_dec0"]).max() <= lb
    assert abs(p["prior_means"].std() - 1.0) < 0.25 and not p["prior_log_vars"].any()


def test_recon_logits_zero_is_I_ln2():
    cfg = O.Config(784, 10, 10)
    X = np.random.RandomState(0).rand(5, 784)
    assert O.recon_loss(cfg, X, np.zeros((5, 784))) == pytest.approx(784 * math.log(2), rel=1e-14)


def test_kl_c_known_answers():
    assert O.kl_categorical(np.zeros((4, 10)), 10) == pytest.approx(0.0, abs=1e-15)
    lg = np.full((3, 10), -200.0)
    lg[:, 2] = 200.0
    assert O.kl_categorical(lg, 10) == pytest.approx(math.log(10), rel=1e-12)


def test_kl_z_known_answers():
    rng = np.random.RandomState(1)
    mu, lv = rng.randn(6, 5), rng.randn(6, 5) * 0.3
    # K=1, prior N(0,I) == NormalFactorial
    one = np.ones((6, 1))
    assert O.kl_mixture_exact(mu, lv, one, np.zeros((1, 5)), np.zeros((1, 5))) == \
        pytest.approx(O.kl_normal(mu, lv), rel=1e-13)
    # posterior == component k, one-hot weights -> 0
    pm, plv = rng.randn(3, 5), rng.randn(3, 5)
    w = np.eye(3)[[1] * 6]
    assert O.kl_mixture_exact(np.tile(pm[1], (6, 1)), np.tile(plv[1], (6, 1)), w, pm, plv) == \
        pytest.approx(0.0, abs=1e-13)
    # exact == relaxed for one-hot weights
    w = np.eye(3)[rng.randint(0, 3, 6)]
    assert O.kl_mixture_exact(mu, lv, w, pm, plv) == \
        pytest.approx(O.kl_mixture_relaxed(mu, lv, w, pm, plv), rel=1e-13)


def test_eps_zero_gives_mean_and_tau_to_zero_gives_onehot():
    rng = np.random.RandomState(2)
    mu, lv = rng.randn(4, 3), rng.randn(4, 3)
    np.testing.assert_array_equal(O.gaussian_reparam(mu, lv, np.zeros((4, 3))), mu)
    lg, g = rng.randn(5, 6), O.sample_gumbel((5, 6), rng)
    z = O.gumbel_softmax(lg, g, 1e-4)
    np.testing.assert_allclose(z, np.eye(6)[np.argmax(lg + g, 1)], atol=1e-12)


def test_adam_first_step_is_lr_sign():
    p = {"a": np.array([1.0, -2.0, 3.0])}
    g = {"a": np.array([0.3, -5.0, 1e-3])}
    m, v = O.adam_tf_init(p)
    p0 = p["a"].copy()
    O.adam_tf(p, g, m, v, 1, lr=0.002)
    # delta = -lr*sqrt(1-b2)/(1-b1) * (1-b1) g / (sqrt((1-b2) g^2) + eps)
    exp = -0.002 * math.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g["a"]) / (
        np.sqrt(0.001 * g["a"] ** 2) + 1e-8)
    np.testing.assert_allclose(p["a"] - p0, exp, rtol=1e-12)
    np.testing.assert_allclose(p["a"] - p0, -0.002 * np.sign(g["a"]), rtol=1e-3)


def torch_loss(pt, cfg, X, eps, kl_ratio, mode, gumbel, tau):
    """Independent autograd formulation written from the reference formulas."""
    def dense(x, n, relu):
        y = x @ pt["W_" + n] + pt["b_" + n]
        return torch.relu(y) if relu else y
    h = X
    if getattr(cfg, "cnn", False):      # base_models.py:176-216 with torch's own conv / pool
        F = torch.nn.functional
        h = X.reshape(-1, 1, 28, 28)
        for name, ci, co, hw, pool in cfg.conv_table():
            W = pt["W_" + name].reshape(3, 3, ci, co).permute(3, 2, 0, 1)      # HWIO -> OIHW
            h = torch.relu(F.conv2d(h, W, pt["b_" + name], padding=1))
            if pool:
                if h.shape[-1] % 2:                                          # SAME: pad bottom / right
                    h = F.pad(h, (0, 1, 0, 1), value=float("-inf"))
                h = F.max_pool2d(h, 2)
        h = h.permute(0, 2, 3, 1).reshape(X.shape[0], -1)                    # flatten (h, w, c)
    for i in range(len(cfg.enc_layers)):
        h = dense(h, "enc%d" % i, True)
    hz = dense(h, "zh", True)
    mean, lv = dense(hz, "mean", False), dense(hz, "logvar", False)
    logits = dense(dense(h, "ch", True), "logits", False)
    Z = mean + torch.exp(lv / 2) * eps
    d = Z
    for i in range(len(cfg.dec_layers)):
        d = dense(d, "dec%d" % i, True)
    xl = dense(d, "out", False)
    if cfg.input_type == "binary":
        recon = torch.nn.functional.binary_cross_entropy_with_logits(
            xl, X, reduction="none").sum(1).mean()
    else:
        recon = 0.5 * ((X - xl) ** 2).sum(1).mean()
    q = torch.softmax(logits, 1)
    klc = (q * (torch.log(q + 1e-20) - math.log(1.0 / cfg.n_classes))).sum(1).mean()
    pm, plv = pt["prior_means"], pt["prior_log_vars"]
    if mode == "exact":
        t = plv[None] - lv[:, None] - 1 + (torch.exp(lv)[:, None] + (mean[:, None] - pm[None]) ** 2) / torch.exp(plv)[None]
        klz = (0.5 * (t.sum(-1) * q).sum(-1)).mean()
    else:
        w = torch.softmax((logits + gumbel) / tau, 1)
        bm, bl = w @ pm, w @ plv
        klz = (0.5 * (bl - lv - 1 + (torch.exp(lv) + (mean - bm) ** 2) / torch.exp(bl)).sum(1)).mean()
    return recon + kl_ratio * (klc + klz), recon, klc, klz


@pytest.mark.parametrize("mode", ["exact", "relaxed"])
@pytest.mark.parametrize("input_type", ["binary", "real"])
def test_backward_matches_autograd(mode, input_type):
    cfg = small_cfg(input_type=input_type)
    rng = np.random.RandomState(7)
    p = O.init_params(cfg, 3)
    p["prior_log_vars"] = rng.randn(3, 4) * 0.4
    for k in p:
        if k.startswith("b_"):
            p[k] = rng.randn(*p[k].shape) * 0.1
    B = 8
    X = rng.rand(B, 32)
    eps = rng.randn(B, 4)
    gum = O.sample_gumbel((B, 3), rng)
    a = O.forward(p, cfg, X, eps, 0.7, mode, gum, 0.5)
    g = O.backward(p, cfg, a)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    loss, recon, klc, klz = torch_loss(pt, cfg, torch.tensor(X), torch.tensor(eps), 0.7, mode,
                                       torch.tensor(gum), 0.5)
    loss.backward()
    assert a["loss"] == pytest.approx(loss.item(), rel=1e-12)
    assert a["recon"] == pytest.approx(recon.item(), rel=1e-12)
    assert a["kl_c"] == pytest.approx(klc.item(), rel=1e-10)
    assert a["kl_z"] == pytest.approx(klz.item(), rel=1e-12)
    assert set(g) == set(p)
    for k in p:
        np.testing.assert_allclose(g[k], pt[k].grad.numpy(), rtol=1e-9, atol=1e-13, err_msg=k)


def test_cnn_trunk_backward_matches_autograd():
    """the conv / max-pool trunk (base_models.py:176-216) in the oracle against torch's
    conv2d / max_pool2d and autograd, float64, whole loss."""
    cfg = O.Config(784, 6, 4, enc_layers=(24,), head_dim=20, dec_layers=(20, 16), cnn=True)
    assert cfg.n_params() == sum(v.size for v in O.init_params(cfg, 0).values())
    rng = np.random.RandomState(11)
    p = O.init_params(cfg, 5)
    B = 3
    X = O.synthetic_images(B, 784, seed=2).astype(np.float64)
    eps = rng.randn(B, 6)
    a = O.forward(p, cfg, X, eps, 1.0, "exact")
    g = O.backward(p, cfg, a)
    assert a["flat"].shape == (B, 2048) and a["pool5"].shape == (B, 4, 4, 128)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    loss, recon, klc, klz = torch_loss(pt, cfg, torch.tensor(X), torch.tensor(eps), 1.0, "exact", None, 1.0)
    loss.backward()
    assert a["loss"] == pytest.approx(loss.item(), rel=1e-12)
    assert set(g) == set(p)
    for k in p:
        np.testing.assert_allclose(g[k], pt[k].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=k)
    # xavier fans of the conv kernels / biases (includes/layers.py:39-51)
    assert np.abs(p["W_conv3"]).max() <= math.sqrt(6.0 / (9 * 64 + 9 * 64))
    assert np.abs(p["b_conv3"]).max() <= math.sqrt(6.0 / (64 + 64))


def test_central_difference_spot_check():
    cfg = small_cfg()
    rng = np.random.RandomState(9)
    p = O.init_params(cfg, 1)
    X, eps = rng.rand(5, 32), rng.randn(5, 4)
    g = O.backward(p, cfg, O.forward(p, cfg, X, eps))
    for k, idx in (("W_enc0", (3, 2)), ("prior_means", (1, 2)), ("prior_log_vars", (2, 0)),
                   ("W_logits", (5, 1)), ("b_dec1", (4,))):
        h = 1e-6
        q = {n: v.copy() for n, v in p.items()}
        q[k][idx] += h
        lp = O.forward(q, cfg, X, eps)["loss"]
        q[k][idx] -= 2 * h
        lm = O.forward(q, cfg, X, eps)["loss"]
        assert g[k][idx] == pytest.approx((lp - lm) / (2 * h), rel=2e-5, abs=1e-9)


def test_adam_matches_tf_formula_over_steps_and_torch_differs_only_by_eps_placement():
    cfg = small_cfg()
    rng = np.random.RandomState(4)
    p = O.init_params(cfg, 2)
    m, v = O.adam_tf_init(p)
    X, eps = rng.rand(6, 32), rng.randn(6, 4)
    losses = []
    for t in range(1, 6):
        a, _ = O.train_step(p, m, v, t, cfg, X, eps, lr=0.002)
        losses.append(a["loss"])
    assert losses[-1] < losses[0]


def test_data_parallel_equivalence():
    """mean over 2 equal shards of per-shard gradients == full-batch gradient
    (SURVEY 8e): the identity the RCCL all-reduce path relies on."""
    cfg = small_cfg()
    rng = np.random.RandomState(5)
    p = O.init_params(cfg, 5)
    X, eps = rng.rand(8, 32), rng.randn(8, 4)
    full = O.backward(p, cfg, O.forward(p, cfg, X, eps))
    parts = [O.backward(p, cfg, O.forward(p, cfg, X[s], eps[s])) for s in (slice(0, 4), slice(4, 8))]
    for k in full:
        np.testing.assert_allclose(0.5 * (parts[0][k] + parts[1][k]), full[k], rtol=1e-10, atol=1e-14)
