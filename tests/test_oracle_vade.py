"""CPU tests of the VaDE restatement in the oracle (code/base_models.py:435-562): its hand-derived
backward -- including the path through get_cluster_probs(Z) into mean / log_var and the prior tables -- against
torch autograd in float64, and its building blocks against the reference-generated golden vectors."""
import numpy as np
import pytest
import torch

import dmvae_oracle as O


def torch_vade_loss(tp, cfg, X, eps, r):
    Xt, et = torch.tensor(X), torch.tensor(eps)
    h = Xt
    if cfg.cnn:                          # base_models.py:455-488 with torch's own conv2d / max_pool2d
        F = torch.nn.functional
        h = Xt.reshape(-1, 1, 28, 28)
        for name, ci, co, hw, pool in cfg.conv_table():
            W = tp["W_" + name].reshape(3, 3, ci, co).permute(3, 2, 0, 1)      # HWIO -> OIHW
            h = torch.relu(F.conv2d(h, W, tp["b_" + name], padding=1))
            if pool:
                if h.shape[-1] % 2:                                          # SAME: pad bottom / right
                    h = F.pad(h, (0, 1, 0, 1), value=float("-inf"))
                h = F.max_pool2d(h, 2)
        h = h.permute(0, 2, 3, 1).reshape(Xt.shape[0], -1)                    # flatten (h, w, c)
    for i in range(len(cfg.enc_layers)):
        h = torch.relu(h @ tp["W_enc%d" % i] + tp["b_enc%d" % i])
    mean, lv = h @ tp["W_mean"] + tp["b_mean"], h @ tp["W_logvar"] + tp["b_logvar"]
    Z = mean + torch.exp(lv / 2) * et
    pm, plv = tp["prior_means"], tp["prior_log_vars"]
    # get_cluster_probs, priors.py:91-102
    u = -(((Z[:, None, :] - pm[None]) ** 2 / torch.exp(plv[None])).sum(-1) + plv.sum(-1)[None]) / 2
    gam = torch.softmax(u, dim=1)
    # kl_from_prior exact branch, priors.py:131-145, weights = cluster_probs
    res = plv[None] - lv[:, None, :] - 1 + (torch.exp(lv[:, None, :]) + (mean[:, None, :] - pm[None]) ** 2) / torch.exp(plv[None])
    klz = (0.5 * (res.sum(-1) * gam).sum(-1)).mean()
    # DiscreteFactorial.kl_from_prior "probs" branch, priors.py:183-201
    klc = (gam * (torch.log(gam + 1e-20) - np.log(1.0 / cfg.n_classes))).sum(1).mean()
    h = Z
    for i in range(len(cfg.dec_layers)):
        h = torch.relu(h @ tp["W_dec%d" % i] + tp["b_dec%d" % i])
    l = h @ tp["W_out"] + tp["b_out"]
    if cfg.input_type == "binary":
        rec = (torch.clamp(l, min=0) - l * Xt + torch.log1p(torch.exp(-l.abs()))).sum(1).mean()
    else:
        rec = 0.5 * ((Xt - l) ** 2).sum(1).mean()
    return rec + r * (klc + klz), rec, klz, klc


@pytest.mark.parametrize("input_type", ["binary", "real"])
@pytest.mark.parametrize("shape", [(20, 5, 4, (12, 9), (9, 12), 7), (30, 3, 6, (16,), (8, 8, 8), 11)])
def test_vade_backward_matches_autograd(shape, input_type):
    I, D, K, enc, dec, B = shape
    cfg = O.VadeConfig(I, D, K, enc, dec, input_type)
    p = O.init_params(cfg, 3)
    rng = np.random.RandomState(0)
    p["prior_log_vars"] = rng.randn(K, D) * 0.3
    X, eps = rng.rand(B, I), rng.randn(B, D)
    a = O.vade_forward(p, cfg, X, eps, 0.7)
    g = O.vade_backward(p, cfg, a)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    loss, rec, klz, klc = torch_vade_loss(tp, cfg, X, eps, 0.7)
    loss.backward()
    assert a["loss"] == pytest.approx(loss.item(), rel=1e-12)
    assert (a["recon"], a["kl_z"], a["kl_c"]) == pytest.approx((rec.item(), klz.item(), klc.item()), rel=1e-12)
    assert set(g) == set(p)
    for k in g:
        np.testing.assert_allclose(g[k], tp[k].grad.numpy(), rtol=1e-9, atol=1e-13, err_msg=k)


def test_vade_building_blocks_against_reference_golden_vectors(golden):
    """cluster_probs = the reference's own get_cluster_probs output; the "probs" branch of the categorical KL = the
    reference's DiscreteFactorial.kl_from_prior on it; the exact mixture KL with those weights."""
    for ci in range(int(golden["n_cases"])):
        g = lambda k: golden["c%d_s0_%s" % (ci, k)]
        B, D, K = g("shape")
        cp = O.cluster_probs(g("Z"), g("prior_means"), g("prior_log_vars"))
        np.testing.assert_allclose(cp, g("cluster_probs"), rtol=1e-10, atol=1e-300)
        klc = np.mean(np.sum(g("w") * (np.log(g("w") + 1e-20) - np.log(1.0 / K)), axis=1))
        assert klc == pytest.approx(float(g("kl_c_probs")), rel=1e-10)


def test_vade_adam_step_moves_every_tensor():
    cfg = O.VadeConfig(24, 4, 3, (10, 8), (8, 10))
    p = O.init_params(cfg, 1)
    m, v = O.adam_tf_init(p)
    rng = np.random.RandomState(2)
    before = {k: x.copy() for k, x in p.items()}
    O.vade_train_step(p, m, v, 1, cfg, rng.rand(9, 24), rng.randn(9, 4))
    for k in p:                                       # every VaDE trainable has a gradient (no dead Y head as in DMVAE)
        assert np.abs(p[k] - before[k]).max() > 0, k
    assert cfg.n_params() == sum(x.size for x in p.values())


def test_vade_cnn_backward_matches_autograd():
    """VaDE(cnn=True), base_models.py:456-488: the conv / pool stack ending in ("fc", 2048 -> 128), mean / log_var straight off
    it; the oracle's forward and hand-derived backward (through the conv trunk) against torch conv2d / max_pool2d + autograd."""
    cfg = O.VadeConfig(784, 5, 4, dec_layers=(12, 10), cnn=True)
    assert cfg.enc_layers == (128,) and [t[:3] for t in cfg.layer_table()[:3]] == [("enc0", 2048, 128), ("mean", 128, 5), ("logvar", 128, 5)]
    p = O.init_params(cfg, 2)
    assert cfg.n_params() == sum(v.size for v in p.values())
    assert np.abs(p["b_enc0"]).max() > 0 and not p["b_mean"].any()         # FullyConnected bias xavier, tf.layers.dense bias zero
    rng = np.random.RandomState(4)
    p["prior_log_vars"] = rng.randn(4, 5) * 0.3
    B = 3
    X, eps = O.synthetic_images(B, 784, seed=5).astype(np.float64), rng.randn(B, 5)
    a = O.vade_forward(p, cfg, X, eps, 0.9)
    g = O.vade_backward(p, cfg, a)
    assert a["flat"].shape == (B, 2048) and a["enc0"].shape == (B, 128)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    loss, rec, klz, klc = torch_vade_loss(tp, cfg, X, eps, 0.9)
    loss.backward()
    assert a["loss"] == pytest.approx(loss.item(), rel=1e-12)
    assert set(g) == set(p)
    for k in g:
        np.testing.assert_allclose(g[k], tp[k].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=k)
