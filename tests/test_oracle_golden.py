"""Oracle (oracle/dmvae_oracle.py) against golden vectors produced by the
reference's own priors.py / includes/utils.py (oracle/make_golden.py)."""
import numpy as np
import pytest

import dmvae_oracle as O


def cases(golden):
    for ci in range(int(golden["n_cases"])):
        for s in (0, 1):
            yield "c%d_s%d_" % (ci, s)


def test_reparam_and_kl_match_reference(golden):
    n = 0
    for pre in cases(golden):
        g = lambda k: golden[pre + k]
        B, D, K = g("shape")
        Z = O.gaussian_reparam(g("mean"), g("log_var"), g("eps"))
        np.testing.assert_allclose(Z, g("Z"), rtol=0, atol=1e-13)
        w = O.softmax(g("logits"))
        np.testing.assert_allclose(w, g("w"), rtol=1e-13)
        klz = O.kl_mixture_exact(g("mean"), g("log_var"), w, g("prior_means"), g("prior_log_vars"))
        np.testing.assert_allclose(klz, g("kl_z_exact"), rtol=1e-12)
        klc = O.kl_categorical(g("logits"), int(K))
        np.testing.assert_allclose(klc, g("kl_c"), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(g("kl_c_probs"), g("kl_c"), rtol=1e-12, atol=1e-14)
        cp = O.cluster_probs(g("Z"), g("prior_means"), g("prior_log_vars"))
        np.testing.assert_allclose(cp, g("cluster_probs"), rtol=1e-10, atol=1e-300)
        for ti, tau in enumerate((1.0, 0.5)):
            zeta = O.gumbel_softmax(g("logits"), g("gumbel").reshape(B, K), tau)
            np.testing.assert_allclose(zeta, g("zeta_t%d" % ti).reshape(B, K), rtol=1e-12)
            klr = O.kl_mixture_relaxed(g("mean"), g("log_var"), zeta,
                                       g("prior_means"), g("prior_log_vars"))
            np.testing.assert_allclose(klr, g("kl_z_relaxed_t%d" % ti), rtol=1e-12)
        np.testing.assert_allclose(O.kl_normal(g("mean"), g("log_var")), g("kl_normal"), rtol=1e-12)
        n += 1
    assert n == 10


def test_noise_samplers_follow_reference_call_order(golden):
    """C (gumbel, shape (n,1,K)) is drawn before Z (randn (n,D)) on the global
    RNG: base_models.py:44-56 over the dict of :256-274."""
    for pre in cases(golden):
        B, D, K = golden[pre + "shape"]
        rng = np.random.RandomState(int(golden[pre + "np_seed"]))
        g = O.sample_gumbel((B, 1, K), rng)
        eps = rng.randn(B, D)
        np.testing.assert_array_equal(g, golden[pre + "gumbel"])
        np.testing.assert_array_equal(eps, golden[pre + "eps"])


def test_sample_gumbel(golden):
    rng = np.random.RandomState(5)
    np.testing.assert_array_equal(O.sample_gumbel((7, 1, 4), rng), golden["gumbel_seed5"])


def test_dataset_epoch_semantics(golden):
    N, Bsz = 23, 5
    data = np.arange(N, dtype=np.float64)[:, None] * np.ones((1, 3))
    classes = np.arange(N) % 4
    rng = np.random.RandomState(11)
    ds = O.Dataset((data, classes), batch_size=Bsz, rng=rng)
    assert ds.epoch_len == int(golden["ds_epoch_len"]) == 5
    for ep in range(2):
        batches = list(ds.get_batches())
        order = np.concatenate([b[:, 0] for b in batches]).astype(np.int64)
        np.testing.assert_array_equal(order, golden["ds_order_ep%d" % ep])
        assert [len(b) for b in batches] == [5, 5, 5, 5, 3]
    np.testing.assert_array_equal(golden["ds_batch_sizes"], [5, 5, 5, 5, 3])


def test_clustering_accuracy(golden):
    acc = O.clustering_accuracy(golden["acc_weights"], golden["acc_classes"])
    assert acc == pytest.approx(float(golden["acc_value"]), abs=1e-15)
    assert float(golden["acc_perm_value"]) == 1.0
