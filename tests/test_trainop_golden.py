"""Epoch-loop semantics against a record of the reference's OWN VAE.train_op (code/base_models.py:112-132)
run with its own Dataset and samplers under a mock session (oracle/make_trainop_golden.py):
batch composition and order over two epochs, one (C, then Z) noise draw per batch, the short last
batch, kl_ratio per batch, loss = sum(batch_loss) / epoch_len."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tg():
    return np.load(os.path.join(ROOT, "tests", "golden", "trainop_golden.npz"))


def _check_epochs(tg, ds, draw, epoch_len, dtype=np.float64, rel=1e-12):
    np.random.seed(int(tg["seed_epoch"]))
    for ep in range(2):
        loss, nb = 0.0, 0
        for bi, batch in enumerate(ds.get_batches()):
            eps_C, eps_Z = draw(len(batch))
            np.testing.assert_array_equal(np.asarray(batch, dtype=dtype), tg["ep%d_b%d_X" % (ep, bi)].astype(dtype))
            np.testing.assert_array_equal(eps_C, tg["ep%d_b%d_eps_C" % (ep, bi)])
            np.testing.assert_array_equal(eps_Z, tg["ep%d_b%d_eps_Z" % (ep, bi)])
            assert float(tg["ep%d_b%d_kl" % (ep, bi)]) == 0.25 + ep
            loss += (float(np.sum(batch)) + 0.5 * len(batch)) / epoch_len
            nb += 1
        assert nb == int(tg["ep%d_n_batches" % ep]) == 5
        assert loss == pytest.approx(float(tg["ep%d_loss" % ep]), rel=rel)


def test_oracle_epoch_stream_matches_reference_train_op(tg):
    import dmvae_oracle as O
    N, I, D, K, Bsz = (int(v) for v in tg["dims"])
    np.random.seed(int(tg["seed_dataset"]))
    ds = O.Dataset((tg["data"], tg["classes"]), batch_size=Bsz)
    _check_epochs(tg, ds, lambda n: (O.sample_gumbel((n, 1, K)), np.random.randn(n, D)), ds.epoch_len)


def test_dropin_host_classes_match_reference_train_op(tg):
    """the product's host side: includes.utils.Dataset + VAE.sample_reparametrization_variables + priors samplers"""
    import base_models
    import priors
    from includes.utils import Dataset
    N, I, D, K, Bsz = (int(v) for v in tg["dims"])
    np.random.seed(int(tg["seed_dataset"]))
    ds = Dataset((tg["data"], tg["classes"]), batch_size=Bsz)
    vae = base_models.VAE("m", "binary", I, D)
    vae.latent_variables = {"C": (priors.DiscreteFactorial("cluster", 1, K), "epsilon_C", {}),
                            "Z": (priors.NormalMixtureFactorial("representation", D, K), "epsilon_Z", {})}

    def draw(n):
        feed = vae.sample_reparametrization_variables(n)
        assert list(feed) == ["epsilon_C", "epsilon_Z"]          # C is drawn first
        return feed["epsilon_C"], feed["epsilon_Z"]
    _check_epochs(tg, ds, draw, ds.epoch_len, np.float32, 1e-6)      # the drop-in keeps the rows in float32 (what the GPU path consumes)
