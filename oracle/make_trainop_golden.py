"""Golden record of ONE epoch loop as the reference's own VAE.train_op runs it
(code/base_models.py:112-132), with the reference's own Dataset (includes/utils.py:428-466) and
latent-variable samplers (priors.py:67-68, :157-158) -- executed from /root/reference under the
NumPy stand-in for `tensorflow`, against a MOCK session that records every feed and returns a
known per-batch "loss".  TEST INFRASTRUCTURE ONLY; build container only; nothing is copied.

Pins SURVEY 8a rows A14 (epoch loop: loss = sum(batch_loss) / epoch_len, one noise draw per batch,
short last batch), A2's call order (C before Z, per batch, after the epoch's shuffle) and A15
(kl_ratio fed per batch).

    python oracle/make_trainop_golden.py      # writes tests/golden/trainop_golden.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"
OUT = os.path.join(HERE, "..", "tests", "golden", "trainop_golden.npz")


class MockSession:
    def __init__(self, model):
        self.model, self.feeds = model, []

    def run(self, fetches, feed_dict=None):
        m = self.model
        assert fetches == [m.loss, m.train_step]
        X = feed_dict[m.X]
        self.feeds.append(dict(X=np.array(X), kl=float(feed_dict[m.kl_ratio]), training=bool(feed_dict[m.is_training]),
                               eps_C=np.array(feed_dict["epsilon_C"]), eps_Z=np.array(feed_dict["epsilon_Z"])))
        return float(np.sum(X) + 0.5 * len(X)), None          # a known function of the batch


def main():
    sys.path.insert(0, HERE)
    import np_tf_ops
    sys.modules["tensorflow"] = np_tf_ops
    from scipy.optimize import linear_sum_assignment
    la = types.ModuleType("sklearn.utils.linear_assignment_")
    la.linear_assignment = lambda c: np.stack(linear_sum_assignment(c), axis=1)
    sys.modules["sklearn.utils.linear_assignment_"] = la
    sys.modules["includes.visualization"] = types.ModuleType("includes.visualization")
    sys.path.insert(0, REF)
    import priors
    import base_models
    from includes.utils import Dataset

    N, I, D, K, Bsz = 23, 7, 3, 4, 5
    rng = np.random.RandomState(77)
    data = rng.rand(N, I)
    classes = np.arange(N) % K
    np.random.seed(123)
    ds = Dataset((data, classes), batch_size=Bsz)
    model = base_models.VAE("m", "binary", I, D)
    model.X, model.loss, model.train_step = "X", "loss", "train_step"
    # the dict base_models.py:256-274 builds: C first, then Z
    model.latent_variables = {
        "C": (priors.DiscreteFactorial("cluster", 1, K), "epsilon_C", {}),
        "Z": (priors.NormalMixtureFactorial("representation", D, K), "epsilon_Z", {}),
    }
    np.random.seed(321)
    blob = {"data": data, "classes": classes, "dims": np.array([N, I, D, K, Bsz], dtype=np.int64),
            "seed_dataset": np.int64(123), "seed_epoch": np.int64(321)}
    for ep in range(2):
        sess = MockSession(model)
        loss = model.train_op(sess, ds, kl_ratio=0.25 + ep)
        blob["ep%d_loss" % ep] = np.float64(loss)
        blob["ep%d_n_batches" % ep] = np.int64(len(sess.feeds))
        for bi, f in enumerate(sess.feeds):
            for k in ("X", "eps_C", "eps_Z"):
                blob["ep%d_b%d_%s" % (ep, bi, k)] = f[k]
            blob["ep%d_b%d_kl" % (ep, bi)] = np.float64(f["kl"])
            assert f["training"] is True
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
