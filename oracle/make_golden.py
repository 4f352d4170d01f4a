"""Golden-vector generator -- runs ONLY in the build container, where
/root/reference exists.  TEST INFRASTRUCTURE ONLY.

Executes the reference's own code/priors.py and code/includes/utils.py
(imported from where they lie; nothing is copied) with the `tensorflow` module
name bound to oracle/np_tf_ops.py, and writes inputs + the reference's outputs
to tests/golden/priors_golden.npz.  Covers SURVEY 8a rows A1 (Dataset), A2
(noise samplers), A6, A6b, A9 (both branches), A10, plus get_cluster_probs and
get_clustering_accuracy.

    python oracle/make_golden.py            # regenerates the fixture
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"
OUT = os.path.join(HERE, "..", "tests", "golden", "priors_golden.npz")


def import_reference():
    sys.path.insert(0, HERE)
    import np_tf_ops
    sys.modules["tensorflow"] = np_tf_ops
    # sklearn.utils.linear_assignment_ was removed from scikit-learn >= 0.23;
    # the reference calls linear_assignment(cost) -> array of (row, col) pairs.
    from scipy.optimize import linear_sum_assignment
    la = types.ModuleType("sklearn.utils.linear_assignment_")
    la.linear_assignment = lambda c: np.stack(linear_sum_assignment(c), axis=1)
    sys.modules["sklearn.utils.linear_assignment_"] = la
    # includes/visualization.py is plotting only (matplotlib/TSNE); never called here.
    sys.modules["includes.visualization"] = types.ModuleType("includes.visualization")
    sys.path.insert(0, REF)
    import priors
    from includes import utils
    return priors, utils


def main():
    priors, utils = import_reference()
    out = {}
    cases = [(4, 3, 5), (8, 16, 5), (100, 10, 10), (16, 64, 10), (6, 40, 50)]
    ci = 0
    for (B, D, K) in cases:
        for seed in (0, 1):
            rng = np.random.RandomState(1000 * ci + seed)
            mean = rng.randn(B, D) * 1.5
            log_var = rng.randn(B, D) * 0.7 - 0.3
            logits = rng.randn(B, K) * 2.0
            pm = rng.randn(K, D)
            plv = rng.randn(K, D) * 0.5
            mix = priors.NormalMixtureFactorial("representation", D, K)
            mix.means = pm
            mix.log_vars = plv
            disc = priors.DiscreteFactorial("cluster", 1, K)
            # reference noise samplers, in the reference's call order (C then Z:
            # dict order of base_models.py:256-274) on a seeded global RNG
            np.random.seed(77 + ci * 10 + seed)
            g = disc.sample_reparametrization_variable(B)          # (B,1,K)
            eps = mix.sample_reparametrization_variable(B)         # (B,D)
            pre = "c%d_s%d_" % (ci, seed)
            out[pre + "shape"] = np.array([B, D, K])
            out[pre + "np_seed"] = np.array(77 + ci * 10 + seed)
            out[pre + "mean"] = mean
            out[pre + "log_var"] = log_var
            out[pre + "logits"] = logits
            out[pre + "prior_means"] = pm
            out[pre + "prior_log_vars"] = plv
            out[pre + "gumbel"] = g
            out[pre + "eps"] = eps
            out[pre + "Z"] = mix.inverse_reparametrize(
                eps, {"mean": mean, "log_var": log_var})
            w = np.exp(logits - logits.max(1, keepdims=True))
            w = w / w.sum(1, keepdims=True)
            out[pre + "w"] = w
            out[pre + "kl_z_exact"] = np.array(mix.kl_from_prior({
                "mean": mean, "log_var": log_var, "weights": w,
                "cluster_sample": False}))
            out[pre + "kl_c"] = np.array(disc.kl_from_prior({"logits": logits}))
            out[pre + "kl_c_probs"] = np.array(disc.kl_from_prior({"probs": w}))
            out[pre + "cluster_probs"] = mix.get_cluster_probs(out[pre + "Z"])
            for ti, tau in enumerate((1.0, 0.5)):
                zeta = disc.inverse_reparametrize(
                    g, {"logits": logits, "temperature": tau})     # (B,1,K)
                out[pre + "zeta_t%d" % ti] = zeta
                out[pre + "kl_z_relaxed_t%d" % ti] = np.array(mix.kl_from_prior({
                    "mean": mean, "log_var": log_var, "weights": zeta,
                    "cluster_sample": True}))
            # K=1 known answer partner: NormalFactorial
            nf = priors.NormalFactorial("n", D)
            out[pre + "kl_normal"] = np.array(nf.kl_from_prior(
                {"mean": mean, "log_var": log_var}))
        ci += 1
    out["n_cases"] = np.array(ci)

    # sample_gumbel on a seeded global RNG (includes/utils.py:17-19)
    np.random.seed(5)
    out["gumbel_seed5"] = utils.sample_gumbel((7, 1, 4))

    # Dataset epoch semantics (includes/utils.py:428-466): rows are tagged by
    # their first column so the emitted order can be read back.
    N, Bsz = 23, 5
    data = np.arange(N, dtype=np.float64)[:, None] * np.ones((1, 3))
    classes = np.arange(N) % 4
    np.random.seed(11)
    ds = utils.Dataset((data, classes), batch_size=Bsz)
    out["ds_epoch_len"] = np.array(ds.epoch_len)
    for ep in range(2):
        order = np.concatenate([b[:, 0] for b in ds.get_batches()])
        out["ds_order_ep%d" % ep] = order.astype(np.int64)
    np.random.seed(11)
    ds2 = utils.Dataset((data, classes), batch_size=Bsz)
    out["ds_batch_sizes"] = np.array([len(b) for b in ds2.get_batches()])

    # clustering accuracy (includes/utils.py:22-34)
    rng = np.random.RandomState(3)
    wts = rng.rand(200, 6)
    cls = rng.randint(0, 6, 200)
    out["acc_weights"] = wts
    out["acc_classes"] = cls
    out["acc_value"] = np.array(utils.get_clustering_accuracy(wts, cls))
    perm = np.array([2, 0, 1, 5, 3, 4])
    onehot = np.eye(6)[perm[cls]]
    out["acc_perm_value"] = np.array(utils.get_clustering_accuracy(onehot, cls))

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
