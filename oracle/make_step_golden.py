"""Full-step golden vectors (SURVEY 8c "Golden fixtures to commit", second list) from the float64
restatement oracle/dmvae_oracle.py.  TEST INFRASTRUCTURE ONLY.

These rows of the path (dense layers, sigmoid cross entropy, autodiff, TF-Adam: SURVEY 8a rows A0,
A3-A5, A7, A8, A12, A13) live in TensorFlow 1.x, which cannot run here and for which the reference
holds no test or golden value: the fixture pins the ORACLE (a later edit of the restatement cannot
drift unnoticed) and gives the GPU tests a fixed target that does not execute the oracle at run
time; it does not pin the oracle to TensorFlow ("parity unpinned").

Cases: (B=8, I=32) and (B=4, I=784), D=4, K=3, small hidden layers; exact and relaxed mixture KL;
every parameter, the batch, the noise, loss / recon / KL terms, every gradient, and the parameters
after 1 and 3 TF-Adam steps on the same batch.

    python oracle/make_step_golden.py        # regenerates tests/golden/step_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dmvae_oracle as O

OUT = os.path.join(HERE, "..", "tests", "golden", "step_golden.npz")
CASES = [dict(B=8, I=32, D=4, K=3, enc=(24, 20), head=28, dec=(28, 20, 16)),
         dict(B=4, I=784, D=4, K=3, enc=(40, 24), head=32, dec=(32, 24, 40))]


def run_case(c, mode, seed):
    cfg = O.Config(c["I"], c["D"], c["K"], c["enc"], c["head"], c["dec"])
    rng = np.random.RandomState(seed)
    p = O.init_params(cfg, seed)
    p["prior_log_vars"] = rng.randn(*p["prior_log_vars"].shape) * 0.3
    for k in p:          # float32-representable values: the GPU engine holds float32 masters
        p[k] = p[k].astype(np.float32).astype(np.float64)
    X = (rng.rand(c["B"], c["I"]) * (rng.rand(c["B"], c["I"]) < 0.3)).astype(np.float32).astype(np.float64)
    eps = rng.randn(c["B"], c["D"]).astype(np.float32).astype(np.float64)
    gum = O.sample_gumbel((c["B"], c["K"]), rng).astype(np.float32).astype(np.float64)
    out = {"X": X, "eps": eps, "gumbel": gum}
    for k, v in p.items():
        out["p0_" + k] = v
    a = O.forward(p, cfg, X, eps, 0.8, mode, gum, 0.5)
    g = O.backward(p, cfg, a)
    for k in ("loss", "recon", "kl_z", "kl_c"):
        out[k] = np.float64(a[k])
    for k in ("mean", "logvar", "logits", "Z", "w", "xlogits"):
        out["fwd_" + k] = a[k]
    for k, v in g.items():
        out["g_" + k] = v
    m, v = O.adam_tf_init(p)
    for t in (1, 2, 3):
        a_t, _ = O.train_step(p, m, v, t, cfg, X, eps, 0.8, 0.002, mode, gum, 0.5)
        out["loss_t%d" % t] = np.float64(a_t["loss"])
        if t == 3:
            for k, val in p.items():
                out["p%d_%s" % (t, k)] = val.copy()
    return out


def main():
    blob = {"n_cases": np.int64(len(CASES))}
    for ci, c in enumerate(CASES):
        blob["c%d_dims" % ci] = np.array([c["B"], c["I"], c["D"], c["K"], c["head"]], dtype=np.int64)
        blob["c%d_enc" % ci] = np.array(c["enc"], dtype=np.int64)
        blob["c%d_dec" % ci] = np.array(c["dec"], dtype=np.int64)
        for mode in ("exact", "relaxed"):
            for k, v in run_case(c, mode, 100 + ci).items():
                # scalars stay float64; tensors are stored as float32 (inputs are float32-representable,
                # outputs are compared at 1e-4..1e-6) to keep the fixture small
                blob["c%d_%s_%s" % (ci, mode, k)] = v if np.ndim(v) == 0 else np.asarray(v, dtype=np.float32)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
