"""GPU parity tests through the drop-in class surface (base_models.DeepMixtureVAE,
priors.*, train.main) -- the tests a maintainer of the reference would write
against code/base_models.py if it had a test suite."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import dmvae_oracle as O

SMALL = dict(enc_layers=(70, 50), head_dim=90, dec_layers=(90, 50, 30))


def build(input_dim=40, latent_dim=6, n_classes=5, **kw):
    import base_models
    args = dict(SMALL, batch_size=16, dtype="fp32", noise="host", seed=3)
    args.update(kw)
    m = base_models.DeepMixtureVAE("dmvae", "binary", input_dim, latent_dim, n_classes, activation="relu",
                                   initializer="xavier", **args).build_graph()
    m.define_train_step(0.002, 1000, 0.9)
    return m


def test_train_op_epoch_matches_reference_semantics_with_host_noise():
    """One epoch of VAE.train_op (base_models.py:112-132) with the reference's NumPy
    noise stream: identical shuffles, identical epsilon, short last batch, loss =
    sum(batch_loss)/epoch_len -- against the oracle's train_epoch on the same stream."""
    from includes.utils import Dataset
    rng = np.random.RandomState(0)
    N = 16 * 3 + 5
    X = (rng.rand(N, 40) * (rng.rand(N, 40) < 0.4)).astype(np.float32)
    cls = rng.randint(0, 5, N)
    m = build()
    cfg = O.Config(40, 6, 5, SMALL["enc_layers"], SMALL["head_dim"], SMALL["dec_layers"])
    p = {k: v.astype(np.float64) for k, v in m.engine.get_parameters().items()}
    mo, vo = O.adam_tf_init(p)
    np.random.seed(42)
    data = Dataset((X, cls), batch_size=16)
    losses = [m.train_op(None, data, 1.0) for _ in range(2)]
    # oracle on the same global-RNG stream
    np.random.seed(42)
    odata = O.Dataset((X.astype(np.float64), cls), batch_size=16)

    def noise(n):
        O.sample_gumbel((n, 1, 5))                 # C is drawn first (and unused), base_models.py:122-124
        return np.random.randn(n, 6).astype(np.float32).astype(np.float64)
    t = 0
    for ep in range(2):
        lo, t = O.train_epoch(p, mo, vo, t, cfg, odata, noise, 1.0, 0.002)
        assert losses[ep] == pytest.approx(lo, abs=2e-3), (ep, losses[ep], lo)
    assert m.engine.read_state().adam_t == 8
    pg = m.engine.get_parameters()
    for k in p:
        assert np.percentile(np.abs(pg[k] - p[k]), 99) <= 5e-4, k


def test_train_op_device_noise_graph_path_and_relaxed_mode():
    from includes.utils import Dataset
    X = O.synthetic_images(256 * 3 + 50, 784, seed=2)
    cls = np.zeros(len(X), np.int64)
    for gumbel in (False, True):
        import base_models
        m = base_models.DeepMixtureVAE("dmvae", "binary", 784, 10, 10, batch_size=256, dtype="bf16", noise="device",
                                       gumbel=gumbel, temperature=0.7, seed=1).build_graph()
        m.define_train_step(0.002, 1000)
        np.random.seed(1)
        data = Dataset((X, cls), batch_size=256)
        l0 = m.train_op(None, data, 1.0)
        l1 = m.train_op(None, data, 1.0)
        l2 = m.train_op(None, data, 0.5)
        assert np.isfinite([l0, l1, l2]).all() and l1 < l0 and 50 < l1 < 560
        assert m.engine.read_state().adam_t == 12


def test_encode_decode_reconstruct_and_accuracy():
    from includes.utils import Dataset
    m = build(784, 10, 10, batch_size=64, enc_layers=(500, 500), head_dim=2000, dec_layers=(2000, 500, 500))
    cfg = O.Config(784, 10, 10)
    p = {k: v.astype(np.float64) for k, v in m.engine.get_parameters().items()}
    X = O.synthetic_images(150, 784, seed=5)             # 2 full batches + a short one
    mean, log_var, logits = m.encode(X)
    a = O.encode(p, cfg, X.astype(np.float64))
    np.testing.assert_allclose(mean, a["mean"], atol=3e-5)
    np.testing.assert_allclose(log_var, a["logvar"], atol=3e-5)
    np.testing.assert_allclose(logits, a["logits"], atol=3e-5)
    rec = m.reconstruct(X)                                # epsilon = 0 -> Z = mean (visualization.py:39-46)
    xl = O.decode(p, cfg, a["mean"])["xlogits"]
    np.testing.assert_allclose(rec, 1 / (1 + np.exp(-xl)), atol=3e-5)
    Z = np.random.RandomState(1).randn(70, 10)
    np.testing.assert_allclose(m.decode(Z), 1 / (1 + np.exp(-O.decode(p, cfg, Z.astype(np.float32).astype(np.float64))["xlogits"])), atol=3e-5)
    cls = np.random.RandomState(2).randint(0, 10, 150)
    np.random.seed(0)
    acc = m.get_accuracy(None, Dataset((X, cls), batch_size=64))
    assert acc == pytest.approx(O.clustering_accuracy(a["logits"], cls), abs=1e-12)
    feed = m.sample_generative_feed(12, Z={"session": None, "c": np.arange(12) % 10})
    assert feed["Z"].shape == (12, 10) and feed["C"].shape == (12, 1, 10) and (feed["C"].sum(-1) == 1).all()
    eps = m.sample_reparametrization_variables(9)
    assert list(eps) == ["epsilon_C", "epsilon_Z"] and eps["epsilon_C"].shape == (9, 1, 10) and eps["epsilon_Z"].shape == (9, 10)


def test_latent_variable_classes_against_reference_golden_vectors(golden):
    import priors
    for ci in range(int(golden["n_cases"])):
        pre = "c%d_s1_" % ci
        g = lambda k: golden[pre + k]
        B, D, K = (int(v) for v in g("shape"))
        mix = priors.NormalMixtureFactorial("representation", D, K)
        mix.means, mix.log_vars = g("prior_means"), g("prior_log_vars")
        disc = priors.DiscreteFactorial("cluster", 1, K)
        Z = mix.inverse_reparametrize(g("eps"), {"mean": g("mean"), "log_var": g("log_var")})
        np.testing.assert_allclose(Z, g("Z"), rtol=3e-6, atol=3e-6)
        klz = mix.kl_from_prior({"mean": g("mean"), "log_var": g("log_var"), "weights": g("w"), "cluster_sample": False})
        assert klz == pytest.approx(float(g("kl_z_exact")), rel=5e-5)
        assert disc.kl_from_prior({"logits": g("logits")}) == pytest.approx(float(g("kl_c")), rel=5e-5, abs=1e-6)
        assert disc.kl_from_prior({"probs": g("w")}) == pytest.approx(float(g("kl_c")), rel=5e-5, abs=1e-6)
        for ti, tau in enumerate((1.0, 0.5)):
            zeta = disc.inverse_reparametrize(g("gumbel"), {"logits": g("logits"), "temperature": tau})
            assert zeta.shape == (B, 1, K)
            np.testing.assert_allclose(zeta, g("zeta_t%d" % ti), rtol=3e-5, atol=1e-7)
            klr = mix.kl_from_prior({"mean": g("mean"), "log_var": g("log_var"), "weights": g("zeta_t%d" % ti), "cluster_sample": True})
            assert klr == pytest.approx(float(g("kl_z_relaxed_t%d" % ti)), rel=1e-4)
        nf = priors.NormalFactorial("n", D)
        assert nf.kl_from_prior({"mean": g("mean"), "log_var": g("log_var")}) == pytest.approx(float(g("kl_normal")), rel=5e-5)


def test_pretraining_stages_match_reference_semantics(tmp_path):
    """base_models.py:304-423.  Stage 1: Adam(vae_lr) on the reconstruction loss at epsilon = 0;
    stage 2: prior tables from a diagonal GMM on the encoder means, then Adam(prior_lr) on the
    latent loss over the c-head variables only.  Each stage is its own optimizer (fresh slots)."""
    from includes.utils import Dataset
    from sklearn.mixture import GaussianMixture
    rng = np.random.RandomState(4)
    N, B = 16 * 3 + 6, 16
    X = (rng.rand(N, 40) * (rng.rand(N, 40) < 0.4)).astype(np.float32)
    cls = rng.randint(0, 5, N)
    m = build()
    m.path = str(tmp_path / "ckpt")
    m.define_pretrain_step(0.003, 0.004)
    cfg = O.Config(40, 6, 5, SMALL["enc_layers"], SMALL["head_dim"], SMALL["dec_layers"])
    p0 = m.engine.get_parameters()
    p = {k: v.astype(np.float64) for k, v in p0.items()}
    data = Dataset((X, cls), batch_size=B, shuffle=False)
    batches = [X[i:i + B].astype(np.float64) for i in range(0, N, B)]
    epoch_len = len(batches)

    # ---- stage 1
    loss_vae = m.pretrain_vae(None, data, 2)
    mo, vo = O.adam_tf_init(p)
    t, ref = 0, []
    for ep in range(2):
        acc = 0.0
        for xb in batches:
            t += 1
            a = O.forward(p, cfg, xb, np.zeros((len(xb), 6)), 0.0)
            g = O.backward(p, cfg, a)
            O.adam_tf(p, g, mo, vo, t, 0.003)
            acc += a["recon"] / epoch_len
        ref.append(acc)
    assert loss_vae == pytest.approx(min(ref), abs=2e-3)
    pg = m.engine.get_parameters()
    for k in p:
        assert np.percentile(np.abs(pg[k] - p[k]), 99) <= 5e-4, ("vae stage", k)
    for k in ("W_ch", "b_ch", "W_logits", "b_logits", "prior_means", "prior_log_vars"):   # no gradient from recon: untouched
        np.testing.assert_array_equal(pg[k], p0[k])
    assert os.path.exists(os.path.join(m.path, "vae", "parameters.npz"))

    # ---- stage 2: GMM initialisation (same global NumPy stream on both sides), then the c-head
    for k in p:        # continue from the GPU's parameters so that the GMM sees identical means
        p[k] = pg[k].astype(np.float64)
    Z = m.encode(data.data)[0]
    np.random.seed(7)
    gmm = GaussianMixture(n_components=5, covariance_type="diag", max_iter=2, n_init=20, weights_init=np.ones(5) / 5).fit(Z)
    np.random.seed(7)
    loss_prior = m.pretrain_prior(None, data, 2)
    p["prior_means"] = gmm.means_.astype(np.float32).astype(np.float64)
    p["prior_log_vars"] = np.log(gmm.covariances_ + 1e-20).astype(np.float32).astype(np.float64)
    mo, vo = O.adam_tf_init(p)
    t, ref = 0, []
    for ep in range(2):
        acc = 0.0
        for xb in batches:
            t += 1
            a = O.forward(p, cfg, xb, np.zeros((len(xb), 6)), 1.0)
            g = O.backward(p, cfg, a)
            for k in g:
                if k not in m.PRIOR_VAR_LIST:
                    g[k] = np.zeros_like(g[k])
            O.adam_tf(p, g, mo, vo, t, 0.004)
            acc += (a["kl_z"] + a["kl_c"]) / epoch_len
        ref.append(acc)
    assert loss_prior == pytest.approx(min(ref), rel=2e-3, abs=2e-3)
    pg2 = m.engine.get_parameters()
    for k in p:
        if k in m.PRIOR_VAR_LIST:
            assert np.percentile(np.abs(pg2[k] - p[k]), 99) <= 5e-4, ("prior stage", k)
        else:
            np.testing.assert_allclose(pg2[k], p[k], rtol=0, atol=1e-6, err_msg=k)     # frozen
    # the full pretrain() hands the main optimizer fresh slots
    m.pretrain(None, data, 0, 0)
    assert m.engine.read_state().adam_t == 0 and float(m.engine.m.abs().sum()) == 0.0


def test_regeneration_and_cluster_samples_match_reference_definitions(tmp_path, monkeypatch):
    """visualization.py:20-129: a reconstruction is sigmoid(decode(mean)) (every epsilon fed as
    zeros); a sample of cluster c is sigmoid(decode(mu_c + sigma_c * n)), n from the NumPy stream."""
    from includes import visualization as V
    from includes.utils import Dataset
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(9)
    m = build(input_dim=49, latent_dim=6, n_classes=5)          # 7 x 7 "images"
    X = (rng.rand(120, 49) * (rng.rand(120, 49) < 0.4)).astype(np.float32)
    data = Dataset((X, rng.randint(0, 5, 120)), batch_size=16, shuffle=False)
    cfg = O.Config(49, 6, 5, SMALL["enc_layers"], SMALL["head_dim"], SMALL["dec_layers"])
    p = {k: v.astype(np.float64) for k, v in m.engine.get_parameters().items()}
    orig, recn = V.regenerate(m, data)
    a = O.forward(p, cfg, X[:100].astype(np.float64), np.zeros((100, 6)))
    np.testing.assert_array_equal(orig, X[:100])
    np.testing.assert_allclose(recn, 1.0 / (1.0 + np.exp(-a["xlogits"])), atol=2e-5)
    left, right = V.mnist_regeneration_plot(m, data)
    assert left.shape == right.shape == (70, 70)
    np.testing.assert_allclose(left[7:14, 14:21], X[12].reshape(7, 7) * 255.0)      # panel (row 1, col 2) = image 12
    np.random.seed(3)
    Zs, dec = V.sample_clusters(m, n=200)
    np.random.seed(3)
    for c in range(5):
        O.sample_gumbel((200, 1, 5))      # sample_generative_feed walks C first (its one-hot draw, unused), then Z: base_models.py:58-64
        z = p["prior_means"][c] + np.random.randn(200, 6) * np.exp(p["prior_log_vars"][c] / 2.0)
        np.testing.assert_allclose(Zs[c], z, atol=1e-6)
        xl = O.decode(p, cfg, z[:100])["xlogits"]
        np.testing.assert_allclose(dec[c], 1.0 / (1.0 + np.exp(-xl)), atol=2e-5)
    fig = V.mnist_sample_plot(m)
    assert fig.shape == (35, 70)
    for name in ("regenerated.png", "sampled.png"):
        blob = open(os.path.join("plots", m.name, "mnist", name), "rb").read()
        assert blob[:8] == b"\x89PNG\r\n\x1a\n" and blob[-8:-4] == b"IEND"


def test_unsupported_surfaces_fail_loudly():
    import base_models
    with pytest.raises(ValueError):                    # the CNN trunk reshapes to 28x28x1 (base_models.py:176)
        base_models.DeepMixtureVAE("m", "binary", 100, 10, 10, cnn=True)
    with pytest.raises(ValueError):                    # VaDE's convolutional encoder (base_models.py:455) likewise
        base_models.VaDE("v", "binary", 100, 10, 10, cnn=True)
    assert base_models.VaDE("v", "binary", 784, 10, 10, cnn=True).enc_layers == (128,)       # :486 ("fc", 2048 -> 128); built in tests/test_gpu_vade.py


def test_cnn_model_trains_through_the_reference_surface():
    """`cnn=True` = the checked-in trunk (base_models.py:156,176-216) behind the same class surface:
    define_train_step / train_op / get_accuracy, parameters named after the reference's layers."""
    import base_models
    from includes.utils import Dataset
    X = O.synthetic_images(512, 784, seed=4)
    y = np.random.RandomState(0).randint(0, 10, 512)
    model = base_models.DeepMixtureVAE("c", "binary", 784, 8, 10, activation="relu", initializer="xavier", cnn=True,
                                       batch_size=128, dtype="bf16", head_dim=256, dec_layers=(256, 128)).build_graph()
    eng = model.engine
    assert eng.cnn and eng.enc_layers == (500,) and eng.tensors["W_conv0"][1:3] == (9, 32)
    data = Dataset((X, y), batch_size=128)
    # Adam at the CLI's 0.002 on four batches of 128 sits at the edge of stability for this 8-layer encoder (the
    # epoch loss can jump, in fp32 and bf16 alike: tools/cnn_surface_probe.py); a quarter of it descends steadily
    model.define_train_step(0.0005, data.epoch_len * 10)
    losses = [model.train_op(None, data, 1.0) for _ in range(5)]
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0] and max(losses) <= 1.02 * losses[0], losses
    acc = model.get_accuracy(None, Dataset((X, y), batch_size=128))
    assert 0.0 <= acc <= 1.0


def test_train_cli_one_epoch(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DMVAE_DATA", str(tmp_path / "nodata"))
    import importlib
    import includes.utils as U
    real = U.load_data
    monkeypatch.setattr(U, "load_data", lambda name, **kw: real(name, n_train=900, n_test=300, **kw))
    sys.argv = ["train.py"]
    train = importlib.import_module("train")
    args = train.parser.parse_args(["--n_epochs", "2", "--batch_size", "256", "--latent_dim", "10", "--seed", "1"])
    loss = train.main(args)
    assert np.isfinite(loss) and 50 < loss < 560
    assert os.path.exists(tmp_path / "saved-models" / "mnist" / "dmvae" / "model" / "parameters.ckpt")
    assert "Max Accuracy" in open(tmp_path / "dmvae_logs.txt").read()
    import json
    recs = [json.loads(l) for l in open(tmp_path / "dmvae_metrics.jsonl")]          # one JSON line per epoch: loss terms, img/s, accuracies
    assert [r["epoch"] for r in recs] == [0, 1] and recs[-1]["loss"] == pytest.approx(loss, rel=1e-6)
    assert all(abs(r["recon"] + r["kl_ratio"] * (r["kl_z"] + r["kl_c"]) - r["loss"]) < 1e-2 and r["images_per_sec"] > 0 and r["rows"] == 1200 for r in recs)
    # a checkpoint this revision cannot read (another format at the same path, a truncated archive, other shapes)
    # must not abort start-up: the reason is printed and training starts from the initial parameters, as in the reference
    ck = tmp_path / "saved-models" / "mnist" / "dmvae" / "model" / "parameters.ckpt"
    blob = open(ck, "rb").read()
    for bad in (blob[: len(blob) // 3], b"PK\x03\x04 not an npz", b""):
        open(ck, "wb").write(bad)
        loss = train.main(train.parser.parse_args(["--n_epochs", "1", "--batch_size", "256", "--latent_dim", "10", "--seed", "1"]))
        assert np.isfinite(loss)
    np.savez(open(ck, "wb"), W_enc0=np.zeros((3, 3), np.float32))          # readable archive, wrong shapes
    loss = train.main(train.parser.parse_args(["--n_epochs", "1", "--batch_size", "256", "--latent_dim", "10", "--seed", "1"]))
    assert np.isfinite(loss)
