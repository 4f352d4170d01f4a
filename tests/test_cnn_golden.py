"""the oracle's CNN trunk against golden vectors produced by executing the reference's own
Convolution / MaxPooling / FullyConnected / DeepNetwork classes (oracle/make_cnn_golden.py)."""
import os

import numpy as np

import dmvae_oracle as O
from make_cnn_golden import draw_weights

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_golden.npz"))


def golden_params():
    cfg = O.Config(784, 4, 3, enc_layers=(500,), head_dim=8, dec_layers=(8,), cnn=True)
    p = O.init_params(cfg, 0)
    ws = draw_weights(int(G["weight_seed"]))
    for i, (name, ci, co, hw, pool) in enumerate(cfg.conv_table()):
        W, b = ws[i]
        assert tuple(G["shape_W%d" % i]) == (3, 3, ci, co) and tuple(G["shape_b%d" % i]) == (co,)
        p["W_" + name] = W.reshape(9 * ci, co)          # HWIO memory order, flattened
        p["b_" + name] = b
    W, b = ws[6]
    assert tuple(G["shape_W6"]) == (2048, 500) and tuple(G["shape_b6"]) == (1, 500)
    p["W_enc0"], p["b_enc0"] = W, b.reshape(-1)
    return cfg, p


def test_cnn_trunk_reproduces_reference_layer_stack():
    cfg, p = golden_params()
    X = G["X"].astype(np.float64)
    a = O.encode(p, cfg, X)
    # reference layer list: cn cn mp cn cn mp cn cn mp fc -> output index of every oracle activation
    where = {"conv0": 0, "conv1": 1, "pool1": 2, "conv2": 3, "conv3": 4, "pool3": 5, "conv4": 6, "conv5": 7, "pool5": 8}
    for name, i in where.items():
        assert tuple(G["shape_out%d" % i]) == a[name].shape, name
        np.testing.assert_allclose(a[name].sum(), float(G["sum_out%d" % i]), rtol=1e-11, err_msg=name)
        np.testing.assert_allclose(np.abs(a[name]).sum(), float(G["abs_out%d" % i]), rtol=1e-11, err_msg=name)
    assert a["flat"].shape == (2, O.Config.CONV_FLAT)
    np.testing.assert_allclose(a["enc0"], G["out"], rtol=1e-11, atol=1e-12)
