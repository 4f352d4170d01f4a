"""Golden vectors for the CNN encoder trunk from the reference's OWN
code/includes/layers.py (Convolution, MaxPooling, FullyConnected) and code/includes/network.py
(DeepNetwork), executed from /root/reference with the `tensorflow` module name bound to
oracle/np_tf_ops.py -- nothing is copied.  TEST INFRASTRUCTURE ONLY; runs in the build container only.

Pins: the layer order and shapes of the spec at base_models.py:181-202 (conv 1-32-32 / pool / 32-64-64 /
pool / 64-128-128 / pool / fc 2048-500), the HWIO weight layout, conv -> bias_add -> activation, SAME
pooling 28 -> 14 -> 7 -> 4, and the (h, w, c) flatten order in front of the fc layer.  The arithmetic of
tf.nn.conv2d / tf.nn.max_pool themselves is np_tf_ops' restatement of the documented TF semantics.

Weights are NOT stored: generator and test draw them from the same seeded numpy RandomState (legacy
stream, frozen across numpy versions); the fixture holds the inputs, every intermediate shape, a
checksum per layer output and the final [B, 500] activations.

    python oracle/make_cnn_golden.py      # writes tests/golden/cnn_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"
OUT = os.path.join(HERE, "..", "tests", "golden", "cnn_golden.npz")

# the spec list of base_models.py:181-202, as data
SPEC = [("cn", {"n_kernels": 32, "prev_n_kernels": 1, "kernel": (3, 3)}),
        ("cn", {"n_kernels": 32, "prev_n_kernels": 32, "kernel": (3, 3)}),
        ("mp", {"k": 2}),
        ("cn", {"n_kernels": 64, "prev_n_kernels": 32, "kernel": (3, 3)}),
        ("cn", {"n_kernels": 64, "prev_n_kernels": 64, "kernel": (3, 3)}),
        ("mp", {"k": 2}),
        ("cn", {"n_kernels": 128, "prev_n_kernels": 64, "kernel": (3, 3)}),
        ("cn", {"n_kernels": 128, "prev_n_kernels": 128, "kernel": (3, 3)}),
        ("mp", {"k": 2}),
        ("fc", {"input_dim": 2048, "output_dim": 500})]


def draw_weights(seed):
    """the weights both sides use: He-scaled normals rounded to float32, biases 0.1 N(0,1), in layer order"""
    rng = np.random.RandomState(seed)
    ws = []
    for kind, kw in SPEC:
        if kind == "cn":
            shape = tuple(kw["kernel"]) + (kw["prev_n_kernels"], kw["n_kernels"])
            W = rng.randn(*shape) * np.sqrt(2.0 / (9 * kw["prev_n_kernels"]))
            b = rng.randn(kw["n_kernels"]) * 0.1
        elif kind == "fc":
            W = rng.randn(kw["input_dim"], kw["output_dim"]) * np.sqrt(2.0 / kw["input_dim"])
            b = rng.randn(1, kw["output_dim"]) * 0.1
        else:
            continue
        ws.append((W.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)))
    return ws


def main():
    sys.path.insert(0, HERE)
    import np_tf_ops
    sys.modules["tensorflow"] = np_tf_ops
    sys.path.insert(0, REF)
    from includes.network import DeepNetwork
    np.random.seed(0)
    net = DeepNetwork("layers", SPEC, activation=np_tf_ops.nn.relu, initializer=np_tf_ops.contrib.layers.xavier_initializer)
    ws = iter(draw_weights(77))
    blob = {"weight_seed": np.int64(77)}
    li = 0
    for layer in net.layers:
        if hasattr(layer, "W"):
            blob["shape_W%d" % li] = np.array(layer.W.shape, dtype=np.int64)
            blob["shape_b%d" % li] = np.array(layer.b.shape, dtype=np.int64)
            W, b = next(ws)
            assert W.shape == tuple(layer.W.shape) and b.shape == tuple(layer.b.shape)
            layer.W, layer.b = W, b
            li += 1
    rng = np.random.RandomState(5)
    X = (rng.rand(2, 784) * (rng.rand(2, 784) < 0.25)).astype(np.float32).astype(np.float64)
    blob["X"] = X.astype(np.float32)
    h = X.reshape(-1, 28, 28, 1)                     # base_models.py:176
    for i, layer in enumerate(net.layers):
        h = layer(h)
        blob["shape_out%d" % i] = np.array(np.shape(h), dtype=np.int64)
        blob["sum_out%d" % i] = np.float64(np.sum(h))
        blob["abs_out%d" % i] = np.float64(np.abs(h).sum())
    blob["out"] = np.asarray(h, dtype=np.float64)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
