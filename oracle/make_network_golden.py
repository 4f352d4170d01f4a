"""Golden vectors for the decoder network from the reference's OWN
code/includes/layers.py (FullyConnected) and code/includes/network.py (DeepNetwork), executed from
/root/reference with the `tensorflow` module name bound to oracle/np_tf_ops.py -- nothing is
copied.  TEST INFRASTRUCTURE ONLY; runs in the build container only.

Pins: the layer spec base_models.py:280-288 builds (three "fc" layers), variable shapes (weight
(in, out), bias (1, out)), the call order flatten -> matmul + b -> activation, layer after layer.
The final linear tf.layers.dense (base_models.py:291-293) is TensorFlow-internal and stays unpinned.

    python oracle/make_network_golden.py      # writes tests/golden/network_golden.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"
OUT = os.path.join(HERE, "..", "tests", "golden", "network_golden.npz")


def main():
    sys.path.insert(0, HERE)
    import np_tf_ops
    sys.modules["tensorflow"] = np_tf_ops
    sys.path.insert(0, REF)
    from includes.network import DeepNetwork
    blob = {}
    cases = [(5, 6, (9, 7, 5)), (8, 10, (64, 48, 32)), (3, 4, (12,))]
    blob["n_cases"] = np.int64(len(cases))
    for ci, (B, D, widths) in enumerate(cases):
        rng = np.random.RandomState(40 + ci)
        np.random.seed(40 + ci)
        spec, prev = [], D
        for w in widths:                      # exactly the list base_models.py:280-288 writes
            spec.append(("fc", {"input_dim": prev, "output_dim": w}))
            prev = w
        net = DeepNetwork("layers", spec, activation=np_tf_ops.nn.relu, initializer=np_tf_ops.contrib.layers.xavier_initializer)
        Z = rng.randn(B, D)
        blob["c%d_Z" % ci] = Z
        blob["c%d_widths" % ci] = np.array(widths, dtype=np.int64)
        for li, layer in enumerate(net.layers):
            blob["c%d_shape_W%d" % (ci, li)] = np.array(layer.W.shape, dtype=np.int64)
            blob["c%d_shape_b%d" % (ci, li)] = np.array(layer.b.shape, dtype=np.int64)
            layer.W = (rng.randn(*layer.W.shape) * np.sqrt(2.0 / layer.W.shape[0])).astype(np.float32).astype(np.float64)
            layer.b = (rng.randn(*layer.b.shape) * 0.1).astype(np.float32).astype(np.float64)
            blob["c%d_W%d" % (ci, li)] = layer.W.astype(np.float32)
            blob["c%d_b%d" % (ci, li)] = layer.b.astype(np.float32)
        blob["c%d_out" % ci] = np.asarray(net(Z), dtype=np.float64)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
