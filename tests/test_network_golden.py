"""Decoder network against golden vectors produced by the reference's own FullyConnected /
DeepNetwork classes (oracle/make_network_golden.py): the oracle on CPU, the fp32 engine on the GPU."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ng():
    return np.load(os.path.join(ROOT, "tests", "golden", "network_golden.npz"))


def _case(ng, ci):
    widths = tuple(int(v) for v in ng["c%d_widths" % ci])
    Z = ng["c%d_Z" % ci]
    Ws = [ng["c%d_W%d" % (ci, i)] for i in range(len(widths))]
    bs = [ng["c%d_b%d" % (ci, i)] for i in range(len(widths))]
    return widths, Z, Ws, bs, ng["c%d_out" % ci]


def test_reference_layer_shapes_and_oracle_decoder(ng):
    import dmvae_oracle as O
    for ci in range(int(ng["n_cases"])):
        widths, Z, Ws, bs, out = _case(ng, ci)
        prev = Z.shape[1]
        for i, w in enumerate(widths):       # includes/layers.py:24-28: weight (in, out), bias (1, out)
            assert tuple(ng["c%d_shape_W%d" % (ci, i)]) == (prev, w)
            assert tuple(ng["c%d_shape_b%d" % (ci, i)]) == (1, w)
            prev = w
        cfg = O.Config(7, Z.shape[1], 3, (5,), 6, widths)
        p = O.init_params(cfg, 0)
        for i in range(len(widths)):
            p["W_dec%d" % i] = Ws[i].astype(np.float64)
            p["b_dec%d" % i] = bs[i].astype(np.float64).reshape(p["b_dec%d" % i].shape)
        acts = O.decode(p, cfg, Z.astype(np.float64))
        np.testing.assert_allclose(acts["dec%d" % (len(widths) - 1)], out, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_gpu_decoder_hidden_layers_match_reference_network(ng):
    import torch
    from dmvae_hip import StepEngine
    for ci in range(int(ng["n_cases"])):
        widths, Z, Ws, bs, out = _case(ng, ci)
        B, D = Z.shape
        eng = StepEngine(7, D, 3, enc_layers=(5,), head_dim=6, dec_layers=widths, dtype="fp32", max_batch=B)
        eng.init_parameters(0)
        eng.set_parameters(dict([("W_dec%d" % i, Ws[i]) for i in range(len(widths))] +
                                [("b_dec%d" % i, bs[i].reshape(-1)) for i in range(len(widths))]))
        eng.decode(torch.as_tensor(np.ascontiguousarray(Z, dtype=np.float32)).cuda())
        torch.cuda.synchronize()
        last = eng.view("dec%d" % (len(widths) - 1), B, widths[-1]).cpu().numpy()
        np.testing.assert_allclose(last, out, rtol=2e-5, atol=2e-5)
