"""Full-step golden fixture (tests/golden/step_golden.npz, written by oracle/make_step_golden.py):
  * CPU: the oracle still reproduces it (the restatement cannot drift unnoticed);
  * GPU (-m gpu): the fp32 engine reproduces it through the C ABI WITHOUT executing the oracle --
    forward tensors, loss terms, every gradient, parameters after 1 and 3 TF-Adam steps."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(ROOT, "tests", "golden", "step_golden.npz"))


def _cases(sg):
    for ci in range(int(sg["n_cases"])):
        B, I, D, K, head = (int(v) for v in sg["c%d_dims" % ci])
        enc, dec = tuple(int(v) for v in sg["c%d_enc" % ci]), tuple(int(v) for v in sg["c%d_dec" % ci])
        for mode in ("exact", "relaxed"):
            yield "c%d_%s_" % (ci, mode), mode, B, I, D, K, enc, head, dec


def _params(sg, pre, tag):
    n = len(pre + tag)
    return {k[n:]: sg[k] for k in sg.files if k.startswith(pre + tag)}


def test_oracle_reproduces_step_golden(sg):
    import dmvae_oracle as O
    for pre, mode, B, I, D, K, enc, head, dec in _cases(sg):
        cfg = O.Config(I, D, K, enc, head, dec)
        p = {k: v.astype(np.float64) for k, v in _params(sg, pre, "p0_").items()}
        X, eps, gum = (sg[pre + k].astype(np.float64) for k in ("X", "eps", "gumbel"))
        a = O.forward(p, cfg, X, eps, 0.8, mode, gum, 0.5)
        for k in ("loss", "recon", "kl_z", "kl_c"):
            assert a[k] == pytest.approx(float(sg[pre + k]), rel=1e-12, abs=1e-12), (pre, k)
        g = O.backward(p, cfg, a)
        for k, v in g.items():
            np.testing.assert_allclose(v, sg[pre + "g_" + k], rtol=2e-6, atol=1e-9, err_msg=pre + k)
        m, v = O.adam_tf_init(p)
        for t in (1, 2, 3):
            a_t, _ = O.train_step(p, m, v, t, cfg, X, eps, 0.8, 0.002, mode, gum, 0.5)
            assert a_t["loss"] == pytest.approx(float(sg[pre + "loss_t%d" % t]), rel=1e-12)
        for k, val in p.items():
            np.testing.assert_allclose(val, sg[pre + "p3_" + k], rtol=2e-6, atol=1e-8, err_msg=pre + k)


@pytest.mark.gpu
def test_gpu_fp32_step_reproduces_step_golden(sg):
    import torch
    from dmvae_hip import StepEngine
    for pre, mode, B, I, D, K, enc, head, dec in _cases(sg):
        eng = StepEngine(I, D, K, enc_layers=enc, head_dim=head, dec_layers=dec, dtype="fp32", max_batch=B, mode=mode, temperature=0.5)
        eng.init_parameters(0)
        eng.set_parameters(_params(sg, pre, "p0_"))
        eng.write_state(kl_ratio=0.8, lr=0.002)
        Xd, ed, gd = (torch.as_tensor(np.ascontiguousarray(sg[pre + k])).cuda() for k in ("X", "eps", "gumbel"))
        eng.load_batch(Xd, None, 0, B)
        eng.forward_backward(B, ed, gd if mode == "relaxed" else None)
        torch.cuda.synchronize()
        st = eng.read_state()
        assert abs(st.last_loss - float(sg[pre + "loss"])) <= 1e-3
        assert abs(st.last_recon - float(sg[pre + "recon"])) <= 1e-3
        assert abs(st.last_klz - float(sg[pre + "kl_z"])) <= 1e-4 * max(1.0, abs(float(sg[pre + "kl_z"])))
        assert abs(st.last_klc - float(sg[pre + "kl_c"])) <= 1e-5
        np.testing.assert_allclose(eng.view("mean", B).cpu().numpy(), sg[pre + "fwd_mean"], atol=2e-5)
        np.testing.assert_allclose(eng.view("log_var", B).cpu().numpy(), sg[pre + "fwd_logvar"], atol=2e-5)
        np.testing.assert_allclose(eng.view("logits", B).cpu().numpy(), sg[pre + "fwd_logits"], atol=2e-5)
        np.testing.assert_allclose(eng.view("weights", B).cpu().numpy(), sg[pre + "fwd_w"], atol=1e-5)
        gg = eng.get_gradients()
        for k, v in gg.items():
            ref = sg[pre + "g_" + k]
            # a ReLU unit whose pre-activation is within f32 rounding of zero may sit on the other side of
            # the gate on the GPU: bound the error against the tensor's scale, as the oracle tests do
            assert np.abs(v - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-12), (pre, k)
        eng.update(1.0)
        for t in (2, 3):
            eng.forward_backward(B, ed, gd if mode == "relaxed" else None)
            eng.update(1.0)
            torch.cuda.synchronize()
            assert abs(eng.read_state().last_loss - float(sg[pre + "loss_t%d" % t])) <= 2e-3
        pg = eng.get_parameters()
        for k, v in pg.items():
            assert np.percentile(np.abs(v - sg[pre + "p3_" + k]), 99.0) <= 2e-4, (pre, k)
