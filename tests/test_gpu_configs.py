"""GPU tests on the geometries of BASELINE.json's configs that tests/test_gpu_step.py does not reach:

  cfg3  Fashion-MNIST shape, K=10, z_dim=128 (batch 16384 per GPU)
  cfg5  4096-d inputs, K=256, z_dim=512, 4 x 4096 MLP encoder / decoder, head 4096 (batch 8192 per GPU)

and FULL-SIZE property tests at the batch sizes the configs name (cfg2 4096, cfg3 16384, cfg4 8192 per GPU,
cfg5 8192 per GPU), where the tile planner, the XCD run cutting and the supertile height choose differently than
at the small batches the oracle comparisons run at.  Reference lines: code/base_models.py:218-302 (graph),
:66-110 (loss, Adam), code/priors.py:104-147 (mixture KL).

Tolerances (north star: "within a stated floating-point tolerance"):
  fp32 engine vs float64 oracle : loss |delta| <= 1e-3 nats per 784 inputs (scaled by input_dim / 784 for the
                                  4096-d config), every gradient tensor <= 1e-4 of its max
  bf16 engine vs oracle / fp32  : loss <= 2e-3 relative, per-tensor gradient Frobenius error <= 8e-2
  properties                    : bit-identical (torch.equal) -- no tolerance
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import dmvae_oracle as O

CFG2 = dict(input_dim=784, latent_dim=64, n_classes=10)
CFG3 = dict(input_dim=784, latent_dim=128, n_classes=10)
CFG4 = dict(input_dim=784, latent_dim=256, n_classes=50)
CFG5 = dict(input_dim=4096, latent_dim=512, n_classes=256, enc_layers=(4096,) * 4, head_dim=4096, dec_layers=(4096,) * 4)
# (geometry, per-GPU batch, learning rate): cfg5 overflows exp(log_var) after one Adam step at train.py's 0.002
FULL = {"cfg2": (CFG2, 4096, 0.002), "cfg3": (CFG3, 16384, 0.002), "cfg4": (CFG4, 8192, 0.002), "cfg5": (CFG5, 8192, 1e-4)}


def make(kw, dtype, B, seed=0, lr=0.002, mode="exact"):
    from dmvae_hip import StepEngine
    eng = StepEngine(dtype=dtype, max_batch=B, mode=mode, deterministic=True, seed=77, **kw)
    eng.init_parameters(seed)
    eng.write_state(lr=lr)
    return eng


def oracle_cfg(kw):
    return O.Config(kw["input_dim"], kw["latent_dim"], kw["n_classes"], kw.get("enc_layers", (500, 500)),
                    kw.get("head_dim", 2000), kw.get("dec_layers", (2000, 500, 500)), "binary")


def fp32_step_vs_oracle(kw, B, own_masks=False, p_override=None, seed=1):
    """forward + loss + backward of the fp32 engine against the float64 oracle on the same parameters, batch
    and noise; own_masks: the oracle uses ITS ReLU masks (the caller made every pre-activation clear zero)."""
    eng = make(kw, "fp32", B)
    cfg = oracle_cfg(kw)
    rng = np.random.RandomState(seed)
    if p_override is not None:
        eng.set_parameters(p_override)
    p = {k: v.astype(np.float64) for k, v in eng.get_parameters().items()}
    I, D = cfg.input_dim, cfg.latent_dim
    X = (rng.rand(B, I) * (rng.rand(B, I) < 0.3)).astype(np.float32)
    eps = rng.randn(B, D).astype(np.float32)
    eng.write_state(kl_ratio=0.8)
    eng.load_batch(torch.as_tensor(X).cuda(), None, 0, B)
    eng.forward_backward(B, torch.as_tensor(eps).cuda())
    torch.cuda.synchronize()
    a = O.forward(p, cfg, X.astype(np.float64), eps.astype(np.float64), 0.8)
    acts = eng.hidden_activations(B)
    masks = {k: (v > 0).cpu().numpy() for k, v in acts.items()}
    flips = sum(int((masks[k] != (a[k] > 0)).sum()) for k in masks)
    units = sum(mk.size for mk in masks.values())
    if own_masks:
        assert flips == 0, flips
    else:
        assert flips <= 1e-4 * units, (flips, units)
    scale_I = max(1.0, I / 784.0)
    for k in masks:
        np.testing.assert_allclose(acts[k].float().cpu().numpy(), a[k], atol=3e-5 * scale_I, err_msg=k)
    g = O.backward(p, cfg, a, None if own_masks else masks)
    st = eng.read_state()
    assert abs(st.last_loss - a["loss"]) <= 1e-3 * scale_I, (st.last_loss, a["loss"])
    assert abs(st.last_recon - a["recon"]) <= 1e-3 * scale_I
    assert abs(st.last_klz - a["kl_z"]) <= 1e-4 * max(1.0, abs(a["kl_z"]))
    assert abs(st.last_klc - a["kl_c"]) <= 1e-5
    np.testing.assert_allclose(eng.view("mean", B).cpu().numpy(), a["mean"], atol=2e-5 * scale_I)
    np.testing.assert_allclose(eng.view("logits", B).cpu().numpy(), a["logits"], atol=2e-5 * scale_I)
    gg = eng.get_gradients()
    for k in g:
        scale = np.abs(g[k]).max() + 1e-12
        assert np.abs(gg[k] - g[k]).max() <= 1e-4 * scale, (k, np.abs(gg[k] - g[k]).max(), scale)
    return eng, a, g


def test_cfg3_geometry_fp32_step_matches_oracle():
    fp32_step_vs_oracle(CFG3, 256)


def test_cfg3_geometry_bf16_step_close_to_oracle():
    kw, B = CFG3, 512
    eng = make(kw, "bf16", B)
    cfg = oracle_cfg(kw)
    rng = np.random.RandomState(2)
    p = {k: v.astype(np.float64) for k, v in eng.get_parameters().items()}
    X = O.synthetic_images(B, 784, seed=3)
    eps = rng.randn(B, 128).astype(np.float32)
    eng.load_batch(torch.as_tensor(X).cuda(), None, 0, B)
    eng.forward_backward(B, torch.as_tensor(eps).cuda())
    torch.cuda.synchronize()
    a = O.forward(p, cfg, X.astype(np.float64), eps.astype(np.float64))
    g = O.backward(p, cfg, a)
    st = eng.read_state()
    assert abs(st.last_loss - a["loss"]) <= 2e-3 * abs(a["loss"]), (st.last_loss, a["loss"])
    gg = eng.get_gradients()
    for k in g:
        rel = np.linalg.norm(gg[k] - g[k]) / (np.linalg.norm(g[k]) + 1e-30)
        assert rel <= 8e-2, (k, rel)


def test_cfg5_geometry_fp32_step_matches_oracle():
    """4096-d inputs, 4 x 4096 trunk / decoder, head 4096, D=512, K=256 at B=128: 0.27 TFLOP in float64 NumPy.
    The latent kernel streams its 1 MB prior tables through LDS in D-chunks here; every GEMM has K = 4096."""
    eng, a, g = fp32_step_vs_oracle(CFG5, 128)
    assert len(g) == 2 * (4 + 5 + 4 + 1) + 2


def test_cfg5_geometry_bf16_tracks_fp32_engine_at_full_batch():
    kw, B, lr = FULL["cfg5"]
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    X = torch.rand((B, 4096), device="cuda", generator=g) * (torch.rand((B, 4096), device="cuda", generator=g) < 0.19)
    eps = torch.randn((B, 512), device="cuda", generator=g)
    out = {}
    for dt in ("fp32", "bf16"):
        eng = make(kw, dt, B, lr=lr)
        eng.load_batch(X, None, 0, B)
        eng.forward_backward(B, eps)
        torch.cuda.synchronize()
        st = eng.read_state()
        out[dt] = (st.last_loss, st.last_recon, st.last_klz, st.last_klc, {k: eng.grad_view(k).clone() for k in ("W_enc0", "W_zh", "W_mean", "W_logits", "W_dec0", "W_out", "prior_means", "prior_log_vars")})
        del eng
        torch.cuda.empty_cache()
    assert abs(out["bf16"][0] - out["fp32"][0]) <= 2e-3 * abs(out["fp32"][0]), (out["bf16"][:4], out["fp32"][:4])
    assert abs(out["bf16"][2] - out["fp32"][2]) <= 2e-2 * abs(out["fp32"][2]) + 1e-3
    for k, gb in out["bf16"][4].items():
        gf = out["fp32"][4][k]
        rel = (gb - gf).norm().item() / (gf.norm().item() + 1e-30)
        assert rel <= 0.15, (k, rel)          # bf16 activations through ten 4096-wide layers


def _snap(eng):
    st = eng.read_state()
    return dict(param=eng.param.clone(), m=eng.m.clone(), v=eng.v.clone(), shadow=eng.param_bf16.clone(),
                adam_t=st.adam_t, last_loss=st.last_loss, epoch_loss=st.epoch_loss, noise_step=st.noise_step, cursor=st.batch_cursor)


def _same(a, b, what):
    for k in ("param", "m", "v", "shadow"):
        assert torch.equal(a[k], b[k]), (what, k, (a[k] != b[k]).sum().item())
    for k in ("adam_t", "last_loss", "epoch_loss", "noise_step", "cursor"):
        assert a[k] == b[k], (what, k, a[k], b[k])


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_full_size_step_properties(name):
    _full_size_step_properties(name)


def _full_size_step_properties(name):
    """At the batch size the config names, three steps (device Philox noise, batch cursor in the device state):
      two eager runs are bit-identical; HIP-graph replay == eager; Adam fused into the dW launch == backward
      then stand-alone Adam; the staged (data-parallel) backward == the whole one; and the bf16 loss stays
      within 2e-3 relative of the fp32 engine on the same batch and noise."""
    kw, B, lr = FULL[name]
    I = kw["input_dim"]
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    N = 4 * B
    data = torch.rand((N, I), device="cuda", generator=g)
    data = data * (torch.rand((N, I), device="cuda", generator=g) < 0.19)
    perm = torch.randperm(N, device="cuda", generator=g).to(torch.int32)
    steps = 3

    def run(form):
        eng = make(kw, "bf16", B, seed=4, lr=lr)
        eng.reset_epoch(N // B, kl_ratio=1.0)
        grads = None
        if form == "graph":
            replay = eng.capture_step(data, perm)
            for _ in range(steps):
                replay()
        else:
            for s in range(steps):
                if form == "fused":
                    eng.train_step(data, perm, use_state_cursor=True)
                elif form == "unfused":
                    eng.train_step(data, perm, use_state_cursor=True, fused=False)
                else:
                    eng.load_batch(data, perm, 0, None, True)
                    for stage in range(3):
                        eng.forward_backward_stage(stage)
                    eng.update(1.0)
                if s == 0 and form in ("unfused", "staged"):
                    torch.cuda.synchronize()
                    grads = eng.grad.clone()
        torch.cuda.synchronize()
        snap = _snap(eng)
        del eng
        torch.cuda.empty_cache()
        return snap, grads

    a, _ = run("fused")
    assert a["adam_t"] == steps and a["cursor"] == steps and np.isfinite(a["last_loss"])
    b, _ = run("fused")
    _same(a, b, "two eager runs")
    del b
    c, _ = run("graph")
    _same(a, c, "graph replay vs eager")
    del c
    d, gd = run("unfused")
    _same(a, d, "fused vs backward + stand-alone Adam")
    del d
    e, ge = run("staged")
    _same(a, e, "staged vs whole backward")
    assert torch.equal(gd, ge)
    del e, gd, ge
    torch.cuda.empty_cache()

    # bf16 against the fp32 engine on one batch with caller-supplied noise
    eps = torch.randn((B, kw["latent_dim"]), device="cuda", generator=g)
    loss = {}
    for dt in ("fp32", "bf16"):
        eng = make(kw, dt, B, seed=4, lr=lr)
        eng.load_batch(data, perm, 0, B)
        eng.forward_backward(B, eps)
        torch.cuda.synchronize()
        loss[dt] = eng.read_state().last_loss
        del eng
        torch.cuda.empty_cache()
    assert abs(loss["bf16"] - loss["fp32"]) <= 2e-3 * abs(loss["fp32"]), loss


@pytest.mark.parametrize("B,D,K", [(100, 10, 10), (256, 64, 10)])
def test_fp32_step_with_the_oracles_own_relu_masks(B, D, K):
    """the other fp32 comparisons hand the oracle's backward the GPU's ReLU masks.  Here the
    biases are nudged until EVERY pre-activation of every ReLU layer is at least 1e-3 away from zero in
    float64 (fp32 rounding is ~1e-5 at these widths), so both sides take the same side of every kink by
    themselves: the oracle runs on its own masks and the mask sets must agree exactly.  cfg1's geometry, and
     cfg2's latent geometry at 256 rows: 1.8 M ReLU units, none handed over."""
    kw = dict(input_dim=784, latent_dim=D, n_classes=K)
    cfg = oracle_cfg(kw)
    rng = np.random.RandomState(1)
    p = O.init_params(cfg, 5)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    X = (rng.rand(B, 784) * (rng.rand(B, 784) < 0.3)).astype(np.float32)
    eps = rng.randn(B, D).astype(np.float32)
    Xd, ed = X.astype(np.float64), eps.astype(np.float64)
    margin = 1e-3
    nudge = np.random.RandomState(7)

    def clear(x, name):
        W, b = p["W_" + name], p["b_" + name]
        for _ in range(1000):
            pre = x @ W + b
            bad = np.where(np.abs(pre).min(axis=0) < margin)[0]
            if bad.size == 0:
                return np.maximum(pre, 0)
            b[bad] = (b[bad] + nudge.choice([-1.0, 1.0], bad.size) * nudge.uniform(0.004, 0.02, bad.size)).astype(np.float32)
        raise AssertionError("could not clear layer " + name)
    h = Xd
    for i in range(len(cfg.enc_layers)):
        h = clear(h, "enc%d" % i)
    zh = clear(h, "zh")
    clear(h, "ch")
    mean = zh @ p["W_mean"] + p["b_mean"]
    logvar = zh @ p["W_logvar"] + p["b_logvar"]
    h = O.gaussian_reparam(mean, logvar, ed)
    for i in range(len(cfg.dec_layers)):
        h = clear(h, "dec%d" % i)
    # same batch / noise stream as fp32_step_vs_oracle draws from seed 1
    fp32_step_vs_oracle(kw, B, own_masks=True, p_override=p, seed=1)


@pytest.mark.parametrize("slices,macro", [(2, 0), (4, 0), (4, 1)])
def test_dw_group_in_k_slices_equals_the_unsliced_group(slices, macro):
    """dW group in K slices (batches >= 8192 rows have the slabs; the plan's rule turns them on from 16384 rows, knobs 10 / 11 force them
    here at cfg4's 8192): every problem cut into `slices` K ranges that store partial products into slabs, the 256-divisible layers on
    the macro tile with bias-only strips (macro = 1), the slabs added in ascending order by slab_reduce (backward alone) or inside
    the Adam kernel (fused step).  Against the unsliced group on the same batch and noise: loss identical (the forward pass is
    untouched), every gradient tensor within 1e-3 of its Frobenius norm (another summation order over bf16 products); inside the
    sliced form: fused step == backward + stand-alone Adam bit for bit, two runs bit-identical."""
    from dmvae_hip import _lib
    kw, B, lr = FULL["cfg4"]
    g = torch.Generator(device="cuda"); g.manual_seed(21)
    X = torch.rand((B, 784), device="cuda", generator=g) * (torch.rand((B, 784), device="cuda", generator=g) < 0.19)
    eps = torch.randn((B, kw["latent_dim"]), device="cuda", generator=g)

    def run(fused):
        eng = make(kw, "bf16", B, seed=4, lr=lr)
        eng.load_batch(X, None, 0, B)
        if fused:
            eng.forward_backward_update(B, eps)
        else:
            eng.forward_backward(B, eps)
            torch.cuda.synchronize()
            grad = eng.grad.clone()
            eng.update(1.0)
        torch.cuda.synchronize()
        out = (eng.read_state().last_loss, eng.param.clone(), eng.m.clone(), eng.v.clone(), eng.param_bf16.clone(), None if fused else grad,
               {k: eng._strided(eng.grad, k).clone() for k in eng.tensors} if not fused else None)
        del eng
        return out

    try:
        _lib.check(_lib.lib.dmvae_debug_set_knob(10, 1))
        base = run(False)
        _lib.check(_lib.lib.dmvae_debug_set_knob(10, slices))
        _lib.check(_lib.lib.dmvae_debug_set_knob(11, macro))
        plain, plain2, fused = run(False), run(False), run(True)
    finally:
        _lib.check(_lib.lib.dmvae_debug_set_knob(10, 0))
        _lib.check(_lib.lib.dmvae_debug_set_knob(11, 1))
    assert plain[0] == base[0] == fused[0]
    for i in (1, 2, 3, 4):
        assert torch.equal(plain[i], plain2[i]) and torch.equal(plain[i], fused[i]), i          # reproducible; fused == unfused
    assert torch.equal(plain[5], plain2[5]) and not torch.equal(plain[5], base[5])                # (the slices ARE another summation order)
    for k in base[6]:
        d = (plain[6][k] - base[6][k]).norm().item() / (base[6][k].norm().item() + 1e-30)
        assert d <= 1e-3, (k, d)


def test_bf16_plan_with_ieee_adam_matches_the_fp32_arithmetic():
    """dmvae_config.adam_ieee: a bf16 plan whose Adam quotient uses the IEEE square root and division.  Fused and
    stand-alone updates stay bit-identical under the switch, and the parameters after one step differ from the default (hardware sqrt /
    reciprocal) by at most a few ulps of the step size."""
    from dmvae_hip import StepEngine
    kw, B = CFG2, 512
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    X = torch.rand((B, 784), device="cuda", generator=g) * (torch.rand((B, 784), device="cuda", generator=g) < 0.19)
    eps = torch.randn((B, 64), device="cuda", generator=g)
    res = {}
    for ieee in (False, True):
        for fused in (False, True):
            eng = StepEngine(dtype="bf16", max_batch=B, deterministic=True, seed=77, adam_ieee=ieee, **kw)
            eng.init_parameters(4)
            eng.load_batch(X, None, 0, B)
            if fused:
                eng.forward_backward_update(B, eps)
            else:
                eng.forward_backward(B, eps)
                eng.update(1.0)
            torch.cuda.synchronize()
            res[(ieee, fused)] = (eng.param.clone(), eng.m.clone(), eng.v.clone())
            del eng
    for ieee in (False, True):
        for a, b in zip(res[(ieee, False)], res[(ieee, True)]):
            assert torch.equal(a, b)
    assert torch.equal(res[(False, True)][1], res[(True, True)][1]) and torch.equal(res[(False, True)][2], res[(True, True)][2])      # m, v: same arithmetic
    d = (res[(False, True)][0] - res[(True, True)][0]).abs().max().item()
    assert 0 < d <= 1.5e-8, d          # the quotients differ by ~1 ulp of a step <= lr = 2e-3; in p - quotient that is at most one ulp of p (|p| < 0.125: 2^-27)


# ---------------------------------------------------------------- full-size INDEPENDENT gradient parity
# The property tests above compare the full-size paths with themselves (fused == unfused, staged == whole, graph == eager) and only
# the loss with the fp32 engine.  Below, at the batch sizes where the tile planner, the XCD runs, the K slices of the dW group, the
# macro-tile slices with their bias-only strips and adam_slabs ARE the product path, every gradient tensor and the parameters after
# a step are compared with something that shares none of that code: the fp32 engine (f32 MFMA kernels, unsliced, stand-alone IEEE
# Adam), the float64 oracle, and the same batch run as independent shards.

@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4"])
def test_full_size_bf16_gradients_and_update_track_the_fp32_engine(name):
    """code/base_models.py:110 (tf.gradients of every trainable) + :102-110 (Adam), at 4096 / 16384 / 8192 rows.
    (i) every gradient tensor of the bf16 backward (cfg3: dW in four K slices on the macro tile + bias-only strips + slab_reduce)
        against the fp32 engine on the same batch and noise: Frobenius error <= 8e-2 per tensor (the test prints the worst one);
    (ii) the parameters after ONE fused bf16 step (cfg3: adam_slabs) against fp32 engine + IEEE Adam.  Step 1 of TF-Adam moves a
        parameter by -lr g / (|g| + eps'), i.e. by +-lr wherever |g| >> 3e-7: the two runs agree to rounding where the bf16 and the
        fp32 gradient have the same sign and differ by 2 lr where a gradient within bf16 error of zero changes sign.  Stated bound:
        no element differs by more than 2 lr (1 + 1e-3), at most 1 % differ by more than lr / 10 (measured 0.21 / 0.17 / 0.25 % at
        cfg2 / cfg3 / cfg4, worst gradient Frobenius error 2.0e-2 / 1.4e-2 / 2.2e-2: profiles/r04_fullsize_parity.txt), and the pad
        elements (zero gradient) do not move at all."""
    kw, B, lr = FULL[name]
    g = torch.Generator(device="cuda"); g.manual_seed(31)
    X = torch.rand((B, 784), device="cuda", generator=g) * (torch.rand((B, 784), device="cuda", generator=g) < 0.19)
    eps = torch.randn((B, kw["latent_dim"]), device="cuda", generator=g)
    res = {}
    for dt in ("fp32", "bf16"):
        eng = make(kw, dt, B, seed=4, lr=lr)
        p0 = eng.param.clone()
        eng.load_batch(X, None, 0, B)
        eng.forward_backward(B, eps)
        torch.cuda.synchronize()
        grads = {k: eng.grad_view(k).clone() for k in eng.tensors}
        loss = eng.read_state().last_loss
        if dt == "fp32":
            eng.update(1.0)
            torch.cuda.synchronize()
            delta = eng.param - p0
        else:
            # the product path -- Adam fused into the dW launch (cfg3: slabs + adam_slabs) -- on a FRESH engine: every forward pass
            # advances adam_t (step_finalize), so a second pass on the engine above would apply step 2's lr_t to zero moments
            del eng
            torch.cuda.empty_cache()
            eng = make(kw, dt, B, seed=4, lr=lr)
            assert torch.equal(eng.param, p0)
            eng.load_batch(X, None, 0, B)
            eng.forward_backward_update(B, eps)
            torch.cuda.synchronize()
            assert eng.read_state().adam_t == 1 and eng.read_state().last_loss == loss
            delta = eng.param - p0
        res[dt] = (grads, delta, loss, (p0 == 0))
        del eng
        torch.cuda.empty_cache()
    gf, gb = res["fp32"][0], res["bf16"][0]
    worst = {}
    for k in gf:
        rel = (gb[k] - gf[k]).norm().item() / (gf[k].norm().item() + 1e-30)
        worst[k] = rel
        assert rel <= 8e-2, (name, k, rel)
    assert abs(res["bf16"][2] - res["fp32"][2]) <= 2e-3 * abs(res["fp32"][2])
    df, db = res["fp32"][1], res["bf16"][1]
    diff = (db - df).abs()
    assert diff.max().item() <= 2 * lr * (1 + 1e-3), diff.max().item()
    frac = (diff > lr / 10).float().mean().item()
    assert frac <= 0.01, (name, frac)
    # elements that never receive a gradient (layout pads: zero parameter, zero gradient) stay put in both runs
    still = (df == 0)
    assert torch.equal(db[still & res["fp32"][3]], torch.zeros_like(db[still & res["fp32"][3]]))
    print("%s: worst gradient Frobenius error %.3e (%s); parameters off by > lr/10: %.3f %%" %
          (name, max(worst.values()), max(worst, key=worst.get), 100 * frac))


def test_fp32_engine_at_the_full_cfg2_batch_matches_the_float64_oracle():
    """The oracle comparisons elsewhere run at B <= 512.  Here: the fp32 engine at the metric's own batch, 4096 x 784 (the grids, supertile
    heights and XCD runs of the full size), against the float64 oracle: loss <= 1e-3, every activation <= 3e-5, every gradient
    tensor <= 1e-4 of its max (~0.1 TFLOP of float64 NumPy: seconds on the GPU box's host cores)."""
    fp32_step_vs_oracle(CFG2, 4096, seed=5)


def test_one_65536_row_step_equals_eight_8192_row_shards():
    """`bench.py --config cfg4-strong` at N = 1: ONE 65 536-row batch on one GPU (dW in K slices on the macro tile, adam_slabs, the
    MFMA form of the latent stage at 512 row blocks).  The loss terms are batch means of per-row quantities (base_models.py:74-79,
    priors.py:145,199), so its gradients must equal the SUM of the gradients of the same rows run as eight independent 8192-row
    shards at inv_B = 1 / 65536 (the data-parallel identity of SURVEY 8e, here as a full-size check of the 65 536-row launch
    geometry).  Different summation orders over bf16 products: per-tensor Frobenius error <= 2e-3, loss <= 1e-5 relative."""
    kw = CFG4
    BIG, SH = 65536, 8192
    g = torch.Generator(device="cuda"); g.manual_seed(41)
    X = torch.rand((BIG, 784), device="cuda", generator=g) * (torch.rand((BIG, 784), device="cuda", generator=g) < 0.19)
    eps = torch.randn((BIG, kw["latent_dim"]), device="cuda", generator=g)
    big = make(kw, "bf16", BIG, seed=4)
    big.load_batch(X, None, 0, BIG)
    big.forward_backward(BIG, eps)
    torch.cuda.synchronize()
    gbig = {k: big.grad_view(k).clone() for k in big.tensors}
    sb = big.read_state()
    loss_big = (sb.last_loss, sb.last_recon, sb.last_klz, sb.last_klc)
    del big
    torch.cuda.empty_cache()
    small = make(kw, "bf16", SH, seed=4)
    acc = {k: torch.zeros_like(small.grad_view(k), dtype=torch.float64) for k in small.tensors}
    loss = np.zeros(4)
    for j in range(BIG // SH):
        small.load_batch(X, None, j * SH, SH)
        small.forward_backward(SH, eps[j * SH:(j + 1) * SH].contiguous(), None, 1.0 / BIG)
        torch.cuda.synchronize()
        for k in acc:
            acc[k] += small.grad_view(k).double()
        st = small.read_state()
        loss += np.array([st.last_loss, st.last_recon, st.last_klz, st.last_klc])
    for a, b in zip(loss_big, loss):
        assert abs(a - b) <= 1e-5 * abs(b) + 1e-6, (loss_big, loss)
    for k in acc:
        rel = (gbig[k].double() - acc[k]).norm().item() / (acc[k].norm().item() + 1e-30)
        assert rel <= 2e-3, (k, rel)


def test_staged_backward_in_k_slices_leaves_earlier_buckets_alone():
    """with the dW group in K slices (>= 16384 rows per GPU) the staged backward of the data-parallel path summed
    the slabs over ONE range from the segment's first weight to its last bias -- every bias now lives in the arena tail, so that
    range covered the weight buckets of EARLIER segments (whose reduce-scatter / all-reduce is already in flight under
    _step_with_exchange) and the not-yet-written biases of later ones.  Here: cfg3's 16 384 rows, slices forced to 2 (small tiles)
    and 4 (macro tile + bias-only strips); after every segment its bucket is cloned and then PERTURBED (as an in-place collective
    would), the untouched tail biases carry a sentinel; later segments must leave all of that exactly as it was, and the
    gradients must equal the whole (unstaged) backward."""
    from dmvae_hip import _lib
    kw, B, lr = FULL["cfg3"]
    g = torch.Generator(device="cuda"); g.manual_seed(51)
    X = torch.rand((B, 784), device="cuda", generator=g) * (torch.rand((B, 784), device="cuda", generator=g) < 0.19)
    eps = torch.randn((B, kw["latent_dim"]), device="cuda", generator=g)
    for slices in (2, 4):
        try:
            _lib.check(_lib.lib.dmvae_debug_set_knob(10, slices))
            whole = make(kw, "bf16", B, seed=4, lr=lr)
            whole.load_batch(X, None, 0, B)
            whole.forward_backward(B, eps)
            torch.cuda.synchronize()
            ref = whole.grad.clone()
            del whole
            eng = make(kw, "bf16", B, seed=4, lr=lr)
            buckets, (tlo, thi) = eng.grad_buckets()
            eng.load_batch(X, None, 0, B)
            SENT = 12345.0
            eng.grad[tlo:thi] = SENT
            bias_of = {0: ["b_dec0", "b_dec1", "b_dec2", "b_out"], 1: ["b_zh", "b_ch", "b_mean", "b_logvar", "b_logits"], 2: ["b_enc0", "b_enc1"]}
            kept = []
            for stage in range(3):
                eng.forward_backward_stage(stage, B, eps)
                torch.cuda.synchronize()
                for lo, hi, want in kept:            # earlier buckets: exactly the perturbed values
                    assert torch.equal(eng.grad[lo:hi], want), (slices, stage, lo, hi)
                lo, hi = buckets[stage]
                assert torch.equal(eng.grad[lo:hi], ref[lo:hi]), (slices, stage)
                for later in range(stage + 1, 3):     # later segments' biases: not written yet
                    for k in bias_of[later]:
                        assert bool((eng.grad_view(k) == SENT).all()), (slices, stage, k)
                for k in bias_of[stage]:
                    assert torch.equal(eng.grad_view(k), eng._strided(ref, k)), (slices, stage, k)
                eng.grad[lo:hi] += 1.0                # what an in-flight in-place collective would do to the bucket
                kept.append((lo, hi, eng.grad[lo:hi].clone()))
            del eng
            torch.cuda.empty_cache()
        finally:
            _lib.check(_lib.lib.dmvae_debug_set_knob(10, 0))
