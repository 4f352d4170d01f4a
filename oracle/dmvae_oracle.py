"""CPU restatement of the DMVAE ELBO training step -- TEST INFRASTRUCTURE ONLY.

This module is the parity oracle for the HIP hot path.  It is imported only by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg; the
product path (deep-mixture-vae_amd/) never imports it and fails loudly when the
HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * prior / latent-variable math (rows A2, A6, A6b, A9, A10 of SURVEY.md 8a) is
    pinned by golden vectors produced by executing the reference's own
    code/priors.py (oracle/make_golden.py, fixtures in tests/golden/).
  * the decoder's hidden FullyConnected stack (row A7, base_models.py:279-289) is pinned by
    executing the reference's own includes/layers.py + includes/network.py
    (oracle/make_network_golden.py).
  * the CNN encoder trunk (base_models.py:176-216) is pinned by executing the
    reference's own Convolution / MaxPooling / FullyConnected / DeepNetwork classes on its spec
    list (oracle/make_cnn_golden.py, tests/golden/cnn_golden.npz); tf.nn.conv2d / max_pool /
    bias_add themselves are np_tf_ops' restatement of the documented TF semantics, and the
    trunk's backward is checked against torch conv2d / max_pool2d autograd in float64.
  * tf.layers.dense layers, sigmoid cross-entropy, autodiff backward and Adam live in
    TensorFlow 1.x (third party, not vendored, not installed): for those rows
    this file restates the documented TF semantics; PARITY UNPINNED by any
    reference run -- checked instead against closed forms and torch-autograd
    float64 (tests/test_oracle_*.py).

All functions are plain NumPy and dtype-generic (float64 for checking,
float32 for the timed CPU baseline).  Reference citations are file:line under
/root/reference/code/.
"""
import math

import numpy as np

# --------------------------------------------------------------------------
# Model description
# --------------------------------------------------------------------------


class Config:
    """Shapes of the MLP DMVAE (base_models.py:218-293, literals 500/500/2000,
    decoder 2000/500/500).  enc/head/dec widths are parameters so that the
    4x4096 configuration of BASELINE.json can be expressed."""

    def __init__(self, input_dim=784, latent_dim=10, n_classes=10,
                 enc_layers=(500, 500), head_dim=2000,
                 dec_layers=(2000, 500, 500), input_type="binary", cnn=False):
        self.cnn = bool(cnn)
        self.input_dim = int(input_dim)
        self.latent_dim = int(latent_dim)
        self.n_classes = int(n_classes)
        self.enc_layers = tuple(int(v) for v in enc_layers)
        self.head_dim = int(head_dim)
        self.dec_layers = tuple(int(v) for v in dec_layers)
        self.input_type = input_type
        if self.cnn:
            # the checked-in encoder trunk (base_models.py:176-216): 28x28x1 images, six 3x3 SAME
            # convolutions, three 2x2 SAME max-pools, then FullyConnected 2048 -> enc_layers[0]
            assert self.input_dim == 784 and len(self.enc_layers) == 1, "cnn trunk: 784 inputs, one fc layer"

    # (name, cin, cout, H=W of its input/output, pool after it)   base_models.py:181-201
    CONV_STACK = (("conv0", 1, 32, 28, False), ("conv1", 32, 32, 28, True),
                  ("conv2", 32, 64, 14, False), ("conv3", 64, 64, 14, True),
                  ("conv4", 64, 128, 7, False), ("conv5", 128, 128, 7, True))
    CONV_FLAT = 4 * 4 * 128      # 7 -> 4 under SAME pooling; "input_dim": 2048 at base_models.py:202

    def conv_table(self):
        return self.CONV_STACK if self.cnn else ()

    def layer_table(self):
        """[(name, fan_in, fan_out, bias_kind)] in forward order.

        bias_kind "zero": tf.layers.dense default bias (base_models.py:221-248,
        291-293); "xavier": FullyConnected bias of shape (1,out) created with
        the same xavier initializer (includes/layers.py:24-28)."""
        t = []
        prev = self.CONV_FLAT if self.cnn else self.input_dim
        for i, h in enumerate(self.enc_layers):
            # cnn: the trunk's last layer is a FullyConnected ("fc" spec, base_models.py:202), bias xavier
            t.append(("enc%d" % i, prev, h, "xavier" if self.cnn else "zero"))
            prev = h
        trunk = prev
        t.append(("zh", trunk, self.head_dim, "zero"))
        t.append(("mean", self.head_dim, self.latent_dim, "zero"))
        t.append(("logvar", self.head_dim, self.latent_dim, "zero"))
        t.append(("ch", trunk, self.head_dim, "zero"))
        t.append(("logits", self.head_dim, self.n_classes, "zero"))
        prev = self.latent_dim
        for i, h in enumerate(self.dec_layers):
            t.append(("dec%d" % i, prev, h, "xavier"))
            prev = h
        t.append(("out", prev, self.input_dim, "zero"))
        return t

    def n_params(self):
        n = sum(9 * ci * co + co for _, ci, co, _, _ in self.conv_table())
        for _, fi, fo, _ in self.layer_table():
            n += fi * fo + fo
        return n + 2 * self.n_classes * self.latent_dim


def xavier_uniform(rng, fan_in, fan_out, shape, dtype=np.float64):
    """tf.contrib.layers.xavier_initializer(uniform=True): U(+-sqrt(6/(in+out)))
    (train.py:197 passes it to every layer)."""
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_params(cfg, seed=0, dtype=np.float64):
    """Parameter creation, SURVEY 8a row A0.

    dense kernels xavier-uniform, dense biases zero (base_models.py:221-248,
    291-293); FullyConnected weight (in,out) AND bias (1,out) xavier-uniform
    (includes/layers.py:24-28: for the bias fan_in=1, fan_out=out); prior
    means ~ N(0,1), prior log_vars = 0 (priors.py:57-65)."""
    rng = np.random.RandomState(seed)
    p = {}
    for name, ci, co, _, _ in cfg.conv_table():
        # Convolution (includes/layers.py:39-51): weight (3,3,cin,cout) and bias (cout,) both from the
        # xavier initializer; TF's fans for a conv kernel are 9*cin / 9*cout, for a 1-D shape n / n.
        # Stored flattened (ky,kx,cin) x cout = the HWIO memory order.
        p["W_" + name] = xavier_uniform(rng, 9 * ci, 9 * co, (9 * ci, co), dtype)
        p["b_" + name] = xavier_uniform(rng, co, co, (co,), dtype)
    for name, fi, fo, bk in cfg.layer_table():
        p["W_" + name] = xavier_uniform(rng, fi, fo, (fi, fo), dtype)
        if bk == "zero":
            p["b_" + name] = np.zeros((fo,), dtype)
        else:
            p["b_" + name] = xavier_uniform(rng, 1, fo, (fo,), dtype)
    p["prior_means"] = rng.randn(cfg.n_classes, cfg.latent_dim).astype(dtype)
    p["prior_log_vars"] = np.zeros((cfg.n_classes, cfg.latent_dim), dtype)
    return p


def param_names(cfg):
    names = []
    for name, _, _, _, _ in cfg.conv_table():
        names += ["W_" + name, "b_" + name]
    for name, _, _, _ in cfg.layer_table():
        names += ["W_" + name, "b_" + name]
    return names + ["prior_means", "prior_log_vars"]


# --------------------------------------------------------------------------
# Latent-variable math (priors.py)
# --------------------------------------------------------------------------


def sample_gumbel(shape, rng=None, eps=1e-20):
    """includes/utils.py:17-19: U~U(0,1); -log(eps - log(U + eps))."""
    rng = np.random if rng is None else rng
    U = rng.uniform(0, 1, shape)
    return -np.log(eps - np.log(U + eps))


def softmax(a):
    a = a - np.max(a, axis=-1, keepdims=True)
    e = np.exp(a)
    return e / np.sum(e, axis=-1, keepdims=True)


def gaussian_reparam(mean, log_var, epsilon):
    """priors.py:86-89: Z = mean + exp(log_var/2) * epsilon."""
    return mean + np.exp(log_var / 2) * epsilon


def gumbel_softmax(logits, gumbel, temperature):
    """priors.py:170-181: softmax((logits + g) / temperature), logits (B,K),
    g (B,K) (reference shape (B,1,K) flattened)."""
    return softmax((logits + gumbel) / temperature)


def kl_mixture_exact(mean, log_var, weights, prior_means, prior_log_vars):
    """priors.py:131-145 (cluster_sample=False, the live branch)."""
    pm = prior_means[None, :, :]
    plv = prior_log_vars[None, :, :]
    mu = mean[:, None, :]
    lv = log_var[:, None, :]
    res = plv - lv - 1 + (np.exp(lv) + np.square(mu - pm)) / np.exp(plv)
    res = np.sum(res, axis=-1)
    res = np.sum(res * weights, axis=-1)
    return np.mean(0.5 * res)


def kl_mixture_relaxed(mean, log_var, weights, prior_means, prior_log_vars):
    """priors.py:119-128 (cluster_sample=True)."""
    pm = weights @ prior_means
    plv = weights @ prior_log_vars
    res = plv - log_var - 1 + (np.exp(log_var) + np.square(mean - pm)) / np.exp(plv)
    return np.mean(0.5 * np.sum(res, axis=1))


def kl_categorical(logits, n_classes, eps=1e-20):
    """priors.py:183-201 with dim=1."""
    q = softmax(logits)
    res = q * (np.log(q + eps) - np.log(1.0 / n_classes))
    return np.mean(np.sum(res, axis=1))


def kl_normal(mean, log_var):
    """NormalFactorial.kl_from_prior, priors.py:38-47 (K=1 known answer)."""
    res = np.exp(log_var) + np.square(mean) - 1.0 - log_var
    return np.mean(0.5 * np.sum(res, axis=1))


def cluster_probs(Z, prior_means, prior_log_vars):
    """priors.py:91-102."""
    z = Z[:, None, :]
    pm = prior_means[None]
    plv = prior_log_vars[None]
    p = -(np.sum(np.square(z - pm) / np.exp(plv), axis=-1) + np.sum(plv, axis=-1)) / 2
    return softmax(p)


# --------------------------------------------------------------------------
# Forward / loss / backward of the whole step
# --------------------------------------------------------------------------


def _dense(x, W, b, relu):
    y = x @ W + b
    return np.maximum(y, 0) if relu else y


def im2col3x3(x):
    """[B,H,W,C] -> [B,H,W,9C]: column (ky*3+kx)*C + c holds x[y+ky-1, x+kx-1, c], zero outside the
    image (tf.nn.conv2d padding='SAME', stride 1, 3x3: one zero pixel on every side)."""
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C), x.dtype)
    xp[:, 1:-1, 1:-1] = x
    return np.concatenate([xp[:, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], axis=3)


def col2im3x3(dcol, C):
    """adjoint of im2col3x3"""
    B, H, W, _ = dcol.shape
    dxp = np.zeros((B, H + 2, W + 2, C), dcol.dtype)
    for ky in range(3):
        for kx in range(3):
            t = ky * 3 + kx
            dxp[:, ky:ky + H, kx:kx + W] += dcol[..., t * C:(t + 1) * C]
    return dxp[:, 1:-1, 1:-1]


def conv3x3_relu(x, Wm, b):
    """includes/layers.py:53-60: relu(bias_add(conv2d(x, W, strides 1, SAME), b)), W flattened HWIO"""
    return np.maximum(im2col3x3(x) @ Wm + b, 0)


def maxpool2_same(x):
    """tf.nn.max_pool ksize 2 strides 2 padding SAME (includes/layers.py:72-75): out = ceil(H/2); the
    pad (bottom / right only, when H is odd) never wins.  Returns (out, first-argmax one-hot [B,H,W,C])
    -- the gradient goes to the first maximum of a window in row-major order."""
    B, H, W, C = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    xp = np.full((B, 2 * Ho, 2 * Wo, C), -np.inf, x.dtype)
    xp[:, :H, :W] = x
    win = np.stack([xp[:, dy::2, dx::2] for dy in range(2) for dx in range(2)], axis=0)   # [4,B,Ho,Wo,C]
    out = win.max(axis=0)
    first = np.argmax(win, axis=0)            # np.argmax returns the first maximum
    route = np.zeros((B, 2 * Ho, 2 * Wo, C), bool)
    for dy in range(2):
        for dx in range(2):
            route[:, dy::2, dx::2] = first == dy * 2 + dx
    return out, route[:, :H, :W]


def maxpool2_same_backward(dout, route):
    B, H, W, C = route.shape
    Ho, Wo = dout.shape[1:3]
    up = np.repeat(np.repeat(dout, 2, axis=1), 2, axis=2)[:, :H, :W]
    return up * route


def cnn_trunk(p, cfg, X, acts):
    """base_models.py:176-216: reshape to (-1,28,28,1), the conv / pool stack, flatten (h, w, c)"""
    h = X.reshape(X.shape[0], 28, 28, 1)
    for i, (name, ci, co, hw, pool) in enumerate(cfg.conv_table()):
        acts[name + "_in"] = h
        h = conv3x3_relu(h, p["W_" + name], p["b_" + name])
        acts[name] = h
        if pool:
            h, acts["route%d" % i] = maxpool2_same(h)
            acts["pool%d" % i] = h
    acts["flat"] = h.reshape(h.shape[0], -1)
    return acts["flat"]


def cnn_trunk_backward(p, cfg, a, dflat, g):
    dh = dflat.reshape(a["pool5"].shape)
    table = cfg.conv_table()
    for i in reversed(range(len(table))):
        name, ci, co, hw, pool = table[i]
        if pool:
            dh = maxpool2_same_backward(dh, a["route%d" % i])
        dy = dh * (a[name] > 0)
        col = im2col3x3(a[name + "_in"])
        g["W_" + name] = col.reshape(-1, 9 * ci).T @ dy.reshape(-1, co)
        g["b_" + name] = dy.reshape(-1, co).sum(0)
        if i > 0:
            dh = col2im3x3(dy @ p["W_" + name].T, ci)
    return g


def encode(p, cfg, X):
    """Encoder trunk + z head + c head, base_models.py:220-249."""
    acts = {"x": X}
    h = cnn_trunk(p, cfg, X, acts) if cfg.cnn else X
    for i in range(len(cfg.enc_layers)):
        h = _dense(h, p["W_enc%d" % i], p["b_enc%d" % i], True)
        acts["enc%d" % i] = h
    acts["zh"] = _dense(h, p["W_zh"], p["b_zh"], True)
    acts["mean"] = _dense(acts["zh"], p["W_mean"], p["b_mean"], False)
    acts["logvar"] = _dense(acts["zh"], p["W_logvar"], p["b_logvar"], False)
    acts["ch"] = _dense(h, p["W_ch"], p["b_ch"], True)
    acts["logits"] = _dense(acts["ch"], p["W_logits"], p["b_logits"], False)
    return acts


def decode(p, cfg, Z, acts=None):
    """Decoder FullyConnected x3 + linear dense, base_models.py:279-293."""
    acts = {} if acts is None else acts
    h = Z
    for i in range(len(cfg.dec_layers)):
        h = _dense(h, p["W_dec%d" % i], p["b_dec%d" % i], True)
        acts["dec%d" % i] = h
    acts["xlogits"] = _dense(h, p["W_out"], p["b_out"], False)
    return acts


def recon_loss(cfg, X, xlogits):
    """base_models.py:72-85; binary branch = tf.nn.sigmoid_cross_entropy_with_
    logits = max(l,0) - l*x + log(1+exp(-|l|))."""
    if cfg.input_type == "binary":
        l = xlogits
        per = np.maximum(l, 0) - l * X + np.log1p(np.exp(-np.abs(l)))
        return np.mean(np.sum(per, axis=1))
    elif cfg.input_type == "real":
        return 0.5 * np.mean(np.sum(np.square(X - xlogits), axis=1))
    raise NotImplementedError


def forward(p, cfg, X, epsilon, kl_ratio=1.0, mode="exact", gumbel=None,
            temperature=1.0):
    """One forward pass of the loss, base_models.py:66-93,158-302.

    mode "exact": the checked-in graph (cluster_sample False, weights =
    softmax(logits)).  mode "relaxed": weights = Gumbel-Softmax sample feeding
    the cluster_sample=True KL branch (report/report.tex:330-333)."""
    a = encode(p, cfg, X)
    a["eps"] = epsilon
    a["Z"] = gaussian_reparam(a["mean"], a["logvar"], epsilon)
    decode(p, cfg, a["Z"], a)
    a["q"] = softmax(a["logits"])
    if mode == "exact":
        a["w"] = a["q"]
        a["kl_z"] = kl_mixture_exact(a["mean"], a["logvar"], a["w"],
                                     p["prior_means"], p["prior_log_vars"])
    elif mode == "relaxed":
        a["w"] = gumbel_softmax(a["logits"], gumbel, temperature)
        a["kl_z"] = kl_mixture_relaxed(a["mean"], a["logvar"], a["w"],
                                       p["prior_means"], p["prior_log_vars"])
    else:
        raise ValueError(mode)
    a["kl_c"] = kl_categorical(a["logits"], cfg.n_classes)
    a["recon"] = recon_loss(cfg, X, a["xlogits"])
    a["latent"] = a["kl_c"] + a["kl_z"]
    a["loss"] = a["recon"] + kl_ratio * a["latent"]
    a["mode"] = mode
    a["kl_ratio"] = kl_ratio
    a["temperature"] = temperature
    return a


def latent_backward(cfg, a, p, dZ):
    """Hand-derived gradients of kl_ratio*(KL_C+KL_Z) and of the reparam,
    SURVEY 8a row A13.  Returns dmean, dlogvar, dlogits, dprior_means,
    dprior_log_vars.  dZ is dLoss/dZ from the decoder."""
    mu, lv, eps = a["mean"], a["logvar"], a["eps"]
    pm, plv = p["prior_means"], p["prior_log_vars"]
    B = mu.shape[0]
    K = cfg.n_classes
    r = a["kl_ratio"]
    e = np.exp(lv)
    q = a["q"]
    e0 = 1e-20
    # KL_C = mean_b sum_k q (log(q+e0) + log K)
    dq = (r / B) * (np.log(q + e0) + q / (q + e0) + math.log(K))
    dlogits = q * (dq - np.sum(q * dq, axis=1, keepdims=True))
    if a["mode"] == "exact":
        w = a["w"]
        ip = np.exp(-plv)                                   # (K,D)
        diff = mu[:, None, :] - pm[None]                    # (B,K,D)
        t = plv[None] - lv[:, None, :] - 1 + (e[:, None, :] + diff ** 2) * ip[None]
        dw = (r / (2 * B)) * np.sum(t, axis=2)              # (B,K)
        dlogits = dlogits + w * (dw - np.sum(w * dw, axis=1, keepdims=True))
        gmu = (r / B) * np.sum(w[:, :, None] * diff * ip[None], axis=1)
        glv = (r / (2 * B)) * (e * (w @ ip) - 1)
        dpm = -(r / B) * np.sum(w[:, :, None] * diff * ip[None], axis=0)
        dplv = (r / (2 * B)) * np.sum(
            w[:, :, None] * (1 - (e[:, None, :] + diff ** 2) * ip[None]), axis=0)
    else:
        w = a["w"]
        tau = a["temperature"]
        bpm = w @ pm
        bplv = w @ plv
        ibp = np.exp(-bplv)
        diff = mu - bpm
        gmu = (r / B) * diff * ibp
        glv = (r / (2 * B)) * (e * ibp - 1)
        dbpm = -(r / B) * diff * ibp
        dbplv = (r / (2 * B)) * (1 - (e + diff ** 2) * ibp)
        dw = dbpm @ pm.T + dbplv @ plv.T                    # (B,K)
        dpm = w.T @ dbpm
        dplv = w.T @ dbplv
        dlogits = dlogits + (1.0 / tau) * w * (dw - np.sum(w * dw, axis=1, keepdims=True))
    dmean = dZ + gmu
    dlogvar = dZ * eps * 0.5 * np.exp(lv / 2) + glv
    return dmean, dlogvar, dlogits, dpm, dplv


def backward(p, cfg, a, masks=None):
    """Gradient of a["loss"] w.r.t. every trainable (what optimizer.minimize
    derives through tf.gradients, base_models.py:110).

    masks: optional {layer name: bool array} overriding the ReLU masks
    (y > 0).  Parity tests pass the masks of the implementation under test:
    a pre-activation within float32 rounding of zero may land on the other
    side in float64, and with ~5e5 units per step that happens."""
    if masks is not None:
        a = dict(a)
        for k, mk in masks.items():
            # only the sign pattern of y is used below
            a[k] = np.where(mk, np.maximum(a[k], 1e-300), 0.0)
    g = {}
    X = a["x"]
    B = X.shape[0]
    if cfg.input_type == "binary":
        l = a["xlogits"]
        dl = (1.0 / (1.0 + np.exp(-l)) - X) / B
    else:
        dl = (a["xlogits"] - X) / B
    nd = len(cfg.dec_layers)
    h = a["dec%d" % (nd - 1)]
    g["W_out"] = h.T @ dl
    g["b_out"] = dl.sum(0)
    dh = dl @ p["W_out"].T
    for i in reversed(range(nd)):
        y = a["dec%d" % i]
        dy = dh * (y > 0)
        xin = a["dec%d" % (i - 1)] if i > 0 else a["Z"]
        g["W_dec%d" % i] = xin.T @ dy
        g["b_dec%d" % i] = dy.sum(0)
        dh = dy @ p["W_dec%d" % i].T
    dZ = dh
    dmean, dlogvar, dlogits, dpm, dplv = latent_backward(cfg, a, p, dZ)
    g["prior_means"] = dpm
    g["prior_log_vars"] = dplv
    g["W_mean"] = a["zh"].T @ dmean
    g["b_mean"] = dmean.sum(0)
    g["W_logvar"] = a["zh"].T @ dlogvar
    g["b_logvar"] = dlogvar.sum(0)
    dzh = (dmean @ p["W_mean"].T + dlogvar @ p["W_logvar"].T) * (a["zh"] > 0)
    g["W_logits"] = a["ch"].T @ dlogits
    g["b_logits"] = dlogits.sum(0)
    dch = (dlogits @ p["W_logits"].T) * (a["ch"] > 0)
    ne = len(cfg.enc_layers)
    trunk = a["enc%d" % (ne - 1)]
    g["W_zh"] = trunk.T @ dzh
    g["b_zh"] = dzh.sum(0)
    g["W_ch"] = trunk.T @ dch
    g["b_ch"] = dch.sum(0)
    dh = dzh @ p["W_zh"].T + dch @ p["W_ch"].T
    for i in reversed(range(ne)):
        y = a["enc%d" % i]
        dy = dh * (y > 0)
        xin = a["enc%d" % (i - 1)] if i > 0 else (a["flat"] if cfg.cnn else X)
        g["W_enc%d" % i] = xin.T @ dy
        g["b_enc%d" % i] = dy.sum(0)
        if i > 0 or cfg.cnn:
            dh = dy @ p["W_enc%d" % i].T
    if cfg.cnn:
        cnn_trunk_backward(p, cfg, a, dh, g)
    return g


# --------------------------------------------------------------------------
# VaDE (base_models.py:435-562): one encoder 784 -> 2000 -> 500 -> 500 of FullyConnected layers,
# mean / log_var heads straight off the trunk, q(c|x) := p(c|z) = get_cluster_probs(Z)
# --------------------------------------------------------------------------


class VadeConfig:
    """Shapes of VaDE's MLP branch (base_models.py:490-499, :530-547): encoder and decoder are DeepNetworks of
    ("fc", ...) specs = FullyConnected layers, whose (1, out) bias is xavier-initialised like the weight
    (includes/layers.py:24-28); the mean / log_var / output layers are tf.layers.dense (zero bias)."""

    CONV_STACK, CONV_FLAT = Config.CONV_STACK, Config.CONV_FLAT

    def __init__(self, input_dim=784, latent_dim=10, n_classes=10, enc_layers=(2000, 500, 500), dec_layers=(500, 500, 2000),
                 input_type="binary", cnn=False):
        self.input_dim, self.latent_dim, self.n_classes = int(input_dim), int(latent_dim), int(n_classes)
        # cnn (base_models.py:456-488): the SAME conv / pool stack as DMVAE's (:181-201) ending in ("fc", 2048 -> 128)
        self.cnn = bool(cnn)
        self.enc_layers = tuple(int(v) for v in ((128,) if (cnn and tuple(enc_layers) == (2000, 500, 500)) else enc_layers))
        self.dec_layers = tuple(int(v) for v in dec_layers)
        self.input_type = input_type
        if self.cnn:
            assert self.input_dim == 784 and len(self.enc_layers) == 1, "cnn trunk: 784 inputs, one fc layer"

    def layer_table(self):
        t, prev = [], (self.CONV_FLAT if self.cnn else self.input_dim)
        for i, h in enumerate(self.enc_layers):
            t.append(("enc%d" % i, prev, h, "xavier"))
            prev = h
        t.append(("mean", prev, self.latent_dim, "zero"))
        t.append(("logvar", prev, self.latent_dim, "zero"))
        prev = self.latent_dim
        for i, h in enumerate(self.dec_layers):
            t.append(("dec%d" % i, prev, h, "xavier"))
            prev = h
        t.append(("out", prev, self.input_dim, "zero"))
        return t

    def conv_table(self):
        return self.CONV_STACK if self.cnn else ()

    def n_params(self):
        return (sum(9 * ci * co + co for _, ci, co, _, _ in self.conv_table()) +
                sum(fi * fo + fo for _, fi, fo, _ in self.layer_table()) + 2 * self.n_classes * self.latent_dim)


def vade_forward(p, cfg, X, epsilon, kl_ratio=1.0):
    """base_models.py:443-562 + :66-93: Z = mean + exp(log_var/2) eps; cluster_probs = get_cluster_probs(Z)
    (priors.py:91-102) are the mixture weights of the exact KL (priors.py:131-145, cluster_sample False) AND the
    probabilities of the categorical KL (priors.py:183-201, "probs" branch)."""
    a = {"x": X, "eps": epsilon}
    h = cnn_trunk(p, cfg, X, a) if cfg.cnn else X           # cnn: X_flat = reshape(X, (-1, 28, 28, 1)), base_models.py:455,482
    for i in range(len(cfg.enc_layers)):
        h = _dense(h, p["W_enc%d" % i], p["b_enc%d" % i], True)
        a["enc%d" % i] = h
    a["mean"] = _dense(h, p["W_mean"], p["b_mean"], False)
    a["logvar"] = _dense(h, p["W_logvar"], p["b_logvar"], False)
    a["Z"] = gaussian_reparam(a["mean"], a["logvar"], epsilon)
    decode(p, cfg, a["Z"], a)
    a["w"] = cluster_probs(a["Z"], p["prior_means"], p["prior_log_vars"])
    a["kl_z"] = kl_mixture_exact(a["mean"], a["logvar"], a["w"], p["prior_means"], p["prior_log_vars"])
    q = a["w"]
    a["kl_c"] = np.mean(np.sum(q * (np.log(q + 1e-20) - np.log(1.0 / cfg.n_classes)), axis=1))
    a["recon"] = recon_loss(cfg, X, a["xlogits"])
    a["latent"] = a["kl_c"] + a["kl_z"]
    a["loss"] = a["recon"] + kl_ratio * a["latent"]
    a["kl_ratio"] = kl_ratio
    return a


def vade_latent_backward(cfg, a, p, dZ_dec):
    """Hand-derived gradient of kl_ratio * (KL_C + KL_Z) for VaDE, where the weights gamma = softmax_k(u),
    u_k = -1/2 [sum_d (z_d - pm_kd)^2 ip_kd + sum_d plv_kd] depend on Z (hence on mean, log_var, eps) and on the prior
    tables.  Returns dmean, dlogvar, dprior_means, dprior_log_vars, and the two latent outputs the kernel hands to
    the dZ GEMM epilogue: gmu' = dKL/dmean (direct + through Z), glv' = dKL/dlog_var (direct + through Z)."""
    mu, lv, eps, Z, g = a["mean"], a["logvar"], a["eps"], a["Z"], a["w"]
    pm, plv = p["prior_means"], p["prior_log_vars"]
    B, K, r, e0 = mu.shape[0], cfg.n_classes, a["kl_ratio"], 1e-20
    e, ip = np.exp(lv), np.exp(-plv)
    diff = mu[:, None, :] - pm[None]                                   # (B,K,D)
    T = np.sum(plv[None] - lv[:, None, :] - 1 + (e[:, None, :] + diff ** 2) * ip[None], axis=2)
    G = (r / B) * (0.5 * T + np.log(g + e0) + g / (g + e0) + math.log(K))          # dL/dgamma
    du = g * (G - np.sum(g * G, axis=1, keepdims=True))                 # softmax backward -> dL/du
    zd = Z[:, None, :] - pm[None]
    dZl = -np.sum(du[:, :, None] * zd * ip[None], axis=1)               # through gamma(Z)
    gmu = (r / B) * np.sum(g[:, :, None] * diff * ip[None], axis=1)
    glv = (r / (2 * B)) * (e * (g @ ip) - 1)
    dpm = -(r / B) * np.sum(g[:, :, None] * diff * ip[None], axis=0) + np.sum(du[:, :, None] * zd * ip[None], axis=0)
    dplv = (r / (2 * B)) * np.sum(g[:, :, None] * (1 - (e[:, None, :] + diff ** 2) * ip[None]), axis=0) \
        + np.sum(du[:, :, None] * 0.5 * (zd ** 2 * ip[None] - 1), axis=0)
    clv = eps * 0.5 * np.exp(lv / 2)
    gmu2, glv2 = gmu + dZl, glv + dZl * clv
    return dZ_dec + gmu2, dZ_dec * clv + glv2, dpm, dplv, gmu2, glv2


def vade_backward(p, cfg, a, masks=None):
    """gradient of a["loss"] w.r.t. every trainable of the VaDE graph (what optimizer.minimize derives)"""
    if masks is not None:
        a = dict(a)
        for k, mk in masks.items():
            a[k] = np.where(mk, np.maximum(a[k], 1e-300), 0.0)
    g = {}
    X = a["x"]
    B = X.shape[0]
    dl = ((1.0 / (1.0 + np.exp(-a["xlogits"])) - X) if cfg.input_type == "binary" else (a["xlogits"] - X)) / B
    nd = len(cfg.dec_layers)
    g["W_out"] = a["dec%d" % (nd - 1)].T @ dl
    g["b_out"] = dl.sum(0)
    dh = dl @ p["W_out"].T
    for i in reversed(range(nd)):
        dy = dh * (a["dec%d" % i] > 0)
        xin = a["dec%d" % (i - 1)] if i > 0 else a["Z"]
        g["W_dec%d" % i] = xin.T @ dy
        g["b_dec%d" % i] = dy.sum(0)
        dh = dy @ p["W_dec%d" % i].T
    dmean, dlogvar, dpm, dplv, _, _ = vade_latent_backward(cfg, a, p, dh)
    g["prior_means"], g["prior_log_vars"] = dpm, dplv
    ne = len(cfg.enc_layers)
    trunk = a["enc%d" % (ne - 1)]
    g["W_mean"], g["b_mean"] = trunk.T @ dmean, dmean.sum(0)
    g["W_logvar"], g["b_logvar"] = trunk.T @ dlogvar, dlogvar.sum(0)
    dh = dmean @ p["W_mean"].T + dlogvar @ p["W_logvar"].T
    for i in reversed(range(ne)):
        dy = dh * (a["enc%d" % i] > 0)
        xin = a["enc%d" % (i - 1)] if i > 0 else (a["flat"] if cfg.cnn else X)
        g["W_enc%d" % i] = xin.T @ dy
        g["b_enc%d" % i] = dy.sum(0)
        if i > 0 or cfg.cnn:
            dh = dy @ p["W_enc%d" % i].T
    if cfg.cnn:
        cnn_trunk_backward(p, cfg, a, dh, g)
    return g


def vade_train_step(p, m, v, t, cfg, X, epsilon, kl_ratio=1.0, lr=0.002):
    a = vade_forward(p, cfg, X, epsilon, kl_ratio)
    g = vade_backward(p, cfg, a)
    adam_tf(p, g, m, v, t, lr)
    return a, g


# --------------------------------------------------------------------------
# Optimizer and epoch semantics
# --------------------------------------------------------------------------


def adam_tf_init(p):
    return ({k: np.zeros_like(v) for k, v in p.items()},
            {k: np.zeros_like(v) for k, v in p.items()})


def adam_tf(p, g, m, v, t, lr=0.002, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """tf.train.AdamOptimizer (TF 1.x) update, base_models.py:95-110:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)
    (eps outside the bias correction).  lr is constant.  t starts
    at 1.  Updates p, m, v in place for the keys present in g."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k in g:
        m[k] = beta1 * m[k] + (1 - beta1) * g[k]
        v[k] = beta2 * v[k] + (1 - beta2) * g[k] * g[k]
        p[k] = p[k] - (lr_t * m[k] / (np.sqrt(v[k]) + epsilon)).astype(p[k].dtype)
    return p, m, v


def train_step(p, m, v, t, cfg, X, epsilon, kl_ratio=1.0, lr=0.002,
               mode="exact", gumbel=None, temperature=1.0):
    """One session.run([loss, train_step]) of base_models.py:126-129."""
    a = forward(p, cfg, X, epsilon, kl_ratio, mode, gumbel, temperature)
    g = backward(p, cfg, a)
    adam_tf(p, g, m, v, t, lr)
    return a, g


class Dataset:
    """includes/utils.py:428-466: shuffle on construction and at every
    get_batches(); emit consecutive batches, last one may be short."""

    def __init__(self, data, batch_size=100, shuffle=True, rng=None):
        data, classes = data
        self.rng = np.random if rng is None else rng
        self.data = np.copy(data)
        self.classes = np.copy(classes)
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.data_dim = self.data.shape[1]
        self.epoch_len = int(math.ceil(len(self.data) / batch_size))
        if shuffle:
            idx = self.rng.permutation(len(self.data))
            self.data = self.data[idx]
            self.classes = self.classes[idx]

    def get_batches(self):
        if self.shuffle:
            idx = self.rng.permutation(len(self.data))
            self.data = self.data[idx]
            self.classes = self.classes[idx]
        for s in range(0, len(self.data), self.batch_size):
            yield self.data[s:s + self.batch_size]

    def __len__(self):
        return self.epoch_len


def train_epoch(p, m, v, t0, cfg, data, noise_fn, kl_ratio=1.0, lr=0.002):
    """VAE.train_op, base_models.py:112-132: loss += batch_loss/epoch_len."""
    loss = 0.0
    t = t0
    for batch in data.get_batches():
        t += 1
        a, _ = train_step(p, m, v, t, cfg, batch, noise_fn(len(batch)), kl_ratio, lr)
        loss += a["loss"] / data.epoch_len
    return loss, t


def clustering_accuracy(weights, classes):
    """includes/utils.py:22-34 with scipy's Hungarian solver in place of the
    removed sklearn.utils.linear_assignment_."""
    from scipy.optimize import linear_sum_assignment
    clusters = np.argmax(weights, axis=-1)
    n = weights.shape[1]
    d = np.zeros((n, n), dtype=np.int64)
    for c, y in zip(clusters, classes):
        d[c, y] += 1
    r, c = linear_sum_assignment(d.max() - d)
    return d[r, c].sum() / float(len(clusters))


def synthetic_images(n, dim=784, seed=0, density=0.19, dtype=np.float32):
    """Deterministic MNIST-like stand-in: x = u * 1[v < density]."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, dim), dtype=np.float32)
    v = rng.random((n, dim), dtype=np.float32)
    return (u * (v < density)).astype(dtype)
