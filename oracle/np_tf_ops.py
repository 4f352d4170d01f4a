"""NumPy evaluation of the dozen TensorFlow ops that the reference's
code/priors.py calls -- TEST INFRASTRUCTURE ONLY, used by oracle/make_golden.py
in the build container to execute the reference's own latent-variable code
(TensorFlow 1.x itself is not installed and cannot be).

Each function is the documented eager meaning of the TF op of the same name on
float64 ndarrays; nothing here belongs to the reference.  Variables created by
tf.get_variable become plain ndarrays that the generator overwrites with the
prior tables of each golden case.
"""
import contextlib
import types

import numpy as np

float32 = np.float32
exp = np.exp
square = np.square
log = np.log
matmul = np.matmul


def reduce_sum(x, axis=None, keep_dims=False, keepdims=False):
    return np.sum(x, axis=axis, keepdims=bool(keep_dims or keepdims))


def reduce_mean(x, axis=None, keep_dims=False, keepdims=False):
    return np.mean(x, axis=axis, keepdims=bool(keep_dims or keepdims))


def reshape(x, shape):
    return np.reshape(x, shape)


def add_n(xs):
    out = xs[0]
    for x in xs[1:]:
        out = out + x
    return out


def _softmax(x, axis=-1):
    x = x - np.max(x, axis=axis, keepdims=True)
    e = np.exp(x)
    return e / np.sum(e, axis=axis, keepdims=True)


nn = types.SimpleNamespace(softmax=_softmax)


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    yield name


def _random_normal(shape, dtype=np.float64):
    return np.random.randn(*shape).astype(np.float64)


def _zeros(shape, dtype=np.float64):
    return np.zeros(shape, np.float64)


initializers = types.SimpleNamespace(random_normal=_random_normal, zeros=_zeros)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    return initializer(tuple(shape))


# ---- additions for code/includes/layers.py + network.py (FullyConnected / DeepNetwork, the decoder
# of base_models.py:279-289): eager NumPy meaning of the ops their class bodies and _call use.
def _relu(x):
    return np.maximum(x, 0.0)


nn.relu = _relu


@contextlib.contextmanager
def name_scope(name, *a, **k):
    yield name


def _flatten(x):
    x = np.asarray(x)
    return x.reshape(x.shape[0], -1)


layers = types.SimpleNamespace(flatten=_flatten)


def _xavier_initializer():
    """tf.contrib.layers.xavier_initializer(): uniform, limit sqrt(6 / (fan_in + fan_out)) -- TensorFlow's
    documented rule restated (the golden generator overwrites the variables anyway; only shapes matter)."""
    def init(shape, dtype=np.float64):
        fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (int(np.prod(shape[:-1])), shape[-1])
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return np.random.uniform(-lim, lim, size=shape).astype(np.float64)
    return init


contrib = types.SimpleNamespace(layers=types.SimpleNamespace(xavier_initializer=_xavier_initializer))


# ---- additions for code/base_models.py's VAE base class (constructor placeholders only): a placeholder is
# an opaque hashable token that the mocked session of oracle/make_trainop_golden.py finds in its feed dict.
class _Token:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<placeholder %s>" % self.name


def placeholder_with_default(value, shape=None, name=None):
    return _Token(name)


def placeholder(dtype=None, shape=None, name=None):
    return _Token(name)


# ---- additions for the Convolution / MaxPooling layers of code/includes/layers.py:39-77 (the CNN trunk of
# base_models.py:178-216).  Written as plain loops over taps / windows -- deliberately NOT the im2col
# formulation of oracle/dmvae_oracle.py, so that the golden vectors check it.
def _same_pads(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2       # TF puts the odd pixel at the end


def _conv2d(inputs, W, strides=(1, 1, 1, 1), padding="SAME"):
    assert padding == "SAME"
    x = np.asarray(inputs, np.float64)
    B, H, Wd, C = x.shape
    kh, kw, ci, co = W.shape
    assert ci == C
    sh, sw = strides[1], strides[2]
    Ho, pt, pb = _same_pads(H, kh, sh)
    Wo, pl, pr = _same_pads(Wd, kw, sw)
    xp = np.zeros((B, H + pt + pb, Wd + pl + pr, C))
    xp[:, pt:pt + H, pl:pl + Wd] = x
    out = np.zeros((B, Ho, Wo, co))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (Ho - 1) * sh + 1:sh, kx:kx + (Wo - 1) * sw + 1:sw]
            out += np.tensordot(patch, W[ky, kx], axes=([3], [0]))
    return out


def _bias_add(x, b):
    return x + np.asarray(b).reshape((1,) * (np.ndim(x) - 1) + (-1,))


def _max_pool(inputs, ksize, strides, padding="SAME"):
    assert padding == "SAME"
    x = np.asarray(inputs, np.float64)
    B, H, Wd, C = x.shape
    kh, kw, sh, sw = ksize[1], ksize[2], strides[1], strides[2]
    Ho, pt, pb = _same_pads(H, kh, sh)
    Wo, pl, pr = _same_pads(Wd, kw, sw)
    xp = np.full((B, H + pt + pb, Wd + pl + pr, C), -np.inf)
    xp[:, pt:pt + H, pl:pl + Wd] = x
    out = np.empty((B, Ho, Wo, C))
    for i in range(Ho):
        for j in range(Wo):
            out[:, i, j] = xp[:, i * sh:i * sh + kh, j * sw:j * sw + kw].max(axis=(1, 2))
    return out


nn.conv2d = _conv2d
nn.bias_add = _bias_add
nn.max_pool = _max_pool
