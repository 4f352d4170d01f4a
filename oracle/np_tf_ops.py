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
