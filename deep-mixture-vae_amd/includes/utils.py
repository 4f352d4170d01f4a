"""Host-side data utilities of the DMVAE drop-in -- the counterpart of the parts of
code/includes/utils.py that the hot path touches: sample_gumbel (:17-19),
get_clustering_accuracy (:22-34), load_data("mnist") (:122-148) and Dataset
(:428-466).  The MoE label generators, MEDataset and the other loaders are out
of scope."""
import gzip
import math
import os
import struct
import types

import numpy as np


def sample_gumbel(shape, eps=1e-20):
    """includes/utils.py:17-19 (same global NumPy RNG, same formula)."""
    U = np.random.uniform(0, 1, shape)
    return -np.log(eps - np.log(U + eps))


def get_clustering_accuracy(weights, classes):
    """includes/utils.py:22-34; scipy's Hungarian solver replaces the removed
    sklearn.utils.linear_assignment_."""
    from scipy.optimize import linear_sum_assignment
    clusters = np.argmax(weights, axis=-1)
    n_classes = weights.shape[1]
    size = len(clusters)
    d = np.zeros((n_classes, n_classes), dtype=np.int64)
    np.add.at(d, (clusters, np.asarray(classes, dtype=np.int64)), 1)
    r, c = linear_sum_assignment(d.max() - d)
    return d[r, c].sum() / (size * 1.0)


def synthetic_images(n, dim=784, seed=0, density=0.19):
    """Deterministic MNIST-like stand-in used when no idx files are present and
    by the benchmark: x = u * 1[v < density], u, v ~ U[0,1)."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, dim), dtype=np.float32)
    v = rng.random((n, dim), dtype=np.float32)
    return (u * (v < density)).astype(np.float32)


def _read_idx(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        magic, = struct.unpack(">I", f.read(4))
        nd = magic & 0xFF
        dims = struct.unpack(">" + "I" * nd, f.read(4 * nd))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(dims)


def _find(root, stem):
    for ext in ("", ".gz"):
        for sep in ("-", "."):
            p = os.path.join(root, stem.replace("-idx", sep + "idx") + ext)
            if os.path.exists(p):
                return p
    return None


def load_data(datagroup, **args):
    """load_data("mnist") of the reference (includes/utils.py:77,122-148): float
    grey levels in [0,1] used as soft Bernoulli targets, NOT binarised (SURVEY
    F6); TF's 55 000 / 10 000 train/test split.  Reads idx files from
    data/<datagroup>/ when present; otherwise (no network in this image) a
    deterministic synthetic stand-in of the same shapes, flagged in
    dataset.synthetic."""
    if datagroup not in ("mnist", "fashion-mnist", "synthetic"):
        raise NotImplementedError("dataset %r: only the MNIST-shaped DMVAE path is built (SURVEY.md 2.1)" % datagroup)
    ds = types.SimpleNamespace(datagroup=datagroup, input_dim=784, input_type="binary", n_classes=10,
                               sample_plot=None, regeneration_plot=None, synthetic=False)
    root = os.path.join(os.environ.get("DMVAE_DATA", "data"), datagroup)
    ti = _find(root, "train-images-idx3-ubyte") if datagroup != "synthetic" else None
    if ti:
        tl, si, sl = (_find(root, s) for s in ("train-labels-idx1-ubyte", "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte"))
        tr = _read_idx(ti).reshape(-1, 784).astype(np.float32) / 255.0
        te = _read_idx(si).reshape(-1, 784).astype(np.float32) / 255.0
        ds.train_data, ds.train_classes = tr[:55000], _read_idx(tl)[:55000].astype(np.int64)
        ds.test_data, ds.test_classes = te, _read_idx(sl).astype(np.int64)
    else:
        ds.synthetic = True
        n_tr, n_te = int(args.get("n_train", 55000)), int(args.get("n_test", 10000))
        allx = synthetic_images(n_tr + n_te, 784, seed=0)
        cls = np.random.RandomState(0).randint(0, 10, n_tr + n_te)
        ds.train_data, ds.train_classes = allx[:n_tr], cls[:n_tr]
        ds.test_data, ds.test_classes = allx[n_tr:], cls[n_tr:]
    ds.train_labels = ds.test_labels = None
    return ds


class Dataset:
    """includes/utils.py:428-466.  Same contract: shuffle on construction and at
    every get_batches() with the global NumPy RNG, consecutive batches, short
    last batch.  Instead of physically permuting the rows each epoch the class
    keeps the cumulative row order; `data` / `classes` present the permuted
    view, and the device-resident copy is gathered by that order on the GPU
    (dmvae_gather_rows)."""

    def __init__(self, data, batch_size=100, shuffle=True):
        data, classes = data
        self._rows = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        self._cls = np.copy(classes)
        self.order = np.arange(len(self._rows))
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.data_dim = self._rows.shape[1]
        self.epoch_len = int(math.ceil(len(self._rows) / batch_size))
        self._device = {}
        if shuffle:
            self.order = self.order[np.random.permutation(len(self._rows))]

    @property
    def data(self):
        return self._rows[self.order]

    @property
    def classes(self):
        return self._cls[self.order]

    def reshuffle(self):
        """the per-epoch shuffle of get_batches (utils.py:450-454); returns the new row order"""
        if self.shuffle:
            self.order = self.order[np.random.permutation(len(self._rows))]
        return self.order

    def get_batches(self):
        order = self.reshuffle()
        for s in range(0, len(order), self.batch_size):
            yield self._rows[order[s:s + self.batch_size]]

    def device_rows(self, device):
        """the unpermuted rows, resident in HBM (uploaded once)"""
        import torch
        key = str(device)
        if key not in self._device:
            self._device[key] = torch.as_tensor(self._rows).to(device)
        return self._device[key]

    def __len__(self):
        return self.epoch_len
