"""Reconstruction / generation outputs of code/includes/visualization.py:20-129
(mnist_regeneration_plot, mnist_sample_plot) for the DMVAE drop-in.

What a "reconstruction" and a "sample" are is taken from the reference:
  * regeneration: the first 100 rows of the data set, every reparameterisation
    variable fed as ZEROS (Z = mean), fetch reconstructed_X       (:36-46)
  * samples: for every cluster c, z ~ N(mu_c, sigma_c^2) from the prior tables
    (sample_generative_feed(1000, Z={"c": c})), the first 100 decoded; panel
    (c, j) shows out[10*c + j]                                      (:75-91)
The figures are written as plain 8-bit greyscale PNGs (own zlib writer: the
pixel grids are the product, no plotting library is needed on the GPU box) to
the reference's paths plots/<model.name>/mnist/{regenerated,sampled}.png; the
grids are also returned.  tsne=True adds the t-SNE scatter of the prior samples (:93-110) as an RGB panel.
"""
import os
import struct
import zlib

import numpy as np


def _write_png(path, img):
    """img: [h, w] greyscale or [h, w, 3] RGB, values in [0, 255]."""
    img = np.clip(np.asarray(img), 0, 255).astype(np.uint8)
    h, w = img.shape[:2]
    ctype = 2 if img.ndim == 3 else 0
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    png += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "wb") as f:
        f.write(png)


def _grid(images, side=28):
    """[100, side*side] -> [10*side, 10*side], row-major 10 x 10 (visualization.py:29-35)."""
    images = np.asarray(images).reshape((10, 10, side, side))
    return images.transpose(0, 2, 1, 3).reshape(10 * side, 10 * side)


def regenerate(model, data, sess=None):
    """(originals, reconstructions) of the first 100 rows, epsilon = 0."""
    orig_X = np.asarray(data.data[:100], dtype=np.float32)
    return orig_X, model.reconstruct(orig_X, epsilon=None)


def mnist_regeneration_plot(model, data, sess=None):
    orig_X, recn_X = regenerate(model, data, sess)
    side = int(round(np.sqrt(orig_X.shape[1])))
    left, right = _grid(orig_X, side) * 255.0, _grid(recn_X, side) * 255.0
    gap = np.full((left.shape[0], side // 2), 255.0)
    figure = np.concatenate([left, gap, right], axis=1)
    _write_png("plots/%s/mnist/regenerated.png" % model.name, figure)
    return left, right


def sample_clusters(model, sess=None, n=1000):
    """per cluster c: z ~ N(mu_c, sigma_c^2) [n, D] and the decoded first 100 (visualization.py:75-86)."""
    sample_Z, decoded = [], []
    for i in range(model.n_classes):
        z = model.sample_generative_feed(n, Z={"session": sess, "c": i})["Z"]
        sample_Z.append(z)
        decoded.append(model.decode(np.asarray(z[:100], dtype=np.float32)))
    return sample_Z, decoded


# ten cluster colours for the scatter (RGB)
_COLORS = np.array([[31, 119, 180], [255, 127, 14], [44, 160, 44], [214, 39, 40], [148, 103, 189], [140, 86, 75],
                    [227, 119, 194], [127, 127, 127], [188, 189, 34], [23, 190, 207]], dtype=np.float64)


def cluster_scatter(sample_Z, side=280):
    """visualization.py:93-110: the per-cluster prior samples (1000 each) reduced to two dimensions by t-SNE when the
    latent space has more than two, drawn as one dot per sample coloured by cluster (colors[k % 10]) -> [side, side, 3]."""
    Z = np.concatenate(sample_Z, axis=0)
    n_per = len(sample_Z[0])
    if Z.shape[1] > 2:
        from sklearn.manifold import TSNE
        Z = TSNE(n_components=2).fit_transform(Z)
    lo, hi = Z.min(0), Z.max(0)
    xy = np.clip(((Z - lo) / np.maximum(hi - lo, 1e-12) * (side - 5) + 2).astype(int), 0, side - 1)
    img = np.full((side, side, 3), 255.0)
    for k in range(len(sample_Z)):
        pts = xy[n_per * k:n_per * (k + 1)]
        img[side - 1 - pts[:, 1], pts[:, 0]] = _COLORS[k % 10]
    return img


def mnist_sample_plot(model, sess=None, tsne=False):
    sample_Z, decoded = sample_clusters(model, sess)
    side = int(round(np.sqrt(decoded[0].shape[1])))
    figure = np.zeros((side * model.n_classes, side * 10))
    for i in range(model.n_classes):
        for j in range(10):
            # the reference indexes out[10*i + j] of the 100 decoded samples of cluster i (:88-91)
            figure[i * side:(i + 1) * side, j * side:(j + 1) * side] = decoded[i][(10 * i + j) % 100].reshape(side, side) * 255
    if tsne:      # the sample grid on the left, the scatter of the prior samples on the right (visualization.py:93-110)
        h = figure.shape[0]
        sc = cluster_scatter(sample_Z, side=h)
        rgb = np.repeat(figure[:, :, None], 3, axis=2)
        _write_png("plots/%s/mnist/sampled.png" % model.name, np.concatenate([rgb, np.full((h, 14, 3), 255.0), sc], axis=1))
        return figure
    _write_png("plots/%s/mnist/sampled.png" % model.name, figure)
    return figure
