"""Diagnostic: the CNN model through the class surface, a few epochs at the CLI's learning rate, over shuffle seeds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmvae_oracle as O
import base_models
from includes.utils import Dataset
X = O.synthetic_images(512, 784, seed=4)
y = np.random.RandomState(0).randint(0, 10, 512)
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for seed in range(12):
    np.random.seed(seed)
    m = base_models.DeepMixtureVAE("c", "binary", 784, 8, 10, activation="relu", initializer="xavier", cnn=True,
                                   batch_size=128, dtype=dtype, head_dim=256, dec_layers=(256, 128)).build_graph()
    data = Dataset((X, y), batch_size=128)
    m.define_train_step(0.002, data.epoch_len * 10)
    print(dtype, seed, ["%.1f" % m.train_op(None, data, 1.0) for _ in range(5)], flush=True)
