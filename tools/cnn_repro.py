"""Diagnostic: run-to-run reproducibility of the bf16 CNN step (split-K atomics are the only legitimate source of
differences: last-bit changes in the conv weight gradients).  Prints the largest loss difference between repeated
identical runs, per step."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmvae_oracle as O
from dmvae_hip import StepEngine
B, steps, reps = 128, 24, 6
X = torch.as_tensor(O.synthetic_images(4 * B, 784, seed=4)).cuda()
perm = torch.arange(4 * B, dtype=torch.int32).cuda()
runs = []
for rep in range(reps):
    eng = StepEngine(784, 8, 10, enc_layers=(500,), head_dim=256, dec_layers=(256, 128), dtype="bf16", max_batch=B, cnn=True, seed=5)
    eng.init_parameters(0)
    eng.reset_epoch(4, kl_ratio=1.0)
    use_graph = rep % 2 == 1
    step = eng.capture_step(X, perm) if use_graph else (lambda: eng.train_step(X, perm, use_state_cursor=True))
    ls = []
    for _ in range(steps):
        step()
        ls.append(eng.read_state().last_loss)
    runs.append(ls)
    print("graph" if use_graph else "eager", " ".join("%.2f" % v for v in ls[:12]), flush=True)
runs = np.array(runs)
print("max |loss difference| between runs, per step:", " ".join("%.3f" % v for v in (runs.max(0) - runs.min(0))))
