"""Where do fused (train_step) and unfused (forward_backward + update) runs diverge?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmvae_oracle as O
from dmvae_hip import StepEngine
kw, B = dict(input_dim=784, latent_dim=64, n_classes=10), 256
rng = np.random.RandomState(11)
Xd = torch.as_tensor(O.synthetic_images(B, 784, seed=5)).cuda()
engs = []
for _ in range(2):
    e = StepEngine(dtype="bf16", max_batch=B, mode="exact", deterministic=True, **kw); e.init_parameters(9); engs.append(e)
for step in range(3):
    ed = torch.as_tensor(rng.randn(B, 64).astype(np.float32)).cuda()
    for fused, e in zip((False, True), engs):
        e.load_batch(Xd, None, 0, B)
        if fused: e.forward_backward_update(B, ed)
        else: e.forward_backward(B, ed); e.update(1.0)
    torch.cuda.synchronize()
    a, b = engs
    print("step", step, "lr_t", a.read_state().lr_t, b.read_state().lr_t, "adam_t", a.read_state().adam_t, b.read_state().adam_t)
    for name in ("param", "m", "v"):
        d = (getattr(a, name) != getattr(b, name)).nonzero().flatten()
        print("  ", name, "differing elements:", d.numel(), "max abs diff %.3e" % (getattr(a, name) - getattr(b, name)).abs().max().item())
        if d.numel():
            idx = d.cpu().numpy()
            for nm, (off, rows, cols, ld) in a.tensors.items():
                lo, hi = off, off + max(1, rows) * ld
                n = int(((idx >= lo) & (idx < hi)).sum())
                if n: print("      in %-16s %d of %d (first rel idx %s)" % (nm, n, hi - lo, (idx[(idx >= lo) & (idx < hi)][:6] - lo).tolist()))
