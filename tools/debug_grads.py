"""Per-tensor gradient error table: HIP step vs float64 oracle (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "deep-mixture-vae_amd"), os.path.join(ROOT, "oracle")]
import dmvae_oracle as O
from dmvae_hip import StepEngine

def run(dtype, B, D=10, seed=1, perturb=False):
    kw = dict(input_dim=784, latent_dim=D, n_classes=10)
    eng = StepEngine(dtype=dtype, max_batch=B, deterministic=True, **kw)
    eng.init_parameters(0)
    cfg = O.Config(784, D, 10)
    rng = np.random.RandomState(seed)
    p = {k: v.astype(np.float64) for k, v in eng.get_parameters().items()}
    if perturb:
        for k in p:
            if k.startswith("b_"):
                p[k] = (rng.randn(*p[k].shape) * 0.05).astype(np.float32).astype(np.float64)
        eng.set_parameters(p)
    X = O.synthetic_images(B, 784, seed=3)
    eps = rng.randn(B, D).astype(np.float32)
    eng.load_batch(torch.as_tensor(X).cuda(), None, 0, B)
    eng.forward_backward(B, torch.as_tensor(eps).cuda())
    torch.cuda.synchronize()
    a = O.forward(p, cfg, X.astype(np.float64), eps.astype(np.float64))
    g = O.backward(p, cfg, a)
    st = eng.read_state()
    print("== %s B=%d  loss gpu %.6f oracle %.6f" % (dtype, B, st.last_loss, a["loss"]))
    gg = eng.get_gradients()
    for k in g:
        d = gg[k] - g[k]
        print("%-16s max|g| %.3e  maxerr/max|g| %.2e  fro rel %.2e" % (
            k, np.abs(g[k]).max(), np.abs(d).max() / (np.abs(g[k]).max() + 1e-30),
            np.linalg.norm(d) / (np.linalg.norm(g[k]) + 1e-30)))

run("fp32", 100, perturb=True)
