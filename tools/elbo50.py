"""ELBO@50ep (BASELINE.json metric, second half): 50 epochs of the cfg2 model on the deterministic
synthetic 28x28 generator (no MNIST files in this image), bf16 throughput mode against the fp32
parity-mode engine on IDENTICAL batches and the identical noise stream (device Philox, same seed:
the draw depends on (seed, step, row, column) only, not on the arithmetic type).
ELBO = -(epoch-mean loss) in nats/image as VAE.train_op computes it (base_models.py:130).
    python tools/elbo50.py [--epochs 50] [--out gpurun_out/elbo50.json]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import StepEngine

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=50)
ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--latent_dim", type=int, default=64)
ap.add_argument("--n_clusters", type=int, default=10)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "elbo50.json"))
args = ap.parse_args()

torch.cuda.set_device(0)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
data = torch.rand((args.rows, 784), device="cuda", generator=gen)
data = data * (torch.rand((args.rows, 784), device="cuda", generator=gen) < 0.19)
bpe = args.rows // args.batch
pg = torch.Generator(device="cuda"); pg.manual_seed(1)
perms = [torch.randperm(args.rows, device="cuda", generator=pg).to(torch.int32) for _ in range(args.epochs)]

curves, times = {}, {}
for dtype in ("bf16", "fp32"):
    eng = StepEngine(784, args.latent_dim, args.n_clusters, dtype=dtype, max_batch=args.batch, seed=1234)
    eng.init_parameters(0)
    curve = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for ep in range(args.epochs):
        eng.reset_epoch(bpe, kl_ratio=1.0)
        for _ in range(bpe):
            eng.train_step(data, perms[ep], use_state_cursor=True)
        torch.cuda.synchronize()
        curve.append(float(eng.read_state().epoch_loss))
    times[dtype] = time.perf_counter() - t0
    curves[dtype] = curve
    print("%s: epoch-mean loss  ep1 %.4f  ep10 %.4f  ep%d %.4f   (%.2f s)" % (dtype, curve[0], curve[min(9, len(curve) - 1)], args.epochs, curve[-1], times[dtype]), flush=True)
rel = abs(curves["bf16"][-1] - curves["fp32"][-1]) / abs(curves["fp32"][-1])
out = {"metric": "ELBO@%dep (nats/image, = -epoch-mean loss), cfg2 model, synthetic 28x28 stand-in for MNIST" % args.epochs,
       "data": "synthetic x = u*1[v<0.19], %d rows, batch %d, per-epoch device permutation (seed 1), device Philox noise (seed 1234)" % (args.rows, args.batch),
       "elbo_bf16": -curves["bf16"][-1], "elbo_fp32_parity_mode": -curves["fp32"][-1], "relative_difference": rel,
       "tolerance": 1e-3, "within_tolerance": bool(rel <= 1e-3),
       "loss_curve_bf16": curves["bf16"], "loss_curve_fp32": curves["fp32"], "wall_seconds": times}
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(out, open(args.out, "w"), indent=1)
print("ELBO@%dep bf16 %.4f  fp32 %.4f  relative difference %.2e (tolerance 1e-3: %s)" % (args.epochs, out["elbo_bf16"], out["elbo_fp32_parity_mode"], rel, "ok" if rel <= 1e-3 else "EXCEEDED"))
