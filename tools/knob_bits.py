"""Does a tuning knob change the step's BITS?  Two engines, same seed, same batches, `steps` eager steps each, one under knob = a, one under knob = b;
prints the number of parameter / moment elements that differ (0 = bit-identical) and the two losses.
python tools/knob_bits.py <cfg1..cfg5>[:batch=N] <knob> <a> <b> [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, ROOT)
from dmvae_hip import StepEngine, _lib as L
import bench
cfg = dict(bench.PRESETS[sys.argv[1].split(":")[0]])
for kv in sys.argv[1].split(":")[1:]: cfg[kv.split("=")[0]] = int(kv.split("=")[1])
which, va, vb = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]); steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
torch.cuda.set_device(0)
tup = lambda t: tuple(int(x) for x in t.split(","))
B, I = cfg["batch"], cfg.get("input_dim", 784)
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.rand((4 * B, I), device="cuda", generator=g); data = data * (torch.rand(data.shape, device="cuda", generator=g) < 0.19)
perm = torch.randperm(4 * B, device="cuda", generator=g).to(torch.int32)
out = []
for v in (va, vb):
    L.check(L.lib.dmvae_debug_set_knob(which, v))
    e = StepEngine(I, cfg["latent_dim"], cfg["n_clusters"], enc_layers=tup(cfg.get("enc_layers", "500,500")), head_dim=cfg.get("head_dim", 2000),
                   dec_layers=tup(cfg.get("dec_layers", "2000,500,500")), dtype="bf16", max_batch=B)
    e.init_parameters(0); e.write_state(lr=cfg.get("lr", 0.002)); e.reset_epoch(4)
    step = e.capture_step(data, perm)
    for _ in range(steps): step()
    torch.cuda.synchronize()
    st = e.read_state()
    out.append((e.param.clone(), e.m.clone(), e.v.clone(), (st.last_loss, st.adam_t)))
    del step, e
L.check(L.lib.dmvae_debug_set_knob(which, va))
(pa, ma, va_, sa), (pb, mb, vb_, sb) = out
print("knob %d: %d vs %d after %d steps: params differ in %d of %d, m in %d, v in %d" % (which, va, vb, steps, int((pa != pb).sum()), pa.numel(), int((ma != mb).sum()), int((va_ != vb_).sum())))
print("state a", sa, "state b", sb)
