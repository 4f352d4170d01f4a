"""Replays the captured step of a BASELINE.json config 300 times (run under rocprofv3 --kernel-trace --stats: tools/step_trace.sh).
python tools/step_trace.py <cfgN>[:batch=B]     DMVAE_KNOBS=k=v,... sets tuning knobs first."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, ROOT)
from dmvae_hip import StepEngine, _lib as L
import bench
cfg = dict(bench.PRESETS[sys.argv[1].split(":")[0]])
for kv in sys.argv[1].split(":")[1:]: cfg[kv.split("=")[0]] = int(kv.split("=")[1])
torch.cuda.set_device(0)
tup = lambda t: tuple(int(x) for x in t.split(","))
B, I = cfg["batch"], cfg.get("input_dim", 784)
data = torch.rand((4 * B, I), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(4 * B, device="cuda").to(torch.int32)
for kv in os.environ.get("DMVAE_KNOBS", "").split(","):
    if kv: L.check(L.lib.dmvae_debug_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1])))
e = StepEngine(I, cfg["latent_dim"], cfg["n_clusters"], enc_layers=tup(cfg.get("enc_layers", "500,500")), head_dim=cfg.get("head_dim", 2000),
               dec_layers=tup(cfg.get("dec_layers", "2000,500,500")), dtype="bf16", max_batch=B)
e.init_parameters(0); e.write_state(lr=cfg.get("lr", 0.002)); e.reset_epoch(4)
rp = e.capture_step(data, perm)
n = 300 if B <= 4096 else 60
for _ in range(n): rp()
torch.cuda.synchronize()
