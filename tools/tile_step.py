"""A/B forced tile shapes of the stand-alone bf16 GEMMs on the step of a BASELINE.json config (interleaved graph replays on one box).
python tools/tile_step.py <cfg1..cfg5> 0,0 128,64 128,128 64,64        (0,0 = the heuristic; dmvae_debug_set_tile)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd")); sys.path.insert(0, ROOT)
from dmvae_hip import StepEngine, _lib as L
import bench
cfg = bench.PRESETS[sys.argv[1]]; values = sys.argv[2:]
torch.cuda.set_device(0)
tup = lambda t: tuple(int(x) for x in t.split(","))
B, I = cfg["batch"], cfg.get("input_dim", 784)
rows = 4 * B
data = torch.rand((rows, I), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(rows, device="cuda").to(torch.int32)
res = {v: [] for v in values}
engs = {}
for v in values:
    bm_, bn_ = (int(x) for x in v.split(",")); L.check(L.lib.dmvae_debug_set_tile(bm_, bn_))
    e = StepEngine(I, cfg["latent_dim"], cfg["n_clusters"], enc_layers=tup(cfg.get("enc_layers", "500,500")), head_dim=cfg.get("head_dim", 2000),
                   dec_layers=tup(cfg.get("dec_layers", "2000,500,500")), dtype="bf16", max_batch=B)
    e.init_parameters(0); e.write_state(lr=cfg.get("lr", 0.002)); e.reset_epoch(4)
    engs[v] = (e, e.capture_step(data, perm))        # the knob is baked into the captured graph
steps = 200 if B <= 4096 else 30
for rnd in range(5):
    for v in values:
        rp = engs[v][1]
        for _ in range(5): rp()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): rp()
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / steps * 1e3)
for v in values:
    r = sorted(res[v]); print("%s tile %s : ms/step median %.4f  min %.4f" % (sys.argv[1], v, r[len(r) // 2], r[0]), flush=True)
