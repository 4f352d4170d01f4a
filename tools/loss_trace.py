"""Per-step loss of the first steps of a configuration (diagnostic): python tools/loss_trace.py I D K B enc head dec [dtype] [lr]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import StepEngine
I, D, K, B = (int(v) for v in sys.argv[1:5])
enc = tuple(int(v) for v in sys.argv[5].split(",")); head = int(sys.argv[6]); dec = tuple(int(v) for v in sys.argv[7].split(","))
dtype = sys.argv[8] if len(sys.argv) > 8 else "bf16"
lr = float(sys.argv[9]) if len(sys.argv) > 9 else 0.002
torch.cuda.set_device(0)
data = torch.rand((4 * B, I), device="cuda"); data = data * (torch.rand_like(data) < 0.19)
perm = torch.randperm(4 * B, device="cuda").to(torch.int32)
e = StepEngine(I, D, K, enc_layers=enc, head_dim=head, dec_layers=dec, dtype=dtype, max_batch=B)
e.init_parameters(0); e.reset_epoch(4); e.write_state(lr=lr)
for s in range(12):
    e.train_step(data, perm, use_state_cursor=True)
    torch.cuda.synchronize()
    st = e.read_state()
    print("step %2d loss %.4f recon %.4f klz %.4f klc %.5f  |param|max %.3f" % (s + 1, st.last_loss, st.last_recon, st.last_klz, st.last_klc, e.param.abs().max().item()), flush=True)
