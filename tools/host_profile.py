"""cProfile of the host side of train_step (config 2 shapes): where the ~2 ms of enqueue time per step go.
   python tools/host_profile.py [steps]"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import DPComm, Trainer, default_args, init_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
B, S, D, N, Lt, h = 512, 40, 768, 2, 100, 16
args = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
torch.manual_seed(0)
model = init_model(args, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
tr = Trainer(model, lr=1e-3, weight_decay=1e-4, comm=DPComm())
b = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=1).items()}
for _ in range(5):
    tr.train_step(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.train_step(b)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
