"""How far ahead of the GPU does the host run?  Times train_step() enqueue (no sync) vs the synchronised step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd.trainer import Trainer, DPComm, init_model, default_args
from segmminterest_amd.synth import make_batch

dev = torch.device("cuda:0")
B, S, D, Lt, N, h = 512, 40, 768, 100, 2, 16
margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
torch.manual_seed(1234)
model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=1234).items()}
tr = Trainer(model, lr=1e-3, weight_decay=1e-4, comm=DPComm())
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step; wall %.3f ms/step; GPU drained %.3f ms after the last enqueue" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    tr.train_step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
