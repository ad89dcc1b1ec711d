"""Probe: wall time of eager device-state steps before and after recorded steps (config 3 shapes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from segmminterest_amd import hipabi as H
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer, default_args, init_model
dev = torch.device("cuda:0")
B, S, Lt, D, N, h = 1024, 20, 1, 512, 4, 16
margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "id", "photo": "id"}, exposure_prob=[1.0] * S)
torch.manual_seed(0)
model = init_model(margs, n_users=1903, n_items=352494, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
tr = Trainer(model, device_state=True)
bs = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, n_users=1903, n_items=352494, seed=i, features=False).items()} for i in range(4)]
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(bs[i % 4])
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for _ in range(3): tr.train_step(bs[0])
print("eager before   %.3f ms" % timeit(tr.train_step))
tr.record(bs[0], warmup=2)
print("recorded       %.3f ms" % timeit(tr.run_recorded))
print("eager after    %.3f ms" % timeit(tr.train_step))
print("recorded again %.3f ms" % timeit(tr.run_recorded))
print("eager again    %.3f ms" % timeit(tr.train_step))
torch.cuda.empty_cache()
print("eager after empty_cache %.3f ms" % timeit(tr.train_step))
