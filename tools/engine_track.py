"""Loss trajectories of the same training run on the exact-fp32 MFMA engine and on the plane engine (dropout off, same batches):
   python tools/engine_track.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import DPComm, Trainer, default_args, init_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda", 0)
B, S, D, N, Lt, h = 64, 40, 256, 3, 20, 8
args = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
batches = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=500 + i).items()} for i in range(6)]
curves = {}
for eng, name in ((H.ENGINE_F32, "f32"), (H.ENGINE_F16X3P, "f16x3p")):
    H.GEMM_ENGINE = eng
    torch.manual_seed(11)
    model = init_model(args, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
    tr = Trainer(model, lr=1e-3, weight_decay=1e-4, comm=DPComm(), dropout=False)
    curves[name] = [float(tr.train_step(batches[i % 6])["loss"].detach()) for i in range(steps)]
for i in sorted(set([0, 1, 5, 10, 25, 50] + list(range(100, steps, 6)) + [steps - 1])):
    a, b = curves["f32"][i], curves["f16x3p"][i]
    print("step %4d  f32 %.6f  f16x3p %.6f  rel diff %.2e" % (i, a, b, abs(a - b) / max(abs(a), 1e-9)))
