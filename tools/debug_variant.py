import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmminterest_amd import hipabi as H
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer, default_args, init_model
variant, ds = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
B, S, Lt, D, N, h = 32, 40, 8, 64, 3, 4
kind = "id" if variant == "noPos" else "image"
over = dict(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)
if variant == "learnable_bias": over["learnable_bias"] = 1
elif variant == "focal_first": over["loss_type_list"] = ["focal", "interestBPR"]
elif variant != "plain": over["ablation_type"] = variant
margs = default_args(**over)
b = {k: v.to(dev) for k, v in make_batch(B, S, 1 if kind == "id" else Lt, D, n_users=50, n_items=500, seed=700, features=kind != "id").items()}
torch.manual_seed(11)
model = init_model(margs, n_users=50, n_items=500, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
tr = Trainer(model, lr=1e-3, weight_decay=1e-4, device_state=bool(ds))
for i in range(3):
    out = tr.train_step(b)
    torch.cuda.synchronize()
    print("step", i, float(out["loss"].detach()), flush=True)
if ds:
    tr.record(b, warmup=1); torch.cuda.synchronize(); print("recorded", flush=True)
    out = tr.run_recorded(b); torch.cuda.synchronize(); print("replayed", float(out["loss"].detach()), flush=True)
