"""Bisect: does record(prev_batch=...) + replays equal the eager steps?  (round 6, fit(recorded=True))"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import build_model, load_case
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer
DEV = "cuda"
cfg, g, _, _ = load_case("img_d32_N2")
mk = lambda s, b=16: {k: v.to(DEV) for k, v in make_batch(b, cfg["S"], cfg["Lt"], cfg["D_in"], seed=s).items()}
train = [mk(s) for s in range(8)]
valid = [mk(100)]

def run(mode, eval_at=(), dropout=True):
    model = build_model(cfg); model.load_state_dict(g["sd"]); model = model.cuda()
    torch.manual_seed(11)
    tr = Trainer(model, lr=1e-3, device_state=True, dropout=dropout)
    prev = None
    losses = []
    for i, b in enumerate(train):
        if i in eval_at:
            tr.valid_model(valid, permutation=0)
        if mode == "eager":
            out = tr.train_step(b)
        elif mode == "rec_prev":
            if i < 3: out = tr.train_step(b)
            elif i == 3: out = tr.record(b, prev_batch=prev)
            else: out = tr.run_recorded(b)
        losses.append(float(out["loss"]))
        prev = b
    torch.cuda.synchronize()
    return losses, model._store.flat.detach().clone()

for dropout in (True, False):
    for eval_at in ((), (3,), (5,)):
        le, pe = run("eager", eval_at, dropout)
        lr, pr = run("rec_prev", eval_at, dropout)
        print("dropout", dropout, "eval_at", eval_at, "losses equal", le == lr, "params equal", bool(torch.equal(pe, pr)), [i for i, (a, b) in enumerate(zip(le, lr)) if a != b])
