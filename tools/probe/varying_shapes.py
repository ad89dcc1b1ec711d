"""Batches of varying size (the last batch of an epoch, validation batches) through one Trainer: no crash, finite losses, and the
same loss for a batch whether or not other shapes ran in between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer, default_args, init_model
S, Lt, D, N = 40, 100, 768, 2
margs = default_args(num_layers_enc=N, d_model=D, nhead=16, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
torch.manual_seed(0)
model = init_model(margs, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
tr = Trainer(model, lr=1e-3, dropout=True)
def mk(B, Lt_, seed):
    return {k: v.cuda() for k, v in make_batch(B, S, Lt_, D, seed=seed).items()}
seq = [(512, 100), (512, 100), (300, 100), (512, 100), (77, 100), (512, 37), (1, 100), (512, 100), (129, 100), (512, 100)]
for i, (B, L) in enumerate(seq):
    out = tr.train_step(mk(B, L, 10 + i))
    l = float(out["loss"].detach())
    assert l == l and abs(l) < 1e3, (i, B, L, l)
    if i % 3 == 2:
        m = tr.valid_model([mk(64, 100, 99)], permutation=0)
        assert all(v == v for v in m.values()), m
    print("step %d B=%d Lt=%d loss %.5f exits %d" % (i, B, L, l, model._store.overflow_count()))
torch.cuda.synchronize()
print("OK")
