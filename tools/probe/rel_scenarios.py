"""Window exits of the delayed scales with and without the loss-relative backward scales, on a run that alternates ONE memorised
batch (collapsed loss, tiny gradients) with fresh batches (normal loss)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer, default_args, init_model
for (B, D, N, lr, steps) in [(16, 128, 3, 3e-3, 120), (16, 64, 2, 1e-2, 120), (32, 256, 3, 3e-3, 120)]:
    S, Lt = 20, 6
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    fixed = {k: v.cuda() for k, v in make_batch(B, S, Lt, D, seed=60).items()}
    fresh = [{k: v.cuda() for k, v in make_batch(B, S, Lt, D, seed=1000 + i).items()} for i in range(steps // 4)]
    out = []
    for rel in (True, False):
        torch.manual_seed(3)
        model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
        tr = Trainer(model, lr=lr, dropout=False)
        model._store.loss_relative = rel
        losses = []
        for i in range(steps):
            b = fresh[i // 4] if i % 4 == 3 else fixed          # three steps on the memorised batch, one on a new one
            losses.append(float(tr.train_step(b)["loss"].detach()))
        out.append((model._store.overflow_count(), losses))
    l = out[0][1]
    print("B %d D %d N %d lr %g -> exits rel %d / plain %d; fixed-batch loss %.2g, fresh-batch loss %.2g" % (B, D, N, lr, out[0][0], out[1][0], l[-2], l[-1]))
