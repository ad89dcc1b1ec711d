"""Development aid: window exits per site of the MLP-variant encoders (SelfMLP / CrossMLP) over a short run."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segmminterest_amd import engine as E, hipabi as H
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer, default_args, init_model
S, Lt, D, B = 40, 100, 768, 128
for abl in ("SelfMLP", "CrossMLP"):
    margs = default_args(num_layers_enc=3, d_model=D, nhead=16, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S, ablation_type=abl)
    torch.manual_seed(0)
    model = init_model(margs, n_users=500, n_items=2000, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
    tr = Trainer(model, lr=1e-3, dropout=True)
    bs = [{k: v.cuda() for k, v in make_batch(B, S, Lt, D, seed=7 + i).items()} for i in range(4)]
    hits = collections.Counter(); worst = {}
    orig = E.ParamStore.update_scales
    def spy(self, arena_t, names, n, backward=False):
        a = arena_t[:n].detach().cpu()
        for r in range(n):
            if names[r] is None: continue
            s_used, flag, m = float(a[r, 0]), int(a[r, 1].view(torch.int32)), float(a[r, H.SITE_HDR:].max())
            under = s_used > 0 and m > 0 and m * s_used < 0.25
            if flag or under:
                k = names[r] + (" under" if under and not flag else " over")
                hits[k] += 1; worst[k] = m * s_used
        return orig(self, arena_t, names, n, backward=backward)
    E.ParamStore.update_scales = spy
    for i in range(30):
        tr.train_step(bs[i % 4])
    torch.cuda.synchronize()
    E.ParamStore.update_scales = orig
    print(abl, dict(hits), {k: "%.3g" % v for k, v in worst.items()})
