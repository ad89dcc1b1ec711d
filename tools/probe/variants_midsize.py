"""Every input mode / ablation variant at a mid-size shape for 30 training steps: no crash, finite losses, window exits."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import Trainer, default_args, init_model
S, Lt, D, B = 40, 100, 768, 128
cases = [dict(user="both", photo="both", abl="ours", N=2, fh=2), dict(user="both", photo="both", abl="ours", N=3, fh=-1),
         dict(user="image", photo="both", abl="ours", N=2, fh=0), dict(user="id", photo="id", abl="ours", N=3, fh=2)]
for abl in ("CrossAtt", "SelfAtt", "noPos", "noUser", "noUser_SelfAtt", "SelfMLP", "CrossMLP", "w/oAtt"):
    cases.append(dict(user="image" if abl != "noPos" else "id", photo="image" if abl != "noPos" else "id", abl=abl, N=3, fh=2))
for c in cases:
    try:
        margs = default_args(num_layers_enc=c["N"], d_model=D, nhead=16, input_type={"user": c["user"], "photo": c["photo"]},
                             exposure_prob=[1.0] * S, ablation_type=c["abl"], fusion_heads=c["fh"],
                             loss_type_list=["interestBPR", "focal", "surviveCE"] if c["abl"] == "ours" else ["interestBPR"])
        torch.manual_seed(0)
        model = init_model(margs, n_users=500, n_items=2000, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
        tr = Trainer(model, lr=1e-3, dropout=True)
        feats = not (c["user"] == "id" and c["photo"] == "id")
        bs = [{k: v.cuda() for k, v in make_batch(B, S, Lt, D, n_users=500, n_items=2000, seed=7 + i, features=feats).items()} for i in range(4)]
        ls = [float(tr.train_step(bs[i % 4])["loss"].detach()) for i in range(30)]
        torch.cuda.synchronize()
        ok = all(l == l and abs(l) < 1e4 for l in ls)
        print("%-60s %s  loss %.4f -> %.4f  window exits %d" % (str(c), "OK " if ok else "BAD", ls[0], ls[-1], model._store.overflow_count()))
    except Exception as e:
        print("%-60s FAILED: %s" % (str(c), str(e)[:200]))
        traceback.print_exc()
