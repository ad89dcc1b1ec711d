"""``__graft_entry__.smoke()``: one tiny train step through the HIP path on the GPU, checked against
the CPU oracle (the only place outside tests/ and bench.py's cpu_baseline leg that touches ``oracle/``,
and only as the checker)."""
from __future__ import annotations

import os
import sys

import torch


def run(device="cuda:0"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import segmm_oracle as O                      # checker only
    from . import hipabi
    from .synth import l1_normalize, make_batch
    from .trainer import Trainer, default_args, init_model

    hipabi.lib()                                  # fail loudly if the extension is missing
    torch.manual_seed(0)
    B, S, Lt, D, d, h, N = 16, 40, 10, 64, 64, 4, 3
    args = default_args(num_layers_enc=N, d_model=d, nhead=h, input_type={"user": "image", "photo": "image"},
                        exposure_prob=[1.0] * S, loss_type_list=["interestBPR", "surviveCE"])
    model = init_model(args, n_users=10, n_items=10, input_dim=D, max_vid_len=S, max_usr_len=Lt)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and "stage_mlp" not in n_:
                p.mul_(8.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch(B, S, Lt, D, seed=7)
    inp = dict(usr_image=l1_normalize(batch["user"]), usr_id=batch["user_identity_id"], usr_mask=batch["user_mask"],
               vid_image=l1_normalize(batch["photo"]), vid_id=batch["photo_identity_id"], vid_mask=batch["photo_mask"],
               gt=batch["label"])
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=args.loss_type_list, loss_weight=args.loss_weight,
               exposure_prob=args.exposure_prob)
    ref, rgrads = O.forward_backward(sd, cfg, {k: v.clone() for k, v in inp.items()})

    model = model.to(device).eval()
    dev_inp = {k: v.to(device) for k, v in inp.items()}
    out = model(usr_image=dev_inp["usr_image"], usr_id=dev_inp["usr_id"], usr_mask=dev_inp["usr_mask"],
                vid_image=dev_inp["vid_image"], vid_id=dev_inp["vid_id"], vid_mask=dev_inp["vid_mask"],
                gt=dev_inp["gt"].clone(), mode="train")
    out["loss"].backward()
    err = float((out["logits"].cpu() - ref["logits"].detach()).abs().max())
    assert err < 1e-4, "logit mismatch vs oracle: %g" % err
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4
    worst = 0.0
    for k, p in model.named_parameters():
        if rgrads[k] is None:
            assert p.grad is None, k
        else:
            scale = max(float(rgrads[k].abs().max()), 1e-7)
            worst = max(worst, float((p.grad.cpu() - rgrads[k]).abs().max()) / scale)
    assert worst < 3e-4, "gradient mismatch vs oracle: %g" % worst
    # one real training step (dropout on, fused AdamW) must run and change the live parameters only
    tr = Trainer(model)
    dev_batch = {k: v.to(device) for k, v in batch.items()}
    before = model._store.flat.clone()
    o2 = tr.train_step(dev_batch)
    torch.cuda.synchronize()
    st = model._store
    assert torch.isfinite(o2["loss"]).item()
    assert not torch.equal(before[:st.n_live], st.flat[:st.n_live])
    assert torch.equal(before[st.n_live:], st.flat[st.n_live:])
    print("[smoke] ok: logits err %.2e, loss %.6f (oracle %.6f), worst grad rel err %.2e, train-step loss %.6f"
          % (err, float(out["loss"]), float(ref["loss"]), worst, float(o2["loss"])))
