"""Shared test helpers: golden-fixture loading."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# metric known answers; config-1 labels of the reference's sample file; data-format fixtures (oracle/gen_golden_io.py)
NOT_MODEL_CASES = ("metrics_kat.npz", "cfg1_labels.npz", "io_dataloader.npz", "io_cliprec.npz")
# train_*: TRAIN-mode fixtures (the reference's dropout masks recorded in call order); they pin the oracle's dropout placement
TRAIN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("train_"))
MODEL_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f not in NOT_MODEL_CASES
                     and not f.startswith("train_"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    cfg["loss_type_list"] = [x.strip() for x in cfg["loss"].split(",")]
    grp = {"sd": {}, "in": {}, "out": {}, "grad": {}, "adam1": {}, "adam3": {}, "inf": {}}
    for k in z.files:
        if "/" in k and not k.startswith("mask"):
            g, n = k.split("/", 1)
            grp[g][n] = torch.from_numpy(z[k])
    nograd = json.loads(str(z["nograd"]))
    extra = {"adam_loss3": float(z["adam_loss3"])} if "adam_loss3" in z.files else {}
    if "mask_p" in z.files:          # train-mode fixture: dropout keep-masks of the reference in call order
        ps = z["mask_p"]
        masks = []
        for k in range(len(ps)):
            shape = tuple(int(x) for x in z["mask_shape/%d" % k])
            n = int(np.prod(shape))
            keep = np.unpackbits(z["mask/%d" % k])[:n].reshape(shape).astype(bool)
            masks.append((float(ps[k]), torch.from_numpy(keep)))
        extra["masks"] = masks
        extra["mask_calls_fwd"] = int(z["mask_calls_fwd"])
    return cfg, grp, nograd, extra


def build_model(cfg, device=None):
    """Builds the segmminterest_amd facade exactly like the reference's init_model builds its model
    (main_for_seq_leave_earlystop_SegMM.py:60-130), from a golden-fixture cfg dict."""
    import argparse
    import segmminterest_amd as M
    S, N, d, h = cfg["S"], cfg["N"], cfg["d"], cfg["h"]
    args = argparse.Namespace(debug=0, num_layers_enc=N, ablation_type=cfg.get("ablation_type", "ours"), d_model=d, nhead=h,
                              input_type={"user": cfg["user"], "photo": cfg["photo"]},
                              learnable_bias=cfg.get("learnable_bias", 0), exposure_prob=cfg["exposure_prob"],
                              fusion_heads=cfg.get("fusion_heads", 2), loss_type_list=cfg["loss_type_list"],
                              loss_weight=cfg["loss_weight"], mask_loss=cfg.get("mask_loss", 0), use_pe=cfg.get("use_pe", 1))

    def backbone(user_id_max, video_id_max, max_usr_len):
        return M.SegFormerX(d_model_in=d, d_model_lvls=[d] * N, num_head_lvls=[h] * N, ff_dim_lvls=[d] * N,
                            input_vid_dim=max(cfg["D_in"], 1), input_usr_dim=max(cfg["D_in"], 1), max_vid_len=S,
                            max_usr_len=max_usr_len, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                            output_layers=[-1], model_cfg=args, user_id_max=user_id_max, video_id_max=video_id_max,
                            use_pe=cfg.get("use_pe", 1))

    nu, ni = cfg.get("n_users", 0), cfg.get("n_items", 0)
    u, p = cfg["user"], cfg["photo"]
    if u == "both" or p == "both":
        um1, ul1, um2, ul2 = {"both": (-1, cfg["Lt"], nu, 1), "id": (nu, 1, nu, 1), "image": (-1, cfg["Lt"], -1, cfg["Lt"])}[u]
        vm1, vm2 = {"both": (-1, ni), "id": (ni, ni), "image": (-1, -1)}[p]
        model = M.MultiScaleTemporalDetrLeaveFocal(backbone(um1, vm1, ul1), backbone(um2, vm2, ul2), None, torch.nn.Identity(), args)
    else:
        um1, ul1 = (nu, 1) if u == "id" else (-1, cfg["Lt"])
        vm1 = ni if p == "id" else -1
        model = M.MultiScaleTemporalDetrLeaveFocal(backbone(um1, vm1, ul1), None, None, torch.nn.Identity(), args)
    if device is not None:
        model = model.to(device)
    model._test_fwd_seed = cfg.get("fwd_seed")      # 'noPos' fixtures: torch is reseeded before every forward
    return model


def call_model(model, inp, mode="train", device=None, fwd_seed=None):
    """``fwd_seed``: the 'noPos' fixtures reseed torch before every forward (the model draws torch.randperm)."""
    kw = {k: (v.to(device) if device is not None else v) for k, v in inp.items()}
    fwd_seed = getattr(model, "_test_fwd_seed", None) if fwd_seed is None else fwd_seed
    if fwd_seed is not None:
        torch.manual_seed(fwd_seed)
    return model(usr_image=kw["usr_image"], usr_id=kw["usr_id"], usr_mask=kw["usr_mask"], vid_image=kw["vid_image"],
                 vid_id=kw["vid_id"], vid_mask=kw["vid_mask"], gt=kw["gt"].clone(), mode=mode)


# ------------------------------------------------------------------ observed gradient errors (DESIGN.md §5 quotes them)
GRAD_ERRS = {}          # test id -> (worst error / tensor maximum, tensor name, tensor maximum)


def note_grad_err(name, err, scale):
    """Whole-model tests call this for every live gradient they compare; with SEGMM_GRAD_ERR_LOG=<file> the session writes the
    worst relative error per test (conftest.pytest_sessionfinish) and which tensor it was."""
    label = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    if scale <= 1e-5:          # gradients that are zero in exact arithmetic (sums of d loss / d logits under the shift-invariant BPR
        return                 # loss): cancellation noise in the reference too, judged by the tests' absolute floor, not listed
    rel = err / max(scale, 1e-30)
    if label not in GRAD_ERRS or rel > GRAD_ERRS[label][0]:
        GRAD_ERRS[label] = (rel, name, scale)
