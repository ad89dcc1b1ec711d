#!/usr/bin/env python
"""Generate golden vectors from the REAL reference (TEST INFRASTRUCTURE ONLY).

Runs only in the build container, where /root/reference exists.  It imports the
reference's own model / loss / metric modules on CPU through ``ref_import``
(SURVEY.md Appendix A), runs them on seeded synthetic batches and writes small
``.npz`` fixtures (inputs + expected outputs only -- no reference source) to
``tests/golden/``.  The committed fixtures are what pins the CPU restatement in
``oracle/segmm_oracle.py`` and, through it, the HIP path.

    python oracle/gen_golden.py            # regenerate every fixture

What is captured per case:
  sd/<name>     state_dict of the reference model (weights perturbed away from the
                N(0,.02)/zero-bias init so that every bias / LN affine term matters)
  in/<name>     usr_image, usr_id, usr_mask, vid_image, vid_id, vid_mask, gt
  out/<name>    eval-mode ``mode="train"`` outputs: logits, every loss scalar, mse, mse2, loss, gt
  grad/<name>   .grad of every live parameter after loss.backward();  nograd = names with grad None
  adam1/ adam3/ parameters after 1 / 3 torch.optim.AdamW(lr=1e-3, weight_decay=1e-4) steps
  inf/logits    ``mode="inference"`` logits
"""
import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from segmminterest_amd.synth import make_batch, l1_normalize  # noqa: E402

ALL_LOSS_W = {'focal': 1.0, 'mse': 0.7, 'hazard': 0.9, 'surviveCE': 1.1, 'interestBPR': 1.0,
              'interestCE': 0.8, 'interestKL': 1.2}

CASES = {
    # name: dict(...)
    "img_d32_N2": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                       loss="interestBPR", adam=True),
    "img_d32_N1": dict(user="image", photo="image", d=32, h=4, N=1, S=40, Lt=10, D_in=48, B=8,
                       loss="interestBPR"),
    "img_d32_N3_alllosses": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=7, D_in=40, B=9,
                                 loss="interestBPR,surviveCE,interestCE,interestKL,huber,hazard", adam=True),
    "img_d32_N2_focalfirst": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                                  loss="focal,interestCE,interestKL,interestBPR", exposure="stat"),
    "img_d32_N2_maskloss": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                                loss="interestCE,interestKL,surviveCE", mask_loss=1),
    "img_d64_h16_N3_Lt100": dict(user="image", photo="image", d=64, h=16, N=3, S=40, Lt=100, D_in=64, B=6,
                                 loss="interestBPR"),
    "img_d32_N2_S20": dict(user="image", photo="image", d=32, h=4, N=2, S=20, Lt=10, D_in=48, B=8,
                           loss="interestBPR,surviveCE", allow_full_len=False),
    "img_d32_N2_lb1": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                           loss="focal,interestBPR", learnable_bias=1, adam=True),
    "id_d32_N2": dict(user="id", photo="id", d=32, h=4, N=2, S=40, Lt=1, D_in=0, B=8,
                      loss="interestBPR", n_users=50, n_items=200, adam=True),
    "id_d64_h16_N4": dict(user="id", photo="id", d=64, h=16, N=4, S=40, Lt=1, D_in=0, B=8,
                          loss="interestBPR", n_users=50, n_items=200),
    "both_fh2": dict(user="both", photo="both", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                     loss="interestBPR", fusion_heads=2, n_users=50, n_items=200, adam=True),
    "both_fh0": dict(user="both", photo="both", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                     loss="interestBPR", fusion_heads=0, n_users=50, n_items=200),
    "both_fhm1": dict(user="both", photo="both", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                      loss="interestBPR", fusion_heads=-1, n_users=50, n_items=200),
    "both_fhm2": dict(user="both", photo="both", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                      loss="interestBPR", fusion_heads=-2, n_users=50, n_items=200),
    "uimg_pid_fh2": dict(user="image", photo="both", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                         loss="interestBPR", fusion_heads=4, n_users=50, n_items=200),
    # ablation variants (--ablation_type, main...SegMM.py:529; encoder.py:108-135,172-173,392-400,428-429,503-511)
    "abl_crossatt_N3": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                            loss="interestBPR", ablation="CrossAtt", adam=True),
    "abl_selfatt_N3": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                           loss="interestBPR", ablation="SelfAtt", adam=True),
    "abl_nouser_selfatt_N2": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                                  loss="interestBPR", ablation="noUser_SelfAtt"),
    "abl_nopos_id_N2": dict(user="id", photo="id", d=32, h=4, N=2, S=40, Lt=1, D_in=0, B=8,
                            loss="interestBPR", n_users=50, n_items=200, ablation="noPos", fwd_seed=4242, adam=True),
    "abl_crossatt_id_N3": dict(user="id", photo="id", d=32, h=4, N=3, S=40, Lt=1, D_in=0, B=8,
                               loss="interestBPR", n_users=50, n_items=200, ablation="CrossAtt"),
    "abl_selfmlp_N4": dict(user="image", photo="image", d=32, h=4, N=4, S=40, Lt=10, D_in=48, B=8,
                           loss="interestBPR", ablation="SelfMLP", adam=True),
    "abl_crossmlp_N5_Lt100": dict(user="image", photo="image", d=32, h=4, N=5, S=40, Lt=100, D_in=48, B=6,
                                  loss="interestBPR", ablation="CrossMLP", adam=True),
    "abl_crossmlp_N2": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                            loss="interestBPR", ablation="CrossMLP"),
    "abl_woatt_N2": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                         loss="interestBPR", ablation="w/oAtt"),
    # --use_pe 0 (main...SegMM.py:515 -> encoder.py:450-471 else branches): no positional-embedding add, one case per input mode
    "nope_img_d32_N3": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                            loss="interestBPR", use_pe=0, adam=True),
    "nope_id_d32_N2": dict(user="id", photo="id", d=32, h=4, N=2, S=40, Lt=1, D_in=0, B=8,
                           loss="interestBPR", n_users=50, n_items=200, use_pe=0, adam=True),
    "nope_both_fh2": dict(user="both", photo="both", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                          loss="interestBPR", fusion_heads=2, n_users=50, n_items=200, use_pe=0),
}

# TRAIN-MODE cases (model.train(): dropout on).  The reference's dropout draws are replaced by recorded deterministic keep-masks
# (torch.nn.functional.dropout is what every nn.Dropout.forward calls) so that the fixture pins WHERE the reference applies
# dropout and with which p: encoder.py:145,149 (raw logits, mask fills included, before the 1/sqrt(dh) scale), :166-167,
# :202,205, :461,471, kn_util/nn_utils/layers/mlp.py:22 (hard-wired 0.1), MLP_Block (:210-252).  Written as train_<name>.npz
# with the masks in CALL ORDER (mask/<k> packed bits, mask_shape/<k>, mask_p[k]).
TRAIN_CASES = {
    "train_img_d32_N3": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                             loss="interestBPR,surviveCE", adam=True),
    "train_id_d32_N2": dict(user="id", photo="id", d=32, h=4, N=2, S=40, Lt=1, D_in=0, B=8,
                            loss="interestBPR", n_users=50, n_items=200, adam=True),
    "train_both_fh2": dict(user="both", photo="both", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                           loss="interestBPR", fusion_heads=2, n_users=50, n_items=200),
    "train_img_d64_h16_N4_Lt100": dict(user="image", photo="image", d=64, h=16, N=4, S=40, Lt=100, D_in=64, B=4,
                                       loss="interestBPR"),
    "train_abl_crossatt_N3": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                                  loss="interestBPR", ablation="CrossAtt"),
    "train_abl_selfatt_N3": dict(user="image", photo="image", d=32, h=4, N=3, S=40, Lt=10, D_in=48, B=8,
                                 loss="interestBPR", ablation="SelfAtt"),
    "train_abl_crossmlp_N5": dict(user="image", photo="image", d=32, h=4, N=5, S=40, Lt=10, D_in=48, B=8,
                                  loss="interestBPR", ablation="CrossMLP"),
    "train_nope_img_d32_N2": dict(user="image", photo="image", d=32, h=4, N=2, S=40, Lt=10, D_in=48, B=8,
                                  loss="interestBPR", use_pe=0),
}


class MaskRecorder:
    """Stand-in for torch.nn.functional.dropout while the REFERENCE runs in train mode: draws the keep-mask from its own
    generator, records (p, mask) per call, returns input * mask / (1 - p) -- the definition of inverted dropout."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.calls = []

    def __call__(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = torch.rand(input.shape, generator=self.g) >= p
        self.calls.append((float(p), keep))
        return input * (keep.to(input.dtype) / (1.0 - p))


def build_reference_model(c, enc, dec):
    """Mirrors init_model (main_for_seq_leave_earlystop_SegMM.py:60-130), with D_in / Lt configurable."""
    import argparse as ap
    S = c["S"]
    if c.get("exposure") == "stat":
        g = torch.Generator().manual_seed(7)
        exposure = (0.5 + 0.5 * torch.rand(S, generator=g)).tolist()
    else:
        exposure = [1.0] * S
    loss_list = [x.strip() for x in c["loss"].split(",")]
    cfg = ap.Namespace(debug=0, num_layers_enc=c["N"], ablation_type=c.get("ablation", "ours"), d_model=c["d"], nhead=c["h"],
                       input_type={"user": c["user"], "photo": c["photo"]},
                       learnable_bias=c.get("learnable_bias", 0), exposure_prob=exposure,
                       fusion_heads=c.get("fusion_heads", 2), loss_type_list=loss_list,
                       loss_weight=dict(ALL_LOSS_W), mask_loss=c.get("mask_loss", 0), use_pe=c.get("use_pe", 1))
    N, d, h = c["N"], c["d"], c["h"]

    def backbone(user_id_max, video_id_max, max_usr_len):
        return enc.SegFormerX(d_model_in=d, d_model_lvls=[d] * N, num_head_lvls=[h] * N, ff_dim_lvls=[d] * N,
                              input_vid_dim=max(c["D_in"], 1), input_usr_dim=max(c["D_in"], 1),
                              max_vid_len=S, max_usr_len=max_usr_len,
                              sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N, output_layers=[-1],
                              model_cfg=cfg, user_id_max=user_id_max, video_id_max=video_id_max, use_pe=c.get("use_pe", 1))

    nu, ni = c.get("n_users", 0), c.get("n_items", 0)
    u, p = c["user"], c["photo"]
    if u == "both" or p == "both":
        um1, ul1, um2, ul2 = {"both": (-1, c["Lt"], nu, 1), "id": (nu, 1, nu, 1),
                              "image": (-1, c["Lt"], -1, c["Lt"])}[u]
        vm1, vm2 = {"both": (-1, ni), "id": (ni, ni), "image": (-1, -1)}[p]
        b1 = backbone(um1, vm1, ul1)
        b2 = backbone(um2, vm2, ul2)
        model = dec.MultiScaleTemporalDetrLeaveFocal(b1, b2, None, torch.nn.Identity(), cfg)
    else:
        um1, ul1 = (nu, 1) if u == "id" else (-1, c["Lt"])
        vm1 = ni if p == "id" else -1
        b1 = backbone(um1, vm1, ul1)
        model = dec.MultiScaleTemporalDetrLeaveFocal(b1, None, None, torch.nn.Identity(), cfg)
    return model, cfg


def perturb(model, seed):
    """Move parameters off the symmetric init so every affine term is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias_weight") or name.endswith("bias_bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g) + (0.02 if "weight" in name else 0.1))
            elif "ln" in name.split(".")[-2] or "pe_lns" in name or (".1." in name and "txt_lvl_projs" in name):
                if name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "stage_mlp" in name or "fusion_module" in name:
                pass         # xavier-initialised head: logits already O(1)
            elif p.dim() == 2 and "_pe" not in name and "usr_proj" not in name and "vid_proj" not in name:
                p.mul_(6.0)  # N(0,.02) -> N(0,.12): attention becomes non-uniform at d=32
            elif "vid_proj" in name or "usr_proj" in name:
                # Linear on L1-normalised features (entries ~1/D_in) vs Embedding rows
                p.mul_(60.0 if isinstance(dict(model.named_modules())[name.rsplit(".", 1)[0]], torch.nn.Linear) else 6.0)
            elif "_pe" in name:
                p.mul_(10.0)


FWD_SEED = [None]      # 'noPos' draws torch.randperm in every forward: reseeded so that the fixture is reproducible


def run_model(model, inp, mode="train"):
    buf = io.StringIO()
    if FWD_SEED[0] is not None:
        torch.manual_seed(FWD_SEED[0])
    with contextlib.redirect_stdout(buf):
        out = model(usr_image=inp["usr_image"], usr_id=inp["usr_id"], usr_mask=inp["usr_mask"],
                    vid_image=inp["vid_image"], vid_id=inp["vid_id"], vid_mask=inp["vid_mask"],
                    gt=inp["gt"].clone(), mode=mode)
    return out


def make_inputs(c, seed):
    b = make_batch(c["B"], c["S"], c["Lt"] if c["user"] != "id" else max(c["Lt"], 1), max(c["D_in"], 1),
                   n_users=max(c.get("n_users", 5), 1), n_items=max(c.get("n_items", 5), 1), seed=seed,
                   allow_full_len=c.get("allow_full_len", True))
    S = c["S"]
    lab = b["label"]
    # force the edge rows the survey lists: leave at segment 0, fully watched short video, view_len == S
    lab[0] = torch.tensor([0] + [-1] * 5 + [-2] * (S - 6))
    lab[1] = torch.tensor([1] * 4 + [-2] * (S - 4))
    if c.get("allow_full_len", True):
        lab[2] = torch.ones(S, dtype=torch.int64)
    lab[3] = torch.tensor([1] * (S - 1) + [0])
    pm = lab != -2
    b["photo_mask"] = pm
    b["photo"] = b["photo"] * pm[:, :, None]
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"],
               gt=lab)
    return inp


def gen_case(name, c, enc, dec, outdir, train=False):
    torch.manual_seed(abs(hash(name)) % (2 ** 31))
    torch.manual_seed(sum(ord(ch) for ch in name))
    model, cfg = build_reference_model(c, enc, dec)
    perturb(model, seed=11 + len(name))
    rec = None
    if train:
        import torch.nn.functional as F
        model.train()
        rec = MaskRecorder(seed=77 + len(name))
        F_dropout, F.dropout = F.dropout, rec
    else:
        model.eval()
    try:
        _gen_case_body(name, c, cfg, model, outdir, rec)
    finally:
        if train:
            F.dropout = F_dropout


def _gen_case_body(name, c, cfg, model, outdir, rec):
    FWD_SEED[0] = c.get("fwd_seed")
    inp = make_inputs(c, seed=1234 + len(name))
    blob = {"cfg": np.array(json.dumps(dict(c, exposure_prob=list(cfg.exposure_prob), loss_weight=cfg.loss_weight,
                                            ablation_type=cfg.ablation_type, train=rec is not None)))}
    for k, v in model.state_dict().items():
        blob["sd/" + k] = v.detach().numpy().copy()
    for k, v in inp.items():
        blob["in/" + k] = v.numpy().copy()

    model.zero_grad()
    out = run_model(model, inp, "train")
    out["loss"].backward()
    for k, v in out.items():
        blob["out/" + k] = v.detach().numpy().copy() if torch.is_tensor(v) else np.array(v)
    nograd = []
    for k, p in model.named_parameters():
        if p.grad is None:
            nograd.append(k)
        else:
            blob["grad/" + k] = p.grad.detach().numpy().copy()
    blob["nograd"] = np.array(json.dumps(nograd))
    if rec is not None:
        blob["mask_calls_fwd"] = np.array(len(rec.calls))          # dropout calls of ONE training forward
    else:
        with torch.no_grad():
            inf = run_model(model, inp, "inference")
        blob["inf/logits"] = inf["logits"].detach().numpy().copy()

    if c.get("adam"):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
        for step in range(1, 4):
            opt.zero_grad()
            o = run_model(model, inp, "train")
            o["loss"].backward()
            opt.step()
            if step in (1, 3):
                for k, p in model.named_parameters():
                    blob["adam%d/%s" % (step, k)] = p.detach().numpy().copy()
        if rec is None:
            blob["adam_loss3"] = run_model(model, inp, "train")["loss"].detach().numpy().copy()
    if rec is not None:
        blob["mask_p"] = np.array([p for p, _ in rec.calls], dtype=np.float64)
        for k, (_, keep) in enumerate(rec.calls):
            blob["mask/%d" % k] = np.packbits(keep.numpy().reshape(-1))
            blob["mask_shape/%d" % k] = np.array(keep.shape, dtype=np.int64)
    path = os.path.join(outdir, name + ".npz")
    np.savez_compressed(path, **blob)
    nlive = sum(1 for k in blob if k.startswith("grad/"))
    print("%-28s live=%3d dead=%3d loss=%.6f  %.0f KB" % (name, nlive, len(nograd), float(out["loss"]),
                                                        os.path.getsize(path) / 1024))


def gen_metrics(ev, outdir):
    """Known-answer vectors for the ranking / AUC metrics (my_evaluation.py:73-231,264-357)."""
    import argparse as ap
    rng = np.random.RandomState(5)
    blob = {}
    B, S = 64, 40
    b = make_batch(B, S, 1, 1, seed=99, features=False)
    gt = b["label"]
    gt[0] = torch.ones(S, dtype=torch.int64)           # view_len == 40 -> dropped by TOP_K_leave
    interests = torch.sigmoid(torch.from_numpy(rng.randn(B, S).astype(np.float32)))
    interests[5] = 0.5                                 # all ties
    interests[6, :10] = interests[6, 10]               # partial ties
    view_lengths = (gt == 1).sum(1, keepdim=True).numpy()
    mask = (gt != -2).numpy()
    blob["interests"] = interests.numpy()
    blob["gt"] = gt.numpy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        for perm in (0, 1):
            np.random.seed(42)
            e = ev.TOP_K_leave(interests.numpy(), view_lengths, mask, permutation=perm)
            blob["topk_perm%d" % perm] = np.array([e["%s@%d" % (m, k)] for k in (1, 3, 5, 10) for m in ("HR", "NDCG")],
                                                  dtype=np.float64)
            np.random.seed(42)
            e = ev.TOP_K_leave_mask(interests.numpy(), view_lengths, mask, permutation=perm)
            blob["topkmask_perm%d" % perm] = np.array([e["%s@%d" % (m, k)] for k in (1, 3, 5, 10) for m in ("HR", "NDCG")],
                                                      dtype=np.float64)
        np.random.seed(42)
        e, mins = ev.TOP_K_leave(interests.numpy(), view_lengths, mask, permutation=0, test=1)
        blob["min_indices"] = mins
        # main_eval_batch on rows with view_len >= 1 (LeaveCTR indexes view_len-1)
        rows = [i for i in range(B) if 1 <= int(view_lengths[i, 0])][:24]
        args = ap.Namespace(TOP_K_mask=0, TOP_K_permutation=0, draw_case=0)
        res = {k: [] for k in ("JaccardSim", "ProbAUC", "LeaveMSE", "LeaveCTR", "LeaveCTR_view", "TOP_K", "view_lengths")}
        it = interests[rows]
        res = ev.main_eval_batch(args, it, gt[rows], (it > 0.5).float(), res, type="inference")
        # the logits= branch (my_evaluation.py:307-318): softmax-inverse leave position -> MAES (a running sum), pred_leave
        lg = torch.from_numpy(np.random.RandomState(7).randn(len(rows), S).astype(np.float32) * 2.0)
        res_l = {"MAES": 0.0, "pred_leave": []}
        res_l = ev.main_eval_batch(args, it, gt[rows], (it > 0.5).float(), res_l, type="inference", logits=lg)
        res_l = ev.main_eval_batch(args, it[:7], gt[rows][:7], (it[:7] > 0.5).float(), res_l, type="inference", logits=lg[:7] * 0.25)
    blob["meb_logits"] = lg.numpy()
    blob["meb_maes"] = np.array(float(res_l["MAES"]), dtype=np.float64)
    blob["meb_pred_leave"] = np.concatenate([np.asarray(x, dtype=np.int64) for x in res_l["pred_leave"]])
    blob["meb_rows"] = np.array(rows)
    for k, v in res.items():
        if k == "TOP_K":
            continue
        blob["meb/" + k] = np.array(v, dtype=np.float64)
    # sklearn AUC / per-user wuAUC (SegRec/main.py:101-117) on integer-score cases
    from sklearn.metrics import roc_auc_score
    scores = rng.randint(0, 20, size=400).astype(np.float64)
    labels = (rng.rand(400) < 0.4).astype(np.int64)
    users = rng.randint(0, 12, size=400)
    labels[users == 3] = 1                       # single-class user: skipped by wuAUC
    blob["auc_scores"], blob["auc_labels"], blob["auc_users"] = scores, labels, users
    blob["auc"] = np.array(roc_auc_score(labels, scores))
    tot, wsum = 0.0, 0.0
    for u in np.unique(users):
        m = users == u
        if len(np.unique(labels[m])) < 2:
            continue
        tot += m.sum() * roc_auc_score(labels[m], scores[m])
        wsum += m.sum()
    blob["wuauc"] = np.array(tot / wsum)
    np.savez_compressed(os.path.join(outdir, "metrics_kat.npz"), **blob)
    print("metrics_kat written")


def main():
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap_.add_argument("--only", default="")
    a = ap_.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(1)
    enc, dec, ev = ref_import.load()
    for name, c in CASES.items():
        if a.only and a.only not in name:
            continue
        gen_case(name, c, enc, dec, a.out)
    for name, c in TRAIN_CASES.items():
        if a.only and a.only not in name:
            continue
        gen_case(name, c, enc, dec, a.out, train=True)
    if not a.only or a.only == "metrics":
        gen_metrics(ev, a.out)


if __name__ == "__main__":
    main()
