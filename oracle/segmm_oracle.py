"""CPU restatement of the reference's segment-interest training path (TEST INFRASTRUCTURE ONLY).

This is the parity oracle: an independent, functional torch-CPU restatement of
what ``MMinterest/models/{encoder,decoder_leave_focal,my_evaluation}.py`` and the
train step of ``MMinterest/main_for_seq_leave_earlystop_SegMM.py`` compute, each
function citing the reference file:line it follows.  It is PINNED: every function
here is checked in ``tests/test_oracle_golden.py`` against golden vectors captured
from the real reference modules (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file, as the checker / the timed CPU baseline -- never the product
path (``segmminterest_amd`` has no import of ``oracle``).

All arithmetic is fp32 by default (``dtype=torch.float64`` gives the high-precision
variant used to bound fp32 drift).  Parameters are addressed by the reference's
``state_dict`` key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

LOSS_NAMES = ("interestBPR", "focal", "surviveCE", "interestCE", "interestKL", "huber", "hazard")


# --------------------------------------------------------------------------- building blocks
def _lin(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _ln(sd, prefix, x, eps=1e-12):
    """LayerNorm(d, eps=1e-12) (encoder.py:39-40,185-186,383-385)."""
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def attn_logits(sd, proj, feat_k, mask_k, feat_q, mask_q, nhead):
    """SegFormerXAttention.get_attn_logits (encoder.py:44-73): Q=proj.0(q), K=proj.1(k),
    per-head QK^T (unscaled), entries outside mask_q (x) mask_k set to -10000."""
    B, Lq, d = feat_q.shape
    Lk = feat_k.shape[1]
    dh = d // nhead
    q = _lin(sd, proj + ".0", feat_q).view(B, Lq, nhead, dh)
    k = _lin(sd, proj + ".1", feat_k).view(B, Lk, nhead, dh)
    logits = torch.einsum("bqhd,bkhd->bhqk", q, k)
    m = (mask_q[:, :, None] & mask_k[:, None, :])[:, None]          # [B,1,Lq,Lk]
    return torch.where(m, logits, torch.full_like(logits, -10000.0))


def attn_mode(abl: str) -> str:
    """Key blocks of the ablation variants (encoder.py:108-161): 'CrossAtt' in type -> the other side's keys only,
    'SelfAtt' in type -> the own side's keys only (and the user branch returns None, :172-173), else both."""
    return "cross" if "CrossAtt" in abl else ("self" if "SelfAtt" in abl else "joint")


def cross_attention(sd, pre, vid, vid_mask, usr, usr_mask, nhead, need_usr=True, drop=None, mode="joint"):
    """SegFormerXAttention.forward, sr_ratio=1 (encoder.py:75-175); ``mode`` from :func:`attn_mode`.

    ``drop`` is None (eval) or a callable applying dropout; note the reference drops the RAW
    logits (mask fills included) BEFORE the 1/sqrt(dh) scale and the softmax (encoder.py:144-146).
    Projections whose result the reference computes but never uses (e.g. v2v under CrossAtt) are skipped.
    """
    B, Lv, d = vid.shape
    Lt = usr.shape[1]
    dh = d // nhead
    do = drop if drop is not None else (lambda t: t)
    # dropout CALL ORDER of the reference (the train-mode fixtures replay its masks by position, tests/test_oracle_golden.py):
    # v_logits (:145) -> t_logits (:149) -> ff_usr (:166) -> ff_vid (:167)
    vals, logits = [], []
    if mode != "cross":
        vals.append(_lin(sd, pre + ".v2v_proj.2", vid))
        logits.append(attn_logits(sd, pre + ".v2v_proj", vid, vid_mask, vid, vid_mask, nhead))
    if mode != "self":
        vals.append(_lin(sd, pre + ".t2v_proj.2", usr))
        logits.append(attn_logits(sd, pre + ".t2v_proj", usr, usr_mask, vid, vid_mask, nhead))
    v_val = torch.cat(vals, 1)
    v_val = v_val.view(B, v_val.shape[1], nhead, dh)
    v_logits = do(torch.cat(logits, -1)) / math.sqrt(dh)
    # the user branch: under SelfAtt the reference computes it (own-side keys) and throws it away (encoder.py:122-135,172-173)
    usr_side = need_usr
    if usr_side:
        if mode == "self":
            vals = [_lin(sd, pre + ".t2t_proj.2", usr)]
            logits = [attn_logits(sd, pre + ".t2t_proj", usr, usr_mask, usr, usr_mask, nhead)]
        else:
            vals = [_lin(sd, pre + ".v2t_proj.2", vid)]
            logits = [attn_logits(sd, pre + ".v2t_proj", vid, vid_mask, usr, usr_mask, nhead)]
            if mode == "joint":
                vals.append(_lin(sd, pre + ".t2t_proj.2", usr))
                logits.append(attn_logits(sd, pre + ".t2t_proj", usr, usr_mask, usr, usr_mask, nhead))
        t_val = torch.cat(vals, 1)
        t_val = t_val.view(B, t_val.shape[1], nhead, dh)
        t_logits = do(torch.cat(logits, -1)) / math.sqrt(dh)
    vid_ = torch.einsum("bhqk,bkhd->bqhd", F.softmax(v_logits, -1), v_val).reshape(B, Lv, d)
    usr_out = None
    if usr_side:
        usr_ = torch.einsum("bhqk,bkhd->bqhd", F.softmax(t_logits, -1), t_val).reshape(B, Lt, d)
        usr_ = do(_lin(sd, pre + ".ff_usr", usr_))
        if mode != "self":
            usr_out = _ln(sd, pre + ".ln_usr", usr + usr_)
    vid_ = do(_lin(sd, pre + ".ff_vid", vid_))
    vid_out = _ln(sd, pre + ".ln_vid", vid + vid_)
    return vid_out, usr_out


def mlp_gelu(sd, pre, x, drop=None):
    """kn_util MLP([d, ff, d], activation='gelu') (kn_util/nn_utils/layers/mlp.py:6-23): erf-GELU,
    inner dropout after the activation."""
    do = drop if drop is not None else (lambda t: t)
    return _lin(sd, pre + ".layers.1", do(F.gelu(_lin(sd, pre + ".layers.0", x))))


def encoder_layer(sd, pre, usr, usr_mask, vid, vid_mask, nhead, need_usr=True, drop=None, mode="joint"):
    """SegFormerXEncoderLayer.forward (encoder.py:189-208): post-LN attention block then post-LN FFN."""
    do = drop if drop is not None else (lambda t: t)
    vid, usr_new = cross_attention(sd, pre + ".cross_attn", vid, vid_mask, usr, usr_mask, nhead, need_usr, drop, mode)
    vid = _ln(sd, pre + ".ln_vid", vid + do(mlp_gelu(sd, pre + ".ff_vid", vid, drop)))
    if usr_new is not None:
        usr_new = _ln(sd, pre + ".ln_usr", usr_new + do(mlp_gelu(sd, pre + ".ff_usr", usr_new, drop)))
    return vid, usr_new


def embedding(sd, pre, usr_feat, vid_feat, drop=None, abl="ours", use_pe=1):
    """SegFormerX._get_embedding (encoder.py:425-473); ``use_pe=0`` (trainer flag --use_pe, main...SegMM.py:515) skips the
    positional-embedding adds (:450-471 else branches), so vid_pe / usr_pe get no gradient.  2-D inputs are id tensors
    ([B,S] item ids broadcast over segments / [B,1] user id), 3-D inputs are features.  'noPos' in the ablation
    type: the segment positions of every row are a fresh ``torch.randperm`` (global CPU generator, :428-429)."""
    do = drop if drop is not None else (lambda t: t)
    if vid_feat.dim() == 2:
        B, Lv = vid_feat.shape
        wdt = sd[pre + ".frameid_proj.weight"].dtype
        if "noPos" in abl:
            pos = torch.stack([torch.randperm(Lv) for _ in range(B)]).to(wdt)[:, :, None]
        else:
            pos = torch.arange(Lv, dtype=wdt)[None, :, None].expand(B, Lv, 1)
        v = torch.cat([F.embedding(vid_feat, sd[pre + ".vid_proj.weight"]),
                       _lin(sd, pre + ".frameid_proj", pos)], -1)
    else:
        v = _lin(sd, pre + ".vid_proj", vid_feat)
    if usr_feat.dim() == 2:
        u = F.embedding(usr_feat, sd[pre + ".usr_proj.weight"])
    else:
        u = _lin(sd, pre + ".usr_proj", usr_feat)
    if use_pe:
        v = v + sd[pre + ".vid_pe.weight"][None, : v.shape[1]]
        u = u + sd[pre + ".usr_pe.weight"][None, : u.shape[1]]
    v = do(_ln(sd, pre + ".vid_ln", v))          # dropout call order: video (:461), then user (:471)
    u = do(_ln(sd, pre + ".usr_ln", u))
    return v, u


def mlp_block(sd, pre, x, drop=None):
    """MLP_Block.forward (encoder.py:210-252) as SegFormerX builds it (:392-400): [Linear, ReLU, Dropout(p)] per hidden
    unit, then the output Linear.  The Sequential indices of the Linears are read off the state_dict keys."""
    do = drop if drop is not None else (lambda t: t)
    idx = sorted({int(k[len(pre) + 1:].split(".")[0]) for k in sd if k.startswith(pre + ".") and k.endswith(".weight")})
    for i in idx[:-1]:
        x = do(F.relu(_lin(sd, "%s.%d" % (pre, i), x)))
    return _lin(sd, "%s.%d" % (pre, idx[-1]), x)


def backbone_forward(sd, pre, usr_feat, usr_mask, vid_feat, vid_mask, N, nhead, S,
                     skip_dead=True, drop=None, abl="ours", use_pe=1):
    """SegFormerX.forward + SegFormerXEncoder.forward with output_layers=[-1]
    (encoder.py:475-520, 302-324).  The encoder records the INPUT of every layer
    (encoder.py:316-319) and [-1] selects the input of the last layer, so layer N-1 is dead and
    layer N-2 is live on the video side only; ``skip_dead=False`` executes them anyway, like the
    reference does, without changing any output.  ``abl`` = model_cfg.ablation_type (:387-400,503-511)."""
    if usr_feat.dim() == 1:                       # encoder.py:478-481
        usr_feat = usr_feat[:, None]
        usr_mask = torch.ones(usr_feat.shape, dtype=torch.bool)
    if vid_feat.dim() == 1:                       # encoder.py:484-486 (40 generalised to S)
        vid_feat = vid_feat[:, None].repeat(1, S)
    usr_mask = usr_mask.bool()
    vid, usr = embedding(sd, pre, usr_feat, vid_feat, drop, abl, use_pe)
    usr_emb = usr
    if abl == "CrossMLP":                         # encoder.py:503-506: MLP over cat(user, video) tokens, pooled to 40 tokens
        z = mlp_block(sd, pre + ".encoder_mlp.mlp", torch.cat((usr, vid), dim=-2), drop)
        return F.adaptive_avg_pool1d(z.permute(0, 2, 1), 40).permute(0, 2, 1), usr_emb
    if abl == "SelfMLP":                          # :507-509
        return mlp_block(sd, pre + ".encoder_mlp.mlp", vid, drop), usr_emb
    if abl == "w/oAtt":                           # :510-511
        return vid, usr_emb
    mode = attn_mode(abl)
    out = None
    for i in range(N):
        if i == N - 1:
            out = vid
            if skip_dead:
                break
        need_usr = (i < N - 2 and mode != "self") or not skip_dead
        vid, usr_new = encoder_layer(sd, "%s.encoder.layers.%d" % (pre, i), usr, usr_mask, vid, vid_mask,
                                     nhead, need_usr, drop, mode)
        if usr_new is not None:
            usr = usr_new
    return out, usr_emb


def interaction_aggregation(sd, x, y, heads):
    """InteractionAggregation.forward (decoder_leave_focal.py:411-423): w_x.x + w_y.y + sum_h x_h^T W_h y_h."""
    out = _lin(sd, "fusion_module.w_x", x) + _lin(sd, "fusion_module.w_y", y)
    if heads > 0:
        B, L, d = x.shape
        hx = d // heads
        W = sd["fusion_module.w_xy"].view(heads, hx, hx)       # output_dim = 1
        xh = x.reshape(B * L, heads, hx)
        yh = y.reshape(B * L, heads, hx)
        xy = torch.einsum("nhi,hij,nhj->n", xh, W, yh).view(B, L, 1)
        out = out + xy
    return out.squeeze(-1)


# --------------------------------------------------------------------------- losses
def focal_elementwise(logits, targets, exposure, alpha=0.5, gamma=2.0):
    """my_sigmoid_focal_loss (decoder_leave_focal.py:35-59), reduction='none'."""
    p = torch.sigmoid(logits) * exposure[None, :]
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
    return alpha_t * loss


def interest_bpr_all(logits, view_len, S, n_valid_global=None):
    """compute_interest_BPR_all (decoder_leave_focal.py:163-221), 40 generalised to S: rows with
    view_len < S; positive = logit at index view_len; negatives = the other S-1 positions (padding
    included); w = softmax(neg); loss = -mean log clamp(sum sigmoid(neg-pos)*w, 1e-8, 1-1e-8)."""
    valid = view_len < S
    z = logits[valid]
    v = view_len[valid]
    n = z.shape[0]
    pos = z[torch.arange(n), v]
    neg_mask = torch.ones_like(z, dtype=torch.bool)
    neg_mask[torch.arange(n), v] = False
    neg = z[neg_mask].view(n, S - 1)
    w = (neg - neg.max()).softmax(dim=1)
    soft = (neg - pos[:, None]).sigmoid() * w
    row = -(soft.sum(1)).clamp(min=1e-8, max=1 - 1e-8).log()
    return row.mean() if n_valid_global is None else row.sum() / n_valid_global


def leave_prob_ce(h_t, y, mask, mask_sum_global=None):
    """compute_leave_prob_CE (decoder_leave_focal.py:68-97): BCE-with-logits on exp(h_t), masked mean."""
    ce = F.binary_cross_entropy_with_logits(torch.exp(h_t), y, reduction="none")
    return (ce * mask).sum() / (mask.sum() if mask_sum_global is None else mask_sum_global)


def interest_leave_ce(logits, gt, mask, kind, use_mask, B_global=None):
    """compute_interest_leave_CE (decoder_leave_focal.py:99-161)."""
    ng = (gt != 0).to(logits.dtype).softmax(dim=1)
    ni = logits.softmax(dim=1)
    Bn = logits.shape[0] if B_global is None else B_global
    if kind == "CE":
        if use_mask:
            return (-(mask * ng * ni.log()).sum(1) / mask.sum(1)).sum() / Bn
        return -(ng * ni.log()).sum(1).sum() / Bn
    if use_mask:
        kl = F.kl_div(ni.log(), ng, reduction="none") * mask
        return (kl.sum(1) / mask.sum(1)).sum() / Bn
    return F.kl_div(ni.log(), ng, reduction="sum") / Bn


def huber(pred, true, delta=1.0, B_global=None):
    """huber_loss (decoder_leave_focal.py:61-66).  NB the call site passes [B] vs [B,1]
    (decoder_leave_focal.py:540) so the error broadcasts to [B,B]."""
    err = pred - true
    h = torch.where(err.abs() < delta, 0.5 * err ** 2, delta * (err.abs() - 0.5 * delta))
    return h.mean() if B_global is None else h.sum() / (B_global * B_global)


def partial_likelihood(hazard, view_len, S, B_global=None):
    """compute_partial_likelihood_loss (decoder_leave_focal.py:273-286), 40 generalised to S."""
    n = view_len.shape[0]
    ll = hazard.new_zeros(())
    for i in range(n):
        t = int(view_len[i])
        if t == S:
            continue
        ll = ll + torch.log(hazard[i, t] + 1e-6) - torch.log(hazard[i, t:].sum() + 1e-6)
    return -ll / (n if B_global is None else B_global)


def compute_loss(logits_raw, gt, cfg, sd=None, global_stats=None):
    """MultiScaleTemporalDetrLeaveFocal.compute_loss (decoder_leave_focal.py:490-572).
    ``gt`` is modified in place by 'focal' exactly like the reference (lines 534-535), which later
    losses in ``loss_type_list`` and the returned 'gt' observe.

    ``global_stats`` (test hook for the data-parallel claim of SURVEY.md §8(e)) = dict(v_all, v2_all,
    norms=[rows with view_len<S, rows, mask count]) of the GLOBAL batch: the rows passed in are then one
    shard, every mean becomes sum/global-count, and the shard losses add up to the full-batch loss."""
    B, S = gt.shape
    gs = global_stats
    Bg = None if gs is None else float(gs["norms"][1])
    dt = logits_raw.dtype
    mask = gt != -2
    logits = logits_raw
    if cfg.get("learnable_bias", 0):
        pos = torch.arange(S, dtype=dt)
        logits = logits + ((pos + 1) * sd["bias_weight"] + sd["bias_bias"])      # lines 497-504
    p = torch.sigmoid(logits)
    h_t = torch.cumsum(torch.log(p), dim=1)                                      # line 511
    survival = torch.exp(h_t)
    hazard = 1 - survival
    gt_binary = (gt == 1).to(dt)
    view_len_f = gt_binary.sum(1, keepdim=True)                                   # [B,1]
    view_len = view_len_f.squeeze(1).long()
    durations = (gt != -2).sum(1)
    hazard_m = torch.where(mask, hazard, torch.zeros_like(hazard))
    survival_m = torch.where(mask, survival, torch.zeros_like(survival))
    exposure = torch.tensor(cfg["exposure_prob"], dtype=dt)
    out = {}
    for name in cfg["loss_type_list"]:
        if name == "focal":
            gt[gt > 0] = 1
            gt[gt == -1] = 0
            el = focal_elementwise(logits, gt.to(dt), exposure)
            out["focal"] = el[mask].sum() / (B if gs is None else Bg)
        elif name == "huber":
            out["huber"] = huber(hazard_m.sum(1), view_len_f if gs is None else gs["v_all"].to(dt)[:, None], B_global=Bg)
        elif name == "hazard":
            out["hazard"] = partial_likelihood(hazard_m, view_len, S, Bg)
        elif name == "surviveCE":
            out["surviveCE"] = leave_prob_ce(h_t, gt_binary, mask, None if gs is None else float(gs["norms"][2]))
        elif name == "interestBPR":
            out["interestBPR"] = interest_bpr_all(logits, view_len, S, None if gs is None else float(gs["norms"][0]))
        elif name == "interestCE":
            out["interestCE"] = interest_leave_ce(logits, gt, mask, "CE", cfg.get("mask_loss", 0), Bg)
        elif name == "interestKL":
            out["interestKL"] = interest_leave_ce(logits, gt, mask, "KL", cfg.get("mask_loss", 0), Bg)
    # mse / mse2 (lines 552-558): [B] vs [B,1] broadcast -> mean over a [B,B] matrix, logged only
    sm2 = survival_m.clone()
    sm2[torch.arange(B), durations - 1] = 1
    vl2 = (gt >= 0).sum(1, keepdim=True).to(dt)
    if gs is None:
        out["mse"] = ((survival_m.sum(1)[None, :] - view_len_f) ** 2).mean()
        out["mse2"] = ((sm2.sum(1)[None, :] - vl2) ** 2).mean()
    else:
        out["mse"] = ((survival_m.sum(1)[None, :] - gs["v_all"].to(dt)[:, None]) ** 2).sum() / (Bg * Bg)
        out["mse2"] = ((sm2.sum(1)[None, :] - gs["v2_all"].to(dt)[:, None]) ** 2).sum() / (Bg * Bg)
    total = 0.0
    for name in cfg["loss_type_list"]:
        coef = cfg["loss_weight"]["mse"] if name == "huber" else cfg["loss_weight"][name]   # lines 561-566
        total = total + out[name] * coef
    out["loss"] = total
    out["logits"] = logits
    out["gt"] = gt
    return out


# --------------------------------------------------------------------------- whole model
def model_forward(sd: Dict[str, torch.Tensor], cfg: dict, inp: Dict[str, torch.Tensor], mode="train",
                  skip_dead=True, drop=None, global_stats=None):
    """MultiScaleTemporalDetrLeaveFocal.forward (decoder_leave_focal.py:574-658)."""
    N, h, S = cfg["N"], cfg["h"], cfg["S"]
    u_t, p_t = cfg["user"], cfg["photo"]
    abl = cfg.get("ablation_type", "ours")
    pe = cfg.get("use_pe", 1)
    if cfg.get("fwd_seed") is not None:      # fixture hook: the 'noPos' goldens reseed torch before every forward
        torch.manual_seed(cfg["fwd_seed"])

    def pick(kind, image, ident, which):
        if kind == "both":
            return image if which == 1 else ident
        return image if kind == "image" else ident

    if u_t != "both" and p_t != "both":
        vid, _ = backbone_forward(sd, "backbone1", pick(u_t, inp["usr_image"], inp["usr_id"], 1), inp["usr_mask"],
                                  pick(p_t, inp["vid_image"], inp["vid_id"], 1), inp["vid_mask"], N, h, S,
                                  skip_dead, drop, abl, pe)
        logits = _lin(sd, "stage_mlp1", vid).squeeze(-1)
    else:
        v1, _ = backbone_forward(sd, "backbone1", pick(u_t, inp["usr_image"], inp["usr_id"], 1), inp["usr_mask"],
                                 pick(p_t, inp["vid_image"], inp["vid_id"], 1), inp["vid_mask"], N, h, S,
                                 skip_dead, drop, abl, pe)
        v2, _ = backbone_forward(sd, "backbone2", pick(u_t, inp["usr_image"], inp["usr_id"], 2), inp["usr_mask"],
                                 pick(p_t, inp["vid_image"], inp["vid_id"], 2), inp["vid_mask"], N, h, S,
                                 skip_dead, drop, abl, pe)
        fh = cfg.get("fusion_heads", 2)
        if fh in (-2, -3):
            logits = _lin(sd, "stage_mlp1", v1 + v2).squeeze(-1)
        elif fh == -1:
            logits = _lin(sd, "stage_mlp1", torch.cat([v1, v2], -1)).squeeze(-1)
        elif fh == 0:
            logits = (_lin(sd, "stage_mlp1", v1) + _lin(sd, "stage_mlp2", v2)).squeeze(-1)
        else:
            logits = interaction_aggregation(sd, v1, v2, fh)
    if mode in ("train", "test"):
        return compute_loss(logits, inp["gt"], cfg, sd, global_stats)
    if cfg.get("learnable_bias", 0):
        pos = torch.arange(S, dtype=logits.dtype)
        logits = logits + ((pos + 1) * sd["bias_weight"] + sd["bias_bias"])
    return {"logits": logits, "gt": inp["gt"]}


def forward_backward(sd, cfg, inp, dtype=torch.float32, skip_dead=True, drop=None):
    """Forward + loss.backward(); returns (outputs, {name: grad or None}).  ``drop`` = None is the
    reference's eval mode; a callable (e.g. ``lambda t: F.dropout(t, 0.1)``) is its train mode."""
    params = {k: v.detach().clone().to(dtype if v.is_floating_point() else v.dtype).requires_grad_(v.is_floating_point())
              for k, v in sd.items()}
    inp = {k: (v if v is None else v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in inp.items()}
    out = model_forward(params, cfg, inp, "train", skip_dead, drop)
    out["loss"].backward()
    grads = {k: (p.grad if p.requires_grad else None) for k, p in params.items()}
    return out, grads


def adamw_step(params, grads, m, v, step, lr=1e-3, wd=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor semantics (the reference's optimiser,
    main_for_seq_leave_earlystop_SegMM.py:226,299): params with grad None are skipped entirely."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        p.mul_(1 - lr * wd)
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)


def train_steps(sd, cfg, inp, n_steps, dtype=torch.float32, skip_dead=True, lr=1e-3, wd=1e-4, drop=None):
    """n_steps of zero_grad -> forward -> backward -> AdamW on one batch (main...SegMM.py:269-300)."""
    params = {k: v.detach().clone().to(dtype) if v.is_floating_point() else v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(p) for k, p in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    losses = []
    for step in range(1, n_steps + 1):
        out, grads = forward_backward(params, cfg, inp, dtype, skip_dead, drop)
        losses.append(float(out["loss"].detach()))
        with torch.no_grad():
            adamw_step(params, grads, m, v, step, lr, wd)
    return params, losses


# --------------------------------------------------------------------------- metrics (numpy, integer ranks)
def _rank_of_target(pred: np.ndarray, target: np.ndarray) -> np.ndarray:
    """1-based rank of column ``target[i]`` in the ASCENDING order of row i; ties resolved
    lowest-index-first (= np.argsort default on these sizes, SURVEY §7 'Eval tie-breaking')."""
    tv = pred[np.arange(pred.shape[0]), target][:, None]
    idx = np.arange(pred.shape[1])[None, :]
    less = (pred < tv) | ((pred == tv) & (idx < target[:, None]))
    return less.sum(1) + 1


def _hr_ndcg(gt_rank):
    ev = {}
    for k in (1, 3, 5, 10):
        hit = (gt_rank <= k).astype(np.float32)
        ev["HR@%d" % k] = hit.mean()
        ev["NDCG@%d" % k] = (hit / np.log2(gt_rank + 1)).mean()
    return ev


def top_k_leave(interests, view_lengths, mask_batch, permutation=1, S=None, masked=False):
    """TOP_K_leave / TOP_K_leave_mask (my_evaluation.py:180-231 / 137-178).  With ``permutation``
    the same np.random stream is consumed (one np.random.permutation(seq_len) per valid row)."""
    bsz, seq_len = interests.shape
    S = seq_len if S is None else S
    vl = view_lengths.astype(np.int64).flatten()
    valid = (vl != mask_batch.sum(1)) if masked else (vl < S)
    vl = vl[valid]
    x = interests[valid]
    mb = mask_batch[valid]
    if masked:
        x = np.where(mb, x, 1.1)
    n = x.shape[0]
    if permutation:
        perm = np.array([np.random.permutation(seq_len) for _ in range(n)]).reshape(n, seq_len)
        pred = np.take_along_axis(x, perm, 1)
        target = np.argmax(perm == vl[:, None], axis=1)
    else:
        pred, target = x, vl
    return _hr_ndcg(_rank_of_target(pred, target))


def auc_rank_sum(labels, scores):
    """ROC-AUC by the Mann-Whitney rank sum with midranks (= sklearn.roc_auc_score, used by
    ProbAUC_batch my_evaluation.py:73-80 and SegRec/helpers/CTRRunner.py:34-35)."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    npos = labels.sum()
    nneg = len(labels) - npos
    return (r[labels].sum() - npos * (npos + 1) / 2.0) / (npos * nneg)


def wuauc(labels, scores, users):
    """Sample-weighted per-user AUC, single-class users skipped (SegRec/main.py:101-117)."""
    tot, w = 0.0, 0.0
    for u in np.unique(users):
        m = users == u
        if len(np.unique(labels[m])) < 2:
            continue
        tot += m.sum() * auc_rank_sum(labels[m], scores[m])
        w += m.sum()
    return tot / w


def eval_rows(interests: torch.Tensor, gt: torch.Tensor):
    """Per-row metrics of main_eval_batch (my_evaluation.py:264-357, test_type='new'):
    JaccardSim (IoU_Sim length_aware, :37-56), LeaveMSE prediction (:82-85), LeaveCTR (:87-90)."""
    surv = torch.exp(torch.cumsum(torch.log(interests), 1))
    mask = gt != -2
    res = {"JaccardSim": [], "LeaveMSE": [], "LeaveCTR": [], "LeaveCTR_view": [], "view_lengths": []}
    for i in range(gt.shape[0]):
        vl = int((gt[i] == 1).sum())
        du = int(mask[i].sum())
        I = (1 - (gt[i] - surv[i]).abs())[:vl].tolist() + [1.0] * (du - vl)
        res["JaccardSim"].append(sum(I) / du)
        res["LeaveMSE"].append(float(surv[i][mask[i]].sum()))
        res["view_lengths"].append(float(vl))
        res["LeaveCTR"].append(1 - float(interests[i, vl - 1]))
        res["LeaveCTR_view"].append(1 - float(surv[i, vl - 1]))
    return res


def prob_auc_batch(interests: torch.Tensor, gt: torch.Tensor):
    """ProbAUC_batch (my_evaluation.py:73-80) on survival probs, labels -1 -> 0."""
    surv = torch.exp(torch.cumsum(torch.log(interests), 1))
    m = gt != -2
    lab = gt[m].clone()
    lab[lab == -1] = 0
    return auc_rank_sum(lab.numpy(), surv[m].numpy())
