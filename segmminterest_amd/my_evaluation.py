"""Host-side mirror of ``MMinterest/models/my_evaluation.py``: ranking / AUC / view-length metrics.

Same function names, arguments and returned keys as the reference (my_evaluation.py:73-231,264-357).
Ranks are computed as integer counts ("how many entries sort before the target", ties broken by the
lower index -- what ``np.argsort`` does on these sizes), so HR@k / NDCG@k are bit-exact;
with ``permutation=1`` the same ``np.random`` stream is consumed (one ``np.random.permutation(seq_len)``
per valid row, seed 42 at import like my_evaluation.py:14-15).

Device path: when the interests / labels are HIP tensors, the ``*_device`` functions and
``main_eval_batch`` compute the integer ranks and the AUC pair counts with the kernels of ``csrc/evalops.h``
(bit-exact integers; only B view lengths go to the host -- for the np.random permutation parity -- and B ranks or
three counters come back), instead of moving [B, S] interests, labels and masks to the host like the reference.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

seed_value = 42
np.random.seed(seed_value)
random.seed(seed_value)
os.environ["PYTHONHASHSEED"] = str(seed_value)


def _rank_of_target(pred, target):
    tv = pred[np.arange(pred.shape[0]), target][:, None]
    idx = np.arange(pred.shape[1])[None, :]
    return ((pred < tv) | ((pred == tv) & (idx < target[:, None]))).sum(1) + 1


def _evaluations(gt_rank, verbose=True):
    evaluations = {}
    for k in [1, 3, 5, 10]:
        hit = (gt_rank <= k).astype(np.float32)
        evaluations["HR@%d" % k] = hit.mean()
        evaluations["NDCG@%d" % k] = (hit / np.log2(gt_rank + 1)).mean()
    if verbose:
        print(evaluations)
    return evaluations


def _topk(interests, view_lengths, mask_batch, permutation, masked, seq_valid=None):
    bsz, seq_len = interests.shape
    vl = view_lengths.astype(np.int64).flatten()
    valid = (vl != mask_batch.sum(axis=1)) if masked else (vl < (seq_len if seq_valid is None else seq_valid))
    vl = vl[valid]
    x = interests[valid, :]
    if masked:
        x = np.where(mask_batch[valid, :], x, 1.1)
    n = x.shape[0]
    if permutation:
        perm = np.array([np.random.permutation(seq_len) for _ in range(n)]).reshape(n, seq_len)
        pred = np.take_along_axis(x, perm, 1)
        target = np.argmax(perm == vl[:, None], axis=1)
    else:
        pred, target = x, vl
    return _rank_of_target(pred, target)


def TOP_K_leave(interests, view_lengths, mask_batch, permutation=1, test=0):
    """Rank of the leave segment among all S positions (my_evaluation.py:180-231)."""
    min_indices = np.argmin(interests, axis=1)
    ev = _evaluations(_topk(interests, view_lengths, mask_batch, permutation, masked=False))
    return (ev, min_indices) if test else ev


def TOP_K_leave_mask(interests, view_lengths, mask_batch, permutation=1):
    """Same with padded positions pushed to the end (interest 1.1) and fully-watched rows dropped
    (my_evaluation.py:137-178)."""
    return _evaluations(_topk(interests, view_lengths, mask_batch, permutation, masked=True))


def TOP_K_leave_device(interests: torch.Tensor, gt: torch.Tensor, permutation=1, masked=False, seq_valid=None, test=0, gather=None):
    """TOP_K_leave / TOP_K_leave_mask on the device: interests [B, S] float32 and labels gt [B, S] int64 stay in HBM.
    Returns the same dict (same float arithmetic on the same integer ranks).  With ``permutation`` the host draws the
    candidate shuffles from np.random exactly as the host path does (one permutation per VALID row, in row order).
    ``gather`` (data parallel): callable int32 [B] -> int32 [G*B] that collects the leave ranks of every
    rank's rows; the metrics are then those of the GLOBAL batch, identical on every rank and -- with ``permutation=0`` --
    bit-identical to a single process evaluating the whole batch (the ranks are integers)."""
    from . import hipabi as H
    B, S = gt.shape
    x = interests.detach()
    if x.dtype != torch.float32:
        x = x.float()
    if x.stride(-1) != 1:
        x = x.contiguous()
    gt = gt.contiguous()
    perm = None
    if permutation:
        vl = (gt == 1).sum(1)
        valid = (vl != (gt != -2).sum(1)) if masked else (vl < (S if seq_valid is None else seq_valid))
        n = int(valid.sum().item())
        pv = np.array([np.random.permutation(S) for _ in range(n)], dtype=np.int32).reshape(n, S)
        perm = torch.zeros((B, S), dtype=torch.int32, device=x.device)
        perm[valid] = torch.from_numpy(pv).to(x.device)
    ranks, _hist = H.rank_leave(x, gt, perm=perm, masked=masked, seq_valid=seq_valid)
    if gather is not None:
        ranks = gather(ranks)
    r = ranks.cpu().numpy().astype(np.int64)
    ev = _evaluations(r[r > 0])
    if test:
        return ev, torch.argmin(x, dim=1).cpu().numpy()
    return ev


def ProbAUC_batch_device(interests: torch.Tensor, gt: torch.Tensor, survival=None):
    """ProbAUC_batch (my_evaluation.py:73-80) from integer pair counts computed on the device."""
    from . import hipabi as H
    x = interests.detach().float()
    if x.stride(-1) != 1:
        x = x.contiguous()
    surv, label = H.survival(x, gt.contiguous())
    if survival is not None:          # test_type == "old": the interests already are survival probabilities
        surv = survival.detach().float().contiguous()
    seg = torch.tensor([0, surv.numel()], dtype=torch.int64, device=surv.device)
    u2, npos, nneg = (int(v) for v in H.auc_counts(surv.view(-1), label.view(-1), seg)[0].tolist())
    return u2 / (2.0 * npos * nneg)


def wuAUC_device(labels: torch.Tensor, scores: torch.Tensor, users: torch.Tensor):
    """Sample-weighted per-user AUC (SegRec/main.py:101-117) with the per-user pair counts computed on the device."""
    from . import hipabi as H
    order = torch.sort(users, stable=True).indices
    u, counts = torch.unique_consecutive(users[order], return_counts=True)
    seg = torch.zeros(u.numel() + 1, dtype=torch.int64, device=users.device)
    seg[1:] = torch.cumsum(counts, 0)
    c = H.auc_counts(scores[order].float().contiguous(), labels[order].to(torch.int8).contiguous(), seg).cpu().numpy().astype(np.float64)
    n = counts.cpu().numpy().astype(np.float64)
    ok = (c[:, 1] > 0) & (c[:, 2] > 0)
    auc = c[ok, 0] / (2.0 * c[ok, 1] * c[ok, 2])
    return float((n[ok] * auc).sum() / n[ok].sum())


def auc_rank_sum(labels, scores):
    """ROC-AUC via the Mann-Whitney rank sum with midranks (what sklearn.roc_auc_score returns)."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    bounds = np.flatnonzero(np.concatenate(([True], s[1:] != s[:-1], [True])))
    for a, b in zip(bounds[:-1], bounds[1:]):
        ranks[a:b] = 0.5 * (a + b - 1) + 1.0
    r = np.empty_like(ranks)
    r[order] = ranks
    npos = labels.sum()
    nneg = len(labels) - npos
    return (r[labels].sum() - npos * (npos + 1) / 2.0) / (npos * nneg)


def wuAUC(labels, scores, users):
    """Sample-weighted per-user AUC (= GAUC), single-class users skipped (SegRec/main.py:101-117)."""
    tot, w = 0.0, 0.0
    for u in np.unique(users):
        m = users == u
        if len(np.unique(labels[m])) < 2:
            continue
        tot += m.sum() * auc_rank_sum(labels[m], scores[m])
        w += m.sum()
    return tot / w


def ProbAUC_batch(probs, labels, masks):
    """my_evaluation.py:73-80."""
    mp = probs[masks == 1]
    ml = labels[masks == 1]
    ml = torch.where(ml == -1, torch.zeros_like(ml), ml)
    return auc_rank_sum(ml.detach().cpu().numpy().flatten(), mp.detach().cpu().numpy().flatten())


def IoU_Sim(logit, label, view_length, duration, type="length_aware"):
    """my_evaluation.py:37-56."""
    I = (1 - (label - logit).abs()).cpu().tolist()
    I_original = I[:view_length]
    I_length_aware = I_original + [1.0] * (duration - view_length)
    if type == "original":
        return float(sum(I_original) / view_length)
    if type == "length_aware":
        return float(sum(I_length_aware) / duration)
    raise ValueError("Invalid Value for IoU type: Supported 'original', 'length_aware'")


def predict_view_length(prob, mask):
    return torch.sum(prob[mask == 1]).item()


def LeaveCTR(interest, survival_prob, view_length):
    return 1 - interest[view_length - 1].item(), 1 - survival_prob[view_length - 1].item()


def draw_hotmap(*args, **kwargs):
    raise NotImplementedError("plotting is outside the training path (my_evaluation.py:233-262)")


def main_eval_batch(args, interests, ground_truths, pred_labels, results_list, type="inference", test_type="new", logits=None):
    """my_evaluation.py:264-357: appends per-batch / per-row metrics to ``results_list``."""
    mask_batch = ground_truths != -2
    if test_type == "old":
        survival_probs = interests
    else:
        survival_probs = torch.exp(torch.cumsum(torch.log(interests), dim=1))
    on_device = interests.is_cuda and ground_truths.is_cuda
    if "ProbAUC" in results_list:
        if on_device:
            results_list["ProbAUC"].append(float(ProbAUC_batch_device(interests, ground_truths,
                                                                      survival=interests if test_type == "old" else None)))
        else:
            results_list["ProbAUC"].append(float(ProbAUC_batch(survival_probs, ground_truths, mask_batch)))
    if "TOP_K" in results_list and on_device:
        out = TOP_K_leave_device(interests, ground_truths, permutation=args.TOP_K_permutation, masked=bool(args.TOP_K_mask),
                                 test=1 if ("TOP1MSE" in results_list and not args.TOP_K_mask) else 0)
        if isinstance(out, tuple):
            evaluations, top1 = out
            results_list["TOP1MSE"].append(top1)
        else:
            evaluations = out
        for k, v in evaluations.items():
            results_list.setdefault(k, []).append(float(v))
    elif "TOP_K" in results_list:
        view_lengths = (ground_truths == 1).sum(dim=1, keepdim=True).cpu().numpy()
        x = interests.cpu().detach().numpy()
        mb = mask_batch.cpu().detach().numpy()
        if args.TOP_K_mask:
            evaluations = TOP_K_leave_mask(x, view_lengths, mb, permutation=args.TOP_K_permutation)
        elif "TOP1MSE" in results_list:
            evaluations, top1 = TOP_K_leave(x, view_lengths, mb, permutation=args.TOP_K_permutation, test=1)
            results_list["TOP1MSE"].append(top1)
        else:
            evaluations = TOP_K_leave(x, view_lengths, mb, permutation=args.TOP_K_permutation)
        for k, v in evaluations.items():
            results_list.setdefault(k, []).append(float(v))
    if logits is not None:
        # my_evaluation.py:307-318: the leave position predicted from the raw logits -- p_leave = normalised 1 / softmax(logits),
        # expected position sum(p_leave * [0 .. S-1]) truncated to int -- against the view lengths: 'MAES' is a RUNNING SUM of
        # batch MAE * batch size, 'pred_leave' collects the int predictions.  (The reference's `print` of the sum is not reproduced;
        # its literal 40 positions is the segment axis S of the logits.)
        lg = logits.detach().float()
        inv_sm = 1.0 / torch.softmax(lg, dim=1)
        leave_p = inv_sm / inv_sm.sum(dim=1, keepdim=True)
        pos = torch.arange(lg.shape[1], dtype=torch.float32, device=lg.device)
        pred_leave = torch.sum(leave_p * pos, dim=1).int().cpu()
        view_len = (ground_truths == 1).sum(dim=1).cpu()
        mae = float((view_len.to(torch.float64) - pred_leave.to(torch.float64)).abs().mean())
        results_list["MAES"] += mae * int(interests.shape[0])
        results_list["pred_leave"].append(pred_leave)
    per_row = [k for k in results_list if k in ("JaccardSim", "LeaveMSE", "LeaveCTR", "LeaveCTR_view")]
    if per_row:
        it, sp, gt_, mk = interests.cpu(), survival_probs.cpu(), ground_truths.cpu(), mask_batch.cpu()
        for interest, survival_prob, label, mask in zip(it, sp, gt_, mk):
            view_length = int((label == 1).sum())
            duration = int((label != -2).sum())
            for eval_type in per_row:
                if eval_type == "JaccardSim":
                    results_list[eval_type].append(IoU_Sim(survival_prob, label, view_length, duration))
                elif eval_type == "LeaveMSE":
                    results_list[eval_type].append(float(predict_view_length(survival_prob, mask)))
                    results_list["view_lengths"].append(float(view_length))
                    if "duration_lengths" in results_list:
                        results_list["duration_lengths"].append(float(duration))
                elif eval_type == "LeaveCTR":
                    results_list[eval_type].append(float(LeaveCTR(interest, survival_prob, view_length)[0]))
                elif eval_type == "LeaveCTR_view":
                    results_list[eval_type].append(float(LeaveCTR(interest, survival_prob, view_length)[1]))
    return results_list
