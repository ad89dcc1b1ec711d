"""Pins the CPU oracle (oracle/segmm_oracle.py) to golden vectors captured from the real
reference modules (oracle/gen_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import MODEL_CASES, TRAIN_CASES, load_case, GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import segmm_oracle as O  # noqa: E402


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_loss_grads(name):
    cfg, g, nograd, _ = load_case(name)
    out, grads = O.forward_backward(g["sd"], cfg, g["in"])
    assert torch.allclose(out["logits"], g["out"]["logits"], atol=2e-6, rtol=1e-5)
    for k, ref in g["out"].items():
        if ref.dim() == 0:
            assert torch.allclose(out[k].detach().float(), ref, rtol=2e-5, atol=1e-6), (k, float(out[k]), float(ref))
    assert torch.equal(out["gt"], g["out"]["gt"])
    # dead / live split identical to the reference's grad-is-None set
    live = {k for k, v in grads.items() if v is not None and k in g["grad"]}
    assert live == set(g["grad"].keys())
    for k in nograd:
        assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k
    for k, ref in g["grad"].items():
        scale = max(float(ref.abs().max()), 1e-6)
        err = float((grads[k] - ref).abs().max())
        assert err <= 2e-5 * scale + 2e-7, (k, err, scale)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_dead_layers_do_not_change_outputs(name):
    cfg, g, _, _ = load_case(name)
    a = O.model_forward(g["sd"], cfg, {k: v.clone() for k, v in g["in"].items()}, "inference", skip_dead=True)
    b = O.model_forward(g["sd"], cfg, {k: v.clone() for k, v in g["in"].items()}, "inference", skip_dead=False)
    assert torch.equal(a["logits"], b["logits"])
    assert torch.allclose(a["logits"], g["inf"]["logits"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if load_case(n)[1]["adam3"]])
def test_adamw_steps(name):
    cfg, g, nograd, extra = load_case(name)
    # Adam normalises g/sqrt(v): a gradient that is analytically zero (e.g. the head bias under
    # the shift-invariant BPR loss) is rounding noise whose SIGN decides a full lr-sized step,
    # so such tensors can only be pinned to within lr*steps.
    def tol(k, base, steps):
        gmax = float(g["grad"][k].abs().max()) if k in g["grad"] else 0.0
        return base if gmax > 1e-5 else 1.1e-3 * steps
    p1, _ = O.train_steps(g["sd"], cfg, g["in"], 1)
    for k, ref in g["adam1"].items():
        assert torch.allclose(p1[k], ref, atol=tol(k, 2e-6, 1), rtol=1e-5), k
    p3, losses = O.train_steps(g["sd"], cfg, g["in"], 3)
    for k, ref in g["adam3"].items():
        assert torch.allclose(p3[k], ref, atol=tol(k, 2e-5, 3), rtol=1e-4), k
    for k in nograd:   # dead parameters are never touched (AdamW skips grad None)
        assert torch.equal(p3[k], g["sd"][k]), k


class MaskReplay:
    """The oracle's ``drop`` hook fed the REFERENCE's recorded dropout masks: call k multiplies by mask k / (1 - p_k).
    ``order``: indices into the reference's call list, in the order the oracle will call (None = the same order)."""

    def __init__(self, masks, order=None, start=0):
        self.masks, self.order, self.i, self.start = masks, order, 0, start

    def __call__(self, t):
        k = self.start + (self.i if self.order is None else self.order[self.i])
        self.i += 1
        p, keep = self.masks[k]
        assert tuple(keep.shape) == tuple(t.shape), "dropout call %d: the reference dropped a %s tensor, the oracle a %s one" % (
            k, tuple(keep.shape), tuple(t.shape))
        return t * (keep.to(t.dtype) / (1.0 - p))


def _live_call_order(cfg):
    """Indices (into the reference's dropout call list of one training forward) of the calls that belong to the LIVE graph,
    in the reference's order.  Per backbone the reference calls dropout at: embedding video, user (encoder.py:461,471); then
    per layer, in execution order: v_logits, t_logits (:145,149), ff_usr, ff_vid (:166,167), MLP vid inner (mlp.py:22), vid
    outer (:202), MLP usr inner, usr outer (:205) -- 8 calls (6 under SelfAtt, whose user branch returns None, :172-173).
    Live: layers 0..N-3 fully (video-side calls only under SelfAtt), layer N-2 video side only, layer N-1 not."""
    N, abl = cfg["N"], cfg.get("ablation_type", "ours")
    two = cfg["user"] == "both" or cfg["photo"] == "both"
    order, k = [], 0
    for _ in range(2 if two else 1):
        order += [k, k + 1]
        k += 2
        if abl in ("CrossMLP", "SelfMLP", "w/oAtt"):
            raise NotImplementedError
        selfatt = "SelfAtt" in abl
        per = 6 if selfatt else 8
        vid_calls = [0, 3, 4, 5]
        for i in range(N):
            if i < N - 2 and not selfatt:
                order += [k + j for j in range(8)]
            elif i <= N - 2:
                order += [k + j for j in vid_calls]
            k += per
    return order, k


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_mode_dropout_placement_matches_reference(name):
    """TRAIN mode: the reference ran with model.train() and its dropout draws replaced by recorded masks
    (oracle/gen_golden.py MaskRecorder).  The oracle executing the reference's whole graph (skip_dead=False), fed the same
    masks BY CALL POSITION, must reproduce logits, every loss and every live gradient: this pins where dropout is applied
    (raw attention logits incl. mask fills before the 1/sqrt(dh) scale, encoder.py:145-150; after ff_usr / ff_vid, :166-167;
    inside the MLP at 0.1, mlp.py:22; after the MLP, :202,205; after the embedding LayerNorm, :461,471) and each site's p."""
    cfg, g, nograd, extra = load_case(name)
    masks, nfwd = extra["masks"], extra["mask_calls_fwd"]
    feed = MaskReplay(masks)
    out, grads = O.forward_backward(g["sd"], cfg, g["in"], skip_dead=False, drop=feed)
    assert feed.i == nfwd, "the oracle called dropout %d times, the reference %d times" % (feed.i, nfwd)
    assert torch.allclose(out["logits"], g["out"]["logits"], atol=2e-6, rtol=1e-5)
    for k, ref in g["out"].items():
        if ref.dim() == 0:
            assert torch.allclose(out[k].detach().float(), ref, rtol=2e-5, atol=1e-6), (k, float(out[k]), float(ref))
    live = {k for k, v in grads.items() if v is not None and k in g["grad"]}
    assert live == set(g["grad"].keys())
    for k in nograd:
        assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k
    for k, ref in g["grad"].items():
        scale = max(float(ref.abs().max()), 1e-6)
        assert float((grads[k] - ref).abs().max()) <= 2e-5 * scale + 2e-7, k
    # the inner MLP dropout is hard-wired to 0.1 whatever the model's dropout (mlp.py:8), every other site uses the model's p
    assert all(abs(p - 0.1) < 1e-12 for p, _ in masks)
    # dropout did something: eval-mode logits differ
    ev = O.model_forward(g["sd"], cfg, {k: v.clone() for k, v in g["in"].items()}, "inference")["logits"]
    assert float((ev - out["logits"].detach()).abs().max()) > 1e-3


@pytest.mark.parametrize("name", [n for n in TRAIN_CASES if "mlp" not in n])
def test_train_mode_live_graph_equals_reference(name):
    """The exact-liveness rule under dropout: the oracle computing ONLY the live graph (skip_dead=True, what the HIP engine
    executes), fed the reference's masks of the live call sites, equals the reference's full train-mode run."""
    cfg, g, nograd, extra = load_case(name)
    order, ncalls = _live_call_order(cfg)
    assert ncalls == extra["mask_calls_fwd"]
    feed = MaskReplay(extra["masks"], order)
    out, grads = O.forward_backward(g["sd"], cfg, g["in"], skip_dead=True, drop=feed)
    assert feed.i == len(order)
    assert torch.allclose(out["logits"], g["out"]["logits"], atol=2e-6, rtol=1e-5)
    assert abs(float(out["loss"]) - float(g["out"]["loss"])) < 2e-5 * max(1.0, abs(float(g["out"]["loss"])))
    for k, ref in g["grad"].items():
        scale = max(float(ref.abs().max()), 1e-6)
        assert float((grads[k] - ref).abs().max()) <= 2e-5 * scale + 2e-7, k


@pytest.mark.parametrize("name", [n for n in TRAIN_CASES if load_case(n)[1]["adam3"]])
def test_train_mode_adamw_steps(name):
    """Three train-mode AdamW steps of the reference (fresh masks every forward, consumed in call order) replayed by the oracle."""
    cfg, g, nograd, extra = load_case(name)
    masks, nfwd = extra["masks"], extra["mask_calls_fwd"]
    assert len(masks) == 4 * nfwd          # the captured forward/backward + three optimizer steps
    params = {k: v.clone() for k, v in g["sd"].items()}
    m = {k: torch.zeros_like(p) for k, p in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    for step in range(1, 4):
        feed = MaskReplay(masks, start=step * nfwd)
        _, grads = O.forward_backward(params, cfg, g["in"], skip_dead=False, drop=feed)
        with torch.no_grad():
            O.adamw_step(params, grads, m, v, step)
        if step in (1, 3):
            for k, ref in g["adam%d" % step].items():
                gmax = float(g["grad"][k].abs().max()) if k in g["grad"] else 0.0
                tol = (2e-6 if step == 1 else 2e-5) if gmax > 1e-5 else 2.2e-3 * step          # noise class: sign of a ~0 gradient, +-lr per step
                assert torch.allclose(params[k], ref, atol=tol, rtol=1e-4), (k, step)


def test_fp64_drift_bound():
    cfg, g, _, _ = load_case("img_d64_h16_N3_Lt100")
    out64, _ = O.forward_backward(g["sd"], cfg, g["in"], dtype=torch.float64)
    assert float((out64["logits"].float() - g["out"]["logits"]).abs().max()) < 2e-5


def test_metric_known_answers():
    z = np.load(os.path.join(GOLDEN, "metrics_kat.npz"))
    interests, gt = z["interests"], z["gt"]
    vl = (gt == 1).sum(1, keepdims=True)
    mask = gt != -2
    keys = ["%s@%d" % (m, k) for k in (1, 3, 5, 10) for m in ("HR", "NDCG")]
    for perm in (0, 1):
        np.random.seed(42)
        e = O.top_k_leave(interests, vl, mask, permutation=perm)
        assert np.array_equal(np.array([e[k] for k in keys], dtype=np.float64), z["topk_perm%d" % perm])
        np.random.seed(42)
        e = O.top_k_leave(interests, vl, mask, permutation=perm, masked=True)
        assert np.array_equal(np.array([e[k] for k in keys], dtype=np.float64), z["topkmask_perm%d" % perm])
    assert np.array_equal(np.argmin(interests, 1), z["min_indices"])
    rows = z["meb_rows"]
    it, g_ = torch.from_numpy(interests[rows]), torch.from_numpy(gt[rows])
    res = O.eval_rows(it, g_)
    for k in ("JaccardSim", "LeaveMSE", "LeaveCTR", "LeaveCTR_view", "view_lengths"):
        assert np.allclose(np.array(res[k]), z["meb/" + k], rtol=1e-6, atol=1e-7), k
    assert abs(O.prob_auc_batch(it, g_) - float(z["meb/ProbAUC"][0])) < 1e-12
    assert abs(O.auc_rank_sum(z["auc_labels"], z["auc_scores"]) - float(z["auc"])) < 1e-12
    assert abs(O.wuauc(z["auc_labels"], z["auc_scores"], z["auc_users"]) - float(z["wuauc"])) < 1e-12


def test_topk_hand_case():
    # SURVEY §8(c) KAT: view_len=[1,3,40] -> HR@1 .5, HR@3 1, NDCG@3 .75
    x = np.ones((3, 40), dtype=np.float32)
    x[0, :4] = [.9, .1, .5, .7]
    x[1, :4] = [.2, .8, .3, .4]
    x[2, :] = .5
    vl = np.array([[1], [3], [40]])
    e = O.top_k_leave(x, vl, np.ones((3, 40), bool), permutation=0)
    assert e["HR@1"] == 0.5 and e["HR@3"] == 1.0 and abs(e["NDCG@3"] - 0.75) < 1e-7
