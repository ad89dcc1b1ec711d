"""Pins the CPU oracle (oracle/segmm_oracle.py) to golden vectors captured from the real
reference modules (oracle/gen_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import MODEL_CASES, load_case, GOLDEN, ROOT

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
