"""SURVEY.md §8(f) rows on the MI355X: on-device evaluation (integer ranks / AUC pair counts, bit-exact against the
numpy oracle), resident-table feature gather, SegRec weighted head."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _abi():
    from segmminterest_amd import hipabi
    hipabi.lib()
    return hipabi


def _labels(B, S, seed):
    from segmminterest_amd.synth import make_labels
    return make_labels(B, S, torch.Generator().manual_seed(seed))[0]


@pytest.mark.parametrize("B,S", [(64, 40), (513, 40), (37, 20), (5, 100)])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("permutation", [0, 1])
def test_rank_leave_bit_exact_vs_oracle(B, S, masked, permutation):
    """Integer ranks of the leave segment: device kernel == numpy oracle (oracle/segmm_oracle.py::top_k_leave restates
    my_evaluation.py:137-231), including ties (quantised interests), padded rows and shuffled candidates."""
    H = _abi()
    from oracle import segmm_oracle as O
    from segmminterest_amd import my_evaluation as E
    gt = _labels(B, S, seed=B + S)
    g = torch.Generator().manual_seed(B * 7 + S)
    x = (torch.rand(B, S, generator=g) * 8).round() / 8 * 0.9 + 0.05          # many exact ties
    vl = (gt == 1).sum(1, keepdim=True).numpy()
    mb = (gt != -2).numpy()
    np.random.seed(123)
    want = O.top_k_leave(x.numpy(), vl, mb, permutation=permutation, S=S, masked=masked)
    np.random.seed(123)
    got = E.TOP_K_leave_device(x.to(DEV), gt.to(DEV), permutation=permutation, masked=masked)
    assert set(got) == set(want)
    for k in want:
        assert float(got[k]) == float(want[k]), (k, got[k], want[k])
    # raw ranks, no permutation: every row against a direct count
    ranks, hist = H.rank_leave(x.to(DEV), gt.to(DEV), masked=masked)
    r = ranks.cpu().numpy()
    xs = np.where(mb, x.numpy(), 1.1) if masked else x.numpy()
    v = vl.reshape(-1)
    valid = (v != mb.sum(1)) if masked else (v < S)
    for b in range(B):
        if not valid[b]:
            assert r[b] == 0
            continue
        t = v[b]
        want_r = 1 + int(((xs[b] < xs[b, t]) | ((xs[b] == xs[b, t]) & (np.arange(S) < t))).sum())
        assert r[b] == want_r
    assert hist.cpu().numpy().sum() == B and hist[0].item() == int((~valid).sum())


def test_auc_counts_and_probauc():
    H = _abi()
    from oracle import segmm_oracle as O
    from segmminterest_amd import my_evaluation as E
    B, S = 300, 40
    gt = _labels(B, S, seed=9)
    g = torch.Generator().manual_seed(4)
    interests = (torch.rand(B, S, generator=g) * 0.98 + 0.01)
    interests[:, ::7] = 0.5                                                  # ties
    surv, label = H.survival(interests.to(DEV), gt.to(DEV))
    ref_surv = torch.exp(torch.cumsum(torch.log(interests), 1))
    assert torch.allclose(surv.cpu(), ref_surv, rtol=1e-4, atol=1e-30)     # exp(h) with |h| up to ~40: a 1-ulp change of h is 4e-6 relative
    lab = label.cpu().numpy().reshape(-1)
    assert ((lab == -1) == (gt.numpy().reshape(-1) == -2)).all() and ((lab == 1) == (gt.numpy().reshape(-1) == 1)).all()
    # integer pair counts against a direct numpy count on the SAME survival values
    s = surv.cpu().numpy().reshape(-1).astype(np.float64)
    pos, neg = s[lab == 1], s[lab == 0]
    seg = torch.tensor([0, B * S], dtype=torch.int64, device=DEV)
    u2, npos, nneg = H.auc_counts(surv.view(-1), label.view(-1), seg)[0].tolist()
    less = (neg[None, :] < pos[:, None]).sum()
    eq = (neg[None, :] == pos[:, None]).sum()
    assert (u2, npos, nneg) == (2 * int(less) + int(eq), len(pos), len(neg))
    auc = u2 / (2.0 * npos * nneg)
    m = lab >= 0
    assert abs(auc - O.auc_rank_sum(lab[m] == 1, s[m])) < 1e-12            # = sklearn.roc_auc_score (midranks)
    assert abs(E.ProbAUC_batch_device(interests.to(DEV), gt.to(DEV)) - auc) == 0.0


def test_wuauc_device_matches_oracle():
    from oracle import segmm_oracle as O
    from segmminterest_amd import my_evaluation as E
    _abi()
    g = torch.Generator().manual_seed(11)
    n = 5000
    users = torch.randint(0, 300, (n,), generator=g)
    labels = (torch.rand(n, generator=g) < 0.3).long()
    labels[users == 7] = 1                                                   # a single-class user: skipped
    scores = (torch.rand(n, generator=g) * 50).round() / 50                  # ties
    want = O.wuauc(labels.numpy(), scores.numpy().astype(np.float64), users.numpy())
    got = E.wuAUC_device(labels.to(DEV), scores.to(DEV), users.to(DEV))
    assert abs(got - want) < 1e-12


def test_gather_l1_matches_host_pipeline():
    H = _abi()
    g = torch.Generator().manual_seed(2)
    n_lines, D, B, S, Lt = 1000, 768, 33, 40, 100
    table = torch.rand(n_lines, D, generator=g)
    idx_v = torch.randint(0, n_lines, (B, S), generator=g)
    idx_u = torch.randint(0, n_lines, (B, Lt), generator=g)
    dur = torch.randint(2, S + 1, (B,), generator=g)
    idx_v[torch.arange(S)[None, :] >= dur[:, None]] = -1                     # collator padding
    idx_u[:, 60:] = -1
    idx_u[0, 3] = n_lines + 5                                                # out of range = padding, never read
    out, mask = H.gather_l1(table.to(DEV), idx_v.to(DEV))
    ref = table[idx_v.clamp(0, n_lines - 1)]
    ref = ref / (ref.abs().sum(-1, keepdim=True) + 1e-6)
    ok = (idx_v >= 0) & (idx_v < n_lines)
    ref[~ok] = 0
    assert torch.equal(mask.cpu(), ok)
    assert torch.allclose(out.cpu(), ref, rtol=2e-6, atol=0)
    out_u, mask_u = H.gather_l1(table.to(DEV), idx_u.to(DEV), normalize=False)
    oku = (idx_u >= 0) & (idx_u < n_lines)
    refu = table[idx_u.clamp(0, n_lines - 1)]
    refu[~oku] = 0
    assert torch.equal(out_u.cpu(), refu) and torch.equal(mask_u.cpu(), oku)


def test_index_batches_train_like_feature_batches():
    """A Trainer fed with index batches + a resident table takes the same step as one fed with the gathered,
    host-padded feature tensors (the reference's DataCollator contract)."""
    _abi()
    from segmminterest_amd.feature_store import ResidentFeatureTable
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    B, S, Lt, D, N = 16, 40, 10, 64, 2
    g = torch.Generator().manual_seed(5)
    table = torch.rand(500, D, generator=g)
    base = make_batch(B, S, Lt, D, seed=77)
    idx_v = torch.randint(0, 500, (B, S), generator=g)
    idx_v[~base["photo_mask"]] = -1
    idx_u = torch.randint(0, 500, (B, Lt), generator=g)
    idx_u[~base["user_mask"]] = -1
    feat = dict(base)
    feat["photo"] = torch.where((idx_v >= 0)[..., None], table[idx_v.clamp(0)], torch.zeros(()))
    feat["user"] = torch.where((idx_u >= 0)[..., None], table[idx_u.clamp(0)], torch.zeros(()))
    idxb = {k: v for k, v in base.items() if k not in ("photo", "user", "photo_mask", "user_mask")}
    idxb["photo_idx"], idxb["user_idx"] = idx_v, idx_u
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    outs = []
    for batch, ft in ((feat, None), (idxb, table)):
        torch.manual_seed(1)
        model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(DEV)
        tr = Trainer(model, dropout=False, feature_table=None if ft is None else ResidentFeatureTable(ft.to(DEV)))
        o = tr.train_step({k: v.to(DEV) for k, v in batch.items()})
        outs.append((float(o["loss"]), o["logits"].detach().cpu(), model._store.flat.detach().cpu().clone()))
    assert abs(outs[0][0] - outs[1][0]) < 1e-6
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-5)
    assert torch.allclose(outs[0][2], outs[1][2], atol=1e-5)


@pytest.mark.parametrize("index_input", [False, True])
def test_input_prefetch_changes_nothing(index_input):
    """Trainer.prefetch (the input stage of the NEXT batch on its own stream, alternating output buffers) leaves every step
    bitwise unchanged: 6 steps over 3 rotating batches with and without it, feature batches and index batches -- incl. a
    prefetched batch that is then NOT the one trained on (its results must be dropped)."""
    _abi()
    from segmminterest_amd.feature_store import ResidentFeatureTable
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    B, S, Lt, D, N = 16, 40, 10, 64, 2
    g = torch.Generator().manual_seed(9)
    table = torch.rand(700, D, generator=g)
    batches = []
    for i in range(3):
        b = make_batch(B, S, Lt, D, seed=100 + i)
        if index_input:
            iv = torch.randint(0, 700, (B, S), generator=g)
            iv[~b["photo_mask"]] = -1
            iu = torch.randint(0, 700, (B, Lt), generator=g)
            iu[~b["user_mask"]] = -1
            b = {k: v for k, v in b.items() if k not in ("photo", "user", "photo_mask", "user_mask")}
            b["photo_idx"], b["user_idx"] = iv, iu
        batches.append({k: v.to(DEV) for k, v in b.items()})
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    res = []
    for prefetch in (False, True):
        torch.manual_seed(1)
        model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(DEV)
        tr = Trainer(model, dropout=False, feature_table=ResidentFeatureTable(table.to(DEV)) if index_input else None)
        losses = []
        for i in range(6):
            nxt = None
            if prefetch:
                nxt = batches[(i + 1) % 3] if i != 3 else batches[0]          # step 3 announces the WRONG next batch
            losses.append(float(tr.train_step(batches[i % 3], next_batch=nxt)["loss"].detach()))
        torch.cuda.synchronize()
        res.append((losses, model._store.flat.detach().cpu().clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


def test_segment_weighted_sum_matches_cliprec():
    H = _abi()
    g = torch.Generator().manual_seed(8)
    Bn, I, S = 31, 7, 40
    pred = torch.randn(Bn, I, S, generator=g)
    w = torch.rand(Bn, I, S, generator=g)
    dur = torch.randint(0, S + 1, (Bn, I), generator=g)
    mask = (torch.arange(S)[None, None, :] < dur[..., None]).float()          # ClipRec.py:171-173
    ref = (pred * w * mask).sum(-1)                                           # ClipRec.py:178-180
    got = H.segment_weighted_sum(pred.to(DEV), w.to(DEV), dur.to(DEV))
    assert torch.allclose(got.cpu(), ref, atol=2e-5)
    got1 = H.segment_weighted_sum(pred.to(DEV))
    assert torch.allclose(got1.cpu(), pred.sum(-1), atol=2e-5)


def test_valid_model_matches_host_metrics():
    """Trainer.valid_model (device ranks) == the reference's host loop over the same batches."""
    _abi()
    from segmminterest_amd import my_evaluation as E
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    B, S, Lt, D, N = 48, 40, 10, 64, 2
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[0.9] * S)
    torch.manual_seed(3)
    model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(DEV)
    tr = Trainer(model)
    batches = [{k: v.to(DEV) for k, v in make_batch(B, S, Lt, D, seed=100 + i).items()} for i in range(3)]
    np.random.seed(7)
    got = tr.valid_model(batches, permutation=1)
    np.random.seed(7)
    acc = {}
    for b in batches:
        out = tr.eval_step(b, mode="train")
        interests = torch.sigmoid(out["logits"]) * torch.tensor(model.exposure_prob, device=DEV)
        gt = out["gt"]
        ev = E.TOP_K_leave(interests.cpu().numpy(), (gt == 1).sum(1, keepdim=True).cpu().numpy(), (gt != -2).cpu().numpy(), permutation=1)
        for k, v in ev.items():
            acc.setdefault(k, []).append(float(v))
        acc.setdefault("valid_loss", []).append(float(out["loss"]))
    for k, v in acc.items():
        assert got[k] == sum(v) / len(v), k


# ------------------------------------------------------------------ (f)-1 / (f)-3 against fixtures made by the reference's own code
def test_index_batch_gather_matches_reference_dataset_rows():
    """oracle/gen_golden_io.py ran FrameDatasetSeq_SegMM._getitem (dataloader_SegMM.py:271-362) on a synthetic corpus; the
    index rows built by IndexBatchBuilder, gathered + L1-normalised on the device, equal the reference's feature rows after
    the trainer's normalisation (main...SegMM.py:272-273), masks included."""
    import json
    import random
    import numpy as np
    from segmminterest_amd.feature_store import IndexBatchBuilder, KeyIndex, ResidentFeatureTable
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_dataloader.npz"))
    rows = json.loads(str(z["rows"]))
    b = IndexBatchBuilder(KeyIndex([str(k) for k in z["keys"]]), json.loads(str(z["user_input_dict"])),
                          json.loads(str(z["user2id"])), json.loads(str(z["item2id"])))
    random.seed(int(z["seed"]))
    np.random.seed(int(z["seed"]))
    batch = b.batch([b.row(r["user_id"], r["video_id"], r["time_ms"], r["duration_ms"], r["playing_time_x"], r["label_1D"],
                           r["history_items"], r["history_playing"], r["history_lengths"]) for r in rows], device=DEV)
    ft = ResidentFeatureTable(torch.from_numpy(z["table"]).to(DEV))
    for key in ("photo", "user"):
        got, mask = ft.gather(key, batch[key + "_idx"])
        ref = torch.from_numpy(z["exp_" + key]).to(DEV)
        ref = ref / (ref.abs().sum(-1, keepdim=True) + 1e-6)
        assert torch.equal(mask.cpu(), torch.from_numpy(z["exp_" + key + "_mask"]))
        assert float((got - ref).abs().max()) <= 1e-7


def test_weighted_head_matches_cliprec_forward():
    """ClipRecBase.forward (SegRec/models/context/ClipRec.py:134-198) run by oracle/gen_golden_io.py: the device kernel
    reproduces its weighted, duration-masked prediction from the per-clip predictions."""
    import numpy as np
    from segmminterest_amd.bridge import weighted_head
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_cliprec.npz"))
    cp, w = torch.from_numpy(z["clip_pred"]).to(DEV), torch.from_numpy(z["weight"]).to(DEV)
    dur = torch.from_numpy(z["duration"]).to(DEV)
    for got, ref in ((weighted_head(cp, w, dur), z["pred_weighted_masked"]), (weighted_head(cp, None, dur), z["pred_ones_masked"]),
                     (weighted_head(cp, w, None), z["pred_weighted_nomask"])):
        assert float((got.cpu() - torch.from_numpy(ref)).abs().max()) <= 2e-5


@pytest.mark.gpu
def test_device_state_optimizer_resumes_from_checkpoint():
    """FusedAdamW.load_state_dict in device_state mode puts the checkpoint's step count into the device-side step state (its
    bias corrections are what segmm_adamw(step = -1) uses): a trainer restored after 3 steps takes the same 4th step."""
    import torch
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D = 16, 20, 6, 32
    margs = default_args(num_layers_enc=2, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=5).items()}
    torch.manual_seed(1)
    model = init_model(margs, n_users=5, n_items=5, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
    tr = Trainer(model, dropout=False, device_state=True)
    for _ in range(3):
        tr.train_step(batch)
    sd_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_o = tr.opt.state_dict()
    tr.train_step(batch)
    want = model._store.flat.detach().clone()
    torch.manual_seed(1)
    model2 = init_model(margs, n_users=5, n_items=5, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
    tr2 = Trainer(model2, dropout=False, device_state=True)          # resets the device state to step 0 ...
    model2.load_state_dict(sd_m)
    tr2.opt.load_state_dict(sd_o)                                     # ... and this puts it at step 3
    assert H.step_get()[1] == 3
    tr2.train_step(batch)
    assert H.step_get()[1] == 4
    got = model2._store.flat.detach()
    assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_recorded_steps_with_validation_between_them():
    """run_recorded() must mirror every host side effect of the eager step.  The recorded step rewrites the weights and
    re-splits the weight planes at the head of the step, so outside it the planes are one optimizer step stale: an evaluation
    pass between recorded steps has to re-split them (ParamStore.refresh_planes keyed on fused_version).  Interleaving eval_step
    with recorded steps must give the same evaluation logits and the same final parameters as the eager device-state run."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D, N, h = 32, 40, 10, 64, 2, 4
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batches = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=200 + i).items()} for i in range(3)]

    def run(graph):
        torch.manual_seed(5)
        model = init_model(margs, n_users=50, n_items=500, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        tr = Trainer(model, lr=1e-2, weight_decay=1e-4, device_state=True)          # large lr: a stale plane set is visible
        if graph:
            tr.record(batches[0], warmup=2)          # 2 eager steps + the recorded one
        else:
            for _ in range(3):
                tr.train_step(batches[0])
        evals = []
        for t in range(6):
            tr.run_recorded(batches[t % 3]) if graph else tr.train_step(batches[t % 3])
            if t % 2 == 1:          # validation after steps 2, 4, 6: the SECOND and third must not run on the first one's planes
                evals.append(tr.eval_step(batches[2], mode="inference")["logits"].detach().clone())
        torch.cuda.synchronize()
        return model._store.flat.detach().clone(), evals

    p_e, ev_e = run(False)
    p_g, ev_g = run(True)
    assert torch.equal(p_e, p_g)
    for a, b in zip(ev_e, ev_g):
        assert torch.equal(a, b)
    assert float((ev_e[0] - ev_e[2]).abs().max()) > 1e-4          # the weights really moved between the validations


@pytest.mark.gpu
def test_device_state_validation_rounds_do_not_exhaust_the_header_arena():
    """with Trainer(device_state=True) the per-step header arena is handed out from row 0 by train_step only;
    evaluation passes between steps draw from the wrapping ring.  Several validation rounds of many batches (far more than
    8192 / ~36 header rows) must run, and training must continue like a run without the validations (not bit for bit: an
    evaluation pass records the eval-mode maxima of the forward sites, which can move a delayed power-of-two scale)."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D = 8, 20, 6, 32
    margs = default_args(num_layers_enc=3, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=9).items()}

    def run(validate):
        torch.manual_seed(2)
        model = init_model(margs, n_users=5, n_items=5, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        tr = Trainer(model, device_state=True)
        for r in range(3):
            tr.train_step(batch)
            if validate:
                m = tr.valid_model([batch] * 150, permutation=0)
                assert m["HR@10"] >= 0.0
        torch.cuda.synchronize()
        return model._store.flat.detach().clone()

    a, b = run(False), run(True)
    err = (a - b).abs()
    # Adam: an element whose gradient is rounding noise moves by +-lr per step with a rounding-dependent sign
    assert float(err.max()) <= 3 * 2.2e-3 and float(err.median()) <= 1e-6


@pytest.mark.gpu
def test_two_device_state_trainers_coexist():
    """The device-side step state is caller-owned memory named by segmm_step_bind (round 5; it used to be one process-global
    __device__ struct and a second device_state trainer superseded the first): two trainers stepping alternately end with the
    parameters each reaches alone, and their device-side step counts are their own."""
    import torch
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D = 8, 20, 6, 32
    margs = default_args(num_layers_enc=2, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=9).items()}

    def fresh(seed):
        torch.manual_seed(seed)
        m = init_model(margs, n_users=5, n_items=5, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        return m, Trainer(m, device_state=True)
    alone = []
    for seed, n in ((1, 4), (2, 2)):
        m, t = fresh(seed)
        for _ in range(n):
            t.train_step(batch)
        alone.append(m._store.flat.detach().clone())
    m1, t1 = fresh(1)
    m2, t2 = fresh(2)          # (draws its dropout seed after t1's: the same draws as in the runs above, seeded per trainer)
    t1.train_step(batch); t2.train_step(batch); t1.train_step(batch); t1.train_step(batch); t2.train_step(batch); t1.train_step(batch)
    torch.cuda.synchronize()
    assert torch.equal(m1._store.flat, alone[0]) and torch.equal(m2._store.flat, alone[1])
    H.step_bind(t1._step_state)
    assert H.step_get()[1] == 4
    H.step_bind(t2._step_state)
    assert H.step_get()[1] == 2


@pytest.mark.gpu
def test_attention_input_modes_agree():
    """SEGMM_ATT_PL (engine ``attn_pl``): 0 = the attention kernels read the fp32 views of the projection outputs (round 4),
    1 = the projection GEMMs write planes ONLY and both attention kernels read them (default), 2 = planes beside fp32, forward
    only.  Different kernels, same mathematics: with lr = 0 (identical parameters throughout) the loss and every gradient of the
    fourth step -- sites calibrated, the planes paths active -- agree within 2e-5 of each tensor's maximum."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D = 16, 20, 8, 64
    margs = default_args(num_layers_enc=3, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batches = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=40 + i).items()} for i in range(2)]
    res = {}
    for mode in (0, 1, 2):
        torch.manual_seed(5)
        m = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        t = Trainer(m, lr=0.0, weight_decay=0.0, device_state=True)
        m._store.attn_pl = mode
        for i in range(4):
            out = t.train_step(batches[i % 2])
        torch.cuda.synchronize()
        res[mode] = (float(out["loss"].detach()), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    gmax = max(float(g.abs().max()) for g in res[0][1].values())          # (floor: gradients that are zero in exact arithmetic)
    for mode in (1, 2):
        assert abs(res[mode][0] - res[0][0]) <= 1e-5 * max(1.0, abs(res[0][0])), mode
        for k, g0 in res[0][1].items():
            assert float((res[mode][1][k] - g0).abs().max()) <= 2e-5 * float(g0.abs().max()) + 1e-6 * gmax, (mode, k)


@pytest.mark.gpu
def test_planes_only_input_needs_plane_consumers_d48():
    """D_in a multiple of 32 but d_model = 48 (3 heads of 16): the embedding weight gradient takes the on-the-fly
    kernel, which reads the fp32 copy of the L1-normalised features -- the planes-only input protocol must stay off, and
    several default-settings training steps must run and match the oracle's first-step gradients."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D_in, d = 8, 20, 6, 64, 48
    margs = default_args(num_layers_enc=2, d_model=d, nhead=3, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D_in, seed=11).items()}
    torch.manual_seed(4)
    model = init_model(margs, n_users=5, n_items=5, input_dim=D_in, max_vid_len=S, max_usr_len=Lt).to(dev)
    tr = Trainer(model, dropout=False)
    losses = [float(tr.train_step(batch)["loss"].detach()) for _ in range(4)]          # raised on step 2 before the fix
    assert all(l == l for l in losses) and losses[-1] < losses[0]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N,B,S,Lt,D,h", [("image", 3, 64, 40, 10, 64, 4), ("id", 4, 64, 20, 1, 64, 4), ("image", 2, 32, 40, 100, 96, 2),
                                               ("both", 2, 16, 40, 10, 64, 4)])
def test_recorded_step_equals_eager_device_state_step(kind, N, B, S, Lt, D, h):
    """Trainer.record / run_recorded (include/segmm_hip.h "Recorded launch sequences"; the loop body of
    main_for_seq_leave_earlystop_SegMM.py:265-300 enqueued by one C call per phase: segmm_step_begin, segmm_embed_fwd,
    segmm_layer_fwd, segmm_head_loss_fwd, segmm_head_loss_bwd, segmm_layer_bwd, segmm_embed_bwd, segmm_step_tail): 20 steps on
    rotating batches with dropout ON leave BIT-IDENTICAL parameters, optimizer moments and losses to the same steps enqueued
    launch by launch from Python in the same mode; an evaluation between recorded steps sees the current weights."""
    import torch
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)
    batches = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, n_users=50, n_items=500, seed=300 + i).items()} for i in range(4)]
    T = 20

    def run(recorded):
        torch.manual_seed(7)
        model = init_model(margs, n_users=50, n_items=500, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        tr = Trainer(model, lr=1e-3, weight_decay=1e-4, device_state=True)
        losses, evals = [], []
        timed = ([], [])
        if recorded:
            tr.record(batches[0], warmup=3)          # 3 eager steps + the recorded one
        else:
            for _ in range(4):
                tr.train_step(batches[0])
        for t in range(T):
            # (steps 8 and 15 of the recorded run are enqueued from Python: the two ways of stepping can be mixed)
            # (... and steps 3 and 12 are TIMED replays: the phases cut at every GEMM / attention command with an event pair around it)
            out = (tr.run_recorded(batches[t % 4], timed=timed if t in (3, 12) else None) if (recorded and t not in (8, 15))
                   else tr.train_step(batches[t % 4]))
            losses.append(float(out["loss"].detach()))
            if t in (5, 11):
                evals.append(tr.eval_step(batches[3], mode="inference")["logits"].detach().clone())
        torch.cuda.synchronize()
        seed, step, _ = H.step_get()
        assert tr.opt.step_count == step == 4 + T
        if recorded:
            assert all(a is not None for _, a in tr._recorded["phases"])          # single GPU: C phases only, no host action
            kinds = [ph.kind for ph, _ in tr._recorded["phases"]]
            assert kinds[0] == H.PHASE_STEP_BEGIN and kinds[-1] == H.PHASE_STEP_TAIL and H.PHASE_LAYER_FWD in kinds and H.PHASE_LAYER_BWD in kinds
            assert len(tr._recorded["relocs"]) >= 3
            tg, ta = timed          # two timed steps: every GEMM (layout, M, N, K) and attention launch of the step, with live events
            assert len(tg) >= 2 * 10 and len(tg) % 2 == 0 and len(ta) >= 2 * 2 and len(ta) % 2 == 0
            assert tg[:len(tg) // 2] == [] or [g[:4] for g in tg[:len(tg) // 2]] == [g[:4] for g in tg[len(tg) // 2:]]
            assert all(g[4].elapsed_time(g[5]) > 0 for g in tg) and all(a[7].elapsed_time(a[8]) > 0 for a in ta)
            assert {a[0] for a in ta} <= {"fwd", "bwd", "bwd4", "bwd4r", "bwd1", "bwd2", "bwd3"} and "fwd" in {a[0] for a in ta}
        return model._store.flat.detach().clone(), tr.opt.m.clone(), tr.opt.v.clone(), losses, evals, seed

    pe, me, ve, le, ee, se = run(False)
    pr, mr, vr, lr_, er, sr = run(True)
    assert se == sr and le == lr_ and len(set(le)) > 1
    assert torch.equal(pe, pr) and torch.equal(me, mr) and torch.equal(ve, vr)
    for a, b in zip(ee, er):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["learnable_bias", "focal_first", "noUser", "noUser_SelfAtt", "noPos"])
def test_recorded_step_covers_every_step_variant_of_the_reference(variant):
    """Round 5: record() accepts the variants it used to refuse because they kept torch ops or host draws inside the step --
    learnable_bias (decoder_leave_focal.py:497-504,649-658: segmm_bias_grad), focal loss first in the list (:534-535: labels rewritten
    in place by segmm_focal_relabel), the noUser ablations (main...SegMM.py:275-280) and noPos (encoder.py:428-429), whose random
    inputs are drawn on the device from the step's own state (the reference's distributions; another bit stream than torch's).
    In the device-state mode the eager step makes the same draws: recorded and eager steps leave BIT-IDENTICAL parameters."""
    import torch
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D, N, h = 32, 40, 8, 64, 3, 4
    kind = "id" if variant == "noPos" else "image"
    over = dict(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)
    if variant == "learnable_bias":
        over["learnable_bias"] = 1
    elif variant == "focal_first":
        over["loss_type_list"] = ["focal", "interestBPR"]
    else:
        over["ablation_type"] = variant
    margs = default_args(**over)
    base = [make_batch(B, S, 1 if kind == "id" else Lt, D, n_users=50, n_items=500, seed=700 + i, features=kind != "id") for i in range(3)]

    def run(recorded):
        torch.manual_seed(11)
        batches = [{k: v.to(dev).clone() for k, v in b.items()} for b in base]          # (focal rewrites the labels in place)
        model = init_model(margs, n_users=50, n_items=500, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        tr = Trainer(model, lr=1e-3, weight_decay=1e-4, device_state=True)
        losses = []
        if recorded:
            tr.record(batches[0], warmup=2)
        else:
            for _ in range(3):
                tr.train_step(batches[0])
        for t in range(9):
            out = tr.run_recorded(batches[t % 3]) if recorded else tr.train_step(batches[t % 3])
            losses.append(float(out["loss"].detach()))
        torch.cuda.synchronize()
        return model._store.flat.detach().clone(), losses, [b["label"].clone() for b in batches]

    pe, le, ge = run(False)
    pr, lr_, gr = run(True)
    assert torch.isfinite(pe).all() and le == lr_ and len(set(le)) > 1
    assert torch.equal(pe, pr)
    for a, b in zip(ge, gr):
        assert torch.equal(a, b)
    if variant == "focal_first":          # the labels were rewritten: no value above 1, no -1 left
        assert int(ge[0].max()) <= 1 and int((ge[0] == -1).sum()) == 0


@pytest.mark.gpu
def test_device_draws_have_the_reference_distributions():
    """segmm_rand_uniform / segmm_rand_ids / segmm_rand_perm_rows (the device-side stand-ins for torch.rand_like, torch.randint and
    torch.randperm in recorded steps): uniform on [0, 1), uniform on [lo, hi), a uniformly random permutation per row."""
    import torch
    from segmminterest_amd import hipabi as H
    dev = torch.device("cuda:0")
    u = torch.empty(1 << 20, device=dev)
    H.rand_uniform(u, 12345, 7)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    u2 = torch.empty_like(u)
    H.rand_uniform(u2, 12346, 7)
    assert not torch.equal(u, u2) and abs(float(((u - 0.5) * (u2 - 0.5)).mean())) < 1e-3
    ids = torch.empty(1 << 18, dtype=torch.int64, device=dev)
    H.rand_ids(ids, 1, 50, 99, 3)
    assert int(ids.min()) == 1 and int(ids.max()) == 49
    cnt = torch.bincount(ids, minlength=50)[1:].float()
    assert float((cnt / cnt.mean() - 1).abs().max()) < 0.06
    rows, S = 4096, 40
    perm = torch.empty(rows, S, device=dev)
    H.rand_perm_rows(perm, rows, S, 777, 5)
    assert torch.equal(perm.sort(1).values, torch.arange(S, device=dev, dtype=torch.float32).expand(rows, S))
    first = torch.bincount(perm[:, 0].long(), minlength=S).float()          # every value equally likely in every position
    assert float((first / first.mean() - 1).abs().max()) < 0.4 and len({tuple(r.tolist()) for r in perm[:64]}) == 64


@pytest.mark.gpu
def test_user_embedding_as_planes_only_gives_the_same_step():
    """Round 5: in the trainer's own step of an N = 2 model the user embedding exists as P32 planes only -- its LayerNorm writes them
    with the scale of the output BOUND (segmm_layernorm_fwd, y = NULL), the fused user projection and its weight gradient read them
    without an fp32 fallback (encoder.py:462-471 feeds encoder.py:95-104).  The step's loss and every gradient agree with the
    step that keeps the fp32 embedding and its delayed-scale planes to the accuracy of the 22-bit operands; an outlier LayerNorm
    gain (x 200 on one column) moves the bound, not the result."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D, h = 64, 40, 100, 256, 8
    margs = default_args(num_layers_enc=2, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, n_users=50, n_items=500, seed=77).items()}

    def run(flag, outlier):
        torch.manual_seed(3)
        model = init_model(margs, n_users=50, n_items=500, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
        if outlier:
            with torch.no_grad():
                model.backbone1.usr_ln.weight[5] *= 200.0
        tr = Trainer(model, lr=1e-3, weight_decay=1e-4, dropout=False)
        model._store.eu_planes_only = flag
        outs = []
        for _ in range(3):          # step 1 calibrates the sites, steps 2 and 3 run on producer-written planes
            o = tr.train_step(batch)
            outs.append((float(o["loss"].detach()), model._store.gflat.detach().clone()))
        return outs

    for outlier in (False, True):
        ref, got = run(False, outlier), run(True, outlier)
        for (l0, g0), (l1, g1) in zip(ref, got):
            assert abs(l0 - l1) <= 2e-5 * max(abs(l0), 1e-3), (l0, l1)
            assert float((g0 - g1).abs().max()) <= 3e-5 * float(g0.abs().max())
        assert torch.isfinite(got[-1][1]).all()


@pytest.mark.gpu
def test_record_refuses_what_a_replay_would_drop():
    """a recorded step replays C-ABI launches only, so record() must refuse every input whose handling needs a torch
    kernel inside the step -- uint8 / float masks, int32 ids, fp16 features -- instead of freezing the record-time result; the
    torch fallbacks themselves raise while a step is being recorded (argsort of ids that are not int64); a change of the
    optimizer's hyperparameters after record() makes run_recorded() refuse (they are recorded by value); and the relocation
    table is built from pointer slots only."""
    import torch
    from segmminterest_amd import engine as E, hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D = 16, 40, 10, 64
    margs = default_args(num_layers_enc=2, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, n_users=50, n_items=500, seed=1).items()}
    torch.manual_seed(3)
    model = init_model(margs, n_users=50, n_items=500, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
    tr = Trainer(model, lr=1e-3, weight_decay=1e-4, device_state=True)
    for key, bad in (("photo_mask", batch["photo_mask"].to(torch.uint8)), ("user_mask", batch["user_mask"].float()),
                     ("photo_identity_id", batch["photo_identity_id"].to(torch.int32)), ("photo", batch["photo"].half())):
        b2 = dict(batch)
        b2[key] = bad
        with pytest.raises(RuntimeError, match="record\\(\\)"):
            tr.record(b2, warmup=1)
        assert H.RECORDER is None
    tr.train_step(batch)          # the refused record() calls left the trainer usable
    # the fallbacks announce themselves while a recorder is installed
    H.RECORDER = H.Recorder(H._stream(), 0)
    try:
        with pytest.raises(RuntimeError, match="argsort"):
            E._argsort_ids(torch.arange(H.ARGSORT_MAX + 8, device=dev, dtype=torch.int32))          # (int64 ids of any count: library kernels)
        with pytest.raises(RuntimeError, match="mask"):
            E._mask_u8(torch.ones(4, 4, dtype=torch.uint8, device=dev))
        assert E._mask_u8(torch.ones(4, 4, dtype=torch.bool, device=dev)).dtype == torch.uint8          # zero-copy: fine
    finally:
        H.RECORDER = None
    big = torch.randint(0, 5000, (2 * H.ARGSORT_MAX + 8,), device=dev)          # beyond one workgroup: the multi-workgroup network, no torch kernel
    assert torch.equal(E._argsort_ids(big, model._store).long(), torch.argsort(big, stable=True))
    tr.record(batch, warmup=2)
    r = tr._recorded
    kinds = set()
    for arr, ci, ai, k, off, lag in r["relocs"]:
        kinds.add(k)
        assert arr[ci].a[ai].p == (batch if not lag else r["prev_batch"])[k].data_ptr() + off
    assert {"photo", "user", "label"} <= kinds
    tr.run_recorded(batch)
    tr.opt.lr = 5e-4          # an LR schedule / a resumed checkpoint: the recorded AdamW launches carry the old value
    with pytest.raises(RuntimeError, match="hyperparameters"):
        tr.run_recorded(batch)
    tr.record(batch, warmup=1)
    tr.run_recorded(batch)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_adamw_table_two_passes_equal_dense_update():
    """segmm_adamw_table (main_for_seq_leave_earlystop_KuaiRand.py:259-261 / optimizer.step at :299 over an nn.Embedding table,
    encoder.py:352-362): phase 0 (rows without a gradient, g = 0, run early) + phase 1 (the batch's rows) leave BIT-IDENTICAL
    parameters and moments to one dense segmm_adamw launch over the table; duplicate and out-of-range ids, flags left zero."""
    import torch
    from segmminterest_amd import hipabi as H
    H.lib()
    dev = "cuda"
    g0 = torch.Generator().manual_seed(3)
    rows, width, off = 5000, 64, 96
    n = rows * width
    flat = torch.randn(off + n + 32, generator=g0).to(dev)
    m = (0.01 * torch.randn(off + n + 32, generator=g0)).to(dev)
    v = (1e-4 * torch.rand(off + n + 32, generator=g0)).to(dev)
    ids = torch.randint(0, rows, (300,), generator=g0)
    ids[:20] = ids[20:40]                                  # duplicates
    ids = torch.cat([ids, torch.tensor([-1, rows, rows + 7])]).to(dev)          # ignored
    grad = torch.zeros(off + n + 32, device=dev)
    gt = grad[off:off + n].view(rows, width)
    valid = ids[(ids >= 0) & (ids < rows)]
    gt[valid] = torch.randn(valid.numel(), width, generator=g0).to(dev)
    for step in (1, 7):
        p1, m1, v1 = flat.clone(), m.clone(), v.clone()
        H.adamw(p1, grad, m1, v1, n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step, p_off=off)
        p2, m2, v2 = flat.clone(), m.clone(), v.clone()
        flags = torch.zeros(rows, dtype=torch.int32, device=dev)
        H.adamw_table(p2, None, m2, v2, off, rows, width, ids, flags, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step, 0)
        assert int(flags.sum()) == int(valid.unique().numel())
        H.adamw_table(p2, grad, m2, v2, off, rows, width, ids, flags, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step, 1)
        assert int(flags.abs().sum()) == 0
        assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.gpu
@pytest.mark.parametrize("device_state", [False, True])
def test_id_table_two_pass_step_equals_dense_step(device_state):
    """Trainer in id mode: the item table's optimizer step split in two (rows outside the batch early, on the auxiliary stream;
    the batch's rows after the backward) against the one-launch dense AdamW (SEGMM_TABLE_TWO_PASS=0): bit-identical parameters
    and moments after 6 steps on rotating batches, eager and as recorded launch sequences."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, D, N, h = 64, 20, 64, 3, 4
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "id", "photo": "id"}, exposure_prob=[1.0] * S)
    batches = [{k: v.to(dev) for k, v in make_batch(B, S, 1, D, n_users=50, n_items=3000, seed=400 + i, features=False).items()} for i in range(3)]

    def run(two_pass):
        os.environ["SEGMM_TABLE_TWO_PASS"] = "1" if two_pass else "0"
        try:
            torch.manual_seed(9)
            model = init_model(margs, n_users=50, n_items=3000, input_dim=D, max_vid_len=S, max_usr_len=1).to(dev)
            tr = Trainer(model, device_state=device_state)
            assert tr.table_two_pass == two_pass
            if device_state:
                tr.record(batches[0], warmup=2)
            else:
                for _ in range(3):
                    tr.train_step(batches[0])
            for t in range(6):
                (tr.run_recorded if device_state else tr.train_step)(batches[t % 3])
            torch.cuda.synchronize()
            if two_pass:
                assert tr.opt._table_flags and all(int(f.abs().sum()) == 0 for f in tr.opt._table_flags.values())
            return model._store.flat.detach().clone(), tr.opt.m.clone(), tr.opt.v.clone()
        finally:
            os.environ.pop("SEGMM_TABLE_TWO_PASS", None)

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("device_state", [False, True])
def test_step_schedule_knobs_do_not_change_the_step(device_state):
    """Round 4's schedule changes only move launches between streams: the head of the step on two streams (SEGMM_BEGIN_OVERLAP),
    the layer's weight gradients beside the attention backward (SEGMM_DEFER_WGRAD) and the LayerNorm column sums on the side
    stream (SEGMM_LN_SIDE) -- on for segment axes > 32 -- must leave parameters and moments BIT-identical to the serial
    schedule, eager and as recorded launch sequences (config-2-like image / image model, S = 40, dropout on)."""
    import torch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, Lt, D, N, h = 12, 40, 20, 64, 3, 4
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batches = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=700 + i).items()} for i in range(3)]
    knobs = ("SEGMM_BEGIN_OVERLAP", "SEGMM_DEFER_WGRAD", "SEGMM_LN_SIDE", "SEGMM_LAZY_HEAD_GRAD", "SEGMM_HEAD_DOT")          # (the last two: the
    # head's gradient formed inside the first LayerNorm backward instead of written out, its logits inside the last LayerNorm forward
    # instead of by a pass over the output -- same products, same summation order)

    def run(on):
        for k in knobs:
            os.environ[k] = "1" if on else "0"
        try:
            torch.manual_seed(21)
            model = init_model(margs, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
            tr = Trainer(model, device_state=device_state)
            assert tr.begin_overlap == on and model._store.defer_wgrad == on and model._store.ln_side == on and model._store.lazy_head_grad == on and model._store.head_dot == on
            if device_state:
                tr.record(batches[0], warmup=2)
            else:
                for _ in range(3):
                    tr.train_step(batches[0])
            for t in range(5):
                (tr.run_recorded if device_state else tr.train_step)(batches[t % 3])
            torch.cuda.synchronize()
            return model._store.flat.detach().clone(), tr.opt.m.clone(), tr.opt.v.clone()
        finally:
            for k in knobs:
                os.environ.pop(k, None)

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)



@pytest.mark.gpu
def test_failed_step_after_the_early_table_pass_leaves_the_step_counts_aligned():
    """a step that fails between the early table pass and opt.step() (here: a forward that raises) has advanced the
    DEVICE step count but not the host's.  The next step must run with the right count again: after one failed and two good steps
    the device count equals opt.step_count, and the parameters of the rows that DID get gradients follow the bias corrections of
    steps 1 and 2 (they equal a trainer that never failed, except for the one extra weight-decay pass on the rows outside the
    failed batch, which is documented as accepted)."""
    import torch
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    dev = torch.device("cuda:0")
    B, S, D, N, h = 32, 20, 64, 2, 4
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "id", "photo": "id"}, exposure_prob=[1.0] * S)
    batches = [{k: v.to(dev) for k, v in make_batch(B, S, 1, D, n_users=50, n_items=2000, seed=500 + i, features=False).items()} for i in range(3)]

    def build():
        torch.manual_seed(4)
        model = init_model(margs, n_users=50, n_items=2000, input_dim=D, max_vid_len=S, max_usr_len=1).to(dev)
        return model, Trainer(model, device_state=True, dropout=False)

    model, tr = build()
    if not tr.table_two_pass:
        pytest.skip("two-pass table update off")
    orig = tr._train_step

    def boom(*a, **k):
        raise RuntimeError("injected forward failure")
    tr._train_step = boom
    with pytest.raises(RuntimeError, match="injected"):
        tr.train_step(batches[0])
    tr._train_step = orig
    assert tr.opt.step_count == 0
    tr.train_step(batches[1])
    tr.train_step(batches[2])
    torch.cuda.synchronize()
    H.step_bind(tr._step_state)
    _, dev_step, _ = H.step_get()
    assert tr.opt.step_count == 2 and dev_step == 2, (tr.opt.step_count, dev_step)
    # a trainer that never failed: the same parameters up to what the one extra g = 0 pass over the item table (weight decay
    # lr * wd = 1e-7 relative on rows outside the failed batch) propagates into the two steps -- far below one lr step (1e-3);
    # with the device count left at 3 instead of 2 the bias corrections would move every parameter by ~30 % of a step
    model2, tr2 = build()
    tr2.train_step(batches[1])
    tr2.train_step(batches[2])
    torch.cuda.synchronize()
    st, st2 = model._store, model2._store
    for n in st.live_names:
        if st._params[n].dim() < 2 or n.startswith("stage_mlp"):          # (shift-invariant / near-zero-gradient parameters: the sign of a ~0 gradient moves them by +-lr)
            continue
        o, k = st.index[n]
        assert float((st.flat[o:o + k] - st2.flat[o:o + k]).abs().max()) < 1e-4, n
