#!/usr/bin/env python
"""Golden vectors for the data formats either side of the hot path (TEST INFRASTRUCTURE ONLY).

Runs only in the build container, where /root/reference exists.  It drives the reference's OWN code on small synthetic
inputs and writes inputs + expected outputs (no reference source) to ``tests/golden/``:

  io_dataloader.npz  ``FrameDatasetSeq_SegMM._getitem`` (MMinterest/utils/dataloader_SegMM.py:271-362, with
                     ``_pad_feature_list`` :251-268, ``_pad_label_list`` :240-249, ``_calculate_frame_ids`` :217-219) over a
                     synthetic corpus: history flatten over watched segments, cap 100 (``random.sample``), more than 40
                     video frames (``np.random.choice``), keys missing from the line map, empty history.
  io_reader.json     ``GeneralModel.Dataset._get_feed_dict`` (SegRec/models/BaseModel.py:228-288): which
                     ``c_interest_weight`` slices a batch row receives from the logits file written by
                     MMinterest/inference/save_logits_for_all_leave_SegMM.py.
  io_cliprec.npz     ``ClipRecBase.forward`` (SegRec/models/context/ClipRec.py:134-198): per-clip predictions (recovered as
                     the gradient of the prediction w.r.t. the interest weights with the duration mask off), interest
                     weights, durations and the weighted predictions with the duration mask on.

    python oracle/gen_golden_io.py
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SEGMM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def gen_dataloader():
    import importlib.util
    import pandas as pd
    if not hasattr(np, "int"):
        np.int = int                      # dataloader_SegMM.py:357 uses the alias numpy removed in 1.24
    spec = importlib.util.spec_from_file_location("ref_dataloader_SegMM", os.path.join(REF, "MMinterest", "utils", "dataloader_SegMM.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    DS = mod.FrameDatasetSeq_SegMM
    rng = np.random.RandomState(7)
    D = 8
    # line map: photos 100..139, each with a number of 5-second frames; some frames deliberately missing
    n_frames = {p: int(rng.randint(1, 13)) for p in range(100, 140)}
    n_frames[105] = 47                    # more than 40 frames: np.random.choice(47, 40, replace=False)
    n_frames[106] = 40
    keys = []
    for p, n in n_frames.items():
        for f in range(n):
            if (p, f) in ((110, 2), (111, 0)):        # holes in the line map (history lookups skip them)
                continue
            keys.append("%d-%d" % (p, f))
    rng.shuffle(keys)
    lineid_map = {k: i for i, k in enumerate(keys)}
    feat = rng.rand(len(keys), D).astype(np.float32)
    users = [11, 12, 13, 14]
    # (a user with NO resolvable frame at all makes the reference raise IndexError in _pad_feature_list -- np.array([]) has
    # no second axis -- so every synthetic user keeps at least one)
    user_input_dict = {"11": ["120_0", "120_1", "121_0", "999_0"], "12": ["122_0"], "13": ["110_2", "130_0"], "14": ["101_0"]}
    user2id = {str(u): i + 1 for i, u in enumerate(users)}
    item2id = {str(p): i + 1 for i, p in enumerate(sorted(n_frames))}

    def lab(n_seg, leave):
        l = [1] * n_seg if leave is None else [1] * leave + [0] + [-1] * (n_seg - leave - 1)
        return "[" + " ".join(str(x) for x in l) + "]"

    rows = []

    def add(user, video, time_ms, playing, hist, leave):
        dur = n_frames[video] * 5000 - 1200           # ceil(dur / 5000) frames
        hi = "[" + " ".join(str(h[0]) for h in hist) + "]"
        hp = "[" + " ".join(str(h[1]) for h in hist) + "]"
        rows.append(dict(user_id=user, video_id=video, time_ms=time_ms, duration_ms=dur, playing_time_x=playing,
                         label_1D=lab(n_frames[video], leave), history_items=hi, history_playing=hp, history_lengths=len(hist)))

    add(11, 101, 1700000001, 9000, [(102, 14000), (103, 5000), (110, 15000)], 1)
    add(12, 105, 1700000002, 230000, [], None)                                         # 47 frames, fully watched, no history
    add(13, 106, 1700000003, 1000, [(111, 9000), (107, 20000)], 0)                     # exactly 40 frames; history with holes
    add(14, 104, 1700000004, 20000, [(p, n_frames[p] * 5000) for p in range(112, 140)], 2)      # > 100 history frames: random.sample
    add(11, 108, 1700000005, 4999, [(109, 1)], None)
    df = pd.DataFrame(rows)
    ds = object.__new__(DS)                 # __init__ reads side files from disk; set the fields it would set
    ds.do_scale_image_to_01 = False
    ds.image_resize = False
    ds.target_hw_shape = None
    ds.shuffle = False
    ds.verbose = False
    ds.photo_max_image, ds.user_max_image = 40, 100
    ds.df = df
    ds.user_input_dict = user_input_dict
    ds.lineid_map = lineid_map
    ds.feat_memmap = feat
    ds.user2id, ds.item2id = user2id, item2id
    random.seed(2024)
    np.random.seed(2024)
    items = list(ds._getitem())
    assert len(items) == len(rows)
    out = dict(table=feat, keys=np.array(keys), seed=np.int64(2024),
               user_input_dict=np.array(json.dumps(user_input_dict)), user2id=np.array(json.dumps(user2id)),
               item2id=np.array(json.dumps(item2id)), rows=np.array(json.dumps(rows)))
    for k in ("photo", "photo_mask", "user", "user_mask", "label"):
        out["exp_" + k] = np.stack([np.asarray(it[k]) for it in items])
    for k in ("photo_id", "photo_identity_id", "user_id", "user_identity_id", "time_ms", "play_time", "duration"):
        out["exp_" + k] = np.array([int(it[k]) for it in items], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "io_dataloader.npz"), **out)
    print("io_dataloader.npz: %d rows, user tokens per row %s" % (len(items), [int(it["user_mask"].sum()) for it in items]))


def _segrec_path():
    p = os.path.join(REF, "SegRec")
    if p not in sys.path:
        sys.path.insert(0, p)


def gen_reader():
    _segrec_path()
    import importlib
    BM = importlib.import_module("models.BaseModel")
    DS = BM.GeneralModel.Dataset
    rng = np.random.RandomState(3)
    clip_weight = {"%d-%d-%d" % (u, i, t): [round(float(x), 4) for x in rng.randn(40)] for (u, i, t) in
                   [(1, 10, 500), (1, 11, 500), (2, 10, 501), (3, 30, 777)]}
    neg_weight = {"%d-%d-%d" % (u, i, t): [round(float(x), 4) for x in rng.randn(40)] for (u, i, t) in
                  [(1, 11, 500), (1, 12, 500), (3, 31, 777), (3, 32, 777)]}
    cases = []

    def run(dataset, user_id, item, neg, time, with_neg_file, id_maps=None):
        ds = object.__new__(DS)
        ds.phase = "test"
        ds.model = types.SimpleNamespace(test_all=False)
        ds.corpus = types.SimpleNamespace(dataset=dataset, n_items=100)
        ds.data = {"user_id": [user_id], "item_id": [item], "time": [time], "neg_items": [np.array(neg)]}
        ds.clip_weight_path = "x"
        ds.clip_weight = clip_weight
        ds.eval_neg_weight_path = "y" if with_neg_file else ""
        ds.clip_neg_weight = neg_weight
        if id_maps:
            ds.id2user, ds.id2item = id_maps
        try:
            fd = ds._get_feed_dict(0)
            w = np.asarray(fd["c_interest_weight"], dtype=np.float64).tolist()
            err = None
        except KeyError as e:
            w, err = None, "KeyError"
        cases.append(dict(dataset=dataset, user_id=user_id, item=item, neg=list(neg), time=time, with_neg_file=with_neg_file,
                          id_maps=id_maps, weights=w, error=err))

    run("KuaiRand_CTR", 1, 10, [11], 500, False)             # key present, training pair: every item gets the TARGET's slice
    run("KuaiRand_CTR", 1, 10, [11, 12], 500, False)
    run("KuaiRand_CTR", 1, 10, [11, 12], 500, True)          # evaluation with a negatives file: own slices for items 1..
    run("KuaiRand_CTR", 3, 30, [31, 33], 777, True)          # a negative is missing from the file -> KeyError
    run("KuaiRand_CTR", 9, 10, [11, 12], 500, False)         # target key absent -> ONE row of ones
    run("KuaiRand_CTR", 9, 10, [11, 12], 500, True)
    ident = ({str(i): i for i in range(50)}, {str(i): i for i in range(50)})
    run("SegMM", 2, 10, [11], 501, False, ident)             # other datasets: ids mapped through id2user / id2item first
    run("SegMM", 2, 13, [11], 501, False, ident)
    json.dump(dict(clip_weight=clip_weight, neg_weight=neg_weight, cases=cases), open(os.path.join(OUT, "io_reader.json"), "w"))
    print("io_reader.json: %d cases" % len(cases))


def gen_cliprec():
    _segrec_path()
    import importlib
    CR = importlib.import_module("models.context.ClipRec")

    class Tiny(torch.nn.Module, CR.ClipRecBase):
        def __init__(self, duration_mask):
            torch.nn.Module.__init__(self)
            self.feature_max = {"user_id": 20, "item_id": 30}
            self.device = torch.device("cpu")
            self.embedding_dim, self.dnn_layers, self.dropout = 8, [16], 0.0
            self.contrastive, self.adjust_interest_weight, self.duration_mask = "", 0, duration_mask
            self._define_params_ClipRec()

    torch.manual_seed(5)
    m0 = Tiny(0)
    with torch.no_grad():
        for p in m0.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    m1 = Tiny(1)
    m1.load_state_dict(m0.state_dict())
    m0.eval()
    m1.eval()
    B, I, Cn = 5, 3, 40
    g = torch.Generator().manual_seed(6)
    feed = {"user_id": torch.randint(1, 20, (B,), generator=g), "item_id": torch.randint(1, 30, (B, I), generator=g),
            "i_item_frames": torch.rand(B, I, Cn, 1024, generator=g) * 0.05,
            "i_duration": torch.randint(0, 45, (B, I), generator=g)}
    w = torch.randn(B, I, Cn, generator=g)
    # per-clip predictions: d prediction / d weight with the duration mask off
    wg = torch.ones(B, I, Cn, requires_grad=True)
    out = CR.ClipRecBase.forward(m0, dict(feed, c_interest_weight=wg))["prediction"]
    out.sum().backward()
    clip_pred = wg.grad.detach().clone()
    with torch.no_grad():
        pred_w = CR.ClipRecBase.forward(m1, dict(feed, c_interest_weight=w))["prediction"]          # weights + duration mask
        pred_plain = CR.ClipRecBase.forward(m1, dict(feed))["prediction"]                            # no weights in the feed: ones
        pred_nomask = CR.ClipRecBase.forward(m0, dict(feed, c_interest_weight=w))["prediction"]
    np.savez_compressed(os.path.join(OUT, "io_cliprec.npz"), clip_pred=clip_pred.numpy(), weight=w.numpy(),
                        duration=feed["i_duration"].numpy(), pred_weighted_masked=pred_w.numpy(), pred_ones_masked=pred_plain.numpy(),
                        pred_weighted_nomask=pred_nomask.numpy())
    print("io_cliprec.npz: clip_pred", tuple(clip_pred.shape))


if __name__ == "__main__":
    gen_dataloader()
    gen_reader()
    gen_cliprec()
