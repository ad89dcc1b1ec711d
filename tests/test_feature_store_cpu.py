"""(f)-1 / (f)-3 pinned to the reference: fixtures written by oracle/gen_golden_io.py from the reference's OWN
``FrameDatasetSeq_SegMM._getitem`` (dataloader_SegMM.py:271-362), ``GeneralModel.Dataset._get_feed_dict``
(SegRec/models/BaseModel.py:228-288) and ``ClipRecBase.forward`` (SegRec/models/context/ClipRec.py:134-198)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from segmminterest_amd.bridge import LogitStore
from segmminterest_amd.feature_store import IndexBatchBuilder, KeyIndex

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture():
    z = np.load(os.path.join(GOLDEN, "io_dataloader.npz"))
    rows = json.loads(str(z["rows"]))
    b = IndexBatchBuilder(KeyIndex([str(k) for k in z["keys"]]), json.loads(str(z["user_input_dict"])),
                          json.loads(str(z["user2id"])), json.loads(str(z["item2id"])))
    return z, rows, b


def _build(z, rows, b):
    random.seed(int(z["seed"]))                  # init_seed of the reference (dataloader_SegMM.py:31-38)
    np.random.seed(int(z["seed"]))
    return b.batch([b.row(r["user_id"], r["video_id"], r["time_ms"], r["duration_ms"], r["playing_time_x"], r["label_1D"],
                          r["history_items"], r["history_playing"], r["history_lengths"]) for r in rows])


def test_index_rows_reproduce_the_reference_dataset():
    z, rows, b = _fixture()
    out = _build(z, rows, b)
    table = torch.from_numpy(z["table"])
    for key, exp_f, exp_m in (("photo", "exp_photo", "exp_photo_mask"), ("user", "exp_user", "exp_user_mask")):
        idx = out[key + "_idx"]
        got = torch.where((idx >= 0)[..., None], table[idx.clamp(min=0)], torch.zeros(()))          # what the gather kernel forms
        assert torch.equal(got, torch.from_numpy(z[exp_f])), key                                   # same rows, same order, zero padding
        assert torch.equal(out[key + "_mask"], torch.from_numpy(z[exp_m])), key
    assert torch.equal(out["label"], torch.from_numpy(z["exp_label"].astype(np.int64)))
    for k in ("photo_id", "photo_identity_id", "user_id", "user_identity_id", "time_ms", "play_time", "duration"):
        assert out[k].tolist() == z["exp_" + k].tolist(), k
    # the cases the fixture was built for are really in it
    assert int(out["user_mask"][3].sum()) == 100 and int(out["photo_mask"][1].sum()) == 40


def test_missing_video_frame_raises_like_the_reference():
    z, rows, b = _fixture()
    r = dict(rows[0], video_id=110, duration_ms=14000)          # frame 110-2 is not in the line map
    with pytest.raises(ValueError):
        b.row(r["user_id"], r["video_id"], r["time_ms"], r["duration_ms"], r["playing_time_x"], r["label_1D"])


def test_label_padding_and_frame_count():
    _, _, b = _fixture()
    assert b.pad_label("[1 1 0 -1]") == [1, 1, 0, -1] + [-2] * 36
    assert b.pad_label("[" + " ".join(["1"] * 45) + "]") == [1] * 40
    assert [b.n_frames(x) for x in (0, 1, 5000, 5001, 14000)] == [0, 1, 1, 2, 3]


def test_reader_rule_matches_segrec():
    d = json.load(open(os.path.join(GOLDEN, "io_reader.json")))

    def store(m):
        st = LogitStore(S=40)
        ks = [[int(x) for x in k.split("-")] for k in m]
        st.add_batch([k[0] for k in ks], [k[1] for k in ks], [k[2] for k in ks], torch.tensor(list(m.values()), dtype=torch.float32))
        return st
    clip, neg = store(d["clip_weight"]), store(d["neg_weight"])
    assert len(d["cases"]) >= 8
    for c in d["cases"]:
        items = np.array([[c["item"]] + c["neg"]])
        maps = dict(id2user=c["id_maps"][0], id2item=c["id_maps"][1]) if c["id_maps"] else {}
        if c["error"]:
            with pytest.raises(KeyError):
                clip.weights([c["user_id"]], items, [c["time"]], neg=neg if c["with_neg_file"] else None, **maps)
            continue
        w = clip.weights([c["user_id"]], items, [c["time"]], neg=neg if c["with_neg_file"] else None, **maps)[0]
        ref = torch.tensor(c["weights"], dtype=torch.float32)
        assert ref.shape[0] in (1, items.shape[1])
        assert torch.equal(w, ref.expand(items.shape[1], 40)), c          # a single row of ones broadcasts over the items


def test_cliprec_fixture_is_the_weighted_masked_sum():
    """The formula the device kernel implements, checked on the reference's numbers: prediction = sum_clip clip_pred * weight *
    (clip < duration) (ClipRec.py:163-181)."""
    z = np.load(os.path.join(GOLDEN, "io_cliprec.npz"))
    cp, w, dur = torch.from_numpy(z["clip_pred"]).double(), torch.from_numpy(z["weight"]).double(), torch.from_numpy(z["duration"])
    mask = (torch.arange(40)[None, None, :] < dur[..., None]).double()
    assert torch.allclose((cp * w * mask).sum(-1), torch.from_numpy(z["pred_weighted_masked"]).double(), atol=2e-5)
    assert torch.allclose((cp * mask).sum(-1), torch.from_numpy(z["pred_ones_masked"]).double(), atol=2e-5)
    assert torch.allclose((cp * w).sum(-1), torch.from_numpy(z["pred_weighted_nomask"]).double(), atol=2e-5)
