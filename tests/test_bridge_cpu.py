"""Inference -> SegRec bridge: the LogitStore reproduces the reference writer's JSON
(inference/save_logits_for_all_leave_SegMM.py:131-146) and the reader's lookup rule (BaseModel.py:259-288)."""
import json

import numpy as np
import torch

from segmminterest_amd.bridge import LogitStore


def _reference_writer(batches):
    """The loop of save_logits_for_all_leave_SegMM.py:131-136, restated."""
    test_logits = {}
    for uid, pid, tms, logits in batches:
        for u, p, t, logit in zip(uid.tolist(), pid.tolist(), tms.tolist(), logits):
            test_logits[f"{u}-{p}-{t}"] = logit.cpu().detach().tolist()
    return test_logits


def _batches():
    g = torch.Generator().manual_seed(3)
    out = []
    for _ in range(3):
        uid = torch.randint(1, 50, (16,), generator=g)
        pid = torch.randint(1, 30, (16,), generator=g)
        tms = torch.randint(10 ** 12, 10 ** 12 + 5, (16,), generator=g)
        out.append((uid, pid, tms, torch.randn(16, 40, generator=g)))
    return out


def test_json_matches_reference_writer(tmp_path):
    bs = _batches()
    st = LogitStore(S=40)
    for b in bs:
        st.add_batch(*b)
    ref = _reference_writer(bs)
    assert st.as_dict() == ref                       # same keys, same python floats, later duplicates win
    p = tmp_path / "logits.json"
    st.save_json(p)
    assert json.load(open(p)) == json.loads(json.dumps(ref))


def test_binary_roundtrip_and_lookup(tmp_path):
    bs = _batches()
    st = LogitStore(S=40)
    for b in bs:
        st.add_batch(*b)
    p = tmp_path / "logits.npz"
    st.save_binary(p)
    st2 = LogitStore.load(p)
    ref = _reference_writer(bs)
    assert st2.as_dict() == ref
    st3 = LogitStore.load(_write_json(tmp_path, ref))
    assert st3.as_dict() == ref
    # reader rule (BaseModel.py:228-288, pinned by tests/test_feature_store_cpu.py::test_reader_rule_matches_segrec):
    # target key present -> EVERY item of the row gets the target's slice; absent -> ones
    uid = np.array([int(k.split("-")[0]) for k in list(ref)[:5]] + [9999])
    tms = np.array([int(k.split("-")[2]) for k in list(ref)[:5]] + [1])
    items = np.array([[int(k.split("-")[1]), 12345] for k in list(ref)[:5]] + [[1, 2]])
    w = st2.weights(uid, items, tms)
    assert w.shape == (6, 2, 40)
    for i, k in enumerate(list(ref)[:5]):
        assert torch.equal(w[i, 0], torch.tensor(ref[k], dtype=torch.float32))
        assert torch.equal(w[i, 1], w[i, 0])
    assert torch.equal(w[5], torch.ones(2, 40))


def _write_json(tmp_path, d):
    p = tmp_path / "ref.json"
    with open(p, "w") as f:
        json.dump(d, f)
    return p
