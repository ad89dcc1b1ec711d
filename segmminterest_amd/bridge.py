"""Inference -> SegRec bridge.

``inference/save_logits_for_all_leave_SegMM.py:105-150`` runs the interest model over every split and writes
``{"<user_id>-<photo_id>-<time_ms>": [S logits]}`` as JSON (and a torch pickle); SegRec reads that file as
``clip_weight`` and feeds ``feed_dict['c_interest_weight']`` ([batch, item_num, 40]: the TARGET item's slice for every item of
the row, ones when the target's key is missing, own slices for the negatives when a negatives file is given --
SegRec/models/BaseModel.py:228-288), which ``ClipRec.forward`` multiplies into the per-segment predictions
(``sum_seg pred * weight * (seg < duration)``, ClipRec.py:163-181).

``LogitStore`` keeps the same mapping as three int64 key columns + one float32 [n, S] matrix, writes the reference's
JSON byte-for-byte (same key format, ``json.dump`` of python floats converted from fp32 like ``tensor.tolist()``) and a
binary ``.npz`` that loads without parsing 40 floats per row from text; ``weights`` answers a whole batch of lookups
with one vectorised search and ``weighted_head`` is the device kernel for the ClipRec sum."""
from __future__ import annotations

import json
from typing import Optional

import numpy as np
import torch


class LogitStore:
    def __init__(self, S: int = 40):
        self.S = S
        self._keys = []          # list of [n_i, 3] int64 blocks
        self._vals = []          # list of [n_i, S] float32 blocks
        self._index = None

    # ---- writer side (one call per inference batch)
    def add_batch(self, user_id, photo_id, time_ms, logits):
        k = np.stack([np.asarray(torch.as_tensor(v).cpu(), dtype=np.int64).reshape(-1) for v in (user_id, photo_id, time_ms)], 1)
        v = torch.as_tensor(logits).detach().float().cpu().numpy().reshape(k.shape[0], -1)
        if v.shape[1] != self.S:
            raise ValueError("logits have %d segments, store expects %d" % (v.shape[1], self.S))
        self._keys.append(k)
        self._vals.append(v)
        self._index = None

    def _cat(self):
        if len(self._keys) > 1:
            self._keys, self._vals = [np.concatenate(self._keys, 0)], [np.concatenate(self._vals, 0)]
        if not self._keys:
            return np.zeros((0, 3), np.int64), np.zeros((0, self.S), np.float32)
        return self._keys[0], self._vals[0]

    def as_dict(self):
        """The reference's in-memory form; later duplicates of a key overwrite earlier ones, like its dict assignment."""
        k, v = self._cat()
        return {"%d-%d-%d" % (int(a), int(b), int(c)): [float(x) for x in row] for (a, b, c), row in zip(k, v)}

    def save_json(self, path):
        with open(path, "w") as fw:
            json.dump(self.as_dict(), fw)

    def save_binary(self, path):
        k, v = self._cat()
        np.savez(path, keys=k, logits=v, S=np.int64(self.S))

    # ---- reader side
    @classmethod
    def load(cls, path):
        if str(path).endswith(".json"):
            with open(path) as f:
                d = json.load(f)
            st = cls(S=len(next(iter(d.values()))) if d else 40)
            if d:
                keys = np.array([[int(x) for x in key.split("-")] for key in d], dtype=np.int64)
                st._keys, st._vals = [keys], [np.array(list(d.values()), dtype=np.float32)]
            return st
        z = np.load(path)
        st = cls(S=int(z["S"]))
        st._keys, st._vals = [z["keys"]], [z["logits"]]
        return st

    def _build_index(self):
        k, v = self._cat()
        # last occurrence of a key wins (dict semantics): stable sort, keep the last of each run
        order = np.lexsort((np.arange(len(k)), k[:, 2], k[:, 1], k[:, 0]))
        ks = k[order]
        last = np.ones(len(ks), bool)
        if len(ks) > 1:
            last[:-1] = (ks[1:] != ks[:-1]).any(1)
        self._index = (ks[last], order[last])

    def _lookup(self, q):
        """rows of the value matrix for the [n, 3] keys ``q`` (-1 where absent)"""
        if self._index is None:
            self._build_index()
        ks, rows = self._index
        out = np.full((q.shape[0],), -1, dtype=np.int64)
        if len(ks):
            dt = np.dtype([("a", np.int64), ("b", np.int64), ("c", np.int64)])          # lexicographic (user, item, time)
            kv = np.ascontiguousarray(ks).view(dt).reshape(-1)
            qv = np.ascontiguousarray(q).view(dt).reshape(-1)
            pos = np.searchsorted(kv, qv)
            pos_c = np.minimum(pos, len(kv) - 1)
            hit = (pos < len(kv)) & (kv[pos_c] == qv)
            out[hit] = rows[pos_c[hit]]
        return out

    def weights(self, user_id, item_ids, time_ms, neg: "Optional[LogitStore]" = None, id2user=None, id2item=None, device=None):
        """``feed_dict['c_interest_weight']`` of a batch exactly as ``GeneralModel.Dataset._get_feed_dict`` builds it
        (SegRec/models/BaseModel.py:228-288): user_id [B], item_ids [B, I] (column 0 = the target item), time_ms [B] ->
        float32 [B, I, S].  Per row, with key = "<user>-<item 0>-<time>":
          * key absent                      -> ones (the reference appends ONE row of ones, which broadcasts over the items);
          * key present, no negatives file  -> EVERY item of the row gets the TARGET's slice;
          * key present, negatives file ``neg`` and I > 2 -> item 0 the target's slice, item j its own slice from ``neg``
            (KeyError if absent, like the reference).
        ``id2user`` / ``id2item`` (dicts keyed by str): the id mapping the non-KuaiRand branch applies first (:268,275)."""
        u = np.asarray(user_id, np.int64).reshape(-1)
        it = np.asarray(item_ids, np.int64)
        t = np.asarray(time_ms, np.int64).reshape(-1)
        if id2user is not None:
            u = np.asarray([int(id2user[str(int(x))]) for x in u], np.int64)
        if id2item is not None:
            it = np.asarray([[int(id2item[str(int(x))]) for x in r] for r in it], np.int64)
        B, I = it.shape
        _, v = self._cat()
        first = self._lookup(np.stack([u, it[:, 0], t], 1))
        out = np.ones((B, I, self.S), np.float32)
        hit = first >= 0
        if hit.any():
            out[hit] = v[first[hit]][:, None, :]
            if neg is not None and I > 2:
                _, nv = neg._cat()
                q = np.stack([np.broadcast_to(u[:, None], (B, I - 1)), it[:, 1:], np.broadcast_to(t[:, None], (B, I - 1))], -1)
                rows = neg._lookup(q.reshape(-1, 3)).reshape(B, I - 1)
                bad = hit[:, None] & (rows < 0)
                if bad.any():
                    b_, j_ = np.argwhere(bad)[0]
                    raise KeyError("Inference, Key %d-%d-%d not found in clip_weight" % (u[b_], it[b_, j_ + 1], t[b_]))
                out[hit, 1:] = nv[rows[hit]]
        w = torch.from_numpy(out)
        return w.to(device) if device is not None else w


def weighted_head(pred: torch.Tensor, weight: Optional[torch.Tensor] = None, duration: Optional[torch.Tensor] = None):
    """ClipRec.forward's ``(clip_predictions * interest_weight * mask).sum(-1)`` (ClipRec.py:163-181) on the device."""
    from . import hipabi as H
    return H.segment_weighted_sum(pred, weight, duration)
