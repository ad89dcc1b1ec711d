"""Inference -> SegRec bridge.

``inference/save_logits_for_all_leave_SegMM.py:105-150`` runs the interest model over every split and writes
``{"<user_id>-<photo_id>-<time_ms>": [S logits]}`` as JSON (and a torch pickle); SegRec reads that file as
``clip_weight`` and feeds ``feed_dict['c_interest_weight']`` ([batch, item_num, 40], missing keys -> ones,
SegRec/models/BaseModel.py:262-408), which ``ClipRec.forward`` multiplies into the per-segment predictions
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

    def weights(self, user_id, item_ids, time_ms, device=None):
        """``c_interest_weight`` for one batch: user_id [B], item_ids [B, I], time_ms [B] -> float32 [B, I, S]; keys not in
        the store get ones (BaseModel.py:259-260,283-288)."""
        if self._index is None:
            self._build_index()
        ks, rows = self._index
        _, v = self._cat()
        u = np.asarray(user_id, np.int64).reshape(-1, 1)
        it = np.asarray(item_ids, np.int64)
        t = np.asarray(time_ms, np.int64).reshape(-1, 1)
        q = np.stack([np.broadcast_to(u, it.shape), it, np.broadcast_to(t, it.shape)], -1).reshape(-1, 3)
        out = np.ones((q.shape[0], self.S), np.float32)
        if len(ks):
            # lexicographic search on (user, item, time) via a structured view
            dt = np.dtype([("a", np.int64), ("b", np.int64), ("c", np.int64)])
            kv = np.ascontiguousarray(ks).view(dt).reshape(-1)
            qv = np.ascontiguousarray(q).view(dt).reshape(-1)
            pos = np.searchsorted(kv, qv)
            pos_c = np.minimum(pos, len(kv) - 1)
            hit = (pos < len(kv)) & (kv[pos_c] == qv)
            out[hit] = v[rows[pos_c[hit]]]
        w = torch.from_numpy(out.reshape(it.shape + (self.S,)))
        return w.to(device) if device is not None else w


def weighted_head(pred: torch.Tensor, weight: Optional[torch.Tensor] = None, duration: Optional[torch.Tensor] = None):
    """ClipRec.forward's ``(clip_predictions * interest_weight * mask).sum(-1)`` (ClipRec.py:163-181) on the device."""
    from . import hipabi as H
    return H.segment_weighted_sum(pred, weight, duration)
