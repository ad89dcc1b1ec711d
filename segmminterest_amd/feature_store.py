"""Device-resident feature table + index batches.

The reference dataset (``FrameDatasetSeq_SegMM._getitem``, dataloader_SegMM.py:271-362) looks every segment of a
row up in a ``[n_lines, D]`` feature file by the key ``"{photo}-{frame}"``, stacks up to S video rows and up to Lt
user-history rows, pads, builds the masks on the host and ships 573 KB per interaction to the GPU, where the trainer
L1-normalises them (main_for_seq_leave_earlystop_SegMM.py:272-273).  Here the table lives in HBM (288 GB per GPU hold
the whole SegMM feature file), a batch carries int64 line indices (-1 = padding slot), and one HBM-bound kernel does
gather + pad + mask + L1 normalisation (``segmm_gather_l1``)."""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence

import numpy as np
import torch

from . import hipabi as H


class KeyIndex:
    """``"{photo}-{frame}" -> line`` map of the feature file (the reference reads it from a JSON side file,
    dataloader_SegMM.py:199-215); unknown keys map to -1, which the gather treats as padding."""

    def __init__(self, keys: Iterable[str]):
        self.line = {k: i for i, k in enumerate(keys)}

    def lines(self, keys: Sequence[str], length: int) -> np.ndarray:
        out = np.full((length,), -1, dtype=np.int64)
        for j, k in enumerate(keys[:length]):
            out[j] = self.line.get(k, -1)
        return out


class ResidentFeatureTable:
    def __init__(self, table: torch.Tensor, user_table: Optional[torch.Tensor] = None, normalize: bool = True):
        if not table.is_cuda or table.dtype != torch.float32 or table.dim() != 2 or not table.is_contiguous():
            raise RuntimeError("feature table must be a contiguous float32 [n_lines, D] HIP tensor")
        self.tables: Dict[str, torch.Tensor] = {"photo": table, "user": table if user_table is None else user_table}
        self.normalize = normalize
        self._out: Dict[tuple, tuple] = {}

    def buffers(self, which: str, idx: torch.Tensor):
        """The persistent (features float32 [B, L, D], mask uint8 [B, L]) output buffers of :meth:`gather` for this index shape."""
        t = self.tables[which]
        key = (which, tuple(idx.shape))
        bufs = self._out.get(key)
        if bufs is None:
            bufs = self._out[key] = (torch.empty(tuple(idx.shape) + (t.shape[1],), dtype=torch.float32, device=t.device),
                                     torch.empty(tuple(idx.shape), dtype=torch.uint8, device=t.device))
        return bufs

    def gather(self, which: str, idx: torch.Tensor, amax=None, po=None):
        """idx int64 [B, L] (-1 = padding) -> (features float32 [B, L, D] L1-normalised, mask bool [B, L]).
        ``amax`` / ``po``: optional partial-maxima slots and plane output of the features (hipabi.gather_l1)."""
        bufs = self.buffers(which, idx)
        return H.gather_l1(self.tables[which], idx, normalize=self.normalize, out=bufs[0], mask=bufs[1], amax=amax, po=po)
