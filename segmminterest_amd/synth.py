"""Synthetic SegMM batches.

The reference's segment features are an external download (SegMM.md:18-21), so
every measurement and parity run here uses seeded synthetic batches with the
exact batch-dictionary contract of the reference's ``DataCollator``
(MMinterest/utils/dataloader_SegMM.py:296,314-316,350-359,370-382):

    user  f32[B,Lt,D_in]  user_mask  bool[B,Lt]
    photo f32[B,S,D_in]   photo_mask bool[B,S]
    label i64[B,S] in {1,0,-1,-2}
    user_identity_id / photo_identity_id / user_id / photo_id / time_ms i64[B]
    play_time / duration i64[B]

Label semantics follow ``construct_label_1D``
(data_process/get_data_SegMM_public.py:45-89) + ``_pad_label_list``
(dataloader_SegMM.py:240-249): 1 before the leave segment, 0 at it, -1 after,
-2 padding; a fully watched video is all 1.
"""
from __future__ import annotations

import torch


def make_labels(B: int, S: int, gen: torch.Generator, full_frac: float = 1.0 / 3.0,
                allow_full_len: bool = True):
    """durations ~ UniformInt[2,S]; leave index v ~ UniformInt[0,dur); ``full_frac`` fully watched.

    allow_full_len=False redraws rows that would be fully watched with dur == S
    (view_len == S): the reference crashes on those for S != 40 (SURVEY §8(a) notes).
    """
    dur = torch.randint(2, S + 1, (B,), generator=gen)
    full = torch.rand(B, generator=gen) < full_frac
    if not allow_full_len:
        full = full & (dur < S)
    v = (torch.rand(B, generator=gen) * dur.float()).long().clamp_(max=S - 1)
    v = torch.minimum(v, dur - 1)
    pos = torch.arange(S)[None, :]
    label = torch.full((B, S), -2, dtype=torch.int64)
    in_video = pos < dur[:, None]
    lab_leave = torch.where(pos < v[:, None], 1, torch.where(pos == v[:, None], 0, -1))
    lab = torch.where(full[:, None], torch.ones_like(lab_leave), lab_leave)
    label = torch.where(in_video, lab, label)
    return label, dur, in_video


def make_batch(B: int, S: int, Lt: int, D_in: int, n_users: int = 1903, n_items: int = 352494,
               seed: int = 1234, ragged_user: bool = True, allow_full_len: bool = True,
               features: bool = True, dtype=torch.float32):
    """Builds one CPU batch dict with the DataCollator key set. Features are U[0,1)
    (the trainer L1-normalises them, main_for_seq_leave_earlystop_SegMM.py:272-273);
    padded feature rows are zero like ``_pad_feature_list`` (dataloader_SegMM.py:251-268)."""
    gen = torch.Generator().manual_seed(seed)
    label, dur, photo_mask = make_labels(B, S, gen, allow_full_len=allow_full_len)
    batch = {}
    if ragged_user and Lt > 1:
        n_tok = torch.randint(1, Lt + 1, (B,), generator=gen)
        # the reference caps history at user_max_image tokens: most rows are full
        n_tok = torch.where(torch.rand(B, generator=gen) < 0.5, torch.full_like(n_tok, Lt), n_tok)
    else:
        n_tok = torch.full((B,), Lt, dtype=torch.int64)
    user_mask = torch.arange(Lt)[None, :] < n_tok[:, None]
    if features:
        photo = torch.rand(B, S, D_in, generator=gen, dtype=dtype)
        photo = photo * photo_mask[:, :, None]
        user = torch.rand(B, Lt, D_in, generator=gen, dtype=dtype)
        user = user * user_mask[:, :, None]
        batch["user"] = user
        batch["photo"] = photo
    batch["user_mask"] = user_mask
    batch["photo_mask"] = photo_mask
    batch["label"] = label
    batch["user_identity_id"] = torch.randint(1, n_users + 1, (B,), generator=gen)
    batch["photo_identity_id"] = torch.randint(1, n_items + 1, (B,), generator=gen)
    batch["user_id"] = batch["user_identity_id"].clone()
    batch["photo_id"] = batch["photo_identity_id"].clone()
    batch["time_ms"] = torch.randint(0, 2 ** 31 - 1, (B,), generator=gen)
    batch["duration"] = dur.clone()
    view_len = (label == 1).sum(1)
    batch["play_time"] = view_len.clone()
    return batch


def l1_normalize(x: torch.Tensor) -> torch.Tensor:
    """x / (||x||_1 + 1e-6) over the feature dim (main_for_seq_leave_earlystop_SegMM.py:272-273)."""
    return x / (x.norm(p=1, dim=-1, keepdim=True) + 1e-6)
