"""Training step of ``MMinterest/main_for_seq_leave_earlystop_SegMM.py:269-300`` on the HIP engine,
single GPU or data-parallel over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference trainer is single-process (DistributedDataParallel is imported at :13 and never used);
data parallelism is what this build adds:

* rows of the global batch are sharded contiguously over ranks, parameters are replicated;
* every cross-row normaliser of the losses (valid-row count of interestBPR, batch size, mask count) is
  made GLOBAL before the loss kernel runs -- one tiny all-reduce of 3 floats and one all-gather of the
  per-row view lengths -- so the sum of the per-rank gradients equals the single-process gradient of
  the whole batch exactly (no averaging step afterwards);
* gradients live in one flat buffer laid out in backward-completion order; each bucket (head, layer
  N-2, ..., layer 0, embedding) is all-reduced (SUM) asynchronously the moment its last gradient has
  been written, overlapping the remaining backward; the fused AdamW waits for the last bucket.

``init_model`` mirrors the reference's model construction (:60-130).
"""
from __future__ import annotations

import argparse
import functools
from typing import Dict, List, Optional

import os

import ctypes
import torch
import torch.nn as nn

from . import engine as E
from . import hipabi as H
from .decoder_leave_focal import MultiScaleTemporalDetrLeaveFocal
from .encoder import SegFormerX


# ------------------------------------------------------------------------------------------------ model factory
def init_model(args, reader=None, n_users: Optional[int] = None, n_items: Optional[int] = None,
               input_dim: int = 1024, max_vid_len: int = 40, max_usr_len: int = 100):
    """init_model(args, reader) of the reference trainer (main_for_seq_leave_earlystop_SegMM.py:60-130).
    ``reader`` only needs ``n_users`` / ``n_items``; feature width and lengths default to the
    reference's hard-coded 1024 / 40 / 100 and are overridable for synthetic configs."""
    n_users = reader.n_users if reader is not None else n_users
    n_items = reader.n_items if reader is not None else n_items
    N, d, h = args.num_layers_enc, args.d_model, args.nhead

    def backbone(user_id_max, video_id_max, usr_len):
        return SegFormerX(d_model_in=d, d_model_lvls=[d] * N, num_head_lvls=[h] * N, ff_dim_lvls=[d] * N,
                          input_vid_dim=input_dim, input_usr_dim=input_dim, max_vid_len=max_vid_len, max_usr_len=usr_len,
                          sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N, output_layers=[-1], model_cfg=args,
                          user_id_max=user_id_max, video_id_max=video_id_max, use_pe=getattr(args, "use_pe", 1))

    u, p = args.input_type["user"], args.input_type["photo"]
    if u == "both" or p == "both":
        um1, ul1, um2, ul2 = {"both": (-1, max_usr_len, n_users, 1), "id": (n_users, 1, n_users, 1),
                              "image": (-1, max_usr_len, -1, max_usr_len)}[u]
        vm1, vm2 = {"both": (-1, n_items), "id": (n_items, n_items), "image": (-1, -1)}[p]
        return MultiScaleTemporalDetrLeaveFocal(backbone(um1, vm1, ul1), backbone(um2, vm2, ul2), None, nn.Identity(), args)
    um1, ul1 = (n_users, 1) if u == "id" else (-1, max_usr_len)
    vm1 = n_items if p == "id" else -1
    return MultiScaleTemporalDetrLeaveFocal(backbone(um1, vm1, ul1), None, None, nn.Identity(), args)


def default_args(**over):
    """The reference's argparse defaults that the model reads (main_for_seq_leave_earlystop_SegMM.py:478-575)."""
    a = argparse.Namespace(debug=0, num_layers_enc=6, ablation_type="ours", d_model=512, nhead=16,
                           input_type={"user": "both", "photo": "both"}, learnable_bias=0, exposure_prob=[1.0] * 40,
                           fusion_heads=2, loss_type_list=["interestBPR"],
                           loss_weight={"focal": 1.0, "mse": 1.0, "hazard": 1.0, "surviveCE": 1.0, "interestBPR": 1.0,
                                        "interestCE": 1.0, "interestKL": 1.0},
                           mask_loss=0, use_pe=1, learning_rate=1e-3, weight_decay=1e-4)
    for k, v in over.items():
        setattr(a, k, v)
    return a


# ------------------------------------------------------------------------------------------------ optimizer
class FusedAdamW:
    """torch.optim.AdamW semantics (lr 1e-3, wd 1e-4, betas (.9,.999), eps 1e-8 at
    main_for_seq_leave_earlystop_SegMM.py:226) as ONE kernel over the contiguous live range of the
    flat parameter buffer.  Dead parameters (no gradient in the reference either) are never touched."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        self.step_count = 0
        self.m = self.v = None
        self._flat_id = None

    def _state(self):
        st = self.model._store
        st.ensure()
        if self.m is None or self._flat_id != st.flat.data_ptr():
            self.m = torch.zeros(st.n_live, device=st.flat.device)
            self.v = torch.zeros(st.n_live, device=st.flat.device)
            self._flat_id = st.flat.data_ptr()
        return st

    def zero_grad(self, set_to_none=True):
        plist = self.__dict__.get("_plist")
        if plist is None:          # the Parameter objects of a model do not change; walking the module tree costs 0.3 ms per step
            plist = self._plist = list(self.model.parameters())
        for p in plist:
            p.grad = None

    def step(self, gbuf=None):
        self.begin_step()
        st = self._state()
        pos = 0
        # id tables whose rows without a gradient were already stepped at the head of the step (table_early): the listed rows
        # now, with their gradients; everything else as contiguous ranges around them
        for o, n, rows, width, ids, flags in sorted(self.__dict__.pop("_early", []), key=lambda e: e[0]):
            self.step_range(pos, o, gbuf)
            H.adamw_table(st.flat, st.gflat if gbuf is None else gbuf, self.m, self.v, o, rows, width, ids, flags, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.wd, -1 if self.__dict__.get("device_state", False) else self.step_count, 1)
            pos = o + n
        self.step_range(pos, st.n_live, gbuf)
        self.end_step()

    def table_early(self, name, ids):
        """First pass of the two-pass update of the id-embedding table ``name`` (segmm_adamw_table): every row that is NOT in
        ``ids`` -- the rows the coming backward gives no gradient -- takes its AdamW step of THIS optimizer step now, with g = 0,
        on the current stream (the trainer calls this at the head of the step on a stream of its own: the forward and backward
        only read the listed rows).  ``step()`` then updates the listed rows and skips the table's range.  Element for element
        the arithmetic of the one-launch dense update."""
        st = self._state()
        o, n = st.index[name]
        rows, width = st._params[name].shape
        fl = self.__dict__.setdefault("_table_flags", {})
        flags = fl.get(name)
        if flags is None or flags.numel() != rows or flags.device != st.flat.device:
            flags = fl[name] = torch.zeros((rows,), dtype=torch.int32, device=st.flat.device)
        H.adamw_table(st.flat, None, self.m, self.v, o, rows, width, ids, flags, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                      -1 if self.__dict__.get("device_state", False) else self.step_count + 1, 0)
        self.__dict__.setdefault("_early", []).append((o, n, rows, width, ids, flags))

    # one optimizer step as several launches over disjoint ranges that together cover the live range: the data-parallel
    # trainer steps each gradient bucket as soon as ITS all-reduce has completed (begin_step, step_range ..., end_step)
    def begin_step(self):
        st = self._state()
        self.step_count += 1
        if not st.__dict__.get("direct_grads", False):
            self._sync_grads(st)

    def _sync_grads(self, st):
        """The kernel reads the FLAT gradient buffer.  After Trainer.train_step the parameters' ``.grad`` ARE views of it
        (engine.deliver_grads).  After any other backward (``loss.backward()``: gradients arrive through autograd, possibly
        rewritten by hooks, clipped or accumulated over several backwards) they are tensors of their own: copy them in."""
        base = st.gflat.data_ptr()
        for n in st.live_names:
            g = st._params[n].grad
            if g is not None:
                o, k = st.index[n]
                if g.data_ptr() != base + 4 * o or not g.is_contiguous():
                    H.torch_fallback("the copy of %s.grad into the flat gradient buffer" % n)
                    st.gflat[o:o + k].view(g.shape).copy_(g)

    def step_range(self, start, end, gbuf=None):
        st = self._state()
        if self.__dict__.get("device_state", False) and self.__dict__.get("_step_state") is not None:
            # the bound step state is a process-global host pointer read when a launch is MADE; a direct opt.step() /
            # step_range() between two trainers' steps must not pick up the other trainer's seed words and bias corrections
            H.step_bind(self._step_state)
        if end > start:
            # device_state: the bias corrections come from the device-side step count (segmm_step_advance), step = -1
            H.adamw(st.flat, st.gflat if gbuf is None else gbuf, self.m, self.v, end - start, self.lr, self.betas[0], self.betas[1],
                    self.eps, self.wd, -1 if self.__dict__.get("device_state", False) else self.step_count, p_off=start)

    def end_step(self):
        self.model._store.fused_version += 1          # the pre-split weight planes of the GEMM engines are now stale

    def state_dict(self):
        """torch.optim.AdamW-format state (keyed by the index of the parameter in model.parameters()),
        so a checkpoint written here resumes under the reference's optimizer and vice versa."""
        st = self._state()
        names = [n for n, _ in self.model.named_parameters()]
        state = {}
        for i, n in enumerate(names):
            if n in st.live_names:
                o, k = st.index[n]
                shape = st._params[n].shape
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[o:o + k].view(shape).clone(),
                            "exp_avg_sq": self.v[o:o + k].view(shape).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        st = self._state()
        names = [n for n, _ in self.model.named_parameters()]
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for i, s in sd["state"].items():
            n = names[int(i)]
            if n in st.live_names:
                o, k = st.index[n]
                self.m[o:o + k].copy_(s["exp_avg"].reshape(-1))
                self.v[o:o + k].copy_(s["exp_avg_sq"].reshape(-1))
                steps.add(int(float(s["step"])))
        self.step_count = max(steps) if steps else 0
        g = sd["param_groups"][0]
        self.lr, self.wd, self.betas, self.eps = g["lr"], g["weight_decay"], tuple(g["betas"]), g["eps"]
        if self.__dict__.get("device_state", False):          # the device-side step count (bias corrections) follows the checkpoint
            H.step_bind(self._step_state)
            seed, _, _ = H.step_get()
            H.step_set(seed, self.step_count, *self.betas)


# ------------------------------------------------------------------------------------------------ communication
class DPComm:
    """The three collectives of a data-parallel step (engine-agnostic; works on any torch.distributed
    backend, ``nccl`` = RCCL on the GPU box, ``gloo`` in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # every collective of the step is issued iff ``active``: more than one rank -- or SEGMM_DP_FORCE=1 with an initialised
        # process group of ONE rank, which sends the single-GPU step through the real RCCL calls (tests/test_dp_gpu.py: the
        # only way to execute the nccl code path on a one-GPU box; results must equal the plain step bit for bit)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("SEGMM_DP_FORCE", "0") == "1")
        self.pending = []
        # gloo has no device collectives for every op: device tensors are staged through the host.
        # Only the test harness uses that (two ranks sharing one GPU); production is nccl = RCCL.
        self.host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"

    def _all_reduce(self, t, async_op=False):
        if self.host_staged and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return None
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _pbuf(self, key, shape, dtype, device):
        """Persistent staging buffer of a collective: the SAME tensor every step, so that a recorded step (Trainer.record),
        whose kernels name the results by address, can be replayed around the collectives."""
        bufs = self.__dict__.setdefault("_bufs", {})
        b = bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype or b.device != device:
            b = bufs[key] = torch.empty(shape, dtype=dtype, device=device)
        return b

    def label_stats_buffers(self, B, device):
        """(v, v2, norms) as views of the persistent send record of :meth:`global_label_stats`: the statistics kernel writes where
        the all-gather reads, nothing is packed."""
        mine = self._pbuf("ls_mine", (2 * B + 3,), torch.float32, device)
        return mine[:B], mine[B:2 * B], mine[2 * B:]

    def global_label_stats(self, v, v2, norms):
        """(v_all, v2_all, norms_global): all-gather the per-row view lengths, sum the 3 normalisers.  Returns the triple
        (single process) or a closure that waits for the asynchronous collective and returns it."""
        if not self.active:
            return v, v2, norms
        # ONE collective: [v | v2 | norms] of every rank (every rank holds the same B_local); the three normalisers are
        # summed locally from the gathered copies, in rank order on every rank (identical results everywhere)
        # The collective is ASYNCHRONOUS: the caller gets a closure and calls it right before the loss kernel, so the
        # all-gather (and the rank skew it absorbs) runs under the backbone forward instead of in front of it.
        B, n, dev, G = v.shape[0], 2 * v.shape[0] + 3, v.device, self.world
        mine = self._pbuf("ls_mine", (n,), v.dtype, dev)
        gathered = self._pbuf("ls_gathered", (G * n,), v.dtype, dev)
        v_all, v2_all = self._pbuf("ls_v", (G * B,), v.dtype, dev), self._pbuf("ls_v2", (G * B,), v.dtype, dev)
        norms_g = self._pbuf("ls_norms", (3,), norms.dtype, dev)
        state = {}

        def start():
            # (the statistics kernel normally wrote straight into `mine` -- label_stats_buffers; a caller's own tensors are copied
            # by the library's copy kernel: no torch kernel inside the data-parallel step)
            if v.data_ptr() != mine.data_ptr():
                cp = H.copy_bytes if mine.is_cuda else (lambda d_, s_: d_.copy_(s_))
                cp(mine[:B], v.contiguous())
                cp(mine[B:2 * B], v2.contiguous())
                cp(mine[2 * B:], norms.contiguous())
            if self.host_staged and mine.is_cuda:
                hg = torch.empty((G * n,), dtype=mine.dtype)
                self.dist.all_gather_into_tensor(hg, mine.cpu(), group=self.group)
                gathered.copy_(hg)
                state["work"] = None
            else:
                state["work"] = self.dist.all_gather_into_tensor(gathered, mine, group=self.group, async_op=True)

        def finish():
            work = state.pop("work", None)
            if work is not None:
                work.wait()          # stream-level wait on the communication stream, no host sync
            if gathered.is_cuda:
                H.label_stats_unpack(gathered, G, B, v_all, v2_all, norms_g)          # one launch: split + rank-ordered sum of the counts
            else:
                g = gathered.view(G, n)
                v_all.view(G, B).copy_(g[:, :B])
                v2_all.view(G, B).copy_(g[:, B:2 * B])
                torch.sum(g[:, 2 * B:], 0, out=norms_g)          # counts: exact in fp32 below 2^24
            return v_all, v2_all, norms_g
        H.host_action(start)
        return lambda: H.host_action(finish)

    def gather_rows(self, ids, rows):
        """Sparse exchange of id-table gradients: returns a CLOSURE that yields (ids of all ranks [G*B], rows of all ranks
        [G*B, w]) in rank order (every rank holds the same B, like :meth:`global_label_stats`).

        ONE asynchronous all-gather of ``[rows | id]`` per rank (the int64 id rides in two float lanes of its row) on a process
        group OF ITS OWN: on the gradient group it would queue behind every bucket all-reduce still in flight on that group's
        RCCL stream.  The caller issues it as soon as the compact rows exist and calls the closure right before the segment sum
        (engine._embed_bwd), so the collective and the rank skew it absorbs run under the rest of the embedding backward."""
        if not self.active:
            return lambda: (ids, rows)
        ids, rows = ids.contiguous(), rows.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.to(torch.int64)
        B, w = rows.shape
        G, dev = self.world, rows.device
        seq = self.__dict__.get("_gr_seq", 0)          # (which exchange of the step: reset by Trainer at the head of the backward)
        self._gr_seq = seq + 1
        packed = self._pbuf(("gr_packed", seq), (B, w + 2), torch.float32, dev)
        gathered = self._pbuf(("gr_gathered", seq), (G * B, w + 2), torch.float32, dev)
        ids_all = self._pbuf(("gr_ids", seq), (G * B,), torch.int64, dev)
        rows_all = self._pbuf(("gr_rows", seq), (G * B, w), torch.float32, dev)
        state = {}

        def start():
            packed[:, :w].copy_(rows)
            packed[:, w:].copy_(ids.view(B, 1).view(torch.float32))          # bit pattern of the id, not a conversion
            if self.host_staged and rows.is_cuda:
                hg = torch.empty((G * B, w + 2), dtype=torch.float32)
                self.dist.all_gather_into_tensor(hg, packed.cpu(), group=self.group)
                gathered.copy_(hg)
                state["work"] = None
            else:
                state["work"] = self.dist.all_gather_into_tensor(gathered, packed, group=self._row_group(), async_op=True)

        def finish():
            work = state.pop("work", None)
            if work is not None:
                work.wait()          # stream-level wait on the row group's communication stream, no host sync
            ids_all.copy_(gathered[:, w:].contiguous().view(torch.int64).view(-1))
            rows_all.copy_(gathered[:, :w])
            return ids_all, rows_all
        H.host_action(start)
        return lambda: H.host_action(finish)

    def _row_group(self):
        """Process group of the row exchange (same ranks as the gradient group; created on first use, by every rank at the
        same point of its first data-parallel backward).  A one-rank forced group (tests) shares the gradient group."""
        g = self.__dict__.get("_rowg")
        if g is None:
            if self.world > 1:
                ranks = self.dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.dist.get_world_size()))
                g = self.dist.new_group(ranks=ranks, backend=self.dist.get_backend(self.group))
            else:
                g = self.group if self.group is not None else self.dist.group.WORLD
            self._rowg = g
        return g

    def gather_ints(self, t):
        """[G*B] int32: the values of every rank in rank order (validation: leave ranks of the global batch)."""
        if not self.active:
            return t
        t = t.contiguous()
        if self.host_staged and t.is_cuda:
            h = torch.empty((self.world * t.shape[0],), dtype=t.dtype)
            self.dist.all_gather_into_tensor(h, t.cpu(), group=self.group)
            return h.to(t.device)
        out = torch.empty((self.world * t.shape[0],), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def reduce_bucket(self, flat_grad, start, end):
        """Asynchronous SUM all-reduce of one contiguous gradient bucket (gradients are already
        normalised by global counts, so SUM -- not mean -- reproduces the single-process gradient)."""
        if not self.active or end <= start:
            return
        w = self._all_reduce(flat_grad[start:end], async_op=True)
        if w is not None:
            self.pending.append(w)

    def take_pending(self):
        """The outstanding asynchronous all-reduces (issue order); the caller waits for them (``work.wait()`` is a
        stream-level wait on the communication stream, no host sync)."""
        p, self.pending = self.pending, []
        return p

    def finish(self):
        for w in self.take_pending():
            w.wait()

    def sum_scalar(self, t):
        if self.active:
            self._all_reduce(t)
        return t


def shard_rows(n_rows: int, world: int, rank: int):
    """Contiguous row block of ``rank``; the first ``n_rows % world`` ranks get one more."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# ------------------------------------------------------------------------------------------------ trainer
class Trainer:
    """zero_grad -> L1-normalise -> forward -> backward (-> bucketed all-reduce) -> AdamW, i.e. the body of
    the reference's hot loop (main_for_seq_leave_earlystop_SegMM.py:269-300) minus its host syncs.
    Gradient clipping is a no-op in the reference (exhausted generator at :298) and is therefore absent."""


    def __init__(self, model, lr=1e-3, weight_decay=1e-4, comm: Optional[DPComm] = None, overlap=True, dropout=True,
                 feature_table=None, sparse_tables=True, device_state=False):
        self.model = model
        # device_state: what changes from step to step (dropout seed words, AdamW's step count and bias corrections, the
        # site-header rows of a step) lives on the device / is laid out per step, so the kernel arguments of a step never
        # change and ``record()`` can note its launch sequence once for ``run_recorded()`` to replay from C.  The eager step in
        # this mode and the recorded step are bit-identical.
        self.device_state = bool(device_state)
        # with a device-resident feature table the batch carries INDEX lists ("photo_idx" [B, S],
        # "user_idx" [B, Lt], -1 = padding) instead of feature tensors; gather + pad + mask + L1 normalisation is one
        # HBM-bound kernel (segmm_gather_l1) and the 573 KB/row host->device copy disappears
        self.feature_table = feature_table
        self.dropout = dropout      # False: run the step in eval mode (deterministic; used by the DP equivalence tests)
        self.opt = FusedAdamW(model, lr=lr, weight_decay=weight_decay)
        self.comm = comm if comm is not None else DPComm()
        self.overlap = overlap
        st = model._store
        model._dp_hook = self.comm.global_label_stats if self.comm.active else None
        model._dp_stats_bufs = self.comm.label_stats_buffers if self.comm.active else None
        st.bucket_hook = self._on_bucket if self.comm.active else None
        # id mode under DP: the embedding tables' gradients travel as B rows per rank (DPComm.gather_rows) and their flat
        # ranges are cut out of the dense all-reduce; ``sparse_tables=False`` keeps the dense all-reduce (A/B, tests)
        self.sparse_tables = bool(sparse_tables) and self.comm.active and os.environ.get("SEGMM_SPARSE_TABLES", "1") != "0"
        st.row_exchange = self.comm.gather_rows if self.sparse_tables else None
        if self.sparse_tables:
            # dist.new_group is COLLECTIVE over the default group: create the row-exchange group here, where every rank
            # constructs its Trainer, not lazily inside the first backward (ranks that skip the id-table path on
            # their first step, or a DPComm on a sub-group, would deadlock)
            self.comm._row_group()
        self.per_bucket_adamw = os.environ.get("SEGMM_BUCKET_ADAMW", "1") != "0"
        self.table_two_pass = os.environ.get("SEGMM_TABLE_TWO_PASS", "1") != "0"
        self.begin_overlap = os.environ.get("SEGMM_BEGIN_OVERLAP", "1") != "0"          # head of the step on two streams (_features)
        self.bucket_bytes = int(float(os.environ.get("SEGMM_DP_BUCKET_MB", "8")) * (1 << 20))      # merge threshold of _on_bucket
        self._bucket_works = []
        self._pending_range = None
        self._norm = {}
        self._norm_amax = None
        self._norm_planes = {}
        self._slot = 0
        self._pf = None
        self._pf_stream = None
        self._norm_fresh = False
        if self.device_state:
            # the device step state (seed words, step count, bias corrections) is a small struct in memory THIS trainer owns
            # (segmm_step_bind names it for the launches that follow): several device_state trainers coexist in one process
            self._step_state = torch.zeros((H.step_state_bytes() + 3) // 4, dtype=torch.int32, device=next(model.parameters()).device)
            H.step_bind(self._step_state)
            self.opt._step_state = self._step_state
            self.opt.device_state = True
            seed0 = int(torch.randint(0, 2 ** 62, (1,)).item())          # torch.manual_seed -> reproducible runs
            if self.comm.rank:          # data-parallel ranks seed torch identically: every rank drops different elements of its rows
                seed0 = (seed0 ^ (self.comm.rank * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1)
            H.step_set(seed0, self.opt.step_count, *self.opt.betas)
            st.live_seed = H.LIVE_SEED | 0x5E6D0001          # the (constant) seed argument of every dropout launch

    def _on_bucket(self, name, after_side=False):
        # a host action of the step: a recorded step (record / run_recorded) calls it again at the same point of the launch order
        H.host_action(functools.partial(self._on_bucket_now, name, after_side))

    def _reset_buckets(self):
        self._bucket_works = []
        self._pending_range = None
        self.comm._gr_seq = 0

    def _wait_buckets(self, idxs):
        for i in idxs:
            for w in self._bucket_works[i][2]:
                w.wait()

    def _on_bucket_now(self, name, after_side=False):
        """Called from inside the backward the moment the gradients of bucket ``name`` are written (``after_side``: part of
        them by launches still in flight on the engine's side stream).  Adjacent buckets are MERGED until ``bucket_bytes`` of
        gradients are pending (a collective call costs ~0.2 ms of host time, whatever its size), then one asynchronous
        all-reduce is issued for the merged range; its works are remembered so that AdamW can step the range as soon as THEY
        are done (train_step, which also flushes what is still pending when the backward returns)."""
        st = self.model._store
        if not self.overlap:
            return
        for b, s, e in st.buckets:
            if b == name:
                p = self._pending_range
                if p is not None and p[1] != s and p[0] != e:
                    self._flush_bucket()          # not adjacent (two backbones interleave): send what is pending first
                    p = None
                self._pending_range = [s, e, after_side] if p is None else [min(p[0], s), max(p[1], e), p[2] or after_side]
                if 4 * (self._pending_range[1] - self._pending_range[0]) >= self.bucket_bytes:
                    self._flush_bucket()
                return

    def _flush_bucket(self):
        p, self._pending_range = self._pending_range, None
        if p is None:
            return
        st = self.model._store
        s, e, after_side = p
        if after_side and st.overlap and st._side_stream is not None:
            # order the collective behind main AND side stream without stalling the main stream: issue it from the side
            # stream's context after making the side stream wait for the main stream's work so far
            main, side = torch.cuda.current_stream(), st.side_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                self._reduce_dense(s, e)
        else:
            self._reduce_dense(s, e)
        self._bucket_works.append((s, e, self.comm.take_pending()))

    def _reduce_dense(self, s, e):
        """All-reduce [s, e) of the flat gradient minus the row-exchanged table ranges."""
        st = self.model._store
        if self.sparse_tables:
            for ts, te in st.table_ranges():
                if ts < e and te > s:
                    self.comm.reduce_bucket(st.gflat, s, max(s, ts))
                    s = min(e, te)
        self.comm.reduce_bucket(st.gflat, s, e)

    def _n_users(self):
        """reader.n_users as the reference's init_model sizes the user table (Embedding(n_users + 1, d)); None in image mode."""
        for bb in (self.model.backbone1, getattr(self.model, "backbone2", None)):
            if bb is not None and bb.id_usr:
                return bb.usr_proj.num_embeddings - 1
        return None

    def _input_act(self, key, buf, planes_only=False):
        """Site header (+ planes, + plane output for the producer kernel) of an input feature tensor the trainer itself
        produces this step (L1 normalisation or table gather).  The Act travels WITH the tensor object (attribute
        ``_segmm_act``), never keyed by its device address: a tensor the trainer did not produce in this step (rand_like for
        'noUser', features handed straight to model()) has none and gets an absmax pass in engine.BackboneRun._input_act."""
        st = self.model._store
        if not st.engine_h:
            return None
        dev = buf.device
        if not self._norm_fresh:
            self._norm_amax = st.hdr_rows(2)          # two clean site headers per step from the store's ring (no fill launch)
            self._norm_fresh = True
        cols = buf.shape[-1]
        rows = buf.numel() // cols
        planes = None
        if st.engine_p and cols % 32 == 0:
            pk = (key, self._slot)
            planes = self._norm_planes.get(pk)
            if planes is None or planes.shape != (rows, 2 * cols) or planes.device != dev:
                planes = self._norm_planes[pk] = torch.empty((rows, 2 * cols), dtype=torch.float16, device=dev)
        act = E.Act(buf, self._norm_amax[0 if key == "user" else 1], rows, cols, planes)
        if planes is not None:
            delayed = (self.model.training and st.scaling != "exact") or st.scaling == "always"
            act.scale_ptr = st.scale_ptr("in." + key, delayed)
            if act.scale_ptr is not None:
                if planes_only and st.input_planes_only and self._plane_consumers_only(key):
                    # L1-normalised rows: 1/cols <= max |y| <= 1.  With the FIXED scale 2^14 the planes can neither overflow nor
                    # fall below the fp16 window, whatever the batch: no fp32 copy is written and the consumers get none
                    act.scale_ptr = st.const_f32(16384.0).data_ptr()
                    act.no_f32 = True
                act.po = H.PO(planes, 2 * cols, act.hdr, act.scale_ptr)
        buf._segmm_act = act
        return act

    def _tables_early(self, batch):
        """Id mode, single process: the dense AdamW over the item table (config 3: 352 494 x 256 floats = 2.5 GB of optimizer
        traffic, 0.54 ms alone at the END of the step) is split in two -- the rows of this batch (the only ones with a gradient,
        known from the batch's ids) are stepped after the backward as before, every other row now, with g = 0, on an auxiliary
        stream under the forward and backward, which never touch those rows.  (Data parallel: the listed rows are those of ALL
        ranks, known only after the row exchange -- the dense update stays.)"""
        # a step that failed between this pass and opt.step() (forward / backward error, OOM) left its entry queued and its rows
        # flagged: drop both before queueing this step's, or the table's range would be stepped twice and the failed batch's rows
        # skipped
        stale = self.opt.__dict__.pop("_early", None)
        if stale:
            E.join_aux(self.model._store)
            for _, _, _, _, _, flags in stale:
                flags.zero_()
            #  the failed step had already advanced the DEVICE step count (segmm_step_advance at its head) while
            # opt.step_count never moved, and THIS step has advanced it once more: put it back to step_count + 1, the number of the
            # step that is running, so its bias corrections are right again.  What cannot be taken back and is accepted: the failed step's early pass applied one g = 0 AdamW
            # update (weight decay + moment decay) to the table rows outside its batch.
            if self.device_state:
                H.step_bind(self._step_state)
                seed, _, _ = H.step_get()
                H.step_set(seed, self.opt.step_count + 1, *self.opt.betas)
        if self.comm.active or not self.table_two_pass or self._param_hooks():
            return
        model, st = self.model, self.model._store
        ids = batch.get("photo_identity_id")
        if ids is None or ids.dtype != torch.int64 or not ids.is_contiguous():
            return
        for pre, bb in (("backbone1.", model.backbone1), ("backbone2.", getattr(model, "backbone2", None))):
            if bb is None or not bb.id_vid:
                continue
            name = pre + "vid_proj.weight"
            if name not in st.index or name not in st.live_names or st._params[name].shape[1] % 4:
                continue
            with E.aux_work(st):
                self.opt.table_early(name, ids.reshape(-1))

    def _plane_consumers_only(self, key) -> bool:
        """Every launch that reads the input features of ``key`` is a plane GEMM (forward projection and its weight gradient):
        the plane engine, a P32 plane entry for the projection weight and d_model a multiple of 32 (else the weight gradient
        takes the on-the-fly kernel, which reads the fp32 copy -- ADVICE r3: d_model = 48 raised on the second step)."""
        st = self.model._store
        if not st.engine_p:
            return False
        kind = "usr" if key == "user" else "vid"
        for pre, bb in (("backbone1.", self.model.backbone1), ("backbone2.", getattr(self.model, "backbone2", None))):
            if bb is None or (bb.id_usr if kind == "usr" else bb.id_vid):
                continue
            if bb.d_model % 32 or (pre + kind + "_proj.weight") not in st.wpt:
                return False
        return True

    def normalize(self, key, x):
        """a1: x / (sum|x| + 1e-6) over the feature dim, into a persistent buffer."""
        bk = (key, self._slot)
        buf = self._norm.get(bk)
        if buf is None or buf.shape != x.shape or buf.device != x.device:
            buf = self._norm[bk] = torch.empty_like(x)
        act = self._input_act(key, buf, planes_only=True)
        # the GEMM that reads ``buf`` needs max|buf| (and, on the plane engine, its fp16 planes): both are folded into this
        # kernel instead of separate passes over the features.  Planes only (training steps after the first): ``buf`` is then
        # just the handle that carries the Act -- its fp32 contents are NOT written
        only = act is not None and act.no_f32
        H.l1norm(x, None if only else buf, amax=None if act is None else act.slots, po=None if act is None else act.po)
        if act is not None:
            E.produced(act)
        return buf

    def _features(self, batch):
        """(usr, usr_mask, vid, vid_mask): L1-normalised feature tensors of a batch, from the tensors it carries or --
        index batches -- gathered from the resident table.  Taken from :meth:`prefetch` when that ran for this very batch."""
        st = self.model._store
        pf, self._pf = self._pf, None
        if pf is not None and pf["batch"] is batch:
            st.ensure()
            torch.cuda.current_stream().wait_event(pf["done"])          # the main stream waits for the prefetch stream's kernels
            usr, um, vid, vm = pf["out"]
            self._norm_amax, self._norm_fresh = pf["amax"], pf["fresh"]
        elif self.begin_overlap and st.flat is not None and st.overlap:
            # The head of the step is a serial chain on one stream: weight maxima + split (32 us at config 2), user features,
            # video features, label statistics -- and the user-token chain, which bounds the forward up to the first attention
            # (the side stream's embedding GEMM -> LayerNorm -> projection), cannot fork before the last of them is enqueued.
            # Only the weight planes and the user features are in ITS way: the video features and the label statistics go to the
            # auxiliary stream (low priority), which the main stream joins before the model's forward.  The stage is HBM-bound
            # (0.6 GB), so what this buys is the small launches and their gaps, not the copies.
            it = self.model.input_type
            both = it["user"] != "id" and it["photo"] != "id"
            rows = st.hdr_rows(2) if st.engine_h else None          # taken (and, at a ring quarter, cleared) in main-stream order
            with E.aux_work(st):
                early = self._features_compute(batch, hdr_rows=rows, only="photo") if both else None
                gt = batch.get("label")
                if (self.model._dp_hook is None and gt is not None and gt.dtype == torch.int64 and gt.is_contiguous() and gt.dim() == 2
                        and getattr(self.model, "_loss_spec", None) is not None):
                    self.model._stats_pre = self.model._label_stats(gt, gt.shape[0], gt.shape[1])
            st.ensure()          # first ParamStore.ensure() of the step: the full check, the weight planes of this step
            usr, um, vid, vm = self._features_compute(batch, hdr_rows=None if both else rows, only="user" if both else None, flip=not both)
            if early is not None:
                vid, vm = early[2], early[3]
            E.join_aux(st)
        else:
            st.ensure()
            usr, um, vid, vm = self._features_compute(batch)
        if st.engine_p and self._norm_amax is not None and self._norm_fresh:
            st.update_scales(self._norm_amax, ["in.user", "in.photo"], 2)          # the input sites' scales of the next step
        return usr, um, vid, vm

    def _features_compute(self, batch, hdr_rows=None, only=None, flip=True):
        """``only``: one of the two inputs ("user" / "photo"; _features enqueues them on different streams); ``flip``: alternate
        the output buffer set (once per step)."""
        it = self.model.input_type
        usr = vid = None
        um, vm = batch.get("user_mask"), batch.get("photo_mask")
        if flip:
            self._slot ^= 1                      # two sets of output buffers, alternated: see prefetch()
            self._norm_fresh = False
        if hdr_rows is not None:
            self._norm_amax, self._norm_fresh = hdr_rows, True
        for key, kind in (("user", it["user"]), ("photo", it["photo"])):
            if kind == "id" or (only is not None and key != only):
                continue
            if key + "_idx" in batch:
                out, mask = self.feature_table.buffers(key, batch[key + "_idx"], self._slot)
                act = self._input_act(key, out)
                self.feature_table.gather(key, batch[key + "_idx"], amax=None if act is None else act.slots,
                                          po=None if act is None else act.po, slot=self._slot)
                if act is not None:
                    E.produced(act)
                x, m = out, mask.view(torch.bool)
            else:
                x, m = self.normalize(key, batch[key]), (um if key == "user" else vm)
            if key == "user":
                usr, um = x, m
            else:
                vid, vm = x, m
        return usr, um, vid, vm

    def prefetch(self, batch):
        """Enqueue the input stage of ``batch`` (a1: L1 normalisation, or the table gather of an index batch; with the fp16 planes
        and maxima the first GEMM needs) on a low-priority stream of its own, NOW -- typically while the current step's backward
        is running.  The next ``train_step(batch)`` with this very dict picks the results up instead of running the stage at the
        head of its critical path (the reference overlaps the same work through its DataLoader workers).  Output buffers
        alternate between two sets, so the step in flight keeps reading its own."""
        st = self.model._store
        it = self.model.input_type
        if st.flat is None or (it["user"] == "id" and it["photo"] == "id"):
            return
        main = torch.cuda.current_stream()
        if self._pf_stream is None:          # lowest priority, like the engine's side stream: it must not starve the step in flight
            self._pf_stream = torch.cuda.Stream(device=st.flat.device, priority=int(os.environ.get("SEGMM_SIDE_PRIORITY", "1")))
        rows = st.hdr_rows(2) if st.engine_h else None          # taken (and, at a ring quarter, cleared) in main-stream order
        ev = torch.cuda.Event()
        ev.record(main)                                            # behind this step's scales_update of the input sites
        self._pf_stream.wait_event(ev)
        with torch.cuda.stream(self._pf_stream):
            out = self._features_compute(batch, hdr_rows=rows)
            done = torch.cuda.Event()
            done.record(self._pf_stream)
        self._pf = dict(batch=batch, out=out, done=done, amax=self._norm_amax, fresh=self._norm_fresh)

    def train_step(self, batch: Dict[str, torch.Tensor], next_batch: Optional[Dict[str, torch.Tensor]] = None):
        """One optimisation step on ``batch``.  ``next_batch``: the batch of the FOLLOWING step, if known -- its input stage is
        enqueued on a side stream right away (:meth:`prefetch`) and overlaps this step's backward."""
        model, st = self.model, self.model._store
        if self.device_state:
            H.step_bind(self._step_state)          # (a host-side pointer: the launches below carry it in their arguments)
        if model.training != bool(self.dropout):
            model.train(self.dropout)          # (walks every sub-module: 0.35 ms of host time when done every step)
        self.opt.zero_grad()
        H.mark(H.PHASE_STEP_BEGIN)
        if self.device_state:          # first launches of the step: advance the device-side step state, fresh header rows
            H.step_advance(*self.opt.betas)
            st.hdr_step_begin()
        st._trusted = False
        try:
            usr, um, vid, vm = self._features(batch)          # first ParamStore.ensure() of the step: the full check
            st._trusted = True
            self._tables_early(batch)
            if next_batch is not None:
                self.prefetch(next_batch)
            out = self._train_step(batch, usr, um, vid, vm)
            r = self.__dict__.get("_recorded")
            if r is not None and H.RECORDER is None:
                r["prev_batch"] = batch          # a recorded step that follows names this step's batch as "the previous one"
            return out
        finally:
            st._trusted = False
            model.__dict__.pop("_stats_pre", None)          # (a step that failed before its forward must not leave this batch's statistics behind)
            if self.device_state:
                st.hdr_step_end()          # evaluation passes between steps draw their headers from the wrapping ring

    def _train_step(self, batch, usr, um, vid, vm):
        model, st = self.model, self.model._store
        usr_id = batch["user_identity_id"]
        if "noUser" in getattr(model.backbone1, "ablation_type", "ours"):
            # 'noUser' / 'noUser_SelfAtt' (main...SegMM.py:275-280, main...KuaiRand.py:254-258): the TRAINING forward sees
            # uniform-random user features and random user ids in [1, n_users); validation keeps the real ones
            live = st.__dict__.get("live_seed")
            n_users = self._n_users()
            if live is not None:
                # device-side step state (the step may be recorded): the draws come from the library's counter hash, into persistent
                # buffers -- the reference's distributions (U[0, 1) features, uniform ids), not torch's bit stream
                if usr is not None:
                    buf = self.__dict__.get("_nouser_feat")
                    if buf is None or buf.shape != usr.shape or buf.device != usr.device:
                        buf = self._nouser_feat = torch.empty_like(usr)
                    H.rand_uniform(buf, live, E.SITE_NOUSER_FEAT)
                    usr = buf
                if n_users is not None:
                    ib = self.__dict__.get("_nouser_ids")
                    if ib is None or ib.shape != usr_id.shape or ib.device != usr_id.device:
                        ib = self._nouser_ids = torch.empty_like(usr_id, dtype=torch.int64)
                    H.rand_ids(ib, 1, max(n_users, 2), live, E.SITE_NOUSER_IDS)
                    usr_id = ib
            else:
                if usr is not None:
                    usr = torch.rand_like(usr)
                if n_users is not None:
                    usr_id = torch.randint(1, max(n_users, 2), usr_id.shape, device=usr_id.device)
        out = model(usr_image=usr, usr_id=usr_id, usr_mask=um, vid_image=vid,
                    vid_id=batch["photo_identity_id"], vid_mask=vm, gt=batch["label"], mode="train")
        if self.comm.active:
            H.host_action(self._reset_buckets)          # (a host action of the recorded data-parallel step)
        else:
            self._reset_buckets()
        # seed the backward with the model's own constant-one tensor: the head recognises it (same storage) and skips both the
        # ones_like fill autograd would launch and the dlogits * 1 multiply
        if self._param_hooks():
            # tensor / post-accumulate-grad hooks on parameters (gradient scaling, clipping, DDP-style wrappers) only fire when
            # the gradients travel through autograd: take the plain path (engine.grads_out), FusedAdamW copies the results in
            if self.comm.active:
                raise RuntimeError("parameter hooks under data parallelism are not supported: the bucket all-reduces are "
                                   "issued from inside the backward, before autograd would run the hooks")
            out["loss"].backward()
        else:
            torch.autograd.backward(out["loss"], grad_tensors=[model.unit_grad(out["loss"].device)])
        if self.comm.active and self.overlap:
            H.host_action(self._flush_bucket)          # the tail of the backward (embedding gradients) that stayed below the merge threshold
        if self.comm.active and self.overlap and self.per_bucket_adamw and self._covers_live(st):
            # AdamW per bucket, in completion order: each launch waits (stream-level) only for its own bucket's all-reduce, so
            # the optimizer of the early buckets runs under the collectives of the late ones and only the last, small bucket
            # (the video-side embedding) is exposed
            # ... in TWO launches: every cross-stream wait costs ~15 us of queue time on the GPU, so the buckets that
            # completed early (a contiguous prefix of the flat buffer: head | layers ...) are stepped together, then the last one
            H.mark(H.PHASE_STEP_TAIL)
            self.opt.begin_step()
            works = self._bucket_works
            order = sorted(range(len(works)), key=lambda i: works[i][0])
            last = len(works) - 1
            early = [i for i in order if i != last]
            if early and works[early[-1]][1] <= works[last][0] and all(works[a][1] == works[b][0] for a, b in zip(early, early[1:])):
                H.host_action(functools.partial(self._wait_buckets, early))
                self.opt.step_range(works[early[0]][0], works[early[-1]][1])
                H.host_action(functools.partial(self._wait_buckets, [last]))
                self.opt.step_range(works[last][0], works[last][1])
            else:          # not a prefix + tail (two backbones interleave their buckets): one launch per bucket
                for i in range(len(works)):
                    H.host_action(functools.partial(self._wait_buckets, [i]))
                    self.opt.step_range(works[i][0], works[i][1])
            self.opt.end_step()
            return out
        if self.comm.active:
            if H.RECORDER is not None:
                raise RuntimeError("record(): the data-parallel step is recorded in its default schedule only (overlapped bucket "
                                   "all-reduces, per-bucket AdamW)")
            if not self.overlap:
                self._reduce_dense(0, st.n_live)
            for _, _, works in self._bucket_works:
                for w in works:
                    w.wait()
            self.comm.finish()
        H.mark(H.PHASE_STEP_TAIL)
        if self.opt.__dict__.get("_early"):
            E.join_aux(st)          # the tables' first pass (aux stream) is complete before their listed rows are stepped
        self.opt.step()
        return out

    # ---- the step as recorded launch sequences replayed from C (include/segmm_hip.h "Recorded launch sequences")
    def record(self, batch: Dict[str, torch.Tensor], warmup: int = 3, prev_batch: Optional[Dict[str, torch.Tensor]] = None):
        """Record the launch sequence of one training step on ``batch``'s shapes: ``warmup`` eager steps (site scales calibrated,
        every persistent buffer allocated, both streams created), then ONE more eager step during which every C-ABI call is
        recorded -- entry point, arguments, stream slot, fork / join points of the two streams -- split into the phases of the
        step (step begin, embedding, each encoder layer, head + loss, their backwards, optimizer tail).  The step's allocations
        come from a private memory pool kept for the life of the recording, the per-step state lives on the device
        (``device_state``): the recorded arguments are valid for every later step, except the batch's own tensors, whose
        addresses are patched per step.  ``run_recorded(batch)`` then enqueues a step with one C call per phase
        (segmm_step_begin, segmm_embed_fwd, segmm_layer_fwd, ...): the eager two-stream schedule without the per-launch host
        work.  Results are bit-identical to ``train_step`` in the same mode.
        ``prev_batch``: the batch of the ``train_step`` that ran immediately before (same shapes, other tensors).  With it NO
        warm-up step runs -- the caller has already stepped (the sites are calibrated) -- so the only optimisation step ``record``
        takes is the recorded one, on ``batch``: what an epoch loop needs (:func:`fit` with ``recorded=True``)."""
        if not self.device_state:
            raise RuntimeError("record() needs Trainer(device_state=True): the per-step state must live on the device")
        model, st = self.model, self.model._store
        if self._param_hooks():
            raise RuntimeError("record(): parameter hooks only fire when the gradients travel through autograd; use train_step")
        for k, v in batch.items():
            if torch.is_tensor(v) and not v.is_contiguous():
                raise RuntimeError("record(): batch[%r] is not contiguous" % k)
            # the zero-copy paths of the step: anything else is converted by a torch kernel, which a replay would drop (the
            # conversions also raise by themselves while recording: hipabi.torch_fallback)
            if torch.is_tensor(v) and v.is_floating_point() and v.dtype != torch.float32:
                raise RuntimeError("record(): batch[%r] is %s; the recorded step takes fp32 features" % (k, v.dtype))
            if torch.is_tensor(v) and k.endswith("_mask") and v.dtype != torch.bool:
                raise RuntimeError("record(): batch[%r] is %s; the recorded step takes bool masks" % (k, v.dtype))
            if torch.is_tensor(v) and not v.is_floating_point() and v.dtype not in (torch.bool, torch.int64):
                raise RuntimeError("record(): batch[%r] is %s; the recorded step takes int64 ids and labels" % (k, v.dtype))
        # the step may name tensors of the PREVIOUS step's batch (id mode: the rows the last backward scattered into the table
        # gradient are cleared by id list): the last warm-up step runs on a copy of the batch, so that such pointers can be told
        # from the current batch's and re-based to the previous batch at replay
        if prev_batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    pv = prev_batch.get(k)
                    if not torch.is_tensor(pv) or pv.shape != v.shape or pv.dtype != v.dtype or (v.numel() and pv.data_ptr() == v.data_ptr()):
                        raise RuntimeError("record(prev_batch=...): prev_batch[%r] must be another tensor of the same shape and dtype" % k)
            prev = prev_batch
        else:
            prev = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            for _ in range(max(warmup - 1, 0)):
                self.train_step(batch)
            self.train_step(prev)
        torch.cuda.synchronize()
        main = H._stream()
        side = st.side_stream().cuda_stream
        pool = torch.cuda.MemPool()
        aux = st.__dict__.get("_aux_stream")
        rec = H.Recorder(main, side, aux.cuda_stream if aux is not None else 0)
        H.RECORDER = rec
        # The recorded step must carry the per-step weight split (segmm_wsplit_p32): ParamStore.refresh_planes skips it when the planes
        # are current -- which they are when an EVALUATION pass ran since the last optimizer step (it split the new weights itself).
        # A step recorded in that state would replay on stale weight planes for ever (found by fit(recorded=True): validation, then
        # record).  Force the split into the recording.
        st._planes_key = None
        try:
            with st.rec_pool(pool):
                out = self.train_step(batch)
        finally:
            H.RECORDER = None
        torch.cuda.synchronize()
        phases = rec.finish()
        # the batch's tensors: every recorded pointer that falls inside one of them is re-based per step
        spans = [(k, v.data_ptr(), v.numel() * v.element_size(), tuple(v.shape), v.dtype) for k, v in batch.items() if torch.is_tensor(v) and v.numel()]
        pspans = [(k, v.data_ptr(), v.numel() * v.element_size()) for k, v in prev.items() if torch.is_tensor(v) and v.numel()]
        relocs = []
        for pi, (ph, arr) in enumerate(phases):
            if arr is None:
                continue
            # only the slots recorded as POINTER arguments (Recorder.pslots): an integer / float slot -- a seed, a byte count, the
            # bit pattern of lr -- that happens to fall inside a batch tensor's address range must never be rewritten
            for ci, ai in rec.pslots[pi]:
                pv = arr[ci].a[ai].p
                for k, base, nb, _, _ in spans:
                    if base <= pv < base + nb:
                        relocs.append((arr, ci, ai, k, pv - base, 0))
                        break
                else:
                    for k, base, nb in pspans:
                        if base <= pv < base + nb:
                            relocs.append((arr, ci, ai, k, pv - base, 1))
                            break
        # host-side state of the eager path that names a batch tensor (id mode: the id list whose table-gradient rows the NEXT
        # backward clears): kept current by run_recorded, so that eager steps and recorded steps can be mixed
        tab_keys = []
        for ek, (gptr, t) in getattr(st, "_tab_rows", {}).items():
            for k, base, nb, _, _ in spans:
                if base <= t.data_ptr() < base + nb and t.data_ptr() == base and t.numel() * t.element_size() == nb:
                    tab_keys.append((ek, gptr, k))
        evs = tuple(torch.cuda.Event() for _ in range(4))
        for e in evs:
            e.record()
        streams = [main, side] + ([aux.cuda_stream] if aux is not None else [])
        self._recorded = dict(phases=phases, keep=rec.keep, pool=pool, prev_batch=batch, tab_keys=tab_keys, out=out, relocs=relocs, spans={k: (sh, dt) for k, _, _, sh, dt in spans},
                              main=main, events=evs, table=H.stream_table(streams, [e.cuda_event for e in evs[:2 * (len(streams) - 1)]]),
                              n_cmds=sum(ph.n_cmds for ph, a in phases if a is not None),
                              hyper=(self.opt.lr, self.opt.wd, tuple(self.opt.betas), self.opt.eps))
        return out

    def _timed_plan(self, r):
        """The recorded phases cut at every GEMM / attention command (bench.py's timed replay): [("host", fn) | ("run", Phase) |
        ("timed", Phase of ONE command, stream slot, record)], record = the tuple hipabi's GEMM_PROFILE / ATTN_PROFILE entries
        start with.  The sub-phases alias the recorded command arrays (the per-step re-basing of batch pointers reaches them)."""
        plan = r.get("timed_plan")
        if plan is not None:
            return plan
        names = {v: k for k, v in H.op_ids().items()}
        plan = []
        for ph, arr in r["phases"]:
            if arr is None:
                plan.append(("host", ph))
                continue
            def sub(lo, hi, ph=ph, arr=arr):
                return H.Phase(kind=ph.kind, backbone=ph.backbone, layer=ph.layer, n_cmds=hi - lo,
                               cmds=ctypes.cast(ctypes.byref(arr, lo * ctypes.sizeof(H.Cmd)), ctypes.POINTER(H.Cmd)))
            lo = 0
            for ci in range(ph.n_cmds):
                c = arr[ci]
                nm = names.get(c.op)
                rec = None
                if nm == "segmm_gemm_p":
                    if not (c.a[20].i & 2):          # (a repair launch does no work normally: not a GEMM of the roofline)
                        rec = ("gemm", (10 + c.a[0].i, c.a[1].i, c.a[2].i, c.a[3].i))
                elif nm in ("segmm_gemm", "segmm_gemm_h"):
                    rec = ("gemm", (c.a[0].i, c.a[1].i, c.a[2].i, c.a[3].i))
                elif nm == "segmm_attn_fwd":
                    rec = ("attn", ("fwd",) + tuple(c.a[k].i for k in range(6)))
                elif nm == "segmm_attn_bwd":
                    phase = c.a[39].i
                    repair = bool(c.a[40].p) and bool(H.AttnPlanes.from_address(c.a[40].p).flags & H.ATTN_REPAIR)
                    kind = "bwd" if phase == 0 else "bwd4r" if repair else "bwd4" if phase >= 4 else "bwd%d" % phase
                    B_, H_, dh_, Lq_, La_, Lb_ = (c.a[k].i for k in range(6))
                    rec = ("attn", (kind, B_, H_, dh_, Lq_, 0 if phase == 6 else La_, 0 if phase == 5 else Lb_))
                if rec is None:
                    continue
                if ci > lo:
                    plan.append(("run", sub(lo, ci)))
                plan.append(("timed", sub(ci, ci + 1), c.stream, rec))
                lo = ci + 1
            if ph.n_cmds > lo:
                plan.append(("run", sub(lo, ph.n_cmds)))
        r["timed_plan"] = plan
        return plan

    def run_recorded(self, batch: Dict[str, torch.Tensor], timed=None):
        """One training step from the recorded launch sequences (see :meth:`record`): patch the batch's addresses, then one C
        call per phase.  Must be called with torch's current stream = the stream ``record`` ran on.
        ``timed`` = (gemm list, attention list): the same launch sequence with the phases cut at every GEMM / attention command and
        a HIP-event pair around each on the command's own stream -- entries like hipabi.GEMM_PROFILE / ATTN_PROFILE; bench.py's
        roofline pass (a few more C calls and event records per step; results identical)."""
        r = self.__dict__.get("_recorded")
        if r is None:
            raise RuntimeError("run_recorded() before record()")
        H.step_bind(self._step_state)
        if (self.opt.lr, self.opt.wd, tuple(self.opt.betas), self.opt.eps) != r["hyper"]:
            # the recorded segmm_adamw / segmm_adamw_table commands carry lr, weight decay, betas and eps BY VALUE
            raise RuntimeError("run_recorded(): the optimizer's hyperparameters changed since record() (%r -> %r): record() again"
                               % (r["hyper"], (self.opt.lr, self.opt.wd, tuple(self.opt.betas), self.opt.eps)))
        for k, (sh, dt) in r["spans"].items():
            v = batch[k]
            if tuple(v.shape) != sh or v.dtype != dt or not v.is_contiguous():
                raise RuntimeError("run_recorded(): batch[%r] is %s %s, the step was recorded for %s %s" % (k, tuple(v.shape), v.dtype, sh, dt))
        pb = r["prev_batch"]
        for arr, ci, ai, k, off, lag in r["relocs"]:
            arr[ci].a[ai].p = (pb if lag else batch)[k].data_ptr() + off
        r["prev_batch"] = batch          # (also keeps the tensors a lag-1 pointer names alive until the next step has run)
        main, table = r["main"], r["table"]
        if H._stream() != main:
            raise RuntimeError("run_recorded(): the current stream is not the stream the step was recorded on")
        if timed is not None:
            streams = r.get("timed_streams")
            if streams is None:
                streams = r["timed_streams"] = [torch.cuda.current_stream()] + [torch.cuda.ExternalStream(int(h)) for h in list(table[0])[1:]]
            for item in self._timed_plan(r):
                if item[0] == "host":
                    item[1]()
                elif item[0] == "run":
                    H.run_phase(item[1], table)
                else:
                    _, ph1, slot, (fam, head) = item
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(streams[slot])
                    H.run_phase(ph1, table)
                    e1.record(streams[slot])
                    timed[0 if fam == "gemm" else 1].append(head + (e0, e1))
        for ph, arr in (r["phases"] if timed is None else ()):
            if arr is None:
                ph()          # a host action of the step (data-parallel collective / wait), at its place in the launch order
            else:
                H.run_phase(ph, table)
        # mirror the host side effects of the eager step (FusedAdamW.begin_step / end_step; the id list of the table rows this
        # step's backward scattered into)
        self.opt.step_count += 1
        st = self.model._store
        st.fused_version += 1
        for ek, gptr, k in r["tab_keys"]:
            st._tab_rows[ek] = (gptr, batch[k].reshape(-1))
        return r["out"]

    def _param_hooks(self) -> bool:
        plist = self.__dict__.get("_hook_plist")
        if plist is None:
            plist = self._hook_plist = list(self.model.parameters())
        for p in plist:
            if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                return True
        return False

    def _covers_live(self, st):
        """The buckets whose hooks fired tile [0, n_live) exactly (always, for a backward over the whole model)."""
        spans = sorted((s, e) for s, e, _ in self._bucket_works)
        pos = 0
        for s, e in spans:
            if s != pos:
                return False
            pos = e
        return pos == st.n_live

    @torch.no_grad()
    def test_model(self, test_batches, eval_type_list, ckpt: Optional["CheckPointer"] = None, threshold: float = 0.5, top_k_mask: int = 0,
                   top_k_permutation: int = 1, save_logits: bool = False, train_videos: Optional[set] = None, draw_case: int = 0):
        """The test phase of the reference trainer (main_for_seq_leave_earlystop_SegMM.py:365-459): reload the BEST checkpoint
        (``ckpt.load_checkpoint(model, optimizer, mode='best')``, :366-367), eval mode, ``mode="inference"`` over the test split,
        interests = sigmoid(logits) * exposure_prob (:402-403), pred_label = interests > threshold (:404), every batch through
        ``main_eval_batch`` (:415), then ``compute_final_result`` (:434).  ``train_videos`` (the reference's ``--eval_cold``
        set of photo ids seen in training, :417-427): also the cold / hot splits.  ``save_logits`` (:412-414): the
        [interests | gt | user_id | photo_id] rows.  Returns a dict {"final", "results_list"[, "cold_final", "hot_final",
        "cold_count_inter", "hot_count_inter"][, "saved_logits"]}."""
        import argparse
        from .my_evaluation import main_eval_batch
        model = self.model
        if ckpt is not None:
            load_dict = ckpt.load_checkpoint(model, self.opt, mode="best")
            model.load_state_dict(load_dict["model"])
        model.eval()
        margs = argparse.Namespace(TOP_K_mask=top_k_mask, TOP_K_permutation=top_k_permutation, draw_case=draw_case)

        def fresh():
            r = {}
            for eval_type in eval_type_list:
                r[eval_type] = []
                r["view_lengths"] = []
            return r

        results_list = fresh()
        cold = train_videos is not None
        cold_results, hot_results, cold_n, hot_n = (fresh(), fresh(), 0, 0) if cold else (None, None, 0, 0)
        saved = []
        exposure = None
        for batch in test_batches:
            out = self.eval_step(batch, mode="inference")
            logits = out["logits"]
            if exposure is None:
                exposure = torch.tensor(model.exposure_prob, dtype=torch.float32, device=logits.device)[: logits.shape[1]]
            interests = torch.sigmoid(logits) * exposure
            pred_label = torch.where(interests > threshold, 1.0, 0.0)
            gt = out["gt"]
            if save_logits:
                saved.append(torch.cat((interests.cpu(), gt.cpu().to(torch.float32), batch["user_id"].reshape(-1, 1).cpu().to(torch.float32),
                                        batch["photo_id"].reshape(-1, 1).cpu().to(torch.float32)), dim=1))
            results_list = main_eval_batch(margs, interests, gt, pred_label, results_list, type="inference")
            if cold:
                pids = batch["photo_id"].reshape(-1).cpu().tolist()
                ci = [i for i, p_ in enumerate(pids) if p_ not in train_videos]
                hi = [i for i, p_ in enumerate(pids) if p_ in train_videos]
                cold_n += len(ci)
                hot_n += len(hi)
                if ci:          # (the reference indexes with an empty tensor and lets main_eval_batch see 0 rows; nothing is appended then)
                    ix = torch.tensor(ci, device=interests.device)
                    cold_results = main_eval_batch(margs, interests[ix], gt[ix], pred_label[ix], cold_results, type="inference")
                if hi:
                    ix = torch.tensor(hi, device=interests.device)
                    hot_results = main_eval_batch(margs, interests[ix], gt[ix], pred_label[ix], hot_results, type="inference")
        res = {"final": compute_final_result(results_list), "results_list": results_list}
        if cold:
            res.update(cold_final=compute_final_result(cold_results), hot_final=compute_final_result(hot_results),
                       cold_count_inter=cold_n, hot_count_inter=hot_n)
        if save_logits:
            res["saved_logits"] = torch.cat(saved, dim=0) if saved else torch.empty((0, 0))
        return res

    def fit(self, train_batches, valid_batches, epochs, **kw):
        """The reference's train / validate / checkpoint / early-stop loop (module-level :func:`fit`)."""
        return fit(self, train_batches, valid_batches, epochs, **kw)

    @torch.no_grad()
    def eval_step(self, batch, mode="inference"):
        model = self.model
        model.eval()
        usr, um, vid, vm = self._features(batch)
        return model(usr_image=usr, usr_id=batch["user_identity_id"], usr_mask=um, vid_image=vid,
                     vid_id=batch["photo_identity_id"], vid_mask=vm, gt=batch["label"], mode=mode)


    @torch.no_grad()
    def valid_model(self, batches, metrics=("valid_loss", "HR@1", "HR@3", "HR@5", "HR@10", "NDCG@1", "NDCG@3", "NDCG@5", "NDCG@10"),
                    permutation=1, top_k_mask=False):
        """``valid_model`` of the reference trainer (main_for_seq_leave_earlystop_SegMM.py:132-186): eval-mode forward
        with ``mode="train"`` (loss + logits), interests = sigmoid(logits) * exposure_prob, leave-rank metrics per batch,
        mean over batches.  The ranks are computed on the device (csrc/evalops.h); only B integers per batch reach the
        host instead of the [B, S] interests, labels and masks."""
        from .my_evaluation import TOP_K_leave_device
        model = self.model
        exposure = None
        acc = {k: [] for k in metrics}
        for batch in batches:
            out = self.eval_step(batch, mode="train")
            logits, gt = out["logits"], out["gt"]
            if exposure is None:
                exposure = torch.tensor(model.exposure_prob, dtype=torch.float32, device=logits.device)[: logits.shape[1]]
            interests = torch.sigmoid(logits) * exposure
            dp = self.comm.active
            ev = TOP_K_leave_device(interests, gt, permutation=permutation, masked=top_k_mask,
                                    gather=self.comm.gather_ints if dp else None)
            for k in acc:
                # data parallel: every rank's loss terms are already divided by the GLOBAL normalisers, so the global value
                # is their sum over ranks; the rank metrics above are those of the global batch
                if k == "valid_loss":
                    acc[k].append(float(self.comm.sum_scalar(out["loss"].detach().clone())))
                elif k in out and k not in ("gt", "logits"):
                    acc[k].append(float(self.comm.sum_scalar(out[k].detach().clone())) if dp and k not in ("mse", "mse2") else float(out[k]))
                elif k in ev:
                    acc[k].append(float(ev[k]))
        return {k: (sum(v) / len(v) if v else float("nan")) for k, v in acc.items()}


def early_stop_reached(metric_history: List[float], early_stop: int) -> bool:
    """The two early-stop tests of the reference loop on the monitored validation metric (higher is better),
    main_for_seq_leave_earlystop_SegMM.py:336-352: (i) none of the last ``early_stop`` validations beat the first of that window,
    (ii) the best validation lies more than ``early_stop`` validations back."""
    m = list(metric_history)
    if early_stop <= 0 or not m:
        return False
    if len(m) > early_stop:
        lst = m[-early_stop:]
        if all(lst[0] >= y for y in lst[1:]):
            return True
    return len(m) - m.index(max(m)) > early_stop


def compute_final_result(results_list):
    """``compute_final_result`` of the reference trainer (main_for_seq_leave_earlystop_SegMM.py:188-210): LeaveMSE = mean squared
    error of the predicted against the true view lengths, every other list its mean; 'TOP_K' and 'view_lengths' carry no number."""
    final = {}
    if "LeaveMSE" in results_list:
        vl, pv = results_list["view_lengths"], results_list["LeaveMSE"]
        final["LeaveMSE"] = float(sum((float(a) - float(b)) ** 2 for a, b in zip(vl, pv)) / len(vl)) if vl else float("nan")
    for eval_type, vals in results_list.items():
        if eval_type in ("TOP_K", "LeaveMSE", "view_lengths"):
            continue
        if not isinstance(vals, list) or not vals:
            continue
        final[eval_type] = sum(vals) / len(vals)
    return final


def fit(trainer: "Trainer", train_batches, valid_batches, epochs: int, valid_step: int = 30, early_stop: int = 0,
        main_metric: str = "NDCG@5", ckpt: Optional["CheckPointer"] = None, logging_step: int = 0, log=None,
        permutation: int = 1, top_k_mask: bool = False, metrics=None, recorded: bool = False, eager_steps: int = 3):
    """The train / validate / checkpoint / early-stop loop of the reference trainer around ``Trainer.train_step``
    (main_for_seq_leave_earlystop_SegMM.py:247-354): one validation over ``valid_batches`` BEFORE training, then per epoch every
    ``valid_step`` local steps a validation whose ``main_metric`` (mean over the validation batches, :179-181) drives
    ``ckpt.save_checkpoint(..., metric_vals={"main_metric": ...})`` (:333) and the early-stop tests (:336-352).
    ``train_batches``: a list of batch dicts, or a callable ``epoch -> iterable of batch dicts`` (an epoch of the DataLoader).
    The loss is read on the host only where the reference's bookkeeping needs a number (validation points, logging steps).
    ``recorded=True`` (needs ``Trainer(device_state=True)``; single process or data parallel): the first ``eager_steps`` steps run
    launch by launch (they calibrate the delayed scales), the next step is RECORDED while it runs (``Trainer.record`` with the
    previous batch: no extra optimisation step), every later batch of the same shapes is enqueued from the recorded launch
    sequences (``run_recorded``: one C call per phase, bit-identical to the eager step); a batch of another shape (the short last
    batch of an epoch) takes the eager step.  Every batch is still stepped on exactly once, in order.
    Returns the history dict the reference calls ``total_valid_loss_metrics`` (+ "stopped_epoch", "global_step")."""
    metrics = list(metrics) if metrics is not None else ["valid_loss", "HR@1", "HR@3", "HR@5", "HR@10", "NDCG@1", "NDCG@3", "NDCG@5", "NDCG@10"]
    hist: Dict[str, list] = {"train_loss": [0.0]}
    for k in metrics:
        hist[k] = []

    def validate():
        vm = trainer.valid_model(valid_batches, metrics=tuple(metrics), permutation=permutation, top_k_mask=top_k_mask)
        for k in metrics:
            hist[k].append(vm[k])
        return vm

    if recorded and not trainer.device_state:
        raise RuntimeError("fit(recorded=True) needs Trainer(device_state=True)")

    def shapes_of(b):
        return {k: (tuple(v.shape), v.dtype) for k, v in b.items() if torch.is_tensor(v) and v.numel()}

    def step(batch, prev, n_done):
        """One optimisation step on ``batch``: eager, recording, or from the recording."""
        if not recorded:
            return trainer.train_step(batch)
        r = trainer.__dict__.get("_recorded")
        if r is not None:
            if shapes_of(batch) == r["spans"] and all(v.is_contiguous() for v in batch.values() if torch.is_tensor(v)):
                return trainer.run_recorded(batch)
            return trainer.train_step(batch)
        if n_done >= eager_steps and prev is not None and shapes_of(prev) == shapes_of(batch):
            return trainer.record(batch, prev_batch=prev)
        return trainer.train_step(batch)

    validate()                                           # "Evaluation Before Training" (:247-249)
    global_step, stop, stopped_epoch = 0, False, None
    prev_batch = None
    for epoch in range(epochs):
        if stop:
            break
        it = train_batches(epoch) if callable(train_batches) else train_batches
        epoch_losses = []
        for local_step, batch in enumerate(it):
            out = step(batch, prev_batch, global_step)
            prev_batch = batch
            # (a recorded step returns the SAME loss tensor every step: keep a copy where the epoch mean is going to be logged)
            epoch_losses.append(out["loss"].detach().clone() if (recorded and log is not None) else out["loss"].detach())
            global_step += 1
            if logging_step and (local_step + 1) % logging_step == 0 and log is not None:
                log("Train_loss: %f, Global_step: %d" % (float(epoch_losses[-1]), global_step))
            if (local_step + 1) % valid_step == 0:
                hist["train_loss"].append(float(epoch_losses[-1]))
                validate()
                avg = hist[main_metric][-1]
                if log is not None:
                    log("Valid_loss: %s, %s: %s, Global_step: %d" % (hist["valid_loss"][-1] if "valid_loss" in hist else None, main_metric, avg, global_step))
                if ckpt is not None and trainer.comm.rank == 0:
                    ckpt.save_checkpoint(model=trainer.model, optimizer=trainer.opt, num_epochs=epoch, metric_vals={"main_metric": avg})
                if early_stop_reached(hist[main_metric], early_stop):
                    stop, stopped_epoch = True, epoch
                    break
        if log is not None and epoch_losses:
            log("Epoch: %d, Epoch Loss: %f" % (epoch, float(torch.stack(epoch_losses).mean())))
    hist["stopped_epoch"] = stopped_epoch
    hist["global_step"] = global_step
    return hist


class CheckPointer:
    """kn_util CheckPointer file format (kn_util/nn_utils/checkpoint.py:11-75): a dict
    {model, optimizer, num_epochs, metrics} in ckpt-latest.pth and ckpt-best-ep{E}-{metric}.pth."""

    def __init__(self, monitor, work_dir, mode="min", **_ignored):
        import os
        self.monitor, self.work_dir, self.mode, self.best_metric = monitor, work_dir, mode, None
        os.makedirs(work_dir, exist_ok=True)
        self.ckpt_latest = os.path.join(work_dir, "ckpt-latest.pth")
        self.ckpt_best = os.path.join(work_dir, "ckpt-best-ep{}-{}.pth")

    def better(self, new, orig):
        if orig is None:
            return True
        return new < orig if self.mode == "min" else new > orig

    def save_checkpoint(self, model, optimizer, num_epochs, metric_vals=None, **_):
        import glob
        import os
        sd = dict(model={k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                  optimizer=optimizer.state_dict(), num_epochs=num_epochs, metrics=metric_vals)
        torch.save(sd, self.ckpt_latest)
        if metric_vals and self.better(metric_vals[self.monitor], self.best_metric):
            self.best_metric = metric_vals[self.monitor]
            for f in glob.glob(self.ckpt_best.format("*", "*")):
                os.remove(f)
            torch.save(sd, self.ckpt_best.format(num_epochs, round(float(self.best_metric), 6)))
            return True
        return False

    def load_checkpoint(self, model, optimizer, mode="latest", **_):
        import glob
        fn = self.ckpt_latest if mode == "latest" else glob.glob(self.ckpt_best.format("*", "*"))[0]
        sd = torch.load(fn, map_location="cpu", weights_only=False)
        model.load_state_dict(sd["model"])
        if optimizer is not None:
            optimizer.load_state_dict(sd["optimizer"])
        return sd
