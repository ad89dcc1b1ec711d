"""Fused HIP execution of the segment-interest model (host orchestration over the C ABI).

This is the host half of the drop-in: it owns the flat parameter layout, the exact-liveness rule of the
reference's encoder (layers 0..N-3 both sides, layer N-2 video side only, layer N-1
dead) and the sequence of kernel launches of the forward and the hand-derived backward.  PyTorch is
used for memory, streams and the autograd glue (two ``torch.autograd.Function``s: one per backbone, one
for head+loss); every FLOP of the path runs in ``libsegmm_hip.so``.

Per encoder layer (video side; the user side mirrors it when live), reference file:line in brackets:

    Yv = Xv . [Wq_v2v | Wq_t2v | Wk_v2v | Wv_v2v (| Wk_v2t | Wv_v2t)]^T + b        one fused GEMM  [encoder.py:95-104,50-62]
    Yu = Xu . [Wk_t2v | Wv_t2v (| Wq_v2t | Wq_t2t | Wk_t2t | Wv_t2t)]^T + b        one fused GEMM
    Av = attention(Q from Yv, K/V blocks from Yv and Yu, masks, logits dropout)    [encoder.py:64-71,138-156]
    R1 = Xv + dropout(Av . Wff^T + b) ; X1 = LN(R1)                                GEMM epilogue + LN [:163-171]
    H  = dropout(gelu(X1 . W0^T + b0)) ; R2 = X1 + dropout(H . W1^T + b1) ; X2 = LN(R2)            [:202-203; mlp.py:17-23]
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import hipabi as H

# dropout site ids (distinct hash streams): site = backbone*4096 + layer*32 + kind
K_EMB_V, K_EMB_U, K_ATT_V, K_ATT_U, K_AO_V, K_AO_U, K_MI_V, K_MI_U, K_MO_V, K_MO_U = range(1, 11)
MLP_INNER_DROPOUT = 0.1      # kn_util MLP default (kn_util/nn_utils/layers/mlp.py:8), never overridden


def _site(bb, layer, kind):
    return bb * 4096 + layer * 32 + kind


def _j(prefix, name):
    return prefix + name


# =============================================================================================== parameter store
MLP_VARIANTS = ("CrossMLP", "SelfMLP", "w/oAtt")       # ablations that replace the encoder (encoder.py:392-400,503-511)
K_MLP0 = 11                                             # dropout kinds 11.. : hidden layers of the ablation MLP_Block
# streams of the device-side draws that replace the reference's host draws in a device-state / recorded step (same hash, own sites)
SITE_NOUSER_FEAT, SITE_NOUSER_IDS, SITE_NOPOS = 3 * 4096 + 1, 3 * 4096 + 2, 3 * 4096 + 3
POOL_BINS = 40                                          # nn.AdaptiveAvgPool1d(40), encoder.py:396


def attn_mode(bb) -> str:
    """Which key blocks a query attends to (encoder.py:108-161): 'joint' = [video | user] keys (the model proper, also
    'noPos' / 'noUser'), 'cross' = the OTHER side's keys only ('CrossAtt'), 'self' = the own side's keys only
    ('SelfAtt', 'noUser_SelfAtt'; the user branch then returns None and user tokens never change, :172-173,316-319)."""
    t = getattr(bb, "ablation_type", "ours")
    return "cross" if "CrossAtt" in t else ("self" if "SelfAtt" in t else "joint")


def layer_plan(mode: str, full: bool) -> Tuple[List[str], List[str]]:
    """(vid_projs, usr_projs): the Linears of ``cross_attn`` that are LIVE and read video tokens / user tokens, in the order
    they are fused into one GEMM each (xxx_proj.0 = query, .1 = key, .2 = value projection; t2v = user keys for video
    queries, v2t = video keys for user queries).  ``full``: the user side of this layer is live as well."""
    if mode == "joint":
        vid = ["v2v_proj.0", "t2v_proj.0", "v2v_proj.1", "v2v_proj.2"] + (["v2t_proj.1", "v2t_proj.2"] if full else [])
        usr = ["t2v_proj.1", "t2v_proj.2"] + (["v2t_proj.0", "t2t_proj.0", "t2t_proj.1", "t2t_proj.2"] if full else [])
    elif mode == "cross":
        vid = ["t2v_proj.0"] + (["v2t_proj.1", "v2t_proj.2"] if full else [])
        usr = ["t2v_proj.1", "t2v_proj.2"] + (["v2t_proj.0"] if full else [])
    else:
        vid, usr = ["v2v_proj.0", "v2v_proj.1", "v2v_proj.2"], []
    return vid, usr


def mlp_linears(bb) -> List[str]:
    """Names (relative to the backbone) of the Linears of the ablation ``encoder_mlp`` in forward order; the last one is the
    output Linear, the others are followed by ReLU (+ Dropout when dropout > 0) -- encoder.py:210-252."""
    return ["encoder_mlp.mlp.%d" % i for i, m in enumerate(bb.encoder_mlp.mlp) if isinstance(m, torch.nn.Linear)]


def backbone_layout(prefix: str, bb) -> Tuple[List[Tuple[str, List[List[str]]]], None]:
    """Live parameter groups of one backbone in BACKWARD-completion order, as (bucket, groups).
    A group is a list of tensors that must be adjacent in memory (a fused GEMM reads them as one)."""
    N = bb.n_layers
    abl = getattr(bb, "ablation_type", "ours")
    mode = attn_mode(bb)
    buckets = []
    usr_live = N >= 2 and mode != "self"       # with N == 1 (or SelfAtt) the user embedding only feeds dead compute
    use_pe = bool(getattr(bb, "use_pe", 1))    # --use_pe 0 (encoder.py:450-471 else branches): the positional tables stay dead
    if abl in MLP_VARIANTS:
        usr_live = abl == "CrossMLP"
        if abl != "w/oAtt":                    # w/oAtt builds encoder_mlp but never calls it (encoder.py:397-400,510-511)
            groups = []
            for n in reversed(mlp_linears(bb)):
                groups += [[prefix + n + ".weight"], [prefix + n + ".bias"]]
            buckets.append((prefix + "mlp", groups))
    else:
        for i in reversed(range(max(N - 1, 0))):
            full = i < N - 2 and mode != "self"
            L = "%sencoder.layers.%d." % (prefix, i)
            ca = L + "cross_attn."
            vidP, usrP = layer_plan(mode, full)
            singles = [ca + "ff_vid", ca + "ln_vid", L + "ff_vid.layers.0", L + "ff_vid.layers.1", L + "ln_vid"]
            if full:
                singles += [ca + "ff_usr", ca + "ln_usr", L + "ff_usr.layers.0", L + "ff_usr.layers.1", L + "ln_usr"]
            groups = [[ca + n + ".weight" for n in vidP], [ca + n + ".bias" for n in vidP]]
            if usrP:
                groups += [[ca + n + ".weight" for n in usrP], [ca + n + ".bias" for n in usrP]]
            for s in singles:
                groups += [[s + ".weight"], [s + ".bias"]]
            buckets.append(("%slayer%d" % (prefix, i), groups))
    # the user side of the embedding completes first in the backward (its weight gradient queues on the side stream while the
    # video side is still being differentiated): its own bucket, so its all-reduce starts before the last weight gradient
    if usr_live:
        emb_u = [[prefix + "usr_proj.weight"]]
        if not bb.id_usr:
            emb_u += [[prefix + "usr_proj.bias"]]
        if use_pe:
            emb_u += [[prefix + "usr_pe.weight"]]
        emb_u += [[prefix + "usr_ln.weight"], [prefix + "usr_ln.bias"]]
        buckets.append((prefix + "embed_u", emb_u))
    emb = [[prefix + "vid_proj.weight"]]
    if bb.id_vid:
        emb += [[prefix + "frameid_proj.weight"], [prefix + "frameid_proj.bias"]]
    else:
        emb += [[prefix + "vid_proj.bias"]]
    if use_pe:
        emb += [[prefix + "vid_pe.weight"]]
    emb += [[prefix + "vid_ln.weight"], [prefix + "vid_ln.bias"]]
    buckets.append((prefix + "embed", emb))
    return buckets


class ParamStore:
    """All parameters of a model re-pointed into ONE flat fp32 buffer: live tensors first (in
    backward-completion order, bucketed), dead tensors after.  ``nn.Parameter.data`` become views, so
    ``state_dict`` / ``load_state_dict`` / any torch optimizer keep working, while fused GEMMs read
    concatenated weights in place and AdamW / the gradient all-reduce see contiguous ranges."""

    def __init__(self, root, standalone_backbone=False):
        self.root = root
        self.standalone = standalone_backbone
        self.flat = None
        self.gflat = None
        self.index: Dict[str, Tuple[int, int]] = {}
        self.buckets: List[Tuple[str, int, int]] = []
        self.n_live = 0
        self.live_names: List[str] = []
        self.scratch: Dict[tuple, torch.Tensor] = {}
        self._params = None
        self.bucket_hook = None      # callable(bucket_name): set by the data-parallel trainer
        # data-parallel id mode: callable(ids [B] int64, rows [B, w]) -> closure yielding (ids of all ranks [G*B], rows of all
        # ranks [G*B, w]) -- the collective is asynchronous, the closure waits for it (DPComm.gather_rows); set by the trainer.  The table gradient is then built from the gathered per-row gradients on every rank and the
        # table's range is left out of the dense gradient all-reduce (table_ranges()).
        self.row_exchange = None
        self._side_stream = None
        self._on_side = False
        self.overlap = os.environ.get("SEGMM_OVERLAP", "1") != "0"
        # weight gradients of a layer's ff / MLP Linears enqueued (side stream) right before the attention backward instead of
        # next to their input-gradient GEMMs; the LayerNorm-backward column sums on the side stream.  Both were measured +-0 / -2 %
        # while the host enqueued the step launch by launch; with the recorded step (host 0.3 ms) and the round-4 attention
        # kernels: config 2 133.5 -> 136.2 k/s together (same box, alternating runs: +1.2 % and +0.4 % alone) -- the attention
        # backward leaves CUs idle while its workgroups wait on memory, and a GEMM tile that gets such a CU is productive -- but
        # config 3 (20 segments: short attention launches, K = 512 GEMMs) 185.4 -> 181.8 k/s.  "auto" (default): on for
        # segment axes > 32 on the plane engine (BackboneRun.backward); SEGMM_DEFER_WGRAD / SEGMM_LN_SIDE = 0 / 1 force them.
        self.ln_pos = os.environ.get("SEGMM_LN_POS", "1") != "0"          # embedding LayerNorm backward leaves per-position sums (_ln_bwd)
        self.lazy_head_grad = os.environ.get("SEGMM_LAZY_HEAD_GRAD", "1") != "0"          # head gradient formed inside the first LayerNorm backward
        self.head_dot = os.environ.get("SEGMM_HEAD_DOT", "1") != "0"          # ... and the head's logits inside the last LayerNorm forward
        self._defer_wgrad_env = os.environ.get("SEGMM_DEFER_WGRAD", "auto")
        self._ln_side_env = os.environ.get("SEGMM_LN_SIDE", "auto")
        self.defer_wgrad = self._defer_wgrad_env not in ("0", "auto")
        self.tail_balance = os.environ.get("SEGMM_TAIL_BALANCE", "1") != "0"
        self.attn_planes_only = int(os.environ.get("SEGMM_ATTN_PLANES_ONLY", "1"))
        # round 5: the attention forward reads the Q / K / V planes the fused projection GEMMs write (csrc/attention_pl.h)
        # 1 (default): with the projection outputs Yv / Yu as planes ONLY (no fp32 copy: the planes-in backward reads them too, a repair
        # launch of the GEMM covers a wrong delayed scale); 2: planes beside the fp32 copy, forward only (measured -1.5 %); 0: off
        self.attn_pl = int(os.environ.get("SEGMM_ATT_PL", "1"))
        self.eu_planes_only = os.environ.get("SEGMM_EU_PLANES_ONLY", "1") != "0"          # user embedding as planes only (N = 2, trainer's step)
        self.head_side = os.environ.get("SEGMM_HEAD_SIDE", "1") != "0"
        self.input_planes_only = os.environ.get("SEGMM_INPUT_PLANES_ONLY", "1") != "0"
        self.attn_two_streams = os.environ.get("SEGMM_ATTN_TWO_STREAMS", "0") == "1"
        # attention backward as D-kernel, then dQ (third stream) next to dK/dV (main stream).  Measured (same box, alternating
        # runs): 80.3 k -> 79.6 k interactions/s, the union of the attention intervals unchanged at 1.10-1.14 ms/step -- the two
        # kernels share the same vector-memory pipeline and simply slow each other down.  OFF by default.
        self.attn_split = os.environ.get("SEGMM_ATTN_SPLIT", "0") != "0"
        self.attn_fused = os.environ.get("SEGMM_ATTN_FUSED", "1") != "0"      # fused dQ+dK+dV kernel (<= 12 key tiles per block)
        self._attn_stream = None
        # forward: user-token chain (input Linear -> LayerNorm -> fused user projection) on the side stream next to the
        # video-token chain.  On-the-fly engines: -0.3 % (off); plane engine (one workgroup per CU: the tail of one kernel and the
        # HBM-bound LayerNorms fill under the other chain's GEMMs): +1.2 %, same-box alternating runs (on)
        self.fwd_side = os.environ.get("SEGMM_FWD_SIDE", "1" if H.GEMM_ENGINE == H.ENGINE_F16X3P else "0") != "0"
        # full layers (N >= 3): the user-token chain of a layer (attention with user queries -> ff -> LayerNorm -> MLP -> LayerNorm,
        # and its backward) is independent of the video-token chain between the fused projections and the next layer: it runs on
        # the side stream.  At config 3 its GEMMs have M = 1024 rows (4 - 12 tiles on 256 CUs) and used to sit on the main stream
        # between the video side's kernels.
        self.usr_side = os.environ.get("SEGMM_USR_SIDE", "1") != "0"
        self.ln_side = self._ln_side_env not in ("0", "auto")          # (see defer_wgrad above)
        # pre-split bf16 planes of the weights for the bf16x6 GEMM engine: W planes (forward) and W^T planes (dgrad
        # in the NT form), refreshed when the parameters change (one split pass per optimizer step)
        # (fp16x3 engine: two fp16 planes scaled by one power of two derived from ``wamax``, the partial maxima of
        # |parameters|, which the GEMMs also need for weights they read as fp32)
        self.engine_p = H.GEMM_ENGINE == H.ENGINE_F16X3P      # plane-operand GEMMs (gemm_planes.h); implies the fp16x3 arithmetic
        self.engine_h = H.GEMM_ENGINE in (H.ENGINE_F16X3, H.ENGINE_F16X3P)
        self.use_planes = H.GEMM_ENGINE in (H.ENGINE_BF16X6, H.ENGINE_F16X3) and os.environ.get("SEGMM_PLANES", "1") != "0"
        self.wamax = None
        # engine_p: P32 planes of every weight matrix a GEMM reads (W: forward operand; W^T: the input-gradient GEMM in NT form),
        # one site header per matrix (own scale: LayerNorm gammas and biases are not part of any of them)
        self.wpt: Dict[str, "H.PT"] = {}
        self.wTpt: Dict[str, "H.PT"] = {}
        self._wmats: List[Tuple[str, int, int, int, bool]] = []      # (first parameter name, flat offset, rows, cols, needs W^T)
        # delayed scaling of producer-written planes (common.h PlaneOut): one persistent scale per tensor SITE (a named
        # activation / gradient of the model), refreshed at the end of every pass from the maxima that pass recorded
        # (segmm_scales_update); a site is "calibrated" once it has been produced at least once
        self.scaling = os.environ.get("SEGMM_SCALING", "delayed")      # delayed | exact (split pass after every producer) | always
        # the scale puts the maxima of the LAST pass at 2^target, in the middle of the window the consumers accept
        # (gemm_planes.h site_planes_ok: 2^-2 <= max * s < 2^16): 7 = 256x of headroom before an element overflows fp16 and
        # 512x before the lo terms of a SHRUNKEN tensor sink into the subnormals; outside the window the consuming GEMM takes
        # its fp32 fallback.  The attention-backward gradients of a nearly converged BPR model jump up to 45x from one batch to
        # the next (tools/overflow_sites.py; target 12 = 16x headroom took the fallback 38 times in 400 steps).
        self.scale_target = int(os.environ.get("SEGMM_SCALE_TARGET", "7"))
        # backward sites: scale predicted from the site's recorded gain x THIS step's max |d loss / d logits| (they are linear in it)
        self.loss_relative = os.environ.get("SEGMM_LOSS_RELATIVE", "1") != "0"
        self.site_index: Dict[str, int] = {}
        self.site_scale = None          # [MAX_SITES + 8] floats: scales, then [MAX_SITES] = count of overflowed tensors
        self.calibrated = set()
        self._site_idx_cache: Dict[tuple, torch.Tensor] = {}
        self.whdr = None
        self.wpl = self.wTpl = None
        self.wgrad_planes = 2 if os.environ.get("SEGMM_WGRAD", "x6") == "x3" else 3      # x3 = opt-in, see DESIGN.md
        self.wplanes = self.wTplanes = None
        self.fused_version = 0
        self._planes_key = None
        self._transposes: List[Tuple[int, int, int]] = []
        self._plane_ranges: List[Tuple[int, int]] = []

    MAX_SITES = 1024

    def site(self, name: str) -> int:
        i = self.site_index.get(name)
        if i is None:
            i = self.site_index[name] = len(self.site_index)
            if i >= self.MAX_SITES:
                raise RuntimeError("more than %d tensor sites" % self.MAX_SITES)
        return i

    # ---- site-header ring.  A header ([SITE_FLOATS]: scale, flag, AMAX_SLOTS partial maxima) must be ZERO when its producer
    # starts folding maxima into it, and is read until the last GEMM that consumes the tensor has run (the backward reads the
    # forward's).  Instead of one torch.zeros per pass (three fill launches per step), rows are handed out sequentially from a
    # ring that is zeroed a quarter at a time, when the cursor enters the quarter: the rows cleared then were handed out
    # >= 3/4 ring (dozens of steps) earlier.  One 2 MB fill every ~10-20 steps instead of 3-4 small ones per step.
    HDR_RING_ROWS = 8192

    def hdr_step_begin(self):
        """Step-arena mode (Trainer(device_state=True): the step is replayed from its recorded launch sequences, so every step must use the
        SAME header rows): hand the rows out from row 0 of the ARENA again and clear what the previous step used -- one fill
        launch per step, inside the step.  The arena is only active until ``hdr_step_end`` (the end of train_step): evaluation
        passes between training steps draw from the wrapping ring below, which recycles its rows (a validation round
        of >~100 batches used to exhaust the arena)."""
        if self.flat is None:
            self.ensure()
        dev = self.flat.device
        arena = self.__dict__.get("_hdr_arena")
        if arena is None or arena.device != dev:
            arena = self._hdr_arena = torch.zeros((self.HDR_RING_ROWS, H.SITE_FLOATS), dtype=torch.float32, device=dev)
            self._arena_used = 0
        elif self._arena_used:
            H.fill_zero(arena[:self._arena_used])
        self._arena_off, self._arena_used = 0, 0
        self.step_arena = True

    def hdr_step_end(self):
        self.step_arena = False

    def hdr_rows(self, n: int) -> torch.Tensor:
        dev = self.flat.device
        q = self.HDR_RING_ROWS // 4
        if self.__dict__.get("step_arena", False):
            if self._arena_off + n > self.HDR_RING_ROWS:
                raise RuntimeError("step arena: more than %d site headers in one step" % self.HDR_RING_ROWS)
            r0 = self._arena_off
            self._arena_off += n
            self._arena_used = max(self._arena_used, self._arena_off)
            return self._hdr_arena[r0:r0 + n]
        if n > q:
            return torch.zeros((n, H.SITE_FLOATS), dtype=torch.float32, device=dev)
        ring = getattr(self, "_hdr_ring", None)
        if ring is None or ring.device != dev:
            ring = self._hdr_ring = torch.zeros((self.HDR_RING_ROWS, H.SITE_FLOATS), dtype=torch.float32, device=dev)
            self._hdr_q, self._hdr_off = 0, 0
        if self._hdr_off + n > q:          # the rest of this quarter is too small: enter the next one, clearing it first
            self._hdr_q, self._hdr_off = (self._hdr_q + 1) % 4, 0
            H.fill_zero(ring[self._hdr_q * q:(self._hdr_q + 1) * q])
        r0 = self._hdr_q * q + self._hdr_off
        self._hdr_off += n
        return ring[r0:r0 + n]

    @contextlib.contextmanager
    def rec_pool(self, pool):
        """Allocation context of the step that is being RECORDED (Trainer.record): every tensor allocated while it is open -- by
        any thread: the autograd engine runs the backward on its own -- comes from the private memory pool ``pool``, which the
        trainer keeps for the life of the recording, so the addresses the recorded commands name stay reserved for the replays."""
        # torch's PUBLIC context (torch.cuda.use_mem_pool) routes the calling thread only; the backward of the recorded step
        # allocates on the autograd engine's thread.  The allocator hooks that route EVERY thread are the ones CUDA-graph capture
        # itself uses (torch/cuda/graphs.py); they are private, so their presence is checked and the failure says what to do.
        need = ("_cuda_beginAllocateToPool", "_cuda_endAllocateToPool", "_cuda_releasePool")
        missing = [n for n in need if not hasattr(torch._C, n)]
        if missing:
            raise RuntimeError("Trainer.record(): torch %s has no torch._C.%s (the all-thread allocate-to-pool hooks of torch 2.1 - 2.10 that "
                               "the recording uses to pin the step's buffers); use the per-launch step (Trainer.train_step / fit(recorded=False)) "
                               "with this torch build" % (torch.__version__, ", torch._C.".join(missing)))
        dev = self.flat.device.index if self.flat.device.index is not None else torch.cuda.current_device()
        torch._C._cuda_beginAllocateToPool(dev, pool.id)          # every allocation of every thread and stream
        try:
            yield
        finally:
            torch._C._cuda_endAllocateToPool(dev, pool.id)
            torch._C._cuda_releasePool(dev, pool.id)

    def const_arange(self, n: int, dtype) -> torch.Tensor:
        """arange(n) on the device, made once (never written afterwards)."""
        c = self.__dict__.setdefault("_consts", {})
        key = (n, dtype, self.flat.device)
        t = c.get(key)
        if t is None:
            t = c[key] = torch.arange(n, device=self.flat.device, dtype=dtype)
        return t

    def const_f32(self, value: float) -> torch.Tensor:
        """A device scalar holding ``value``, made once (a fixed plane scale)."""
        c = self.__dict__.setdefault("_consts", {})
        key = ("f32", float(value), self.flat.device)
        if key not in c:
            c[key] = torch.full((1,), float(value), dtype=torch.float32, device=self.flat.device)
        return c[key]

    def const_ones(self, shape, dtype) -> torch.Tensor:
        c = self.__dict__.setdefault("_consts", {})
        key = ("ones", tuple(shape), dtype, self.flat.device)
        t = c.get(key)
        if t is None:
            t = c[key] = torch.ones(tuple(shape), device=self.flat.device, dtype=dtype)
        return t

    def scales(self) -> torch.Tensor:
        if self.site_scale is None or self.site_scale.device != self.flat.device:
            # [0, MAX): scales | [MAX, MAX + 8): counters | [MAX + 8, 2 MAX + 8): loss-relative gains of the backward sites | gmax
            self.site_scale = torch.zeros((2 * self.MAX_SITES + 16,), dtype=torch.float32, device=self.flat.device)
            self.calibrated = set()
        return self.site_scale

    def scale_ptr(self, name: str, delayed: bool):
        """Device address of the delayed scale of site ``name`` -- or None when its planes must come from an exact split pass
        (exact mode, or the site has never been produced: its scale is unknown)."""
        if not (delayed and self.engine_p) or name not in self.calibrated:
            return None
        return self.scales().data_ptr() + 4 * self.site(name)

    def gains(self) -> torch.Tensor:
        return self.scales()[self.MAX_SITES + 8:2 * self.MAX_SITES + 8]

    def gmax(self) -> torch.Tensor:
        return self.scales()[2 * self.MAX_SITES + 8:2 * self.MAX_SITES + 9]

    def update_scales(self, arena_t, site_names, n_rows, backward=False):
        """End of a pass: fold the pass's partial maxima into the site scales (one tiny launch).  ``backward``: also record every
        site's size relative to this step's max |d loss / d logits| (the head's loss_finish launch predicts the next step's
        scale from it)."""
        if n_rows == 0 or not self.engine_p:
            return
        key = tuple(site_names[:n_rows])
        idx = self._site_idx_cache.get(key)
        if idx is None or idx.device != arena_t.device:
            idx = self._site_idx_cache[key] = torch.tensor([-1 if n is None else self.site(n) for n in key], dtype=torch.int32,
                                                           device=arena_t.device)
        sc = self.scales()
        # the head measured gmax in THIS step (HeadLossFn.forward sets the flag, the next BackboneFn.forward clears it: every
        # backbone of a two-tower model records its gains, not only the first one whose backward arena closes)
        rel = backward and self.loss_relative and self.__dict__.get("_gmax_fresh", False)
        H.scales_update(arena_t, idx, n_rows, sc, sc[self.MAX_SITES:], self.scale_target, gain=self.gains() if rel else None,
                        gmax=self.gmax() if rel else None)
        self.calibrated.update(n for n in key if n is not None)

    def overflow_count(self) -> int:
        """Number of plane tensors whose delayed scale overflowed so far (host sync; diagnostics and tests)."""
        return int(self.scales()[self.MAX_SITES].item())

    # -- second HIP stream for weight/bias gradients (see class SideWork)
    # The side / auxiliary streams are ONE pair per device and process, shared by every store: HIP maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues as they are created, and a second model in the same process (bench.py's exact-fp32 leg, a
    # test that builds two trainers) that brought its own pair ended up sharing queues with the main stream -- its step ran 13 %
    # slower than the same model alone (49.5 k -> 43.6 k interactions/s on the fp32 engine).
    _shared_streams = {}

    @classmethod
    def _shared_stream(cls, device, kind):
        key = (str(device), kind)
        s = cls._shared_streams.get(key)
        if s is None:
            # LOWEST priority: the side stream's weight-gradient GEMMs fill idle CUs, they must not starve the main
            # stream's kernels (a 49 us LayerNorm backward was seen taking 470 us next to a same-priority GEMM)
            s = cls._shared_streams[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("SEGMM_SIDE_PRIORITY", "1")))
        return s

    def side_stream(self):
        if self._side_stream is None or self._side_stream.device != self.flat.device:
            self._side_stream = self._shared_stream(self.flat.device, "side")
        return self._side_stream

    def aux_stream(self):
        a = self.__dict__.get("_aux_stream")
        if a is None or a.device != self.flat.device:
            a = self._aux_stream = self._shared_stream(self.flat.device, "aux")
        return a

    def attn_stream(self):
        if self._attn_stream is None or self._attn_stream.device != self.flat.device:
            self._attn_stream = torch.cuda.Stream(device=self.flat.device)
        return self._attn_stream

    # -- layout
    def _layout(self):
        if self.standalone:
            return backbone_layout("", self.root)
        return self.root._param_buckets()

    def ensure(self):
        # inside Trainer.train_step the layout and the weight planes were checked once at the start of the step and nothing
        # outside the step can touch the parameters until it returns: the later calls of the same step are free
        if self.__dict__.get("_trusted") and self.flat is not None:
            return
        params = self._params
        if params is None:
            params = self._params = dict(self.root.named_parameters())
        if self.flat is not None:
            ok = True
            base = self.flat.data_ptr()
            for name, (off, n) in self.index.items():
                if params[name].data_ptr() != base + 4 * off:
                    ok = False
                    break
            if ok:
                self.refresh_planes()
                return
        self._build(params)
        self.refresh_planes()

    def refresh_planes(self):
        if not (self.use_planes or self.engine_h):
            return
        # staleness: in-place updates through torch (optimizer.step, load_state_dict) bump the parameters' version
        # counters; the fused AdamW kernel bumps ``fused_version`` itself
        key = (self.flat._version, self.fused_version, self.flat.data_ptr(), sum(self._params[n]._version for n in self.live_names))
        if key == self._planes_key:
            return
        dev = self.flat.device
        if self.engine_p:
            self._refresh_p32(dev)
        # only parameters that ARE GEMM operands are scanned / split (an id-mode item table of 90 M floats is neither)
        ranges = self._plane_ranges
        if self.engine_h and not (self.engine_p and self._all_p32):
            if self.wamax is None or self.wamax.device != dev:
                self.wamax = torch.zeros((H.AMAX_PARTS,), dtype=torch.float32, device=dev)
            per = H.AMAX_PARTS // max(len(ranges), 1)
            for i, (off, n) in enumerate(ranges):
                H.absmax(self.flat, 1, n, n, off=off, out=self.wamax[i * per:(i + 1) * per])
        if self.use_planes:
            npl, dt = (2, torch.float16) if self.engine_h else (3, torch.bfloat16)
            if self.wplanes is None or self.wplanes.shape != (npl, self.n_live) or self.wplanes.device != dev or self.wplanes.dtype != dt:
                self.wplanes = torch.empty((npl, self.n_live), dtype=dt, device=dev)
                self.wTplanes = torch.empty((npl, self.n_live), dtype=dt, device=dev)
            for off, n in ranges:
                if self.engine_h:
                    H.split2h(self.flat, self.wplanes, n, self.wamax, x_off=off, p_off=off)
                else:
                    H.split3(self.flat, self.wplanes, n, x_off=off, p_off=off)
            for off, R, Cc in self._transposes:
                if self.engine_h:
                    H.split2h_transpose(self.flat, R, Cc, Cc, self.wTplanes, self.wamax, x_off=off, p_off=off)
                else:
                    H.split3_transpose(self.flat, R, Cc, Cc, self.wTplanes, x_off=off, p_off=off)
        self._planes_key = key

    def _refresh_p32(self, dev):
        """Per optimizer step: absmax + exact split of every GEMM weight matrix into P32 planes (and of its transpose)."""
        mats = self._wmats
        if self.whdr is None or self.whdr.device != dev or self.whdr.shape[0] != max(len(mats), 1):
            self.whdr = torch.zeros((max(len(mats), 1), H.SITE_FLOATS), dtype=torch.float32, device=dev)
            self.wpl = torch.empty((2 * self.n_live,), dtype=torch.float16, device=dev)
            self.wTpl = torch.empty((2 * self.n_live,), dtype=torch.float16, device=dev)
            self.wpt, self.wTpt = {}, {}
            for i, (name, off, R, Cc, tr) in enumerate(mats):
                self.wpt[name] = H.PT(self.wpl, self.whdr[i], R, Cc, ld2=2 * Cc, p_off=2 * off)
                if tr:
                    self.wTpt[name] = H.PT(self.wTpl, self.whdr[i], Cc, R, ld2=2 * R, p_off=2 * off)
            # descriptor table of segmm_wsplit_p32: {int64 offset; int32 R, C, transpose, first tile, tile columns}
            import struct
            recs, tile0 = b"", 0
            for (name, off, R, Cc, tr) in mats:
                tcols = Cc // 32
                recs += struct.pack("<qiiiii", off, R, Cc, int(tr), tile0, tcols) + b"\0" * 4          # 32-byte records
                tile0 += ((R + 31) // 32) * tcols
            self._wdesc = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(dev) if mats else None
            self._wtiles = tile0
        if mats:          # (no zero-fill: wabsmax rewrites every slot of every header)
            H.wsplit_p32(self.flat, self._wdesc, len(mats), self._wtiles, self.whdr, self.wpl, self.wTpl)

    def _build(self, params):
        first = next(iter(params.values()))
        dev = first.device
        if dev.type != "cuda":
            raise RuntimeError("segmminterest_amd: the model must be on a HIP device (got %s); there is no CPU path" % dev)
        for n_, p in params.items():
            if p.dtype != torch.float32:
                raise RuntimeError("parameter %s is %s; the path computes in fp32" % (n_, p.dtype))
        order, buckets, off = [], [], 0
        seen = set()
        for bname, groups in self._layout():
            start = off
            for grp in groups:
                off = (off + 31) & ~31        # 32 floats: the P32 plane blocks (and 16-byte alignment of every plane format)
                for name in grp:
                    if name not in params:
                        raise KeyError("layout names unknown parameter %s" % name)
                    if name in seen:
                        raise KeyError("parameter %s listed twice" % name)
                    seen.add(name)
                    n = params[name].numel()
                    if len(grp) > 1 and n % 4:
                        raise RuntimeError("fused group member %s has %d elements (not a multiple of 4)" % (name, n))
                    order.append((name, off, n))
                    off += n
            off = (off + 7) & ~7
            buckets.append((bname, start, off))
        n_live = off
        live_names = [n for n, _, _ in order]
        for name, p in params.items():
            if name not in seen:
                off = (off + 3) & ~3
                order.append((name, off, p.numel()))
                off += p.numel()
        flat = torch.zeros(((off + 3) & ~3,), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, o, n in order:
                p = params[name]
                flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
        self.flat = flat
        self.gflat = torch.zeros((n_live,), dtype=torch.float32, device=dev)
        self._tab_rows = {}          # id-table gradient: (gradient view address, rows scattered into it by the last backward)
        self.index = {name: (o, n) for name, o, n in order}
        self.buckets = buckets
        self.n_live = n_live
        self.live_names = live_names
        self.scratch = {}
        # weight matrices whose transpose is needed by the input-gradient GEMMs: every fused projection group and
        # every d x d Linear inside an encoder layer (the embedding / head weights have no dgrad GEMM)
        self._transposes = []
        for _, groups in self._layout():
            for grp in groups:
                n0 = grp[0]
                if n0.endswith(".weight") and (".encoder.layers." in "." + n0 or ".encoder_mlp.mlp." in "." + n0) \
                        and params[n0].dim() == 2 and "ln_" not in n0:
                    self._transposes.append((self.index[n0][0], sum(params[n].shape[0] for n in grp), params[n0].shape[1]))
        # flat ranges read by GEMMs as the weight operand: the encoder-layer buckets and the input
        # projections when they are Linears (image mode); Embedding tables, positional tables and the head are not
        ranges = [(s0, e0 - s0) for bname, s0, e0 in buckets
                  if not (bname.endswith("embed") or bname.endswith("embed_u")) and bname != "head" and e0 > s0]
        for name in live_names:
            if name.endswith("vid_proj.weight") or name.endswith("usr_proj.weight"):
                mod = dict(self.root.named_modules()).get(name.rsplit(".", 1)[0])
                if isinstance(mod, torch.nn.Linear):
                    o, n = self.index[name]
                    ranges.append((o, (n + 7) & ~7))
        # merge into at most 8 ranges (one slice of the partial-maxima array each)
        ranges.sort()
        merged = []
        for o, n in ranges:
            if merged and o <= merged[-1][0] + merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], o + n - merged[-1][0]))
            else:
                merged.append((o, n))
        while len(merged) > 8:
            a, b = merged[-2], merged[-1]
            merged[-2:] = [(a[0], b[0] + b[1] - a[0])]
        self._plane_ranges = [(o, min(n, n_live - o)) for o, n in merged]
        self._planes_key = None
        # engine_p: the weight matrices as GEMMs read them -- fused projection groups as ONE [sum rows, cols] matrix
        self._wmats, self._all_p32 = [], True
        tr_offs = {o for o, _, _ in self._transposes}
        mods = dict(self.root.named_modules())
        for bname, groups in self._layout():
            for grp in groups:
                n0 = grp[0]
                if not n0.endswith(".weight") or params[n0].dim() != 2 or "ln_" in n0 or "_pe." in n0 or bname == "head":
                    continue
                in_layer = ".encoder.layers." in "." + n0 or ".encoder_mlp.mlp." in "." + n0
                is_lin = isinstance(mods.get(n0.rsplit(".", 1)[0]), torch.nn.Linear)
                if not (in_layer or (is_lin and (n0.endswith("vid_proj.weight") or n0.endswith("usr_proj.weight")))):
                    continue
                R, Cc = sum(params[n].shape[0] for n in grp), params[n0].shape[1]
                o = self.index[n0][0]
                tr = o in tr_offs
                if Cc % 32 == 0 and (not tr or R % 32 == 0):
                    self._wmats.append((n0, o, R, Cc, tr))
                else:
                    self._all_p32 = False
        self.whdr = None

    def table_ranges(self) -> List[Tuple[int, int]]:
        """[start, end) of the live id-embedding tables inside the flat gradient buffer (id mode: vid_proj / usr_proj
        are nn.Embedding): the ranges a data-parallel step exchanges as rows instead of all-reducing densely."""
        mods = dict(self.root.named_modules())
        out = []
        for name in self.live_names:
            if name.endswith("vid_proj.weight") or name.endswith("usr_proj.weight"):
                if isinstance(mods.get(name.rsplit(".", 1)[0]), torch.nn.Embedding):
                    o, n = self.index[name]
                    out.append((o, o + n))
        return sorted(out)

    # -- access
    def p(self, name) -> torch.Tensor:
        return self._params[name].data

    def g(self, name, gbuf=None) -> torch.Tensor:
        o, n = self.index[name]
        gb = self.gflat if gbuf is None else gbuf
        return gb[o:o + n].view(self._params[name].shape)

    def buf(self, key, shape, dtype=torch.float32) -> torch.Tensor:
        """Persistent scratch (single stream => safe to reuse across launches and steps)."""
        k = (key, tuple(shape), dtype)
        t = self.scratch.get(k)
        if t is None:
            t = self.scratch[k] = torch.empty(shape, dtype=dtype, device=self.flat.device)
        return t


def _empty(ref, *shape, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=ref.device)


_SPLIT_TARGET = int(os.environ.get("SEGMM_SPLIT_TARGET", "1024"))     # workgroups a split-K weight gradient aims for


_BN_ENV = os.environ.get("SEGMM_GEMM_BN", "")


def _splits_for(M, N, K):
    """Split-K factor of a weight-gradient GEMM (TN); mirrors the tile-width choice of segmm_gemm_h (capi.hip)."""
    wide = (H.GEMM_ENGINE == H.ENGINE_F16X3 and N > 128 and _BN_ENV != "128"
            and (_BN_ENV == "256" or ((M + 127) // 128) * ((N + 255) // 256) >= 36))
    bn = 256 if wide else 128
    tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
    ktiles = (K + 31) // 32
    return max(1, min(32, ktiles, (_SPLIT_TARGET + tiles - 1) // tiles))


_SPLIT_TARGET_P = int(os.environ.get("SEGMM_SPLIT_TARGET_P", "256"))     # workgroups a plane-operand weight gradient aims for


_SPLIT_TARGET_FEW = int(os.environ.get("SEGMM_SPLIT_TARGET_FEW", "256"))     # ... of the few-tile matrices that take gemm_pl_tn4 (capi.hip: <= 9 tiles, width >= 768)


def _splits_for_p(M, N, K):
    """Split-K factor of a plane-operand weight-gradient GEMM (256 x 256 tiles, one workgroup per CU): fill the 256 CUs once."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    ktiles = (K + 31) // 32
    target = _SPLIT_TARGET_FEW if (tiles <= 9 and M >= 768 and N >= 768) else _SPLIT_TARGET_P
    return max(1, min(64, ktiles, target // tiles if tiles <= target else 1))


@contextlib.contextmanager
def side_work(store):
    """Weight- and bias-gradient launches of a Linear are independent of its input-gradient GEMM.  They are
    enqueued on a second HIP stream (forked from the main stream here, joined by ``join_side``) so that
    their workgroups fill the CUs that the last, partial round of a 960-workgroup dgrad GEMM leaves idle
    (profiles/README.md: 1.875 rounds on 512 slots).  Buffers these launches read carry the layer index in
    their scratch name, so the main stream never overwrites them before the join."""
    if not store.overlap or store._on_side:          # (nested: the caller already runs on the side stream)
        yield
        return
    main = torch.cuda.current_stream()
    side = store.side_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    if H.RECORDER is not None:
        H.RECORDER.pseudo(H.OP_FORK, 1)
    store._on_side = True
    try:
        with torch.cuda.stream(side):
            yield
    finally:
        store._on_side = False


def side_or_defer(store, fn, deferred):
    """Run ``fn`` (weight/bias-gradient launches) on the side stream now, or park it in ``deferred`` to be flushed later
    with ``flush_deferred`` -- used to move GEMM work next to the latency-bound attention backward instead of next to
    other GEMMs (two MFMA-bound kernels side by side gain nothing, a GEMM next to the attention kernels is free)."""
    if deferred is not None and store.overlap:
        deferred.append(fn)
        return
    with side_work(store):
        fn()


def flush_deferred(store, deferred):
    if deferred:
        with side_work(store):
            for fn in deferred:
                fn()
        deferred.clear()


@contextlib.contextmanager
def aux_work(store):
    """A third HIP stream (lowest priority) for work that is independent of the whole forward / backward -- the early pass of
    the id-table optimizer (FusedAdamW.table_early): forked from the main stream here, joined by ``join_aux``."""
    main = torch.cuda.current_stream()
    aux = store.aux_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    aux.wait_event(ev)
    if H.RECORDER is not None:
        H.RECORDER.pseudo(H.OP_FORK, 2)
    with torch.cuda.stream(aux):
        yield


def join_aux(store):
    if store.__dict__.get("_aux_stream") is not None:
        torch.cuda.current_stream().wait_stream(store._aux_stream)
        if H.RECORDER is not None:
            H.RECORDER.pseudo(H.OP_JOIN, 2)


def join_side(store):
    if store.overlap and store._side_stream is not None:
        torch.cuda.current_stream().wait_stream(store._side_stream)
        if H.RECORDER is not None:
            H.RECORDER.pseudo(H.OP_JOIN, 1)


class Act:
    """A tensor that some GEMM reads: the fp32 values ``t`` ([rows, cols] row-major), its site header ``hdr`` (scale, overflow
    flag, partial maxima -- None on the f32 / bf16x6 engines) and, on the plane engine, the P32 fp16 planes ``planes``
    ([rows, 2 cols]; None when cols is not a multiple of 32: such operands go through the on-the-fly kernel)."""
    __slots__ = ("t", "hdr", "rows", "cols", "planes", "filled", "po", "scale_ptr", "no_f32")

    def __init__(self, t, hdr, rows, cols, planes=None):
        self.t, self.hdr, self.rows, self.cols, self.planes, self.filled = t, hdr, rows, cols, planes, False
        self.po = None              # hipabi.PO when the producer writes the planes itself (delayed scale)
        self.scale_ptr = None
        self.no_f32 = False         # the producers wrote planes only (and repaired them if need be): consumers get no fp32 fallback

    @property
    def slots(self):
        return None if self.hdr is None else self.hdr[H.SITE_HDR:]

    def pt(self, c0=0, ncols=None):
        ncols = self.cols - c0 if ncols is None else ncols
        return H.PT(self.planes, self.hdr, self.rows, ncols, ld2=2 * self.cols, p_off=2 * c0, f32=None if self.no_f32 else self.t,
                    ldf=self.cols, f_off=c0)


class AmaxArena:
    """fp16x3 engines: zeroed site headers ([SITE_FLOATS] rows: scale, flag, AMAX_SLOTS partial maxima), one per tensor that
    a GEMM will read; its producer kernel folds max|x| into the slots (hipabi / common.h), the consuming GEMMs derive the
    tensor's power-of-two scale from them.  One allocation + one fill per forward and per backward; ``new()`` returns None
    on the other engines, which turns every amax argument into a no-op."""

    def __init__(self, store, n):
        self.t = store.hdr_rows(n) if store.engine_h else None          # clean rows of the store's header ring
        self.i = 0
        self.sites: List[Optional[str]] = []

    def new(self, site=None):
        if self.t is None:
            return None
        if self.i >= self.t.shape[0]:
            raise RuntimeError("AmaxArena exhausted (%d rows)" % self.t.shape[0])
        r = self.t[self.i]
        self.i += 1
        self.sites.append(site)
        return r

    def close(self, store, backward=False):
        """End of the pass: the site scales of the next pass."""
        if self.t is not None:
            store.update_scales(self.t, self.sites, self.i, backward=backward)


def new_act(store, arena, rows, cols, t=None, key=None, planes=True, site=None, delayed=False):
    """An Act with a fresh site header; ``t`` given or allocated (``key``: persistent scratch name instead of a new tensor).
    ``planes=False``: no GEMM reads it on the plane engine (only its fp32 values / maxima are wanted).
    ``site`` names the tensor for delayed scaling; with ``delayed`` and a calibrated site the Act carries a plane output
    (``po``) for its producer kernel, otherwise ``finish_act`` makes the planes with an exact split pass."""
    dev = store.flat.device
    if t is None:
        t = store.buf(key, (rows, cols)) if key is not None else torch.empty((rows, cols), dtype=torch.float32, device=dev)
    want = planes
    planes = None
    if want and store.engine_p and cols % 32 == 0:
        planes = store.buf(key + ":pl", (rows, 2 * cols), torch.float16) if key is not None else \
            torch.empty((rows, 2 * cols), dtype=torch.float16, device=dev)
    a = Act(t, arena.new(site), rows, cols, planes)
    if planes is not None and site is not None:
        a.scale_ptr = store.scale_ptr(site, delayed)
        if a.scale_ptr is not None:
            a.po = H.PO(planes, 2 * cols, a.hdr, a.scale_ptr)
    return a


def produced(act):
    """The producer kernel of ``act`` was given its plane output (``po``): nothing left for finish_act to do."""
    if act.po is not None:
        act.filled = True
    return act


def finish_act(store, act):
    """Called when the producer(s) of ``act`` have been enqueued: makes the planes unless the producer wrote them itself."""
    if act.planes is not None and not act.filled:
        H.split_p32(act.t, act.rows, act.cols, act.cols, act.planes, 2 * act.cols, act.hdr, mode=0)
        act.filled = True
    return act


def _needs_f32(act, what):
    """A launch is about to read ``act.t``: refuse if its producers wrote planes only (engine._layer_bwd, planes-only protocol)."""
    if getattr(act, "no_f32", False):
        raise RuntimeError("%s reads the fp32 copy of an operand whose producers wrote planes only (attention gradients: set "
                           "SEGMM_ATTN_PLANES_ONLY=0; L1-normalised input features: SEGMM_INPUT_PLANES_ONLY=0 -- needed together with "
                           "SEGMM_FEW_TILES / a non-plane GEMM engine)" % what)


def _wgrad(store, dY, y_off, X, x_off, Mrows, n_out, n_in, gW, accumulate=False, gb=None):
    """gW[n_out, n_in] (+)= dY[:, y_off:y_off+n_out]^T . X[:, x_off:x_off+n_in]  (split-K over tokens); dY, X: Act.
    ``gb``: the bias gradient [n_out] = column sums of the same dY columns -- formed inside the weight-gradient kernel on the
    plane engine (one more MFMA pair per k-step in a third of the workgroups), by a column-sum pass otherwise."""
    if store.engine_p and dY.planes is not None and X.planes is not None and n_out % 32 == 0 and n_in % 32 == 0:
        splits = _splits_for_p(n_out, n_in, Mrows)
        ws = store.buf("splitk_ws_side" if store._on_side else "splitk_ws", (max(splits, 1) * (n_out * n_in + n_out),)) if splits > 1 else None
        H.gemm_p(H.LAYOUT_TN, n_out, n_in, Mrows, dY.pt(y_off, n_out), X.pt(x_off, n_in), gW, n_in, splits=splits, workspace=ws,
                 accumulate=accumulate, colsum_out=gb)
        return
    _needs_f32(dY, "the on-the-fly weight-gradient GEMM")
    _needs_f32(X, "the on-the-fly weight-gradient GEMM")
    if gb is not None:
        _colsum(store, dY.t, dY.cols, Mrows, n_out, gb, x_off=y_off, accumulate=accumulate)
    splits = _splits_for(n_out, n_in, Mrows)
    ws = store.buf("splitk_ws_side" if store._on_side else "splitk_ws", (max(splits, 1) * n_out * n_in,)) if splits > 1 else None
    H.gemm(H.LAYOUT_TN, n_out, n_in, Mrows, dY.t, dY.cols, X.t, X.cols, gW, n_in, splits=splits, workspace=ws,
           accumulate=accumulate, a_off=y_off, b_off=x_off, nplanes=store.wgrad_planes if H.GEMM_ENGINE == H.ENGINE_BF16X6 else 3,
           a_amax=dY.slots, b_amax=X.slots)


# Launches with fewer 256 x 256 output tiles than this go to the 128 x 128 on-the-fly kernel.  Round 2: 48 (the plane kernel
# had one tile shape and left most CUs idle).  Round 3: 0 -- gemm_pl_nt8 picks 256 x 128 tiles for such launches and is faster
# on every config-3 shape (user-side GEMMs with M = 1024: 44-189 us -> 37-111 us; 160.0 -> 168.6 k interactions/s)
_FEW_TILES = int(os.environ.get("SEGMM_FEW_TILES", "0"))


def _few_tiles(M, N):
    return ((M + 255) // 256) * ((N + 255) // 256) < _FEW_TILES


def _lin_fwd(store, M, N, K, X, wname, out, ldo, c_act=None, **kw):
    """out[M,N] = X[M,K] . W[N,K]^T (+ epilogue); X: Act; W = the parameter (or fused group starting at) ``wname``;
    ``c_act``: the Act that ``out`` belongs to (receives the partial maxima of |out|)."""
    w = store.wpt.get(wname) if store.engine_p else None
    if not (w is not None and X.planes is not None and not _few_tiles(M, N)):
        _needs_f32(X, "the on-the-fly forward GEMM")
    if w is not None and X.planes is not None and _few_tiles(M, N):
        # a handful of 256 x 256 tiles would leave most of the 256 CUs idle (config 3: 1024 user tokens): the 128 x 128
        # on-the-fly kernel has 4x the workgroups; it takes the fp32 operands and the same partial maxima
        H.gemm(H.LAYOUT_NT, M, N, K, X.t, K, store.p(wname), K, out, ldo, a_amax=X.slots, b_amax=w.hdr[H.SITE_HDR:],
               c_amax=None if c_act is None else c_act.slots, **kw)
        return
    if out is None and not (w is not None and X.planes is not None and not _few_tiles(M, N) and c_act is not None and c_act.po is not None):
        raise RuntimeError("a planes-only output needs the plane GEMM with a calibrated output site (%s)" % wname)
    if w is not None and X.planes is not None:
        if c_act is not None and c_act.po is not None:
            H.gemm_p(H.LAYOUT_NT, M, N, K, X.pt(), w, out, ldo, c_pt=c_act.pt(), c_scale_ptr=c_act.scale_ptr, write_c=out is not None, **kw)
            if out is None:
                # planes only (the projection outputs the planes-in attention kernels read): no fp32 copy a consumer could fall
                # back on, so the REPAIR launch follows -- workgroups that read the site header and leave, unless the delayed scale
                # turned out wrong, in which case the planes are rewritten with the exact scale of the recorded maxima
                H.gemm_p(H.LAYOUT_NT, M, N, K, X.pt(), w, None, ldo, c_pt=c_act.pt(), c_scale_ptr=c_act.scale_ptr, write_c=False, repair=True, **kw)
            c_act.filled = True
        else:
            H.gemm_p(H.LAYOUT_NT, M, N, K, X.pt(), w, out, ldo, c_hdr=None if c_act is None else c_act.hdr, **kw)
        return
    if store.use_planes and K % 8 == 0:
        kw["b_planes"] = (store.wplanes, store.index[wname][0])
    H.gemm(H.LAYOUT_NT, M, N, K, X.t, K, store.p(wname), K, out, ldo, a_amax=X.slots, b_amax=store.wamax,
           c_amax=None if c_act is None else c_act.slots, **kw)


def _lin_dgrad(store, M, n_in, n_out, dY, wname, out, c_act=None, **kw):
    """out[M,n_in] = dY[M,n_out] . W[n_out,n_in] (+ epilogue); dY: Act.  With W^T planes this is the NT form (both operands
    k-contiguous), otherwise the NN layout on the fp32 weights."""
    wT = store.wTpt.get(wname) if store.engine_p else None
    if not (wT is not None and dY.planes is not None and not _few_tiles(M, n_in)):
        _needs_f32(dY, "the on-the-fly input-gradient GEMM")
    if wT is not None and dY.planes is not None and _few_tiles(M, n_in):
        H.gemm(H.LAYOUT_NN, M, n_in, n_out, dY.t, n_out, store.p(wname), n_in, out, n_in, a_amax=dY.slots, b_amax=wT.hdr[H.SITE_HDR:],
               c_amax=None if c_act is None else c_act.slots, **kw)
        return
    if wT is not None and dY.planes is not None:
        if c_act is not None and c_act.po is not None:
            H.gemm_p(H.LAYOUT_NT, M, n_in, n_out, dY.pt(), wT, out, n_in, c_pt=c_act.pt(), c_scale_ptr=c_act.scale_ptr, **kw)
            c_act.filled = True
        else:
            H.gemm_p(H.LAYOUT_NT, M, n_in, n_out, dY.pt(), wT, out, n_in, c_hdr=None if c_act is None else c_act.hdr, **kw)
        return
    ca = None if c_act is None else c_act.slots
    if store.use_planes and n_out % 8 == 0:
        H.gemm(H.LAYOUT_NT, M, n_in, n_out, dY.t, n_out, None, n_out, out, n_in, b_planes=(store.wTplanes, store.index[wname][0]),
               a_amax=dY.slots, b_amax=store.wamax, c_amax=ca, **kw)
    else:
        H.gemm(H.LAYOUT_NN, M, n_in, n_out, dY.t, n_out, store.p(wname), n_in, out, n_in, a_amax=dY.slots, b_amax=store.wamax,
               c_amax=ca, **kw)


def _colsum(store, X, ld, M, N, out, x_off=0, w=None, accumulate=False):
    ws = store.buf("colsum_ws_side" if store._on_side else "colsum_ws", (H.colsum_chunks(M) * N,))
    H.colsum(X, ld, M, N, out, ws, w=w, accumulate=accumulate, x_off=x_off)


def _ln_bwd(store, dy, x, mean, rstd, gname, bname, gbuf, dx, dx_drop, rows, d, drop_y=(0.0, 0), drop_b=(0.0, 0), seed=0,
            amax=None, dsum_to=None, po=None, pos_period=0, dy_outer=None):
    """LayerNorm backward + its affine gradients.  ``dsum_to``: gradient tensor that receives the column sums of the
    forwarded gradient (dx_drop, or dx): the bias gradient of the Linear feeding this LayerNorm's residual branch,
    accumulated inside the same kernel instead of by a second pass over [rows, d].
    ``pos_period`` = L (embedding LayerNorms, rows = B * L): the launch takes the per-position grid and the per-wave sums of dx
    it leaves are returned ([4 * parts, d], partial row p = position p mod L; None when no such grid exists) -- the
    positional-embedding gradient then is a sum over ~40 partial rows per position instead of a pass over dx."""
    # (worth it when the second pass it saves is long: short sequences / small tensors keep the plain column sum)
    pparts = H.layernorm_bwd_pos_parts(rows, pos_period, d) if (pos_period >= 8 and store.ln_pos and rows * d >= (1 << 23)) else 0
    parts = pparts if pparts > 0 else H.layernorm_bwd_parts(rows, d)
    # partial buffers named after the parameter so that their reductions MAY run on the side stream (SEGMM_LN_SIDE=1;
    # measured 2 % slower than keeping these tiny launches on the main stream, so off by default)
    pg = store.buf("ln_pg:" + gname, (parts, d))
    pb = store.buf("ln_pb:" + gname, (parts, d))
    ps = store.buf("ln_ps:" + gname, (parts, d)) if dsum_to is not None else None
    pp = None
    if pparts > 0:
        pp = store.buf("ln_pp:" + gname, (4 * pparts, d))
        H.layernorm_bwd_pos(dy, x, mean, rstd, store.p(gname), dx, dx_drop, pg, pb, pp, pos_period, drop_y_p=drop_y[0], drop_y_site=drop_y[1],
                            drop_b_p=drop_b[0], drop_b_site=drop_b[1], seed=seed, amax=amax, part_dsum=ps, po=po)
    else:
        if dy_outer is not None:          # dy[row, c] = dl[row] * w[c] (the interest head's gradient), formed inside the launch
            H.layernorm_bwd_outer(dy_outer[0], dy_outer[1], x, mean, rstd, store.p(gname), dx, dx_drop, pg, pb, drop_y_p=drop_y[0],
                                  drop_y_site=drop_y[1], drop_b_p=drop_b[0], drop_b_site=drop_b[1], seed=seed, amax=amax, part_dsum=ps, po=po)
        else:
            H.layernorm_bwd(dy, x, mean, rstd, store.p(gname), dx, dx_drop, pg, pb, drop_y_p=drop_y[0], drop_y_site=drop_y[1],
                            drop_b_p=drop_b[0], drop_b_site=drop_b[1], seed=seed, amax=amax, part_dsum=ps, po=po)
    outs = [store.g(gname, gbuf), store.g(bname, gbuf)] + ([dsum_to] if ps is not None else [])
    with (side_work(store) if store.ln_side else contextlib.nullcontext()):
        ws = store.buf("colsum3_ws_side" if store._on_side else "colsum3_ws", (3 * H.colsum_chunks(parts) * d,))
        Xs = [pg, pb] + ([ps] if ps is not None else [])
        H.colsum3(Xs, d, parts, d, outs, ws)          # one launch pair instead of three
    return pp


def _attn_bwd(store, *args, **kw):
    """Attention backward.  SEGMM_ATTN_SPLIT=1: Dvec = rowsum(dO * O) first (one small kernel), then the dQ kernel on a third
    stream CONCURRENTLY with the dK/dV kernel on the main stream (no gain measured, see ParamStore.attn_split).  The
    partial-maxima slots they share are integer atomic maxima (order-independent), the outputs are disjoint column blocks."""
    B_, H_, dh_, Lq_, La_, Lb_ = args[:6]
    if store.attn_fused and max((La_ + 15) // 16, (Lb_ + 15) // 16) <= 12:
        # dQ + dK + dV in ONE kernel per key block: one workgroup per (b, h, block) with the query side staged in LDS and
        # D = rowsum(dO * O) formed during the staging (attention.h: attn_bwd_fused_kernel); 848 -> ~540 us at config 2
        pl = kw.get("planes")
        if store.attn_two_streams and store.overlap and La_ > 0 and Lb_ > 0 and not (pl is not None and (pl.flags & H.ATTN_REPAIR)):
            # the two key blocks' launches are independent (disjoint outputs; shared maxima slots are integer atomic maxima):
            # block a (the shorter one at config 2) on the side stream, block b on the main stream
            with side_work(store):
                H.attn_bwd(*args, phase=5, **kw)
            H.attn_bwd(*args, phase=6, **kw)
            join_side(store)
            return
        H.attn_bwd(*args, phase=4, **kw)
        return
    if not (store.overlap and store.attn_split):
        H.attn_bwd(*args, **kw)
        return
    H.attn_bwd(*args, phase=1, **kw)
    main, att = torch.cuda.current_stream(), store.attn_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    att.wait_event(ev)
    with torch.cuda.stream(att):
        H.attn_bwd(*args, phase=2, **kw)
    H.attn_bwd(*args, phase=3, **kw)
    main.wait_stream(att)


def vq_tiles(L):
    return (L + 15) // 16


def _mask_u8(m: torch.Tensor) -> torch.Tensor:
    if m.dtype != torch.bool or not m.is_contiguous():
        H.torch_fallback("the conversion of a %s mask to contiguous bool" % str(m.dtype).replace("torch.", ""))
        m = m.to(torch.bool).contiguous()
    return m.view(torch.uint8)


def _as(t: torch.Tensor, dtype, what: str) -> torch.Tensor:
    """``t`` as a contiguous tensor of ``dtype`` -- zero-copy when it already is one (the recordable case)."""
    if t.dtype != dtype or not t.is_contiguous():
        H.torch_fallback("the conversion of %s (%s%s) to contiguous %s" % (what, str(t.dtype).replace("torch.", ""),
                                                                       "" if t.is_contiguous() else ", strided", str(dtype).replace("torch.", "")))
        t = t.contiguous().to(dtype)
    return t


# =============================================================================================== backbone
class BackboneRun:
    """One forward of one backbone; keeps what the backward needs."""

    def __init__(self, store: ParamStore, bb, prefix: str, bb_index: int):
        self.store, self.bb, self.pre, self.bi = store, bb, prefix, bb_index
        self.d, self.H, self.N = bb.d_model, bb.nhead, bb.n_layers
        self.dh = self.d // self.H
        self.sv = {}
        self.abl = getattr(bb, "ablation_type", "ours")
        self.mode = attn_mode(bb)
        self.n_mlp = len(mlp_linears(bb)) if self.abl in MLP_VARIANTS else 0

    def _input_act(self, x, rows, cols):
        """External fp32 input (feature tensor) as a GEMM operand.  Its partial maxima come with the tensor when its producer
        has them (Trainer attaches the whole Act -- header, planes -- as ``_segmm_act``: it travels WITH the tensor object, never
        keyed by address), otherwise from one absmax pass."""
        st = self.store
        a = getattr(x, "_segmm_act", None)
        if a is not None and a.t is x and a.hdr.device == x.device:
            return finish_act(st, a)
        hdr = None
        if st.engine_h:
            hdr = self.am.new()
            H.absmax(x, rows, cols, cols, out=hdr[H.SITE_HDR:])
        planes = torch.empty((rows, 2 * cols), dtype=torch.float16, device=x.device) if st.engine_p and cols % 32 == 0 else None
        return finish_act(st, Act(x, hdr, rows, cols, planes))

    def _as_act(self, arena, t, rows, cols):
        """A plain fp32 tensor (no producer-side maxima) as a GEMM operand: absmax pass (+ split pass)."""
        st = self.store
        a = new_act(st, arena, rows, cols, t=t)
        if a.hdr is not None:
            H.absmax(t, rows, cols, cols, out=a.slots)
        return finish_act(st, a)

    # ---------------------------------------------------------------- forward
    def forward(self, usr_feat, usr_mask, vid_feat, vid_mask, train: bool, seed: int):
        st, bb, P, d = self.store, self.bb, self.pre, self.d
        p_drop = float(bb.dropout_p) if train else 0.0
        p_inner = MLP_INNER_DROPOUT if train else 0.0
        self.p_drop, self.p_inner, self.seed = p_drop, p_inner, seed
        # (head weight, head bias, logits buffer) left by the model for a single-backbone Linear(d, 1) head: the output LayerNorm
        # of the last live layer then computes the raw logits too (_side_post); st._head_dot_done tells the head
        self._head_dot = st.__dict__.pop("_head_dot", None)
        st._head_dot_done = None
        # producer-written planes with delayed scales in training passes; exact split passes otherwise (evaluation stays
        # bitwise reproducible and independent of what ran before)
        self.delayed = st.engine_p and ((train and st.scaling != "exact") or st.scaling == "always")
        vm = _mask_u8(vid_mask)
        B, S = vm.shape
        if S > bb.max_vid_len:
            raise RuntimeError("S=%d exceeds max_vid_len=%d" % (S, bb.max_vid_len))
        if bb.id_usr:
            if usr_feat.dim() != 1:
                raise RuntimeError("id-mode user input must be [B] ids (encoder.py:478-481)")
            Lt = 1
            um = self.store.const_ones((B, 1), torch.uint8)          # encoder.py:481 (made once: a constant)
        else:
            if usr_feat.dim() != 3:
                raise RuntimeError("image-mode user input must be [B,Lt,D]")
            Lt = usr_feat.shape[1]
            um = _mask_u8(usr_mask)
            if Lt > bb.max_usr_len:
                raise RuntimeError("Lt=%d exceeds max_usr_len=%d" % (Lt, bb.max_usr_len))
        Mv, Mu = B * S, B * Lt
        self.B, self.S, self.Lt, self.Mv, self.Mu = B, S, Lt, Mv, Mu
        self.vm, self.um = vm, um
        sv = self.sv
        ref = vm
        am = self.am = AmaxArena(st, 8 + 14 * max(self.N - 1, 0) + 2 * (self.n_mlp + 1))
        layered = self.abl not in MLP_VARIANTS and self.N >= 2
        use_pe = bool(getattr(bb, "use_pe", 1))          # --use_pe 0: no positional-embedding add (encoder.py:450-471)
        usr_is_operand = (layered and self.mode != "self") or self.abl == "CrossMLP"      # does any GEMM read the user embedding?
        # ---- embedding (encoder.py:425-473).  The user-token chain (input Linear -> LayerNorm -> the first layer's fused
        # user-token projection) and the video-token chain are independent until the first attention: with SEGMM_FWD_SIDE=1
        # the user chain is enqueued on the side stream.  Measured on one box, alternating runs: 83.8 k -> 83.6 k
        # interactions/s -- the GEMMs of both chains share the same power-limited matrix pipes, so the knob is OFF by default.
        # Every buffer is allocated HERE, on the main stream (the caching allocator must never hand a side-stream block to
        # the next step while main-stream kernels of this step still read it).
        H.mark(H.PHASE_EMBED_FWD, self.bi)
        pre_u = _empty(ref, Mu, d)
        meu, reu = _empty(ref, Mu), _empty(ref, Mu)
        # The user embedding as PLANES ONLY (round 5): in the trainer's own step of a model whose first layer is not full (N = 2:
        # BASELINE configs 2 / 4 / 5) nothing reads its fp32 values -- it is the operand of the fused user projection and of that
        # projection's weight gradient -- so the LayerNorm writes the planes alone, with the scale of its output bound (no history,
        # no overflow, no repair: segmm_layernorm_fwd with y = NULL), 157 MB less on the chain that bounds the forward
        eu_po = bool(self.delayed and st.eu_planes_only and st.__dict__.get("_trusted") and not bb.id_usr and layered and self.N == 2 and
                     self.mode != "self" and d % 32 == 0 and _FEW_TILES == 0)
        if eu_po:
            Eu = Act(None, am.new(P + "Eu"), Mu, d, torch.empty((Mu, 2 * d), dtype=torch.float16, device=st.flat.device))
            Eu.po, Eu.no_f32 = H.PO(Eu.planes, 2 * d, Eu.hdr, None), True
        else:
            Eu = new_act(st, am, Mu, d, planes=usr_is_operand, site=P + "Eu", delayed=self.delayed)
        Yu0 = None
        fwd_side = st.overlap and st.fwd_side and not bb.id_usr and layered and self.mode != "self"
        if bb.id_usr:
            uids = _as(usr_feat, torch.int64, "the user ids")
            sv["usr_ids"] = uids
            H.embed_id_usr(uids, st.p(P + "usr_proj.weight"), d, st.p(P + "usr_pe.weight") if use_pe else None, pre_u, B)
            H.layernorm_fwd(pre_u, st.p(P + "usr_ln.weight"), st.p(P + "usr_ln.bias"), Eu.t, meu, reu, drop_p=p_drop, seed=seed,
                            site=_site(self.bi, 0, K_EMB_U), amax=Eu.slots, po=Eu.po)
            finish_act(st, produced(Eu))
        else:
            xu = _as(usr_feat, torch.float32, "the user features")
            Din_u = xu.shape[-1]
            sv["usr_x"] = self._input_act(xu, Mu, Din_u)

            def usr_chain():
                _lin_fwd(st, Mu, d, Din_u, sv["usr_x"], P + "usr_proj.weight", pre_u, d,
                         bias=st.p(P + "usr_proj.bias"), **(dict(residual=st.p(P + "usr_pe.weight"), ldr=d, res_period=Lt) if use_pe else {}))
                H.layernorm_fwd(pre_u, st.p(P + "usr_ln.weight"), st.p(P + "usr_ln.bias"), Eu.t, meu, reu, drop_p=p_drop, seed=seed,
                                site=_site(self.bi, 0, K_EMB_U), amax=Eu.slots, po=Eu.po)
                finish_act(st, produced(Eu))
                if Yu0 is not None:
                    self._usr_proj_fwd(0, Eu, Yu0)
            if fwd_side:
                Yu0 = self._proj_act(0, "Yu", Mu, len(layer_plan(self.mode, 0 < self.N - 2)[1]) * d)
                with side_work(st):
                    usr_chain()
            else:
                usr_chain()
        sv["pre_u"], sv["meu"], sv["reu"] = pre_u, meu, reu
        pre_v = _empty(ref, Mv, d)
        if bb.id_vid:
            ids = _as(vid_feat, torch.int64, "the item ids")
            sv["vid_ids"] = ids
            fpos = None
            if "noPos" in self.abl:      # a fresh shuffle of the segment positions per row and per call, from torch's CPU
                # generator exactly like the reference (encoder.py:428-429: B x torch.randperm(Lv)) -- or, when the step's state
                # lives on the device (Trainer(device_state=True): the step may be recorded), drawn on the device: the same
                # distribution (a uniformly random permutation per row), another bit stream
                live = st.__dict__.get("live_seed")
                if live is not None and train and S <= 64:
                    fpos = st.buf("nopos_fpos%d" % self.bi, (B, S))
                    H.rand_perm_rows(fpos, B, S, live, SITE_NOPOS + 16 * self.bi)
                else:
                    H.torch_fallback("the noPos ablation's torch.randperm draws")
                    fpos = torch.stack([torch.randperm(S) for _ in range(B)]).float().to(ids.device).contiguous()
            sv["frame_pos"] = fpos
            H.embed_id_vid(ids, st.p(P + "vid_proj.weight"), d // 2, st.p(P + "frameid_proj.weight"),
                           st.p(P + "frameid_proj.bias"), st.p(P + "vid_pe.weight") if use_pe else None, pre_v, B, S, frame_pos=fpos)
        else:
            x = _as(vid_feat, torch.float32, "the video features")
            Din = x.shape[-1]
            sv["vid_x"] = self._input_act(x, Mv, Din)
            _lin_fwd(st, Mv, d, Din, sv["vid_x"], P + "vid_proj.weight", pre_v, d,
                     bias=st.p(P + "vid_proj.bias"), **(dict(residual=st.p(P + "vid_pe.weight"), ldr=d, res_period=S) if use_pe else {}))
        mev, rev = _empty(ref, Mv), _empty(ref, Mv)
        Ev = new_act(st, am, Mv, d, planes=layered or self.abl in ("SelfMLP", "CrossMLP"), site=P + "Ev", delayed=self.delayed)
        H.layernorm_fwd(pre_v, st.p(P + "vid_ln.weight"), st.p(P + "vid_ln.bias"), Ev.t, mev, rev, drop_p=p_drop, seed=seed,
                        site=_site(self.bi, 0, K_EMB_V), amax=Ev.slots, po=Ev.po)
        finish_act(st, produced(Ev))
        sv["pre_v"], sv["mev"], sv["rev"] = pre_v, mev, rev
        Xv, Xu = Ev, Eu
        sv["layers"] = []
        if self.abl in MLP_VARIANTS:
            out = self._mlp_variant_fwd(Ev, Eu)
            am.close(st)
            return out.view(B, -1, d), Eu.t.view(B, Lt, d)
        for i in range(max(self.N - 1, 0)):
            H.mark(H.PHASE_LAYER_FWD, self.bi, i)
            Xv, Xu = self._layer_fwd(i, Xv, Xu, Yu_ready=Yu0 if i == 0 else None)
        am.close(st)
        # (planes-only user embedding: there is no fp32 tensor to hand out -- the trainer's step, the only caller, ignores it)
        return Xv.t.view(B, S, d), (Eu.t.view(B, Lt, d) if Eu.t is not None else Xv.t.new_empty(0))

    # ---------------------------------------------------------------- MLP ablations (encoder.py:392-400,503-511)
    def _mlp_fwd(self, X, M, tok):
        """encoder_mlp on M tokens: [Linear -> ReLU -> Dropout] x n_hidden, Linear.  ``tok`` (0 video, 1 user) separates the
        dropout streams of the two token sets of CrossMLP.  Returns (Z, saved hidden activations as Acts)."""
        st, d, P = self.store, self.d, self.pre
        lins = mlp_linears(self.bb)
        if len(lins) - 1 > 20:
            raise RuntimeError("encoder_mlp with %d hidden layers: dropout-site space holds 20" % (len(lins) - 1))
        hs = [X]
        for k, n in enumerate(lins[:-1]):
            Hk = new_act(st, self.am, M, d, site="%smlp%d.%d.H" % (P, tok, k), delayed=self.delayed)
            _lin_fwd(st, M, d, d, hs[-1], P + n + ".weight", Hk.t, d, bias=st.p(P + n + ".bias"), c_act=Hk,
                     activation=H.ACT_RELU, drop_p=self.p_drop, seed=self.seed, site=_site(self.bi, tok, K_MLP0 + k))
            hs.append(finish_act(st, Hk))
        Z = _empty(X.t, M, d)
        _lin_fwd(st, M, d, d, hs[-1], P + lins[-1] + ".weight", Z, d, bias=st.p(P + lins[-1] + ".bias"))
        return Z, hs

    def _mlp_bwd(self, dZ, hs, M, tok, gbuf, accumulate, tag):
        """Reverse of _mlp_fwd; weight/bias gradients are ACCUMULATED when ``accumulate`` (second token set of CrossMLP).
        Returns the gradient wrt the MLP input."""
        st, d, P = self.store, self.d, self.pre
        lins = mlp_linears(self.bb)
        g = self._as_act(self.amb, dZ, M, d)
        for k in reversed(range(len(lins))):
            n = lins[k]
            _wgrad(st, g, 0, hs[k], 0, M, d, d, st.g(P + n + ".weight", gbuf), accumulate=accumulate, gb=st.g(P + n + ".bias", gbuf))
            gin = new_act(st, self.amb, M, d, key="mlp_g%d%s" % (k & 1, tag), planes=k > 0, site="%smlp%d.%d.dH" % (P, tok, k),
                          delayed=self.delayed)
            if k > 0:       # through Dropout and ReLU of hidden layer k-1: aux = its saved output (> 0 iff live and kept)
                _lin_dgrad(st, M, d, d, g, P + n + ".weight", gin.t, c_act=gin, activation=H.ACT_DRELU, aux=hs[k].t, ldaux=d,
                           drop_p=self.p_drop, seed=self.seed, site=_site(self.bi, tok, K_MLP0 + k - 1))
                finish_act(st, gin)
            else:
                _lin_dgrad(st, M, d, d, g, P + n + ".weight", gin.t)
            g = gin
        return g.t

    def _mlp_variant_fwd(self, Ev, Eu):
        sv, B, S, Lt, d = self.sv, self.B, self.S, self.Lt, self.d
        if self.abl == "w/oAtt":
            return Ev.t
        Zv, sv["mlp_v"] = self._mlp_fwd(Ev, self.Mv, 0)
        if self.abl == "SelfMLP":
            return Zv
        Zu, sv["mlp_u"] = self._mlp_fwd(Eu, self.Mu, 1)
        out = _empty(Ev.t, B * POOL_BINS, d)
        H.pool_tokens(Zu, Lt, Zv, S, out, B, d, POOL_BINS)       # AdaptiveAvgPool1d(40) over cat(user, video) tokens
        return out

    def _side_post_alloc(self, i, side, ref, M, out_is_operand):
        """Buffers and Acts of _side_post, allocated by the caller on the MAIN stream (the chain itself may run on the side
        stream: the caching allocator must never see a block whose first owner is the side stream)."""
        st, d, am = self.store, self.d, self.am
        sn = "%sL%d.%s." % (self.pre, i, side)
        return dict(R1=_empty(ref, M, d), m1=_empty(ref, M), r1=_empty(ref, M),
                    X1=new_act(st, am, M, d, site=sn + "X1", delayed=self.delayed), G=_empty(ref, M, d),
                    Hh=new_act(st, am, M, d, site=sn + "H", delayed=self.delayed), R2=_empty(ref, M, d), m2=_empty(ref, M), r2=_empty(ref, M),
                    X2=new_act(st, am, M, d, planes=out_is_operand, site=sn + "X2", delayed=self.delayed))

    def _side_post(self, i, L, side, X, A, M, kinds, out_is_operand, bufs=None, head_dot=None):
        """R1 = X + drop(A.Wff^T+b); X1 = LN(R1); H = drop(gelu(X1.W0^T+b0)); R2 = X1 + drop(H.W1^T+b1); X2 = LN(R2).
        X, A: Acts; returns (X2 Act, saved)."""
        st, d, seed = self.store, self.d, self.seed
        k_ao, k_mi, k_mo = kinds
        ca = L + "cross_attn."
        b = bufs if bufs is not None else self._side_post_alloc(i, side, X.t, M, out_is_operand)
        R1, m1, r1, X1, G, Hh, R2, m2, r2, X2 = (b[k] for k in ("R1", "m1", "r1", "X1", "G", "Hh", "R2", "m2", "r2", "X2"))
        _lin_fwd(st, M, d, d, A, ca + "ff_%s.weight" % side, R1, d, bias=st.p(ca + "ff_%s.bias" % side),
                 residual=X.t, ldr=d, res_period=M, drop_p=self.p_drop, seed=seed, site=_site(self.bi, i, k_ao))
        H.layernorm_fwd(R1, st.p(ca + "ln_%s.weight" % side), st.p(ca + "ln_%s.bias" % side), X1.t, m1, r1, amax=X1.slots, po=X1.po)
        finish_act(st, produced(X1))
        ff = L + "ff_%s.layers." % side
        _lin_fwd(st, M, d, d, X1, ff + "0.weight", Hh.t, d, bias=st.p(ff + "0.bias"), c_act=Hh,
                 activation=H.ACT_GELU, aux=G, ldaux=d, drop_p=self.p_inner, seed=seed, site=_site(self.bi, i, k_mi))
        finish_act(st, Hh)
        _lin_fwd(st, M, d, d, Hh, ff + "1.weight", R2, d, bias=st.p(ff + "1.bias"),
                 residual=X1.t, ldr=d, res_period=M, drop_p=self.p_drop, seed=seed, site=_site(self.bi, i, k_mo))
        if head_dot is not None:          # the backbone's output LayerNorm: the Linear(d, 1) head's logits from the same launch
            H.layernorm_fwd_dot(R2, st.p(L + "ln_%s.weight" % side), st.p(L + "ln_%s.bias" % side), X2.t, m2, r2, head_dot[0], head_dot[1],
                                head_dot[2], amax=X2.slots, po=X2.po)
            st._head_dot_done = head_dot[2].data_ptr()
        else:
            H.layernorm_fwd(R2, st.p(L + "ln_%s.weight" % side), st.p(L + "ln_%s.bias" % side), X2.t, m2, r2, amax=X2.slots, po=X2.po)
        finish_act(st, produced(X2))
        return X2, dict(A=A, R1=R1, X1=X1, m1=m1, r1=r1, G=G, Hh=Hh, R2=R2, m2=m2, r2=r2)

    def _attn_views(self, full, Yv, Yu, nv, nu):
        """Column slices of the fused projection buffers (or of their gradients) as the attention kernels take them:
        for the video queries and, when the layer is full, for the user queries.  An empty key block is (None, ..., 0)."""
        d, S, Lt, mode = self.d, self.S, self.Lt, self.mode
        vidP, usrP = layer_plan(mode, full)
        cv = {n: (Yv, k * d) for k, n in enumerate(vidP)}
        cu = {n: (Yu, k * d) for k, n in enumerate(usrP)}
        ldv, ldu = nv * d, max(nu, 1) * d
        g = lambda c, n: c.get(n)
        vq = dict(Qa=g(cv, "v2v_proj.0"), Qb=g(cv, "t2v_proj.0"), ldq=ldv, Ka=g(cv, "v2v_proj.1"), Va=g(cv, "v2v_proj.2"), ldka=ldv,
                  Kb=g(cu, "t2v_proj.1"), Vb=g(cu, "t2v_proj.2"), ldkb=ldu, La=0 if mode == "cross" else S, Lb=0 if mode == "self" else Lt)
        uq = None
        if full:
            uq = dict(Qa=g(cu, "v2t_proj.0"), Qb=g(cu, "t2t_proj.0"), ldq=ldu, Ka=g(cv, "v2t_proj.1"), Va=g(cv, "v2t_proj.2"), ldka=ldv,
                      Kb=g(cu, "t2t_proj.1"), Vb=g(cu, "t2t_proj.2"), ldkb=ldu, La=S, Lb=0 if mode == "cross" else Lt)
        return vq, uq

    def _attn_calls(self):
        """(Lq, La, Lb) of every attention call a pass of this backbone makes (video queries of every live layer; user queries
        of the full layers) -- what ``attn_fwd_pl_takes`` / the planes-in backward of ``capi.hip`` are asked with."""
        S, Lt, mode = self.S, self.Lt, self.mode
        calls = [(S, 0 if mode == "cross" else S, 0 if mode == "self" else Lt)]
        if self.N >= 3 and mode != "self":          # layers 0 .. N-3 are full: the user tokens query too
            calls.append((Lt, S, 0 if mode == "cross" else Lt))
        return calls

    def _attn_planes_in(self):
        """Do the attention kernels of this pass read the Q / K / V planes of the fused projection GEMMs (csrc/attention_pl.h)?
        Training passes with delayed scales on the plane engine, and ONLY when every attention call of the pass is one the planes-in
        forward takes -- the gate of ``attn_fwd_pl_takes`` (capi.hip) restated per call: Lq <= 112 (7 query tiles), La + Lb <= 192
        (12 key tiles), lengths % 4, knob ATT_FWD_PL.  A layer whose projection outputs exist as planes ONLY cannot fall back to the
        fp32-operand kernels, so a single refused call keeps the fp32 buffers for the whole pass (SEGMM_ATT_PL=0: never)."""
        st = self.store
        if not (self.delayed and st.engine_p and st.attn_pl and st.attn_fused and self.dh % 16 == 0 and self.dh <= 48 and
                self.d % 32 == 0 and _FEW_TILES == 0):
            return False
        if H.knob("ATT_FWD_PL") == 0:
            return False
        return all(Lq <= 112 and La + Lb <= 192 and La % 4 == 0 and Lb % 4 == 0 for (Lq, La, Lb) in self._attn_calls())

    def _proj_act(self, i, which, rows, cols):
        """The fused projection output Yv / Yu of layer i as an Act.  With the planes-in attention and a calibrated site its producer
        GEMM writes the P32 planes (delayed scale of the site) and NOTHING else: the Act has no fp32 tensor (``t is None``).  On the
        site's first pass (no scale yet), in evaluation and for shapes the planes-in kernels do not take: a plain fp32 buffer whose
        header records the maxima the scale will come from."""
        st = self.store
        site = "%sL%d.%s" % (self.pre, i, which)
        want = self._attn_planes_in()
        if want and st.attn_pl == 1:
            sp = st.scale_ptr(site, self.delayed)
            # (both projection outputs of the layer or neither: an attention call reads Q / K / V from both)
            full = i < self.N - 2 and self.mode != "self"
            has_u = len(layer_plan(self.mode, full)[1]) > 0
            other = st.scale_ptr("%sL%d.%s" % (self.pre, i, "Yu" if which == "Yv" else "Yv"), self.delayed) if (which == "Yu" or has_u) else True
            if sp is not None and other is not None:
                planes = torch.empty((rows, 2 * cols), dtype=torch.float16, device=st.flat.device)
                a = Act(None, self.am.new(site), rows, cols, planes)
                a.scale_ptr, a.po, a.no_f32 = sp, H.PO(planes, 2 * cols, a.hdr, sp), True
                return a
        return new_act(st, self.am, rows, cols, planes=want, site=site, delayed=self.delayed)

    def _usr_proj_fwd(self, i, Xu, Yu):
        """Yu = Xu . [fused user-token projections of layer i]^T + b."""
        st, d = self.store, self.d
        full = i < self.N - 2 and self.mode != "self"
        usrP = layer_plan(self.mode, full)[1]
        ca = "%sencoder.layers.%d.cross_attn." % (self.pre, i)
        nu = len(usrP)
        # (c_act also on a site's first pass, when it has no scale yet: the GEMM then only records the maxima the scale comes from)
        _lin_fwd(st, self.Mu, nu * d, d, Xu, ca + usrP[0] + ".weight", Yu.t, nu * d, bias=st.p(ca + usrP[0] + ".bias"),
                 c_act=Yu if Yu.planes is not None else None)
        if Yu.po is None:
            Yu.planes = None          # (no calibrated scale yet / evaluation: the attention reads the fp32 views)

    def _layer_fwd(self, i, Xv, Xu, Yu_ready=None):
        st, d, P, am = self.store, self.d, self.pre, self.am
        B, S, Lt, Mv, Mu, Hh, dh = self.B, self.S, self.Lt, self.Mv, self.Mu, self.H, self.dh
        full = i < self.N - 2 and self.mode != "self"
        vidP, usrP = layer_plan(self.mode, full)
        nv, nu = len(vidP), len(usrP)
        L = "%sencoder.layers.%d." % (P, i)
        ca = L + "cross_attn."
        Yv_a = self._proj_act(i, "Yv", Mv, nv * d)
        _lin_fwd(st, Mv, nv * d, d, Xv, ca + vidP[0] + ".weight", Yv_a.t, nv * d, bias=st.p(ca + vidP[0] + ".bias"),
                 c_act=Yv_a if Yv_a.planes is not None else None)
        if Yv_a.po is None:
            Yv_a.planes = None
        Yu_a = None
        if Yu_ready is not None:      # computed on the side stream together with the user embedding (forward())
            Yu_a = Yu_ready
            join_side(st)
        elif nu:
            Yu_a = self._proj_act(i, "Yu", Mu, nu * d)
            self._usr_proj_fwd(i, Xu, Yu_a)
        Yv, Yu = Yv_a.t, (Yu_a.t if Yu_a is not None else None)
        vq, uq = self._attn_views(full, Yv, Yu, nv, nu)
        # input planes of the attention forward: the producer GEMMs wrote them (delayed scales); the kernel judges the site headers
        # itself and stages an unusable site from the fp32 views
        pl_v = (Yv_a.planes, Yv_a.hdr, 2 * nv * d) if Yv_a.po is not None else None
        pl_u = (Yu_a.planes, Yu_a.hdr, 2 * nu * d) if (Yu_a is not None and Yu_a.po is not None) else None
        pin_v = pin_u = None
        if pl_v is not None and (pl_u is not None or vq["Lb"] == 0):
            pin_v = dict(q=pl_v, a=pl_v if vq["La"] else None, b=pl_u if vq["Lb"] else None)
        if full and pl_v is not None and pl_u is not None:
            pin_u = dict(q=pl_u, a=pl_v, b=pl_u if uq["Lb"] else None)
        lse_v = _empty(Xv.t, 2, B, Hh, S)
        Av = new_act(st, am, Mv, d, site="%sL%d.vid.A" % (P, i), delayed=self.delayed)
        rec = dict(full=full, Xv=Xv, Xu=Xu, Yv=Yv, Yu=Yu, lse_v=lse_v, pin_v=pin_v, pin_u=pin_u)
        if (Yv is None or (nu and Yu is None)) and (pin_v is None or (full and pin_u is None)):
            raise RuntimeError("planes-only projection outputs without input planes for every attention call of the layer")
        X2u = None
        usr_ctx = None
        if full:
            # the user-token chain: every buffer allocated here, on the main stream, then the launches on the side stream
            lse_u = _empty(Xv.t, 2, B, Hh, Lt)
            Au = new_act(st, am, Mu, d, site="%sL%d.usr.A" % (P, i), delayed=self.delayed)
            bufs_u = self._side_post_alloc(i, "usr", Xv.t, Mu, True)

            def usr_chain():
                H.attn_fwd(B, Hh, dh, Lt, uq["La"], uq["Lb"], uq["Qa"], uq["Qb"], uq["ldq"], uq["Ka"], uq["Va"], uq["ldka"], uq["Kb"],
                           uq["Vb"], uq["ldkb"], self.um, self.vm, self.um, Au.t, d, lse_u, drop_p=self.p_drop,
                           seed=self.seed, site=_site(self.bi, i, K_ATT_U), amax_o=Au.slots, po=Au.po, pin=pin_u)
                finish_act(st, produced(Au))
                return self._side_post(i, L, "usr", Xu, Au, Mu, (K_AO_U, K_MI_U, K_MO_U), out_is_operand=True, bufs=bufs_u)
            if st.usr_side and st.overlap:
                with side_work(st):          # (both fused projections are enqueued on the main stream: the fork orders the chain behind them)
                    X2u, sv_u = usr_chain()
                usr_ctx = True
        H.attn_fwd(B, Hh, dh, S, vq["La"], vq["Lb"], vq["Qa"], vq["Qb"], vq["ldq"], vq["Ka"], vq["Va"], vq["ldka"], vq["Kb"], vq["Vb"],
                   vq["ldkb"], self.vm, self.vm, self.um, Av.t, d, lse_v, drop_p=self.p_drop, seed=self.seed,
                   site=_site(self.bi, i, K_ATT_V), amax_o=Av.slots, po=Av.po, pin=pin_v)
        finish_act(st, produced(Av))
        hd = self.__dict__.get("_head_dot") if i == self.N - 2 else None          # last live layer: its video-side output IS the backbone's
        X2v, sv_v = self._side_post(i, L, "vid", Xv, Av, Mv, (K_AO_V, K_MI_V, K_MO_V), out_is_operand=i < self.N - 2, head_dot=hd)
        rec["v"] = sv_v
        if full:
            if usr_ctx:
                join_side(st)          # the next layer's projections (main stream) read the user chain's output
            else:
                X2u, sv_u = usr_chain()
            rec["lse_u"], rec["u"] = lse_u, sv_u
        self.sv["layers"].append(rec)
        return X2v, (X2u if full else Xu)

    # ---------------------------------------------------------------- backward
    def _ln_bwd_act(self, key, dy, x, mean, rstd, gname, bname, gbuf, dx, M, drop_b, dsum_to):
        """LayerNorm backward whose FORWARDED gradient (through the residual-branch dropout when p > 0) is a GEMM operand:
        returns it as an Act.  ``dx`` receives the plain input gradient (residual path); without dropout the two coincide."""
        st, d = self.store, self.d
        has_drop = drop_b[0] > 0
        a = new_act(st, self.amb, M, d, t=None if has_drop else dx, key=key, site=self.pre + key, delayed=self.delayed)
        lazy = self.__dict__.get("_lazy_dy")          # (backward(): the head left its gradient as an outer product, unmaterialised)
        outer = None
        if lazy is not None and dy.data_ptr() == lazy[0]:
            outer, self._lazy_dy = (lazy[1], lazy[2]), None
        _ln_bwd(st, dy, x, mean, rstd, gname, bname, gbuf, dx, a.t if has_drop else None, M, d, drop_b=drop_b, seed=self.seed,
                amax=a.slots, dsum_to=dsum_to, po=a.po, dy_outer=outer)
        return finish_act(st, produced(a))

    def _side_post_bwd(self, i, L, side, sv, dX2, M, kinds, gbuf, tag, deferred=None):
        """Reverse of _side_post.  Returns (dR1, dA): gradient wrt the residual input X and wrt the attention output."""
        st, d, seed, am = self.store, self.d, self.seed, self.amb
        k_ao, k_mi, k_mo = kinds
        ca = L + "cross_attn."
        ff = L + "ff_%s.layers." % side
        dR2 = st.buf("dR2" + tag, (M, d))
        dM = self._ln_bwd_act("dM" + tag, dX2, sv["R2"], sv["m2"], sv["r2"], L + "ln_%s.weight" % side, L + "ln_%s.bias" % side, gbuf,
                              dR2, M, (self.p_drop, _site(self.bi, i, k_mo)), st.g(ff + "1.bias", gbuf))
        side_or_defer(st, lambda: _wgrad(st, dM, 0, sv["Hh"], 0, M, d, d, st.g(ff + "1.weight", gbuf)), deferred)
        dG = new_act(st, am, M, d, key="dG" + tag, site=self.pre + "dG" + tag, delayed=self.delayed)
        _lin_dgrad(st, M, d, d, dM, ff + "1.weight", dG.t, c_act=dG, activation=H.ACT_DGELU, aux=sv["G"], ldaux=d,
                   drop_p=self.p_inner, seed=seed, site=_site(self.bi, i, k_mi))
        finish_act(st, dG)

        def _w0():
            _wgrad(st, dG, 0, sv["X1"], 0, M, d, d, st.g(ff + "0.weight", gbuf), gb=st.g(ff + "0.bias", gbuf))
        side_or_defer(st, _w0, deferred)
        dX1 = st.buf("dX1" + tag, (M, d))
        _lin_dgrad(st, M, d, d, dG, ff + "0.weight", dX1, residual=dR2, ldr=d, res_period=M)
        dR1 = st.buf("dR1" + tag, (M, d))
        dZ = self._ln_bwd_act("dZ" + tag, dX1, sv["R1"], sv["m1"], sv["r1"], ca + "ln_%s.weight" % side, ca + "ln_%s.bias" % side, gbuf,
                              dR1, M, (self.p_drop, _site(self.bi, i, k_ao)), st.g(ca + "ff_%s.bias" % side, gbuf))
        side_or_defer(st, lambda: _wgrad(st, dZ, 0, sv["A"], 0, M, d, d, st.g(ca + "ff_%s.weight" % side, gbuf)), deferred)
        dA = st.buf("dA" + tag, (M, d))
        _lin_dgrad(st, M, d, d, dZ, ca + "ff_%s.weight" % side, dA)
        return dR1, dA

    def _layer_bwd(self, i, rec, dXv_out, dXu_out, gbuf):
        st, d, P = self.store, self.d, self.pre
        B, S, Lt, Mv, Mu, Hh, dh = self.B, self.S, self.Lt, self.Mv, self.Mu, self.H, self.dh
        full = rec["full"]
        vidP, usrP = layer_plan(self.mode, full)
        nv, nu = len(vidP), len(usrP)
        L = "%sencoder.layers.%d." % (P, i)
        ca = L + "cross_attn."
        Yv, Yu = rec["Yv"], rec["Yu"]
        # one site per fused dY buffer: both attentions fold into it.  Only the fused backward kernel writes planes itself.
        fused = st.attn_fused and max((vq_tiles(S), vq_tiles(Lt))) <= 12
        dly = self.delayed and fused
        dYv = new_act(st, self.amb, Mv, nv * d, key="dYv%d" % i, site="%sdYv%d" % (P, i), delayed=dly)
        dYu = new_act(st, self.amb, Mu, nu * d, key="dYu%d" % i, site="%sdYu%d" % (P, i), delayed=dly) if nu else None
        vq, uq = self._attn_views(full, Yv, Yu, nv, nu)
        dvq, duq = self._attn_views(full, dYv.t, dYu.t if nu else None, nv, nu)
        Dv = st.buf("attnD", (B * Hh * max(S, Lt),))
        sl_v, sl_u = dYv.slots, (dYu.slots if nu else None)

        # planes only: dQ / dK / dV are read by the projection GEMMs alone, as planes; their fp32 copies only ever fed the consumers'
        # overflow fallback (0.57 GB of stores per step at config 2).  Without them, a REPAIR pass of the same launches follows
        # the producers (leaves at once unless a site's planes are unusable) and the consumers get no fp32 fallback.
        # (short query sides -- config 3: 20 and 1 queries -- keep the copies: the tensors are small and the extra launches of the
        # repair pass cost as much as the stores they save; SEGMM_ATTN_PLANES_ONLY=2 forces the protocol for every shape)
        ponly = dly and st.attn_planes_only and dYv.po is not None and (not nu or dYu.po is not None) and \
            (S > 32 or st.attn_planes_only > 1)

        def planes_of(dq, dka_, dkb_, views, pflags=0):
            """segmm_attn_planes_t for one fused-backward call: query-side buffer dq, key-block buffers dka_ / dkb_ (Acts)."""
            if not dly or dq.po is None:
                return None
            pl = H.AttnPlanes()

            def pp(act, view):          # plane address of the column slice ``view`` = (fp32 tensor, column offset)
                return None if (act is None or act.po is None or view is None) else act.planes.data_ptr() + 4 * view[1]
            pl.flags = pflags
            pl.dqa, pl.dqb, pl.lddq2 = pp(dq, views["Qa"]), pp(dq, views["Qb"]), 2 * dq.cols
            pl.hdr_q, pl.sin_q = dq.hdr.data_ptr(), dq.scale_ptr
            if dka_ is not None and dka_.po is not None and views["Ka"] is not None:
                pl.dka, pl.dva, pl.lddka2 = pp(dka_, views["Ka"]), pp(dka_, views["Va"]), 2 * dka_.cols
                pl.hdr_ka, pl.sin_ka = dka_.hdr.data_ptr(), dka_.scale_ptr
            if dkb_ is not None and dkb_.po is not None and views["Kb"] is not None:
                pl.dkb, pl.dvb, pl.lddkb2 = pp(dkb_, views["Kb"]), pp(dkb_, views["Vb"]), 2 * dkb_.cols
                pl.hdr_kb, pl.sin_kb = dkb_.hdr.data_ptr(), dkb_.scale_ptr
            return pl
        deferred = [] if st.defer_wgrad else None
        attn_u = None
        dR1u = None
        usr_on_side = full and st.usr_side and st.overlap
        if full:
            Dv_u = st.buf("attnD_u", (B * Hh * max(S, Lt),))

            def usr_bwd():
                dR1u_, dAu = self._side_post_bwd(i, L, "usr", rec["u"], dXu_out, Mu, (K_AO_U, K_MI_U, K_MO_U), gbuf, "u%d" % i, deferred)
                flush_deferred(st, deferred)

                def attn_u_(pflags):
                    _attn_bwd(st, B, Hh, dh, Lt, uq["La"], uq["Lb"], uq["Qa"], uq["Qb"], uq["ldq"], uq["Ka"], uq["Va"], uq["ldka"], uq["Kb"],
                               uq["Vb"], uq["ldkb"], self.um, self.vm, self.um, rec["lse_u"], rec["u"]["A"].t, d, dAu, d, Dv_u,
                               duq["Qa"], duq["Qb"], duq["ldq"], duq["Ka"], duq["Va"], duq["ldka"], duq["Kb"], duq["Vb"], duq["ldkb"],
                               drop_p=self.p_drop, seed=self.seed, site=_site(self.bi, i, K_ATT_U),
                               amax_q=sl_u, amax_ka=sl_v, amax_kb=sl_u, planes=planes_of(dYu, dYv, dYu, duq, pflags), pin=rec.get("pin_u"))
                attn_u_(H.ATTN_PLANES_ONLY if ponly else 0)
                return dR1u_, attn_u_
            if usr_on_side:
                # the user-token chain of the layer's backward (LayerNorms, MLP and ff gradients, attention with user queries) on the
                # side stream, next to the video-token chain: the two only meet in the fused dY buffers, where they write disjoint
                # column blocks (the shared maxima slots are integer atomic maxima)
                with side_work(st):
                    dR1u, attn_u = usr_bwd()
        dR1v, dAv = self._side_post_bwd(i, L, "vid", rec["v"], dXv_out, Mv, (K_AO_V, K_MI_V, K_MO_V), gbuf, "v%d" % i, deferred)
        flush_deferred(st, deferred)          # the three weight-gradient GEMMs of this side run under the attention backward
        def attn_v(pflags):
            _attn_bwd(st, B, Hh, dh, S, vq["La"], vq["Lb"], vq["Qa"], vq["Qb"], vq["ldq"], vq["Ka"], vq["Va"], vq["ldka"], vq["Kb"], vq["Vb"],
                       vq["ldkb"], self.vm, self.vm, self.um, rec["lse_v"], rec["v"]["A"].t, d, dAv, d, Dv, dvq["Qa"], dvq["Qb"], dvq["ldq"],
                       dvq["Ka"], dvq["Va"], dvq["ldka"], dvq["Kb"], dvq["Vb"], dvq["ldkb"], drop_p=self.p_drop, seed=self.seed,
                       site=_site(self.bi, i, K_ATT_V), amax_q=sl_v, amax_ka=sl_v, amax_kb=sl_u,
                       planes=planes_of(dYv, dYv, dYu, dvq, pflags), pin=rec.get("pin_v"))
        attn_v(H.ATTN_PLANES_ONLY if ponly else 0)
        if full:
            if usr_on_side:
                join_side(st)
            else:
                dR1u, attn_u = usr_bwd()
        if ponly:          # every producer of the two sites is enqueued: judge the sites (one tiny launch), then the repair pass
            H.site_fixup(dYv.hdr if dYv.po is not None else None, dYu.hdr if (nu and dYu.po is not None) else None,
                         stats=st.scales()[st.MAX_SITES:])
            attn_v(H.ATTN_PLANES_ONLY | H.ATTN_REPAIR)
            if attn_u is not None:
                attn_u(H.ATTN_PLANES_ONLY | H.ATTN_REPAIR)
            dYv.no_f32 = dYv.po is not None
            if nu:
                dYu.no_f32 = dYu.po is not None
        if dly:          # every column block of a dY buffer must have been written WITH planes, else fall back to the split pass
            produced(dYv)
            if nu:
                produced(dYu)
        finish_act(st, dYv)
        if nu:
            finish_act(st, dYu)
        # fused projection weights / inputs
        with side_work(st):
            _wgrad(st, dYv, 0, rec["Xv"], 0, Mv, nv * d, d, _group_view(st, ca + vidP[0] + ".weight", nv * d * d, gbuf),
                   gb=_group_view(st, ca + vidP[0] + ".bias", nv * d, gbuf))
            if nu:
                _wgrad(st, dYu, 0, rec["Xu"], 0, Mu, nu * d, d, _group_view(st, ca + usrP[0] + ".weight", nu * d * d, gbuf),
                       gb=_group_view(st, ca + usrP[0] + ".bias", nu * d, gbuf))
        dXv_in = st.buf("dXv_in%d" % (i & 1), (Mv, d))
        _lin_dgrad(st, Mv, d, nv * d, dYv, ca + vidP[0] + ".weight", dXv_in, residual=dR1v, ldr=d, res_period=Mv)
        dXu_in = None
        if nu:
            dXu_in = st.buf("dXu_in%d" % (i & 1), (Mu, d))
            if full:
                _lin_dgrad(st, Mu, d, nu * d, dYu, ca + usrP[0] + ".weight", dXu_in, residual=dR1u, ldr=d, res_period=Mu)
            else:
                _lin_dgrad(st, Mu, d, nu * d, dYu, ca + usrP[0] + ".weight", dXu_in)
        return dXv_in, dXu_in

    def backward(self, d_vid_out: torch.Tensor, gbuf: Optional[torch.Tensor] = None, on_bucket=None):
        """Fills the gradients of every live parameter of this backbone (views of ``gbuf``/store.gflat).
        ``on_bucket(name)`` is called as soon as a bucket's gradients are complete (DP overlap hook)."""
        st, bb, P, d, sv = self.store, self.bb, self.pre, self.d, self.sv
        B, S, Lt, Mv, Mu = self.B, self.S, self.Lt, self.Mv, self.Mu
        dXv = d_vid_out.contiguous().view(-1, d)
        dXu = None
        # The interest head (Linear(d, 1), single backbone, trainer's direct gradient delivery) hands over d_vid UNWRITTEN, with
        # (address, d logits per row, head weight) on the store: the first consumer -- the last layer's LayerNorm backward --
        # forms dl[row] * w[c] itself (segmm_layernorm_bwd_outer: one launch and a 63 MB round trip less on the critical chain
        # loss -> first input-gradient GEMM).  Any other first consumer gets it written out first.
        lazy = st.__dict__.pop("_lazy_dy", None)
        self._lazy_dy = None
        if lazy is not None and lazy[0] == dXv.data_ptr():
            if self.abl in MLP_VARIANTS or max(self.N - 1, 0) == 0:
                H.rowscale_bcast(lazy[1], lazy[2], dXv, d, dXv.shape[0], d)
            else:
                self._lazy_dy = lazy
        if st._defer_wgrad_env == "auto":          # (plane engine only: on the exact-fp32 engine the deferral costs 1.5 %)
            st.defer_wgrad = S > 32 and st.engine_p
        if st._ln_side_env == "auto":
            st.ln_side = S > 32 and st.engine_p
        self.amb = AmaxArena(st, 4 + 8 * max(self.N - 1, 0) + 2 * (self.n_mlp + 2))
        if self.abl in MLP_VARIANTS:
            if self.abl == "CrossMLP":
                dZu, dZv = st.buf("pool_du", (Mu, d)), st.buf("pool_dv", (Mv, d))
                H.pool_tokens_bwd(dXv, dZu, Lt, dZv, S, B, d, POOL_BINS)
                dXu = self._mlp_bwd(dZu, sv["mlp_u"], Mu, 1, gbuf, False, "u")
                dXv = self._mlp_bwd(dZv, sv["mlp_v"], Mv, 0, gbuf, True, "v")
            elif self.abl == "SelfMLP":
                dXv = self._mlp_bwd(dXv, sv["mlp_v"], Mv, 0, gbuf, False, "v")
            if self.abl != "w/oAtt" and on_bucket is not None:
                on_bucket(P + "mlp", after_side=True)
        else:
            for i in reversed(range(max(self.N - 1, 0))):
                H.mark(H.PHASE_LAYER_BWD, self.bi, i)
                dXv, dXu = self._layer_bwd(i, sv["layers"][i], dXv, dXu, gbuf)
                if on_bucket is not None:
                    # no join: the hook issues the bucket's all-reduce from the side stream's context (ordered behind both
                    # streams), the main stream goes straight on with the next layer's input-gradient GEMMs
                    on_bucket("%slayer%d" % (P, i), after_side=True)
        # ---- embedding backward.  User side first: its weight gradient (the larger one) queues on the side stream
        # behind the projection weight gradients still running there, the video side's runs on the main stream.
        H.mark(H.PHASE_EMBED_BWD, self.bi)
        if dXu is not None:
            dpre_u = new_act(st, self.amb, Mu, d, key="dpre_u", planes=not bb.id_usr, site=P + "dpre_u", delayed=self.delayed)
            pp_u = _ln_bwd(st, dXu, sv["pre_u"], sv["meu"], sv["reu"], P + "usr_ln.weight", P + "usr_ln.bias", gbuf, dpre_u.t, None, Mu, d,
                           drop_y=(self.p_drop, _site(self.bi, 0, K_EMB_U)), seed=self.seed, amax=dpre_u.slots, po=dpre_u.po, pos_period=Lt)
            finish_act(st, produced(dpre_u))
            self._embed_bwd("usr", dpre_u, B, Lt, gbuf, pp_u)
            if on_bucket is not None:
                # the user-side embedding gradients: LayerNorm / positional parts were written on the main stream, the weight
                # gradient is still in flight on the side stream -- the hook orders the all-reduce behind BOTH without making the
                # main stream (which goes on with the video side) wait for the side stream
                on_bucket(P + "embed_u", after_side=True)
        dpre_v = new_act(st, self.amb, Mv, d, key="dpre_v", planes=not bb.id_vid, site=P + "dpre_v", delayed=self.delayed)
        pp_v = _ln_bwd(st, dXv, sv["pre_v"], sv["mev"], sv["rev"], P + "vid_ln.weight", P + "vid_ln.bias", gbuf, dpre_v.t, None, Mv, d,
                       drop_y=(self.p_drop, _site(self.bi, 0, K_EMB_V)), seed=self.seed, amax=dpre_v.slots, po=dpre_v.po, pos_period=S)
        finish_act(st, produced(dpre_v))
        self._embed_bwd("vid", dpre_v, B, S, gbuf, pp_v)
        join_side(st)
        self.amb.close(st, backward=True)
        if on_bucket is not None:
            on_bucket(P + "embed")

    def _embed_bwd(self, side, dpre_act, B, L, gbuf, part_pos=None):
        st, bb, P, d, sv = self.store, self.bb, self.pre, self.d, self.sv
        dpre = dpre_act.t
        M = B * L
        is_id = bb.id_vid if side == "vid" else bb.id_usr
        if getattr(bb, "use_pe", 1):
            gpe = st.g(P + "%s_pe.weight" % side, gbuf)
        else:          # --use_pe 0: the table is dead (grad None); the per-position sums still feed the bias / frame-id gradients
            gpe = st.buf("gpe_scratch_" + side, (L, d))
        if part_pos is not None:          # the LayerNorm backward left per-wave, per-position sums of dpre (12 MB instead of a pass over dpre)
            H.colsum_pos(part_pos, L, gpe)
        else:
            _colsum(st, dpre, L * d, B, L * d, gpe)          # dpe[s,:] = sum_b dpre[b,s,:]: a column sum of the [B, L*d] view
        if gpe.shape[0] > L:
            H.fill_zero(gpe[L:])
        gtab = st.g(P + "%s_proj.weight" % side, gbuf)
        if is_id:
            ids = sv["%s_ids" % side]
            width = d // 2 if side == "vid" else d
            # the dense table gradient (360 MB at config 3) is zero except for the rows the PREVIOUS step scattered into it
            # (AdamW reads it, nobody else writes it): clear those rows instead of filling the whole table again
            prev = st._tab_rows.get(P + side)
            if gbuf is None and prev is not None and prev[0] == gtab.data_ptr() and prev[1].device == gtab.device:
                H.zero_rows(gtab, prev[1])
            else:
                H.fill_zero(gtab)
            st._tab_rows.pop(P + side, None)
            if st.row_exchange is not None:
                # Data parallel: a rank touches at most B rows of the table, so the ranks exchange
                # their B compact row gradients [B, width] + ids (all-gather, ~1 MB) instead of all-reducing the dense
                # [n_items, width] gradient (360 MB at config 3); every rank then runs the same deterministic sorted
                # segment sum over the G*B gathered rows and ends with the bitwise-identical global table gradient.
                rows = st.buf("idrows_" + side, (B, width))
                ar, ar32 = st.const_arange(B, torch.int64), st.const_arange(B, torch.int32)
                H.fill_zero(rows)                                                          # the kernel accumulates into its output
                H.embed_id_bwd(dpre, L, d, 0, width, ar32, ar, rows, B)                    # rows[b] = sum_s dpre[b, s, :width]
                # (the ids travel from a persistent copy: the exchange is a host action that a recorded step calls again every
                # step with the tensors it was given at record time -- the batch's own tensor changes from step to step)
                ids_p = st.buf("idrows_ids_" + side, (ids.numel(),), torch.int64)
                H.copy_bytes(ids_p, ids.reshape(-1))
                pending = st.row_exchange(ids_p, rows)          # asynchronous all-gather: waited for below, after the work that needs no rows
            else:
                pending = None
            if side == "vid":
                dh_ = d // 2
                _colsum(st, gpe, d, L, dh_, st.g(P + "frameid_proj.bias", gbuf), x_off=dh_)
                if sv.get("frame_pos") is not None:      # noPos: positions differ per row -> weighted sum over all tokens
                    _colsum(st, dpre, d, M, dh_, st.g(P + "frameid_proj.weight", gbuf).view(-1), x_off=dh_,
                            w=sv["frame_pos"].view(-1))
                else:
                    pos = st.const_arange(L, torch.float32)
                    _colsum(st, gpe, d, L, dh_, st.g(P + "frameid_proj.weight", gbuf).view(-1), x_off=dh_, w=pos)
            if pending is not None:
                ids_all, rows_all = pending()
                order = _argsort_ids(ids_all, st)
                H.embed_id_bwd(rows_all, 1, width, 0, width, order, ids_all, gtab, ids_all.numel())
                touched = ids_all
            else:
                order = _argsort_ids(ids, st)
                H.embed_id_bwd(dpre, L, d, 0, width, order, ids, gtab, B)
                touched = ids
            # (a dense all-reduce of the table under data parallelism adds the OTHER ranks' rows: no row list then)
            if gbuf is None and (st.bucket_hook is None or st.row_exchange is not None):
                st._tab_rows[P + side] = (gtab.data_ptr(), touched.reshape(-1))          # (zero_rows skips ids outside the table)
        else:
            x = sv["%s_x" % side]
            Din = x.cols
            # The backward ends with the side stream still working through the big projection weight gradients while the
            # main stream runs dry: the video-side embedding weight gradient (the main stream's last GEMM-sized job
            # before the user side) therefore runs on the MAIN stream, the user-side one on the side stream.
            ctx = contextlib.nullcontext() if (side == "vid" and st.tail_balance) else side_work(st)
            with ctx:
                _wgrad(st, dpre_act, 0, x, 0, M, d, Din, gtab)
                # bias gradient = sum over all tokens of dpre = sum over positions of the positional-embedding
                # gradient just computed ([L, d] instead of a second pass over [B*L, d])
                _colsum(st, gpe, d, L, d, st.g(P + "%s_proj.bias" % side, gbuf))


def _argsort_ids(ids, store=None):
    """Stable argsort of a batch's table ids as int32 by the library's kernels: one launch up to 8192 ids, the multi-workgroup
    network over a persistent workspace of the store beyond (the gathered id list of a data-parallel node: G x B ids -- a
    recorded step replays it like any other launch).  Only ids that are not int64 take torch's argsort."""
    ids = ids.reshape(-1).contiguous()
    if ids.dtype == torch.int64 and ids.numel() <= (1 << 24):
        ws = None
        if ids.numel() > H.ARGSORT_MAX and store is not None:
            ws = store.buf("argsort_ws", (H.argsort_ws_words(ids.numel()),), torch.int64)
        return H.argsort_ids(ids, ws=ws)
    H.torch_fallback("the argsort of %d table ids (%s)" % (ids.numel(), str(ids.dtype).replace("torch.", "")))
    return torch.argsort(ids, stable=True).to(torch.int32)


def _group_view(store, first_name, numel, gbuf):
    """Gradient view spanning a fused (adjacent) parameter group that starts at ``first_name``."""
    o, _ = store.index[first_name]
    gb = store.gflat if gbuf is None else gbuf
    return gb[o:o + numel]


# =============================================================================================== autograd glue
_STEP_SEED = [0x5E6D0001]


def next_seed(store=None) -> int:
    """A fresh dropout seed per training forward, drawn from torch's CPU generator so that
    ``torch.manual_seed`` makes train-mode runs reproducible.  Data-parallel ranks seed torch identically (identical
    replicas), so the rank is mixed in: every rank drops different elements of its own rows.
    With a device-side step state the ARGUMENT is the same every step (bit 63 set: the kernels XOR in the seed words that
    segmm_step_advance derives on the device), so that a recorded step can be replayed."""
    live = None if store is None else store.__dict__.get("live_seed")          # set by Trainer(device_state=True)
    if live is not None:
        return live
    s = int(torch.randint(0, 2 ** 62, (1,)).item())
    rank = int(os.environ.get("RANK", "0"))
    return (s ^ (rank * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1) if rank else s


class BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, store, bb, prefix, bb_index, usr_feat, usr_mask, vid_feat, vid_mask, train, seed, *params):
        run = BackboneRun(store, bb, prefix, bb_index)
        store.direct_grads = False          # decided by the backward of THIS pass (HeadLossFn.backward, trainer-seeded or not)
        store._gmax_fresh = False           # set by the head of THIS pass once it has measured max |d loss / d logits|
        vid, usr = run.forward(usr_feat, usr_mask, vid_feat, vid_mask, train, seed)
        ctx.run = run
        ctx.store = store
        ctx.names = [n for n in store.live_names if n.startswith(prefix)] if prefix else list(store.live_names)
        ctx.mark_non_differentiable(usr)
        ctx.set_materialize_grads(False)      # no [B, Lt, d] zero tensor for the non-differentiable user embedding output
        return vid, usr

    @staticmethod
    def backward(ctx, d_vid, d_usr):
        run, store = ctx.run, ctx.store
        gbuf = _pick_gbuf(store, ctx.names)
        if d_vid is None:      # the video states took no part in the differentiated scalar
            d_vid = torch.zeros((run.B, run.S if run.abl != "CrossMLP" else POOL_BINS, run.d), device=store.flat.device)
        run.backward(d_vid, gbuf, on_bucket=store.bucket_hook if gbuf is None else None)
        ctx.run = None
        return (None,) * 10 + grads_out(store, ctx.names, gbuf)


def _pick_gbuf(store, names):
    """Use the store's flat gradient buffer unless a parameter still holds a .grad (gradient accumulation over several
    backward calls without zero_grad): then compute into a fresh buffer, which ``deliver_grads`` adds to the held gradients."""
    if any(store._params[n].grad is not None for n in names):
        if store.bucket_hook is not None:
            raise RuntimeError("gradient accumulation under data parallelism is not supported: call zero_grad() before every "
                               "backward (the bucket all-reduces are issued from inside the backward)")
        return torch.zeros_like(store.gflat)
    return None


def grads_out(store, names, gbuf):
    """The parameter gradients of one autograd.Function.backward, as the tuple it returns for its parameter inputs.

    Trainer.train_step's backward (recognised by the head through the trainer's constant-one seed: ``store.direct_grads``)
    takes the fast path: ``deliver_grads`` hands every parameter a VIEW of the flat gradient buffer and autograd gets None.
    Any other backward -- ``loss.backward()``, ``torch.autograd.grad(loss, params)``, ``backward(inputs=...)`` -- returns the
    views THROUGH autograd like a plain nn.Module: AccumulateGrad copies them into ``.grad``, tensor hooks and
    post-accumulate-grad hooks on the parameters fire, ``autograd.grad`` receives them."""
    if store.__dict__.get("direct_grads", False):
        deliver_grads(store, names, gbuf)
        return (None,) * len(names)
    return tuple(store.g(n, gbuf) for n in names)


def deliver_grads(store, names, gbuf):
    """Hand the gradients of ``names`` to their parameters' ``.grad`` directly: views of the flat gradient buffer, so that the
    fused AdamW, the bucket all-reduces and any torch optimizer all see ONE buffer.  (Returned through autograd they would be
    cloned tensor by tensor by AccumulateGrad -- it never steals a view: 21 device copies and 33 MB per step at config 2.)
    A parameter that already holds a gradient (accumulation; ``gbuf`` is then a fresh buffer) gets the new one ADDED in place --
    into the flat buffer when that is what its ``.grad`` aliases, which is what the fused AdamW reads."""
    for n in names:
        p = store._params[n]
        g = store.g(n, gbuf)
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)


def backbone_apply(store, bb, prefix, usr_feat, usr_mask, vid_feat, vid_mask, training, bb_index=0, seed=None):
    store.ensure()
    if seed is None:
        seed = next_seed(store) if training else 0
    names = [n for n in store.live_names if n.startswith(prefix)] if prefix else list(store.live_names)
    params = [store._params[n] for n in names]
    return BackboneFn.apply(store, bb, prefix, bb_index, usr_feat, usr_mask, vid_feat, vid_mask, bool(training), int(seed), *params)
