"""ctypes binding of ``libsegmm_hip.so`` (C ABI declared in ``include/segmm_hip.h``).

PyTorch is only plumbing here: it owns device memory and the HIP stream; every wrapper passes raw
device pointers, sizes and ``torch.cuda.current_stream()`` through the C ABI.  There is NO fallback:
if the shared library is missing or a call fails, a ``RuntimeError`` is raised (the product path
must never run on a silent eager/CPU substitute).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEGMM_LIB") or os.path.join(_HERE, "libsegmm_hip.so")      # SEGMM_LIB: A/B builds of the kernels
ABI_VERSION = 29

_lib = None

_i, _i64, _f, _u64, _u32, _p = C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p

# name -> argtypes; the single source of truth for the exported symbol set (tests check it against the header)
SIGNATURES = {
    "segmm_l1norm": [_p, _p, _p, _i64, _i, _p, _p, _i, _p, _p, _p],
    "segmm_gemm": [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p, _i, _f, _u64, _u32, _i, _p, _i, _i, _p],
    "segmm_gemm_x": [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p, _i, _f, _u64, _u32, _i, _p, _i,
                     _p, _i64, _p, _i64, _i, _p],
    "segmm_gemm_h": [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p, _i, _f, _u64, _u32, _i, _p, _i,
                     _p, _i64, _p, _i64, _p, _i, _p, _i, _p, _p],
    "segmm_gemm_p": [_i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p, _i, _f, _u64,
                     _u32, _i, _p, _i, _p, _p],
    "segmm_scales_update": [_p, _p, _i, _p, _p, _i, _p, _p, _p],
    "segmm_probe_mfma_rate": [_i, _i, _p, _p, _p],
    "segmm_attn_mode": [_i],
    "segmm_config_set": [_p, _i],
    "segmm_config_dump": [_p, _i],
    "segmm_site_fixup": [_p, _p, _p, _p, _p, _p],
    "segmm_step_bind": [_p],
    "segmm_step_state_bytes": [],
    "segmm_step_set": [_u64, _i, _f, _f, _p],
    "segmm_step_advance": [_f, _f, _p],
    "segmm_step_get": [_p, _p, _p, _p],
    "segmm_loss_finish": [_p, _i, _p, _p, _p, _p, _i64, _p, _p, _i, _p, _i, _p],
    "segmm_split_p32": [_p, _i64, _i, _i, _p, _i, _p, _i, _p],
    "segmm_split_p32_transpose": [_p, _i, _i, _i, _p, _i, _p, _p],
    "segmm_wsplit_p32": [_p, _p, _i, _i, _p, _p, _p, _p],
    "segmm_absmax": [_p, _i64, _i, _i, _p, _i, _p],
    "segmm_split2h": [_p, _p, _i64, _i64, _p, _i, _p],
    "segmm_split2h_transpose": [_p, _i, _i, _i, _p, _i64, _p, _i, _p],
    "segmm_split3": [_p, _p, _i64, _i64, _p],
    "segmm_split3_transpose": [_p, _i, _i, _i, _p, _i64, _p],
    "segmm_layernorm_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _u64, _u32, _p, _p, _i, _p, _p, _p],
    "segmm_layernorm_fwd_dot": [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _u64, _u32, _p, _p, _i, _p, _p, _p, _p, _p, _p],
    "segmm_layernorm_bwd_parts": [_i64, _i],
    "segmm_layernorm_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _u32, _f, _u32, _u64, _p, _p, _i, _p, _p, _p],
    "segmm_layernorm_bwd_outer": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _u32, _f, _u32, _u64, _p, _p, _i, _p, _p, _p],
    "segmm_layernorm_bwd_pos_parts": [_i64, _i, _i],
    "segmm_layernorm_bwd_pos": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _u32, _f, _u32, _u64, _p, _p, _i, _p, _p, _p, _i, _p],
    "segmm_colsum_pos": [_p, _i, _i, _i, _p, _p],
    "segmm_colsum_chunks": [_i64],
    "segmm_colsum": [_p, _i, _p, _i64, _i, _p, _i, _p, _p],
    "segmm_attn_fwd": [_i] * 6 + [_p, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _p, _f, _u64, _u32, _p, _p, _p],
    "segmm_attn_bwd": [_i] * 6 + [_p, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _i,
                                  _p, _p, _i, _f, _u64, _u32, _p, _p, _p, _i, _p, _p],
    "segmm_rowdot": [_p, _i, _p, _p, _p, _i64, _i, _i, _p],
    "segmm_rowscale_bcast": [_p, _p, _p, _i, _i64, _i, _i, _p],
    "segmm_vecsum": [_p, _i64, _p, _i, _p],
    "segmm_rowdot_pair": [_p, _i, _p, _i, _p, _i64, _i, _i, _p],
    "segmm_rowscale_mat": [_p, _p, _i, _p, _i, _i64, _i, _i, _p],
    "segmm_embed_id_vid": [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i64, _p],
    "segmm_embed_id_usr": [_p, _p, _i, _p, _p, _i, _i64, _p],
    "segmm_embed_id_bwd": [_p, _i, _i, _i, _i, _p, _p, _p, _i, _i64, _p],
    "segmm_pe_grad": [_p, _i, _i, _i, _i, _p, _i, _p],
    "segmm_argsort_ids": [_p, _i, _p, _p],
    "segmm_argsort_ids_ws": [_p, _i, _p, _p, _p],
    "segmm_label_stats_unpack": [_p, _i, _i, _p, _p, _p, _p],
    "segmm_zero_rows": [_p, _i, _p, _i, _i64, _p],
    "segmm_label_stats": [_p, _i, _i, _i, _p, _p, _p, _p],
    "segmm_loss_fwd_bwd": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p],
    "segmm_adamw": [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i, _p],
    "segmm_adamw_table": [_p, _p, _p, _p, _i64, _i, _p, _i, _p, _f, _f, _f, _f, _f, _i, _i, _p],
    "segmm_dropout_mult": [_p, _i64, _f, _u64, _u32, _p],
    "segmm_rank_leave": [_p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p],
    "segmm_auc_counts": [_p, _p, _p, _i, _p, _p],
    "segmm_survival": [_p, _i, _p, _p, _p, _i, _i, _p],
    "segmm_gather_l1": [_p, _i64, _i, _p, _i64, _i, _p, _p, _p, _p, _i, _p, _p, _p],
    "segmm_segment_weighted_sum": [_p, _p, _p, _i64, _i, _p, _p],
    "segmm_colsum3": [_p, _p, _p, _i, _i64, _i, _p, _p, _p, _p, _p],
    "segmm_pool_tokens": [_p, _i, _p, _i, _p, _i, _i, _i, _p],
    "segmm_pool_tokens_bwd": [_p, _p, _i, _p, _i, _i, _i, _i, _p],
    "segmm_bias_grad": [_p, _i, _i, _p, _p, _p],
    "segmm_focal_relabel": [_p, _i64, _p],
    "segmm_rand_uniform": [_p, _i64, _u64, _u32, _p],
    "segmm_rand_ids": [_p, _i64, _i64, _i64, _u64, _u32, _p],
    "segmm_rand_perm_rows": [_p, _i, _i, _u64, _u32, _p],
    "segmm_fill_zero": [_p, _i64, _p],
    "segmm_copy_bytes": [_p, _p, _i64, _p],
    "segmm_cmd_op_count": [],
    "segmm_run_phase": [_p, _p, _i, _p],
    "segmm_step_begin": [_p, _p, _i, _p],
    "segmm_embed_fwd": [_p, _p, _i, _p],
    "segmm_layer_fwd": [_p, _p, _i, _p],
    "segmm_head_loss_fwd": [_p, _p, _i, _p],
    "segmm_head_loss_bwd": [_p, _p, _i, _p],
    "segmm_layer_bwd": [_p, _p, _i, _p],
    "segmm_embed_bwd": [_p, _p, _i, _p],
    "segmm_step_tail": [_p, _p, _i, _p],
}


def lib():
    """The shared library (loaded once; raises if it has not been built) -- or, while a step is being recorded, a proxy that
    records every dispatchable call before making it."""
    L = _lib_real()
    rec = RECORDER
    if rec is not None:
        return _RecLib(L, rec)
    return L


def _lib_real():
    """Loads the shared library once; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("segmminterest_amd: %s is missing -- build it with __graft_entry__.build() "
                           "(hipcc --offload-arch=gfx950); there is no fallback path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.segmm_last_error.restype = C.c_char_p
    L.segmm_last_error.argtypes = []
    L.segmm_abi_version.restype = _i
    L.segmm_abi_version.argtypes = []
    L.segmm_cmd_op_name.restype = C.c_char_p
    L.segmm_cmd_op_name.argtypes = [_i]
    for name, at in SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = at
        fn.restype = _i
    if L.segmm_abi_version() != ABI_VERSION:
        raise RuntimeError("libsegmm_hip.so ABI %d != expected %d: rebuild" % (L.segmm_abi_version(), ABI_VERSION))
    _lib = L
    return L


# =============================================================================================== recorded launch sequences
# include/segmm_hip.h "Recorded launch sequences": the per-op calls of ONE eager training step are recorded (entry point, its
# arguments, the stream slot) and replayed every step from C by segmm_run_phase / the named phase entry points -- one
# foreign-function call per part of the step instead of one per kernel (+ the Python that sequences them).
CMD_MAX_ARGS = 48
OP_FORK, OP_JOIN = -1, -2
(PHASE_STEP_BEGIN, PHASE_EMBED_FWD, PHASE_LAYER_FWD, PHASE_HEAD_LOSS_FWD, PHASE_HEAD_LOSS_BWD, PHASE_LAYER_BWD, PHASE_EMBED_BWD,
 PHASE_STEP_TAIL) = range(8)
PHASE_ENTRY = ("segmm_step_begin", "segmm_embed_fwd", "segmm_layer_fwd", "segmm_head_loss_fwd", "segmm_head_loss_bwd",
               "segmm_layer_bwd", "segmm_embed_bwd", "segmm_step_tail")


class CmdArg(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double), ("p", C.c_void_p)]


class Cmd(C.Structure):
    """segmm_cmd_t"""
    _fields_ = [("op", C.c_int32), ("stream", C.c_int32), ("a", CmdArg * CMD_MAX_ARGS)]


class Phase(C.Structure):
    """segmm_phase_t"""
    _fields_ = [("kind", C.c_int32), ("backbone", C.c_int32), ("layer", C.c_int32), ("n_cmds", C.c_int32), ("cmds", C.POINTER(Cmd))]


_op_ids = None


def op_ids():
    """{entry point name: op id} of the library's dispatch table (segmm_cmd_op_name)."""
    global _op_ids
    if _op_ids is None:
        L = lib()
        _op_ids = {L.segmm_cmd_op_name(i).decode(): i for i in range(L.segmm_cmd_op_count())}
    return _op_ids


RECORDER = None          # a Recorder while Trainer.record() runs its one eager step


class Recorder:
    """Collects the C-ABI calls of one eager step, split into phases by :func:`mark`."""

    def __init__(self, main_stream: int, side_stream: int, aux_stream: int = 0):
        self.streams = {int(main_stream): 0, int(side_stream): 1}
        if aux_stream:
            self.streams[int(aux_stream)] = 2
        self.phases = []          # [kind, backbone, layer, [(op, slot, [(field, value)])]]
        self.keep = []            # host structs / arrays the recorded pointer arguments name
        self.pslots = {}          # finish(): index in its result -> [(command, argument slot)] of the pointer arguments
        self.ops = op_ids()

    def mark(self, kind, backbone=0, layer=0):
        self.phases.append([int(kind), int(backbone), int(layer), []])

    def _cmds(self):
        if not self.phases or callable(self.phases[-1]):
            raise RuntimeError("recorder: a launch before the first phase marker")
        return self.phases[-1][3]

    def pseudo(self, op, slot=1):
        self._cmds().append((op, slot, []))

    def callback(self, fn):
        """A host action between launches (a data-parallel collective, a wait for one): replayed by calling ``fn`` again at the
        same point of the launch order.  The launches that follow continue the current phase in a new fragment."""
        last = self.phases[-1] if self.phases else [PHASE_STEP_BEGIN, 0, 0, []]
        self.phases.append(fn)
        self.phases.append([last[0], last[1], last[2], []])

    def call(self, name, args):
        at = SIGNATURES[name]
        if len(args) != len(at) or at[-1] is not _p:
            raise RuntimeError("recorder: %s called with %d arguments" % (name, len(args)))
        slot = self.streams.get(int(args[-1] or 0))
        if slot is None:
            raise RuntimeError("recorder: %s was enqueued on a stream that is neither the step's main, side nor auxiliary stream; this "
                               "launch sequence cannot be replayed (prefetch / third-stream knobs must be off)" % name)
        vals = []
        for k, (ct, v) in enumerate(zip(at[:-1], args[:-1])):
            if ct is _f:
                vals.append(("f", float(v)))
            elif ct is _p:
                if v is None:
                    vals.append(("p", None))
                elif isinstance(v, int):
                    vals.append(("p", v or None))
                else:          # byref(struct) / cast(array) / ctypes instance: a host object that must outlive the replay
                    obj = getattr(v, "_obj", v)
                    self.keep.append((v, obj))
                    addr = v.value if isinstance(v, C.c_void_p) else C.addressof(obj)
                    vals.append(("p", addr))
            else:
                iv = int(v)
                vals.append(("i", iv - (1 << 64) if iv >= (1 << 63) else iv))
        self._cmds().append((self.ops[name], slot, vals))

    def finish(self):
        """-> list of (Phase struct, Cmd array); the arrays are referenced by the Phase structs (keep both)."""
        out = []
        for item in self.phases:
            if callable(item):
                out.append((item, None))
                continue
            kind, bb, layer, cmds = item
            if not cmds:          # (an empty fragment behind a callback)
                continue
            arr = (Cmd * max(len(cmds), 1))()
            pslots = []          # (command, argument slot) of every non-null POINTER argument: what a replay may re-base
            for ci, (c, (op, slot, vals)) in enumerate(zip(arr, cmds)):
                c.op, c.stream = op, slot
                for k, (fld, v) in enumerate(vals):
                    setattr(c.a[k], fld, v)
                    if fld == "p" and v:
                        pslots.append((ci, k))
            self.pslots[len(out)] = pslots
            ph = Phase(kind=kind, backbone=bb, layer=layer, n_cmds=len(cmds), cmds=C.cast(arr, C.POINTER(Cmd)))
            out.append((ph, arr))
        return out


class _RecLib:
    """``lib()`` while recording: every dispatchable entry point is recorded, then executed."""

    def __init__(self, real, rec):
        self._real, self._rec = real, rec

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        rec = self._rec
        if name not in rec.ops:
            return fn

        def call(*args):
            rec.call(name, args)
            return fn(*args)
        return call


def torch_fallback(what):
    """A torch kernel (dtype / layout conversion, sort, copy) is about to run inside the step: fine in an eager step, fatal
    while the step is being recorded -- the launch would be missing from the replay, which would keep using the record-time
    result without any error."""
    if RECORDER is not None:
        raise RuntimeError("record(): %s runs a torch kernel inside the step; a recorded step would silently drop it "
                           "(pass bool masks, int64 ids / labels and contiguous fp32 features, at most %d ids per table lookup)" % (what, ARGSORT_MAX))


def mark(kind, backbone=0, layer=0):
    """Phase boundary of the step (no-op unless a step is being recorded)."""
    if RECORDER is not None:
        RECORDER.mark(kind, backbone, layer)


def host_action(fn):
    """Run ``fn()`` now; while a step is being recorded, also note it as a host action of the step (Recorder.callback)."""
    if RECORDER is not None:
        RECORDER.callback(fn)
    return fn()


def stream_table(streams, events):
    """(ctypes array of hipStream_t, count, ctypes array of hipEvent_t) for :func:`run_phase`: ``streams`` = raw handles of the main,
    side (and auxiliary) stream, ``events`` = two raw event handles per non-main stream."""
    sa = (C.c_void_p * len(streams))(*[int(x) for x in streams])
    ea = (C.c_void_p * max(len(events), 1))(*[int(x) for x in events])
    return sa, len(streams), ea


def run_phase(phase: "Phase", table):
    L = _lib_real()
    sa, n, ea = table
    _check(getattr(L, PHASE_ENTRY[phase.kind])(C.addressof(phase), sa, n, ea), PHASE_ENTRY[phase.kind])


def fill_zero(t: torch.Tensor):
    """``t.zero_()`` through the C ABI (a recordable command; contiguous tensors only)."""
    if not t.is_contiguous():
        raise RuntimeError("fill_zero: non-contiguous view")
    if t.numel():
        _check(lib().segmm_fill_zero(t.data_ptr(), t.numel() * t.element_size(), _stream()), "segmm_fill_zero")


def copy_bytes(dst: torch.Tensor, src: torch.Tensor):
    """``dst.copy_(src)`` of same-sized contiguous tensors through the C ABI (a recordable, re-basable command)."""
    nb = src.numel() * src.element_size()
    if not (dst.is_contiguous() and src.is_contiguous()) or dst.numel() * dst.element_size() != nb:
        raise RuntimeError("copy_bytes: contiguous tensors of the same byte size")
    if nb:
        _check(lib().segmm_copy_bytes(dst.data_ptr(), src.data_ptr(), nb, _stream()), "segmm_copy_bytes")


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _lib_real().segmm_last_error().decode()))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw-handle query: no Stream object per launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("segmminterest_amd kernels need device tensors (got %s); no CPU fallback" % t.device)


def _f32c(t, name="tensor"):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("%s must be contiguous float32 (got %s, contiguous=%s)" % (name, t.dtype, t.is_contiguous()))
    return t


GEMM_PROFILE = None       # bench.py sets this to a list to time every GEMM launch with HIP events
ATTN_PROFILE = None       # ... and every attention launch: (kind, B, H, dh, Lq, La, Lb, event0, event1)
KERNEL_PROFILE = None     # ... and selected HBM-bound launches: (name, algorithmic bytes, event0, event1)


class _kprof:
    """``with _kprof(name, nbytes):`` -- HIP events around a launch when bench.py asked for them (KERNEL_PROFILE is a list)."""

    def __init__(self, name, nbytes):
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        self.on = KERNEL_PROFILE is not None
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            KERNEL_PROFILE.append((self.name, self.nbytes, self.e0, e1))


def attn_peak_tflops():
    """Dense MFMA peak of the instruction the attention kernels use (roofline denominator of bench.py)."""
    return 157.3          # v_mfma_f32_16x16x4_f32 (exact fp32 products)


ATTN_PIN_CALLS = 0          # attention calls that were handed input planes (bench.py names the kernels it measured by this)


def attn_kernel_name():
    if ATTN_PIN_CALLS:
        return ("attn_fwd_pl_kernel + attn_bwd_pl_kernel (csrc/attention_pl.h): Q / K / V read from the projection GEMMs' P32 planes "
                "(forward: K / V by LDS-DMA into a conflict-free rotated image, 9 key tiles over both key blocks; backward: dQ + dK + dV in "
                "one kernel per key block, row fragments straight to registers, K^T read back transposed from LDS), 3 x "
                "v_mfma_f32_16x16x16_f16 per product, O / dQ / dK / dV written as planes (the backward: planes only + a repair launch)")
    return ("attn_fwd (v_mfma_f32_16x16x4_f32, exact fp32 products) + attn_bwd_fused16 / attn_bwd_fused (dQ + dK + dV in one kernel per key "
            "block, S and dP computed once: 10 dh Lq T FLOP instead of 14; fp16x3 products -- 3 x v_mfma_f32_16x16x16_f16 -- for single-chunk "
            "launches with more than 32 queries, exact fp32 products otherwise)")
ENGINE_F32, ENGINE_BF16X6, ENGINE_F16X3, ENGINE_F16X3P = 0, 1, 2, 3
# default engine of gemm(): the scaled two-term fp16 split on the fp16 matrix cores (22-bit operands, three exact
# partial products, fp32 accumulation: measured error vs fp64 at or below the f32-MFMA kernel's on every layout).
# SEGMM_GEMM=bf16x6 selects the exact 3-way bf16 split (six products), SEGMM_GEMM=f32 the f32-input MFMA kernel
# (A/B and parity cross-checks)
# "f16x3p" (default): the same fp16x3 arithmetic with operands PRE-SPLIT into fp16 planes by their producers and staged by
# LDS-DMA (gemm_p / csrc/gemm_planes.h); raw gemm() calls under it run the on-the-fly fp16x3 kernel.
GEMM_ENGINE = {"f32": 0, "bf16x6": 1, "f16x3": 2, "f16x3p": 3}[os.environ.get("SEGMM_GEMM", "f16x3p")]
AMAX_SLOTS = 256          # partial maxima per tensor written by the fused producers (SEGMM_AMAX_SLOTS)
AMAX_PARTS = 1024         # ... and by the stand-alone absmax() pass
LAYOUT_NT, LAYOUT_NN, LAYOUT_TN = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_RELU, ACT_DRELU = 0, 1, 2, 3, 4


class PO:
    """Plane output of a producer kernel: (planes tensor, element offset, ld2, site header, device address of the delayed scale)."""
    __slots__ = ("planes", "p_off", "ld2", "hdr", "scale_ptr")

    def __init__(self, planes, ld2, hdr, scale_ptr, p_off=0):
        self.planes, self.ld2, self.hdr, self.scale_ptr, self.p_off = planes, ld2, hdr, scale_ptr, p_off

    def args(self):
        return (self.planes.data_ptr() + 2 * self.p_off, self.ld2, self.hdr.data_ptr(), self.scale_ptr)


_NO_PO = (None, 0, None, None)


def _po(po):
    return _NO_PO if po is None else po.args()


def l1norm(x, out=None, inv_scale=None, amax=None, po=None):
    """``amax``: optional zeroed [AMAX_SLOTS] vector that receives the partial maxima of |out| (fp16x3 GEMM operand);
    ``po``: optional plane output (PO)."""
    _dev(x)
    D = x.shape[-1]
    rows = x.numel() // D
    _check(lib().segmm_l1norm(_ptr(x), _ptr(out), _ptr(inv_scale), rows, D, _ptr(amax), *_po(po), _stream()), "segmm_l1norm")


def gemm(layout, M, N, K, A, lda, B, ldb, Cout, ldc, bias=None, row_scale=None, residual=None, ldr=0, res_period=0,
         activation=0, aux=None, ldaux=0, drop_p=0.0, seed=0, site=0, splits=1, workspace=None, accumulate=False,
         a_off=0, b_off=0, c_off=0, engine=None, a_planes=None, b_planes=None, nplanes=3, a_amax=None, b_amax=None, c_amax=None):
    """Raw strided GEMM; ``*_off`` are element offsets into the tensors (column slices of fused buffers).
    ``a_planes`` / ``b_planes`` = (16-bit planes tensor [nplanes, ...], element offset): pre-split operand (NT only).
    ``a_amax`` / ``b_amax`` (fp16x3 engine): float32 vectors of partial maxima of |A| / |B|; computed here with
    :func:`absmax` when the caller has none.  ``c_amax``: zeroed [AMAX_SLOTS] vector that receives the partial
    maxima of |C| (ignored by the other engines: nothing consumes it there)."""
    _dev(Cout)
    es = 4
    prof = GEMM_PROFILE
    if prof is not None:          # bench.py: HIP events on the launch stream around the dominant kernel
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    eng = GEMM_ENGINE if engine is None else int(engine)
    if eng == ENGINE_F16X3P:
        eng = ENGINE_F16X3
    if eng == ENGINE_F16X3:
        if a_amax is None:
            if a_planes is not None:
                raise RuntimeError("pre-split fp16 planes need the partial maxima they were made with")
            a_amax = absmax(A, K if layout == LAYOUT_TN else M, M if layout == LAYOUT_TN else K, lda, off=a_off)
        if b_amax is None:
            if b_planes is not None:
                raise RuntimeError("pre-split fp16 planes need the partial maxima they were made with")
            b_amax = absmax(B, N if layout == LAYOUT_NT else K, K if layout == LAYOUT_NT else N, ldb, off=b_off)
        ap = None if a_planes is None else a_planes[0].data_ptr() + 2 * a_planes[1]
        aps = 0 if a_planes is None else a_planes[0].stride(0)
        bp = None if b_planes is None else b_planes[0].data_ptr() + 2 * b_planes[1]
        bps = 0 if b_planes is None else b_planes[0].stride(0)
        _check(lib().segmm_gemm_h(layout, M, N, K, None if A is None else A.data_ptr() + a_off * es, lda,
                                  None if B is None else B.data_ptr() + b_off * es, ldb,
                                  Cout.data_ptr() + c_off * es, ldc, _ptr(bias), _ptr(row_scale), _ptr(residual), ldr,
                                  res_period, activation, _ptr(aux), ldaux, float(drop_p), int(seed), int(site),
                                  int(splits), _ptr(workspace), int(bool(accumulate)), ap, aps, bp, bps,
                                  a_amax.data_ptr(), a_amax.numel(), b_amax.data_ptr(), b_amax.numel(), _ptr(c_amax),
                                  _stream()), "segmm_gemm_h")
    elif a_planes is None and b_planes is None and nplanes == 3:
        _check(lib().segmm_gemm(layout, M, N, K, A.data_ptr() + a_off * es, lda, B.data_ptr() + b_off * es, ldb,
                                Cout.data_ptr() + c_off * es, ldc, _ptr(bias), _ptr(row_scale), _ptr(residual), ldr,
                                res_period, activation, _ptr(aux), ldaux, float(drop_p), int(seed), int(site),
                                int(splits), _ptr(workspace), int(bool(accumulate)), eng, _stream()), "segmm_gemm")
    else:
        if eng != ENGINE_BF16X6:
            raise RuntimeError("pre-split operands / nplanes=2 need the bf16x6 engine")
        ap = None if a_planes is None else a_planes[0].data_ptr() + 2 * a_planes[1]
        aps = 0 if a_planes is None else a_planes[0].stride(0)
        bp = None if b_planes is None else b_planes[0].data_ptr() + 2 * b_planes[1]
        bps = 0 if b_planes is None else b_planes[0].stride(0)
        _check(lib().segmm_gemm_x(layout, M, N, K, None if A is None else A.data_ptr() + a_off * es, lda,
                                  None if B is None else B.data_ptr() + b_off * es, ldb,
                                  Cout.data_ptr() + c_off * es, ldc, _ptr(bias), _ptr(row_scale), _ptr(residual), ldr,
                                  res_period, activation, _ptr(aux), ldaux, float(drop_p), int(seed), int(site),
                                  int(splits), _ptr(workspace), int(bool(accumulate)), ap, aps, bp, bps, int(nplanes),
                                  _stream()), "segmm_gemm_x")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append((layout, M, N, K, e0, e1))


SITE_HDR = 8                                  # floats in front of the partial maxima of a plane tensor's site header
SITE_FLOATS = SITE_HDR + AMAX_SLOTS


class PT:
    """A GEMM operand as its producer leaves it: P32 fp16 planes (``planes`` [rows, 2 * cols] halves, or a wider buffer
    with ``p_off`` / ``ld2`` selecting a column slice), the site header ``hdr`` ([SITE_FLOATS] floats: scale, overflow flag,
    partial maxima) and -- optionally -- the fp32 copy ``f32`` (``f_off`` / ``ldf``) that consumers fall back to."""
    __slots__ = ("planes", "p_off", "ld2", "hdr", "f32", "f_off", "ldf", "rows", "cols")

    def __init__(self, planes, hdr, rows, cols, ld2=None, p_off=0, f32=None, ldf=None, f_off=0):
        self.planes, self.hdr, self.rows, self.cols = planes, hdr, rows, cols
        self.ld2 = 2 * cols if ld2 is None else ld2
        self.p_off = p_off
        self.f32, self.ldf, self.f_off = f32, (cols if ldf is None else ldf), f_off

    def cols_slice(self, c0, ncols):
        """Columns [c0, c0 + ncols) of the same rows (c0, ncols multiples of 32)."""
        return PT(self.planes, self.hdr, self.rows, ncols, self.ld2, self.p_off + 2 * c0, self.f32, self.ldf, self.f_off + c0)

    def pptr(self):
        return self.planes.data_ptr() + 2 * self.p_off

    def fptr(self):
        return None if self.f32 is None else self.f32.data_ptr() + 4 * self.f_off


def new_site(device, n=1):
    """Zeroed site headers [n, SITE_FLOATS]."""
    return torch.zeros((n, SITE_FLOATS), dtype=torch.float32, device=device)


def split_p32(x, rows, cols, ld, planes, ld2, hdr, mode=0, x_off=0, p_off=0):
    """fp32 view -> P32 planes; mode 0: exact scale from the header's (complete) partial maxima; 1: scale hdr[0] as given."""
    _check(lib().segmm_split_p32(x.data_ptr() + 4 * x_off, rows, cols, ld, planes.data_ptr() + 2 * p_off, ld2, hdr.data_ptr(), int(mode),
                                 _stream()), "segmm_split_p32")


def split_p32_transpose(x, R, Cc, ld, planes, ld2, hdr, x_off=0, p_off=0):
    _check(lib().segmm_split_p32_transpose(x.data_ptr() + 4 * x_off, R, Cc, ld, planes.data_ptr() + 2 * p_off, ld2, hdr.data_ptr(),
                                           _stream()), "segmm_split_p32_transpose")


def wsplit_p32(flat, desc, n_mats, n_tiles, hdr, wpl, wTpl):
    """Absmax + exact P32 split (+ transposed split) of every weight matrix described in ``desc`` (see segmm_wsplit_p32)."""
    _check(lib().segmm_wsplit_p32(flat.data_ptr(), desc.data_ptr(), int(n_mats), int(n_tiles), hdr.data_ptr(), wpl.data_ptr(),
                                  wTpl.data_ptr(), _stream()), "segmm_wsplit_p32")


def to_planes(x, rows, cols, ld=None, x_off=0, keep_f32=True):
    """Stand-alone adapter: absmax pass + exact split pass -> PT (tests, external inputs, first use of a site)."""
    ld = cols if ld is None else ld
    hdr = new_site(x.device)[0]
    absmax(x, rows, cols, ld, off=x_off, out=hdr[SITE_HDR:])
    planes = torch.empty((rows, 2 * cols), dtype=torch.float16, device=x.device)
    split_p32(x, rows, cols, ld, planes, 2 * cols, hdr, mode=0, x_off=x_off)
    return PT(planes, hdr, rows, cols, f32=x if keep_f32 else None, ldf=ld, f_off=x_off)


def gemm_p(layout, M, N, K, A: "PT", B: "PT", Cout, ldc, c_pt: "PT" = None, write_c=True, bias=None, row_scale=None, residual=None,
           ldr=0, res_period=0, activation=0, aux=None, ldaux=0, drop_p=0.0, seed=0, site=0, splits=1, workspace=None,
           accumulate=False, c_off=0, c_hdr=None, c_scale_ptr=None, colsum_out=None, repair=False):
    """Plane-operand GEMM (segmm_gemm_p).  ``c_pt``: optional plane output (its hdr[0] holds the scale to write with);
    ``c_hdr``: site header that only receives the partial maxima of |C| (no plane output).  ``repair``: the REPAIR launch of a
    planes-only NT output (``Cout=None, write_c=False``): same arguments, workgroups leave at once unless the output site's planes
    are unusable under the recorded scale, in which case they are rewritten with the exact scale of the recorded maxima."""
    prof = GEMM_PROFILE if not repair else None          # (a repair launch does no work normally: not a GEMM of the roofline)
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(lib().segmm_gemm_p(layout, M, N, K, A.pptr(), A.ld2, A.hdr.data_ptr(), A.fptr(), A.ldf, B.pptr(), B.ld2, B.hdr.data_ptr(),
                              B.fptr(), B.ldf, None if Cout is None else Cout.data_ptr() + 4 * c_off, ldc,
                              None if c_pt is None else c_pt.pptr(), 0 if c_pt is None else c_pt.ld2,
                              (None if c_hdr is None else c_hdr.data_ptr()) if c_pt is None else c_pt.hdr.data_ptr(), c_scale_ptr,
                              int(bool(write_c)) | (2 if repair else 0),
                              _ptr(bias), _ptr(row_scale), _ptr(residual),
                              ldr, res_period, activation, _ptr(aux), ldaux, float(drop_p), int(seed), int(site), int(splits),
                              _ptr(workspace), int(bool(accumulate)), _ptr(colsum_out), _stream()), "segmm_gemm_p")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append((10 + layout, M, N, K, e0, e1))          # 10 +: a plane-engine launch (bench.py's per-shape table)


def scales_update(arena, site_idx, n_rows, site_scale, stats, target=12, gain=None, gmax=None):
    _check(lib().segmm_scales_update(arena.data_ptr(), site_idx.data_ptr(), int(n_rows), site_scale.data_ptr(), stats.data_ptr(), int(target),
                                     _ptr(gain), _ptr(gmax), _stream()), "segmm_scales_update")


LIVE_SEED = 1 << 63          # dropout seed argument bit: XOR the device-side step words into the seed (segmm_step_advance)


def step_state_bytes():
    return int(lib().segmm_step_state_bytes())


def step_bind(state):
    """``segmm_step_bind``: the device-side step state (a caller-owned device tensor of step_state_bytes() bytes, or None: the
    library's own default state) that the launches which follow use -- the pointer travels in their arguments."""
    _check(_lib_real().segmm_step_bind(None if state is None else state.data_ptr()), "segmm_step_bind")


def step_set(seed, step, beta1=0.9, beta2=0.999):
    _check(lib().segmm_step_set(int(seed) & (2 ** 63 - 1), int(step), float(beta1), float(beta2), _stream()), "segmm_step_set")


def step_advance(beta1=0.9, beta2=0.999):
    _check(lib().segmm_step_advance(float(beta1), float(beta2), _stream()), "segmm_step_advance")


def step_get():
    """(seed words, step count, (bc1, sqrt(bc2))) of the device-side step state; synchronises the stream."""
    import ctypes
    seed, step, bc = ctypes.c_uint64(0), ctypes.c_int(0), (ctypes.c_float * 2)()
    _check(lib().segmm_step_get(ctypes.addressof(seed), ctypes.addressof(step), ctypes.addressof(bc), _stream()), "segmm_step_get")
    return int(seed.value), int(step.value), (float(bc[0]), float(bc[1]))


def config_set(name, value):
    """``segmm_config_set``: set the tuning knob ``name`` (without the SEGMM_ prefix; see :func:`config_dump`); returns the previous value."""
    _KNOBS.clear()
    r = int(lib().segmm_config_set(name.encode(), int(value)))
    if r < 0 and name not in config_dump():
        _check(r, "segmm_config_set")
    return r


_KNOBS = {}


def knob(name):
    """Current value of one tuning knob (cached; ``config_set`` drops the cache)."""
    if not _KNOBS:
        _KNOBS.update({k: v[0] for k, v in config_dump().items()})
    return _KNOBS[name]


def config_dump():
    """{knob name: (value, doc)} of the library's tuning table (``segmm_config_dump``)."""
    n = int(lib().segmm_config_dump(None, 0)) + 1
    buf = C.create_string_buffer(n)
    lib().segmm_config_dump(buf, n)
    out = {}
    for ln in buf.value.decode().splitlines():
        k, rest = ln.split("=", 1)
        v, doc = rest.split("#", 1)
        out[k.replace("SEGMM_", "", 1)] = (int(v), doc.strip())
    return out


def attn_mode(mode=-1):
    """``segmm_attn_mode``: 0 exact-fp32 attention kernels, 1 fp16x3 where faster (default), 2 fp16x3 wherever built;
    returns the previous mode, ``mode < 0`` only queries."""
    return int(lib().segmm_attn_mode(int(mode)))


def mfma_sustained_tflops(ms_target=25.0):
    """Diagnostic: sustained fp16 matrix-core rate (TFLOP/s) of this GPU on random operand bits, registers only
    (``segmm_probe_mfma_rate``: 256 workgroups x 8 waves, the plane GEMM's accumulator order).  bench.py reports it beside
    the datasheet peak: the power management clocks a random-data MFMA stream down, so no real-data GEMM can reach the
    datasheet figure on this part."""
    import ctypes
    scratch = torch.zeros(1, dtype=torch.float32, device="cuda")
    fl = ctypes.c_double(0.0)
    iters = 4000

    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(lib().segmm_probe_mfma_rate(256, int(n), scratch.data_ptr(), ctypes.addressof(fl), _stream()), "segmm_probe_mfma_rate")
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    ms = run(iters)                                   # warm-up + calibration
    iters = max(1000, int(iters * ms_target / max(ms, 1e-3)))
    ms = run(iters)
    return fl.value / (ms * 1e-3) * 1e-12


def absmax(x, rows, cols, ld, off=0, out=None):
    """Partial maxima of |x| over the [rows, cols] view (row stride ld) starting ``off`` elements into x."""
    if out is None:
        out = torch.empty(AMAX_PARTS, dtype=torch.float32, device=x.device)
    _check(lib().segmm_absmax(x.data_ptr() + 4 * off, rows, cols, ld, out.data_ptr(), out.numel(), _stream()), "segmm_absmax")
    return out


def split2h(x, planes, n, amax, x_off=0, p_off=0):
    """planes[0/1, p_off + i] = fp16 hi / lo of x.flat[x_off + i] * s(amax); planes is a [2, size] 16-bit tensor."""
    _check(lib().segmm_split2h(x.data_ptr() + 4 * x_off, planes.data_ptr() + 2 * p_off, n, planes.stride(0),
                               amax.data_ptr(), amax.numel(), _stream()), "segmm_split2h")


def split2h_transpose(x, R, Cc, ld, planes, amax, x_off=0, p_off=0):
    """planes[0/1, p_off + c*R + r] = fp16 hi / lo of x.flat[x_off + r*ld + c] * s(amax)."""
    _check(lib().segmm_split2h_transpose(x.data_ptr() + 4 * x_off, R, Cc, ld, planes.data_ptr() + 2 * p_off, planes.stride(0),
                                         amax.data_ptr(), amax.numel(), _stream()), "segmm_split2h_transpose")


def split3(x, planes, n, x_off=0, p_off=0):
    """planes[p, p_off + i] = p-th bf16 term of x.flat[x_off + i]; planes is a [3, size] bf16 tensor."""
    _check(lib().segmm_split3(x.data_ptr() + 4 * x_off, planes.data_ptr() + 2 * p_off, n, planes.stride(0), _stream()), "segmm_split3")


def split3_transpose(x, R, Cc, ld, planes, x_off=0, p_off=0):
    """planes[p, p_off + c*R + r] = p-th bf16 term of x.flat[x_off + r*ld + c]."""
    _check(lib().segmm_split3_transpose(x.data_ptr() + 4 * x_off, R, Cc, ld, planes.data_ptr() + 2 * p_off, planes.stride(0),
                                        _stream()), "segmm_split3_transpose")


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps=1e-12, drop_p=0.0, seed=0, site=0, amax=None, po=None):
    _dev(x, y)
    d = x.shape[-1]
    _check(lib().segmm_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd),
                                     x.numel() // d, d, eps, float(drop_p), int(seed), int(site), _ptr(amax), *_po(po), _stream()),
           "segmm_layernorm_fwd")


def layernorm_fwd_dot(x, gamma, beta, y, mean, rstd, dot_w, dot_b, dot_out, eps=1e-12, drop_p=0.0, seed=0, site=0, amax=None, po=None):
    """layernorm_fwd that also leaves dot_out[row] = y[row, :] . dot_w (+ dot_b[0])."""
    _dev(x, y)
    d = x.shape[-1]
    _check(lib().segmm_layernorm_fwd_dot(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd),
                                         x.numel() // d, d, eps, float(drop_p), int(seed), int(site), _ptr(amax), *_po(po),
                                         _ptr(dot_w), _ptr(dot_b), _ptr(dot_out), _stream()),
           "segmm_layernorm_fwd_dot")


def layernorm_bwd_parts(rows, d):
    return lib().segmm_layernorm_bwd_parts(rows, d)


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, drop_y_p=0.0, drop_y_site=0,
                  drop_b_p=0.0, drop_b_site=0, seed=0, amax=None, part_dsum=None, po=None):
    _dev(dy, x, dx)
    d = x.shape[-1]
    _check(lib().segmm_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx), _ptr(dx_drop),
                                     _ptr(part_dgamma), _ptr(part_dbeta), _ptr(part_dsum), x.numel() // d, d, float(drop_y_p),
                                     int(drop_y_site), float(drop_b_p), int(drop_b_site), int(seed), _ptr(amax), *_po(po), _stream()),
           "segmm_layernorm_bwd")


def layernorm_bwd_outer(dy_row, dy_col, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, drop_y_p=0.0, drop_y_site=0,
                        drop_b_p=0.0, drop_b_site=0, seed=0, amax=None, part_dsum=None, po=None):
    """layernorm_bwd with the incoming gradient dy[row, c] = dy_row[row] * dy_col[c] formed inside the launch."""
    _dev(dy_row, x, dx)
    d = x.shape[-1]
    _check(lib().segmm_layernorm_bwd_outer(_ptr(dy_row), _ptr(dy_col), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx), _ptr(dx_drop),
                                           _ptr(part_dgamma), _ptr(part_dbeta), _ptr(part_dsum), x.numel() // d, d, float(drop_y_p),
                                           int(drop_y_site), float(drop_b_p), int(drop_b_site), int(seed), _ptr(amax), *_po(po), _stream()),
           "segmm_layernorm_bwd_outer")


def layernorm_bwd_pos_parts(rows, period, d):
    return lib().segmm_layernorm_bwd_pos_parts(rows, period, d)


def layernorm_bwd_pos(dy, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, part_pos, period, drop_y_p=0.0, drop_y_site=0,
                      drop_b_p=0.0, drop_b_site=0, seed=0, amax=None, part_dsum=None, po=None):
    """LayerNorm backward on the per-position grid: also leaves the per-wave sums of dx in ``part_pos`` [4 * parts, d]."""
    _dev(dy, x, dx)
    d = x.shape[-1]
    _check(lib().segmm_layernorm_bwd_pos(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx), _ptr(dx_drop),
                                         _ptr(part_dgamma), _ptr(part_dbeta), _ptr(part_dsum), x.numel() // d, d, float(drop_y_p),
                                         int(drop_y_site), float(drop_b_p), int(drop_b_site), int(seed), _ptr(amax), *_po(po),
                                         _ptr(part_pos), int(period), _stream()),
           "segmm_layernorm_bwd_pos")


def colsum_pos(part, period, out):
    """out[s, :] = sum of the rows p = s (mod period) of ``part`` [P, d]."""
    _dev(part, out)
    _check(lib().segmm_colsum_pos(_ptr(part), part.shape[0], int(period), part.shape[1], _ptr(out), _stream()), "segmm_colsum_pos")


def colsum_chunks(M):
    return lib().segmm_colsum_chunks(M)


def colsum(X, ld, M, N, out, workspace, w=None, accumulate=False, x_off=0, out_off=0):
    _dev(X, out, workspace)
    _check(lib().segmm_colsum(X.data_ptr() + 4 * x_off, ld, _ptr(w), M, N, out.data_ptr() + 4 * out_off,
                              int(bool(accumulate)), _ptr(workspace), _stream()), "segmm_colsum")


class AttnPlanes(C.Structure):
    """segmm_attn_planes_t"""
    _fields_ = [("o", _p), ("ldo2", _i), ("hdr_o", _p), ("sin_o", _p),
                ("dqa", _p), ("dqb", _p), ("lddq2", _i), ("dka", _p), ("dva", _p), ("lddka2", _i),
                ("dkb", _p), ("dvb", _p), ("lddkb2", _i), ("hdr_q", _p), ("hdr_ka", _p), ("hdr_kb", _p),
                ("sin_q", _p), ("sin_ka", _p), ("sin_kb", _p), ("flags", _i),
                ("qa_in", _p), ("qb_in", _p), ("ldq2_in", _i), ("hdr_q_in", _p),
                ("ka_in", _p), ("va_in", _p), ("ldka2_in", _i), ("hdr_ka_in", _p),
                ("kb_in", _p), ("vb_in", _p), ("ldkb2_in", _i), ("hdr_kb_in", _p)]


ATTN_PLANES_ONLY, ATTN_REPAIR = 1, 2


def site_fixup(*hdrs, stats=None):
    """Between planes-only producers and their repair launches: judge each site (hdr[2] = needs repair, hdr[0] = the exact
    scale to repair with); ``stats[0]`` counts the repaired sites."""
    h = [None if x is None else x.data_ptr() for x in hdrs] + [None] * 4
    _check(lib().segmm_site_fixup(h[0], h[1], h[2], h[3], _ptr(stats), _stream()), "segmm_site_fixup")


def _fill_pin(pl, pin, Qa, Qb, Ka, Va, Kb, Vb):
    """Input-plane fields of a segmm_attn_planes_t from ``pin = dict(q=(planes, hdr, ld2), a=..., b=...)`` and the (tensor,
    column offset) pairs of the fp32 views (only the offsets are used: 2 fp16 per column)."""
    def PP(side, x):
        return None if (x is None or pin.get(side) is None) else pin[side][0].data_ptr() + 4 * x[1]
    pl.qa_in, pl.qb_in = PP("q", Qa if Qa is not None else Qb), PP("q", Qb if Qb is not None else Qa)
    pl.ldq2_in, pl.hdr_q_in = pin["q"][2], pin["q"][1].data_ptr()
    if pin.get("a") is not None and Ka is not None:
        pl.ka_in, pl.va_in, pl.ldka2_in, pl.hdr_ka_in = PP("a", Ka), PP("a", Va), pin["a"][2], pin["a"][1].data_ptr()
    if pin.get("b") is not None and Kb is not None:
        pl.kb_in, pl.vb_in, pl.ldkb2_in, pl.hdr_kb_in = PP("b", Kb), PP("b", Vb), pin["b"][2], pin["b"][1].data_ptr()


def attn_fwd(B, H, dh, Lq, La, Lb, Qa, Qb, ldq, Ka, Va, ldka, Kb, Vb, ldkb, mq, mka, mkb, O, ldo, lse,
             drop_p=0.0, seed=0, site=0, amax_o=None, po=None, pin=None):
    """Q*/K*/V* are (tensor, element_offset) pairs: column slices of the fused projection buffers.  A key block may be
    empty (La == 0 or Lb == 0, its pairs None): the CrossAtt / SelfAtt ablations attend to one block only.
    ``pin`` (optional): the INPUT planes of the same slices, ``dict(q=(planes, hdr, ld2), a=(planes, hdr, ld2), b=...)`` --
    the P32 plane tensors of the buffers the query views / the key block a views / the key block b views slice (planes
    [rows, ld2] fp16) with their site headers; the planes-in forward (csrc/attention_pl.h) then runs where the shape qualifies."""
    def P(x):
        return 0 if (x is None or x[0] is None) else x[0].data_ptr() + 4 * x[1]
    prof = ATTN_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    pl = None
    if po is not None:
        a = po.args()
        pl = AttnPlanes(o=a[0], ldo2=a[1], hdr_o=a[2], sin_o=a[3])
    if pin is not None:
        pl = pl if pl is not None else AttnPlanes()
        _fill_pin(pl, pin, Qa, Qb, Ka, Va, Kb, Vb)
    if pl is not None:
        pl = C.byref(pl)
    _check(lib().segmm_attn_fwd(B, H, dh, Lq, La, Lb, P(Qa), P(Qb), ldq, P(Ka), P(Va), ldka, P(Kb), P(Vb), ldkb,
                                _ptr(mq), _ptr(mka), _ptr(mkb), _ptr(O), ldo, _ptr(lse), float(drop_p), int(seed),
                                int(site), _ptr(amax_o), pl, _stream()), "segmm_attn_fwd")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append(("fwd", B, H, dh, Lq, La, Lb, e0, e1))


def attn_bwd(B, H, dh, Lq, La, Lb, Qa, Qb, ldq, Ka, Va, ldka, Kb, Vb, ldkb, mq, mka, mkb, lse, O, ldo, dO, lddo, Dvec,
             dQa, dQb, lddq, dKa, dVa, lddka, dKb, dVb, lddkb, drop_p=0.0, seed=0, site=0, amax_q=None, amax_ka=None,
             amax_kb=None, phase=0, planes=None, pin=None):
    """``phase``: 0 whole backward; 1 Dvec only; 2 dQ only; 3 dK/dV only (2 and 3 may run concurrently after 1).
    ``pin`` (fused backward, phase >= 4): the INPUT planes of Q / K / V like attn_fwd's; the Q / K / V pairs then only name
    column offsets and their tensors may be None (a caller whose projection GEMMs write planes only): ``(None, offset)``."""
    def P(x):
        return 0 if (x is None or x[0] is None) else x[0].data_ptr() + 4 * x[1]
    if pin is not None:
        global ATTN_PIN_CALLS
        ATTN_PIN_CALLS += 1
        planes = planes if planes is not None else AttnPlanes()
        _fill_pin(planes, pin, Qa, Qb, Ka, Va, Kb, Vb)
    prof = ATTN_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(lib().segmm_attn_bwd(B, H, dh, Lq, La, Lb, P(Qa), P(Qb), ldq, P(Ka), P(Va), ldka, P(Kb), P(Vb), ldkb,
                                _ptr(mq), _ptr(mka), _ptr(mkb), _ptr(lse), _ptr(O), ldo, _ptr(dO), lddo, _ptr(Dvec), P(dQa),
                                P(dQb), lddq, P(dKa), P(dVa), lddka, P(dKb), P(dVb), lddkb, float(drop_p), int(seed),
                                int(site), _ptr(amax_q), _ptr(amax_ka), _ptr(amax_kb), int(phase),
                                None if planes is None else C.byref(planes), _stream()), "segmm_attn_bwd")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        repair = planes is not None and (planes.flags & ATTN_REPAIR)
        kind = "bwd" if phase == 0 else "bwd4r" if repair else "bwd4" if phase >= 4 else "bwd%d" % phase
        # (phase 5 / 6 = one key block of the fused backward: its FLOPs are the block's; a repair launch does no work normally)
        prof.append((kind, B, H, dh, Lq, 0 if phase == 6 else La, 0 if phase == 5 else Lb, e0, e1))


ARGSORT_MAX = 8192


def label_stats_unpack(gathered, G, B, v_all, v2_all, norms):
    """``segmm_label_stats_unpack``: G gathered records [v | v2 | norms] -> v_all, v2_all [G * B], norms summed in rank order."""
    _check(lib().segmm_label_stats_unpack(gathered.data_ptr(), int(G), int(B), v_all.data_ptr(), v2_all.data_ptr(), norms.data_ptr(), _stream()),
           "segmm_label_stats_unpack")


def argsort_ws_words(n):
    """64-bit words of workspace ``argsort_ids`` needs for ``n`` ids (0 up to ARGSORT_MAX: one workgroup sorts in LDS)."""
    if n <= ARGSORT_MAX:
        return 0
    w = 2 * ARGSORT_MAX
    while w < n:
        w <<= 1
    return w


def argsort_ids(ids, out=None, ws=None):
    """Stable argsort of int64 ids as int32 (``torch.argsort(ids, stable=True)`` is five ATen launches): one launch up to
    ARGSORT_MAX ids; beyond, the multi-workgroup bitonic network over ``ws`` (int64 tensor of ``argsort_ws_words(n)`` words;
    allocated here when not given -- a recorded step passes a persistent one)."""
    n = ids.numel()
    if out is None:
        out = torch.empty((n,), dtype=torch.int32, device=ids.device)
    if n <= ARGSORT_MAX:
        _check(lib().segmm_argsort_ids(ids.data_ptr(), int(n), out.data_ptr(), _stream()), "segmm_argsort_ids")
        return out
    need = argsort_ws_words(n)
    if ws is None:
        ws = torch.empty((need,), dtype=torch.int64, device=ids.device)
    if ws.numel() < need or ws.dtype != torch.int64:
        raise RuntimeError("argsort_ids: %d ids need %d int64 words of workspace" % (n, need))
    _check(lib().segmm_argsort_ids_ws(ids.data_ptr(), int(n), out.data_ptr(), ws.data_ptr(), _stream()), "segmm_argsort_ids_ws")
    return out


def zero_rows(table, ids):
    """table[ids, :] = 0 (2-D float32 table, int64 ids; ids outside the table are skipped), one launch."""
    ids = ids.reshape(-1)
    _check(lib().segmm_zero_rows(table.data_ptr(), int(table.shape[1]), ids.data_ptr(), int(ids.numel()), int(table.shape[0]), _stream()),
           "segmm_zero_rows")


def loss_finish(parts, B, coef, losses, total, dlogits=None, site_scale=None, gain=None, n_sites=0, gmax=None, target=7):
    """losses[12] = column sums of parts[B, 12]; total[0] = coef . losses (one launch).  With ``dlogits``: also the step's
    max |d loss / d logits| and the loss-relative delayed scales of the backward sites (segmm_loss_finish)."""
    _check(lib().segmm_loss_finish(_ptr(parts), int(B), _ptr(coef), _ptr(losses), _ptr(total), _ptr(dlogits),
                                   0 if dlogits is None else int(dlogits.numel()), _ptr(site_scale), _ptr(gain), int(n_sites), _ptr(gmax),
                                   int(target), _stream()), "segmm_loss_finish")


def rowdot(x, ld, w, bias, out, rows, d, accumulate=False, x_off=0, w_off=0):
    _check(lib().segmm_rowdot(x.data_ptr() + 4 * x_off, ld, w.data_ptr() + 4 * w_off, _ptr(bias), _ptr(out), rows, d,
                              int(bool(accumulate)), _stream()), "segmm_rowdot")


def rowscale_bcast(g, w, dx, ld, rows, d, accumulate=False, w_off=0, dx_off=0):
    _check(lib().segmm_rowscale_bcast(_ptr(g), w.data_ptr() + 4 * w_off, dx.data_ptr() + 4 * dx_off, ld, rows, d,
                                      int(bool(accumulate)), _stream()), "segmm_rowscale_bcast")


def rowdot_pair(a, lda, b, ldb, out, rows, d, accumulate=False, a_off=0, b_off=0):
    _check(lib().segmm_rowdot_pair(a.data_ptr() + 4 * a_off, lda, b.data_ptr() + 4 * b_off, ldb, _ptr(out), rows, d,
                                   int(bool(accumulate)), _stream()), "segmm_rowdot_pair")


def rowscale_mat(g, X, ldx, out, ldo, rows, d, accumulate=False):
    _check(lib().segmm_rowscale_mat(_ptr(g), _ptr(X), ldx, _ptr(out), ldo, rows, d, int(bool(accumulate)), _stream()),
           "segmm_rowscale_mat")


def vecsum(v, n, out, accumulate=False):
    _check(lib().segmm_vecsum(_ptr(v), n, _ptr(out), int(bool(accumulate)), _stream()), "segmm_vecsum")


def embed_id_vid(item_id, table, dhalf, frame_w, frame_b, pe, out, B, S, frame_pos=None):
    _check(lib().segmm_embed_id_vid(_ptr(item_id), _ptr(table), dhalf, _ptr(frame_w), _ptr(frame_b), _ptr(pe),
                                    _ptr(frame_pos), _ptr(out), B, S, table.shape[0], _stream()), "segmm_embed_id_vid")


def embed_id_usr(user_id, table, d, pe, out, B):
    _check(lib().segmm_embed_id_usr(_ptr(user_id), _ptr(table), d, _ptr(pe), _ptr(out), B, table.shape[0], _stream()),
           "segmm_embed_id_usr")


def embed_id_bwd(dpre, tokens_per_row, ld, col0, width, order, ids, dtable, B):
    _check(lib().segmm_embed_id_bwd(_ptr(dpre), tokens_per_row, ld, col0, width, _ptr(order), _ptr(ids), _ptr(dtable),
                                    B, dtable.shape[0], _stream()), "segmm_embed_id_bwd")


def pe_grad(dpre, ld, B, S, d, dpe, accumulate=False):
    _check(lib().segmm_pe_grad(_ptr(dpre), ld, B, S, d, _ptr(dpe), int(bool(accumulate)), _stream()), "segmm_pe_grad")


def label_stats(gt, B, S, rewritten, v, v2, norms):
    _check(lib().segmm_label_stats(_ptr(gt), B, S, int(rewritten), _ptr(v), _ptr(v2), _ptr(norms), _stream()),
           "segmm_label_stats")


def loss_fwd_bwd(B, S, logits, gt, bias_w, bias_b, exposure, coef, enabled, rew_ce, rew_kl, use_mask, norms, v_all,
                 v2_all, Bg, logits_out, dlogits, parts):
    coef_a = (C.c_float * 9)(*[float(c) for c in coef])
    en_a = (C.c_int * 9)(*[int(e) for e in enabled])
    _check(lib().segmm_loss_fwd_bwd(B, S, _ptr(logits), _ptr(gt), _ptr(bias_w), _ptr(bias_b), _ptr(exposure),
                                    C.cast(coef_a, C.c_void_p), C.cast(en_a, C.c_void_p), int(rew_ce), int(rew_kl),
                                    int(use_mask), _ptr(norms), _ptr(v_all), _ptr(v2_all), Bg, _ptr(logits_out),
                                    _ptr(dlogits), _ptr(parts), _stream()), "segmm_loss_fwd_bwd")


def adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, p_off=0):
    with _kprof("adamw", 28 * int(n)):
        _adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, p_off)


def adamw_table(p, g, m, v, off, n_rows, width, ids, flags, lr, beta1, beta2, eps, weight_decay, step, phase):
    """segmm_adamw_table on the table that starts ``off`` floats into the flat buffers (phase 0: rows without a gradient, g = 0;
    phase 1: the rows listed in ``ids``)."""
    with _kprof("adamw", (24 if phase == 0 else 28) * int(n_rows if phase == 0 else ids.numel()) * int(width)):
        _check(lib().segmm_adamw_table(p.data_ptr() + 4 * off, None if g is None else g.data_ptr() + 4 * off, m.data_ptr() + 4 * off,
                                       v.data_ptr() + 4 * off, int(n_rows), int(width), ids.data_ptr(), ids.numel(), flags.data_ptr(),
                                       float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), int(phase),
                                       _stream()), "segmm_adamw_table")


def _adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, p_off=0):
    _check(lib().segmm_adamw(p.data_ptr() + 4 * p_off, g.data_ptr() + 4 * p_off, m.data_ptr() + 4 * p_off,
                             v.data_ptr() + 4 * p_off, n, lr, beta1, beta2, eps, weight_decay, step, _stream()),
           "segmm_adamw")


def bias_grad(dl, B, S, gbw, gbb):
    _check(lib().segmm_bias_grad(_ptr(dl), int(B), int(S), _ptr(gbw), _ptr(gbb), _stream()), "segmm_bias_grad")


def focal_relabel(gt):
    _check(lib().segmm_focal_relabel(_ptr(gt), gt.numel(), _stream()), "segmm_focal_relabel")


def rand_uniform(out, seed, site):
    _check(lib().segmm_rand_uniform(_ptr(out), out.numel(), int(seed), int(site), _stream()), "segmm_rand_uniform")


def rand_ids(out, lo, hi, seed, site):
    _check(lib().segmm_rand_ids(_ptr(out), out.numel(), int(lo), int(hi), int(seed), int(site), _stream()), "segmm_rand_ids")


def rand_perm_rows(out, rows, S, seed, site):
    _check(lib().segmm_rand_perm_rows(_ptr(out), int(rows), int(S), int(seed), int(site), _stream()), "segmm_rand_perm_rows")


def dropout_mult(out, n, p, seed, site):
    _check(lib().segmm_dropout_mult(_ptr(out), n, float(p), int(seed), int(site), _stream()), "segmm_dropout_mult")


# ------------------------------------------------------------------ evaluation, gather, SegRec head
def rank_leave(x, gt, perm=None, masked=False, seq_valid=None):
    """(ranks int32 [B], hist int32 [S+1]) of the leave segment; see segmm_rank_leave in include/segmm_hip.h."""
    _dev(x, gt)
    B, S = gt.shape
    if x.dtype != torch.float32 or x.stride(-1) != 1 or gt.dtype != torch.int64 or not gt.is_contiguous():
        raise RuntimeError("rank_leave: x float32 with unit inner stride, gt contiguous int64")
    if perm is not None and (perm.dtype != torch.int32 or not perm.is_contiguous() or perm.shape != gt.shape):
        raise RuntimeError("rank_leave: perm must be a contiguous int32 [B, S] tensor")
    ranks = torch.empty(B, dtype=torch.int32, device=x.device)
    hist = torch.zeros(S + 1, dtype=torch.int32, device=x.device)
    _check(lib().segmm_rank_leave(_ptr(x), x.stride(0), _ptr(gt), _ptr(perm), B, S, int(bool(masked)),
                                  S if seq_valid is None else int(seq_valid), _ptr(ranks), _ptr(hist), _stream()), "segmm_rank_leave")
    return ranks, hist


def auc_counts(score, label, seg_off):
    """int64 [n_seg, 3] = (U2, npos, nneg) per segment; label int8 (1 / 0 / other = ignored)."""
    _dev(score, label, seg_off)
    if score.dtype != torch.float32 or label.dtype != torch.int8 or seg_off.dtype != torch.int64:
        raise RuntimeError("auc_counts: score float32, label int8, seg_off int64")
    n_seg = seg_off.numel() - 1
    out = torch.zeros((max(n_seg, 0), 3), dtype=torch.int64, device=score.device)
    _check(lib().segmm_auc_counts(_ptr(score.contiguous()), _ptr(label.contiguous()), _ptr(seg_off.contiguous()), n_seg, _ptr(out),
                                  _stream()), "segmm_auc_counts")
    return out


def survival(interest, gt):
    """(surv float32 [B, S], label int8 [B, S]) for ProbAUC."""
    _dev(interest, gt)
    B, S = gt.shape
    surv = torch.empty((B, S), dtype=torch.float32, device=interest.device)
    label = torch.empty((B, S), dtype=torch.int8, device=interest.device)
    _check(lib().segmm_survival(_ptr(interest), interest.stride(0), _ptr(gt.contiguous()), _ptr(surv), _ptr(label), B, S, _stream()),
           "segmm_survival")
    return surv, label


def gather_l1(table, idx, normalize=True, out=None, mask=None, amax=None, po=None):
    """out[..., :] = (L1-normalised) table[idx[...]], mask[...] = idx in range; idx int64 of any shape."""
    _dev(table, idx)
    D = table.shape[1]
    rows = idx.numel()
    if out is None:
        out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=table.device)
    if mask is None:
        mask = torch.empty(tuple(idx.shape), dtype=torch.uint8, device=table.device)
    with _kprof("gather_l1", rows * (8 * D + 9)):
        _check(lib().segmm_gather_l1(_ptr(_f32c(table, "table")), table.shape[0], D, _ptr(idx.contiguous()), rows, int(bool(normalize)),
                                     _ptr(out), _ptr(mask), _ptr(amax), *_po(po), _stream()), "segmm_gather_l1")
    return out, mask.view(torch.bool)


def segment_weighted_sum(pred, weight=None, duration=None):
    """sum_seg pred * weight * (seg < duration) over the last axis."""
    _dev(pred)
    S = pred.shape[-1]
    rows = pred.numel() // S
    out = torch.empty(pred.shape[:-1], dtype=torch.float32, device=pred.device)
    _check(lib().segmm_segment_weighted_sum(_ptr(_f32c(pred, "pred")), _ptr(None if weight is None else _f32c(weight, "weight")),
                                            _ptr(None if duration is None else duration.contiguous()), rows, S, _ptr(out), _stream()),
           "segmm_segment_weighted_sum")
    return out


def pool_tokens(U, Lu, V, Lv, out, B, d, bins):
    """AdaptiveAvgPool1d(bins) over the tokens of cat(U[B,Lu,d], V[B,Lv,d]) (CrossMLP ablation)."""
    _dev(U, V, out)
    _check(lib().segmm_pool_tokens(_ptr(U), Lu, _ptr(V), Lv, _ptr(out), B, d, bins, _stream()), "segmm_pool_tokens")


def pool_tokens_bwd(dOut, dU, Lu, dV, Lv, B, d, bins):
    _dev(dOut, dU, dV)
    _check(lib().segmm_pool_tokens_bwd(_ptr(dOut), _ptr(dU), Lu, _ptr(dV), Lv, B, d, bins, _stream()), "segmm_pool_tokens_bwd")


def colsum3(Xs, ld, M, N, outs, ws):
    """Column sums of up to three same-shaped [M, N] matrices in one launch pair; ws: 3 * colsum_chunks(M) * N floats."""
    X = list(Xs) + [None] * (3 - len(Xs))
    O = list(outs) + [None] * (3 - len(outs))
    _check(lib().segmm_colsum3(_ptr(X[0]), _ptr(X[1]), _ptr(X[2]), ld, M, N, _ptr(O[0]), _ptr(O[1]), _ptr(O[2]), _ptr(ws), _stream()),
           "segmm_colsum3")
