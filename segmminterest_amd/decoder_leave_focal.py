"""Host-side mirror of ``MMinterest/models/decoder_leave_focal.py``: the top-level model plugin.

``MultiScaleTemporalDetrLeaveFocal(backbone1, backbone2, head, frame_pooler, model_cfg)`` keeps the
reference's constructor, ``forward`` signature, returned dict keys and ``state_dict`` names
(decoder_leave_focal.py:425-658).  Routing of id / image inputs to one or two backbones, the
interest head (Linear, sum / concat / two-Linear fusion, or the bilinear ``InteractionAggregation``)
and every loss run in the HIP library; autograd sees two kinds of nodes only (``engine.BackboneFn``
and ``HeadLossFn``).
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn

from . import engine as E
from . import hipabi as H
from .encoder import _xavier_like_kn_util, _FusedOnly

LOSS_ORDER = ("interestBPR", "focal", "surviveCE", "interestCE", "interestKL", "huber", "hazard", "mse", "mse2")
_LIDX = {n: i for i, n in enumerate(LOSS_ORDER)}


class InteractionAggregation(_FusedOnly):
    """Parameter container of decoder_leave_focal.py:392-409 (bilinear fusion of two backbones)."""

    def __init__(self, x_dim, y_dim, output_dim=1, num_heads=1):
        super().__init__()
        if output_dim != 1:
            raise NotImplementedError("output_dim must be 1 (decoder_leave_focal.py:471)")
        self.num_heads = num_heads
        self.output_dim = output_dim
        self.w_x = nn.Linear(x_dim, output_dim)
        self.w_y = nn.Linear(y_dim, output_dim)
        self.w_x.apply(_xavier_like_kn_util)
        self.w_y.apply(_xavier_like_kn_util)
        if num_heads > 0:
            assert x_dim % num_heads == 0 and y_dim % num_heads == 0, "Input dim must be divisible by num_heads!"
            self.head_x_dim = x_dim // num_heads
            self.head_y_dim = y_dim // num_heads
            self.w_xy = nn.Parameter(torch.empty(num_heads * self.head_x_dim * self.head_y_dim, output_dim))
            nn.init.xavier_normal_(self.w_xy)


class LossSpec:
    """Static description of ``compute_loss`` for one model_cfg (decoder_leave_focal.py:490-572).

    One difference in error behaviour: a batch in which EVERY row is fully watched has no BPR negative -- the reference raises there
    (``neg_pred.max()`` of an empty tensor, decoder_leave_focal.py:213); the device-side loss returns ``interestBPR`` = 0 with zero
    gradients (it has no host synchronisation to raise from)."""

    def __init__(self, model_cfg, S_hint=40):
        lst = list(model_cfg.loss_type_list)
        unknown = [x for x in lst if x not in LOSS_ORDER[:7]]
        if unknown:
            raise ValueError("unknown loss types %s" % unknown)
        self.loss_list = lst
        lw = model_cfg.loss_weight
        self.coef = [0.0] * 9
        self.enabled = [0] * 9
        for name in lst:
            k = _LIDX[name]
            self.enabled[k] = 1
            self.coef[k] = float(lw["mse"] if name == "huber" else lw[name])      # :561-566
        self.enabled[_LIDX["mse"]] = self.enabled[_LIDX["mse2"]] = 1
        fpos = lst.index("focal") if "focal" in lst else None
        self.has_focal = fpos is not None
        self.rew_ce = int(fpos is not None and "interestCE" in lst and fpos < lst.index("interestCE"))
        self.rew_kl = int(fpos is not None and "interestKL" in lst and fpos < lst.index("interestKL"))
        self.use_mask = int(getattr(model_cfg, "mask_loss", 0))
        self.exposure = [float(x) for x in model_cfg.exposure_prob]


class HeadLossFn(torch.autograd.Function):
    """Interest head + compute_loss.  Outputs: total loss (differentiable), per-loss scalars [12]
    (detached), logits incl. position bias [B,S] (detached)."""

    @staticmethod
    def forward(ctx, model, want_loss, v1, v2, gt, *params):
        st = model._store
        H.mark(H.PHASE_HEAD_LOSS_FWD)
        B, S, d = v1.shape
        M = B * S
        v1c = v1.contiguous()
        v2c = v2.contiguous() if v2 is not None else None
        pre = model.__dict__.pop("_raw_pre", None)          # (model.forward: logits buffer handed to the backbone's last LayerNorm)
        if pre is not None and v2 is None and pre.numel() == M and st.__dict__.get("_head_dot_done") == pre.data_ptr():
            raw, T = pre, None          # the Linear(d, 1) head ran inside that launch (segmm_layernorm_fwd_dot)
        else:
            raw = torch.empty(M, dtype=torch.float32, device=v1.device)
            T = model._head_fwd(v1c, v2c, raw, M, d)
        st._head_dot_done = None
        ctx.model, ctx.dims, ctx.T = model, (B, S, d), T
        ctx.has_v2 = v2 is not None
        ctx.set_materialize_grads(False)          # no zero tensors for the two non-differentiable outputs (two fill launches)
        ctx.save_for_backward(v1c, v2c if v2c is not None else v1c)
        if not want_loss:
            logits = raw.view(B, S)
            if model.bias_weight is not None:
                pos = torch.arange(1, S + 1, device=raw.device, dtype=torch.float32)
                logits = logits + pos * st.p("bias_weight")[:, :S] + st.p("bias_bias")[:, :S]
            ctx.mark_non_differentiable(logits)
            return logits.new_zeros(()), logits.new_zeros(12), logits
        spec = model._loss_spec
        gt = E._as(gt, torch.int64, "the labels")
        stats = model._label_stats(gt, B, S) if model._stats is None else model._stats
        if callable(stats):          # data-parallel: the all-gather was issued before the backbones; wait for it here
            stats = stats()
        v_all, v2_all, norms = stats
        model._stats = None
        expo = model._exposure_tensor(S, raw.device)
        logits = torch.empty(B, S, device=raw.device)
        dlogits = torch.empty(B, S, device=raw.device)
        parts = torch.empty(B, 12, device=raw.device)
        bw = st.p("bias_weight") if model.bias_weight is not None else None
        bb = st.p("bias_bias") if model.bias_weight is not None else None
        H.loss_fwd_bwd(B, S, raw, gt, bw, bb, expo, spec.coef, spec.enabled, spec.rew_ce, spec.rew_kl, spec.use_mask, norms,
                       v_all, v2_all, v_all.numel(), logits, dlogits, parts)
        losses = torch.empty(12, device=raw.device)
        total = torch.empty((), device=raw.device)
        # sum_b parts and sum_i coef_i * loss_i in one launch -- which, on the plane engine in training, also takes max|dlogits| and
        # sets the delayed scales of the backward tensors relative to it (engine.ParamStore.update_scales(backward=True))
        rel = st.engine_p and st.loss_relative and ((model.training and st.scaling != "exact") or st.scaling == "always")
        st._gmax_fresh = bool(rel)
        H.loss_finish(parts, B, model._coef_tensor(raw.device), losses, total, dlogits=dlogits if rel else None,
                      site_scale=st.scales() if rel else None, gain=st.gains() if rel else None, n_sites=st.MAX_SITES if rel else 0,
                      gmax=st.gmax() if rel else None, target=st.scale_target)
        ctx.dlogits = dlogits
        ctx.mark_non_differentiable(losses, logits)
        return total, losses, logits

    @staticmethod
    def backward(ctx, g_total, g_losses, g_logits):
        model = ctx.model
        st = model._store
        B, S, d = ctx.dims
        M = B * S
        v1, v2 = ctx.saved_tensors
        if not ctx.has_v2:
            v2 = None
        names = model._head_param_names()
        if g_total is None:          # the loss took no part in the differentiated scalar
            return (None,) * (5 + len(names))
        H.mark(H.PHASE_HEAD_LOSS_BWD)
        st.__dict__.pop("_lazy_dy", None)          # (a marker no backbone backward consumed -- frozen backbone -- must not meet a later tensor at the same address)
        gbuf = E._pick_gbuf(st, names)
        unit = getattr(model, "_unit_grad", None)
        # the trainer's constant-one seed (Trainer.train_step) also selects the direct gradient delivery for the whole backward
        # (engine.grads_out); every other backward goes through autograd like a plain nn.Module
        st.direct_grads = unit is not None and g_total.data_ptr() == unit.data_ptr()
        if st.direct_grads:
            dl = ctx.dlogits.view(M)          # d loss / d loss == 1
        else:
            dl = (ctx.dlogits * g_total).view(M)
        dv1 = torch.empty(M, d, device=v1.device)
        dv2 = torch.empty(M, d, device=v1.device) if v2 is not None else None
        model._head_bwd(v1, v2, dl, dv1, dv2, ctx.T, M, d, B, S, gbuf)
        if gbuf is None and st.bucket_hook is not None:
            st.bucket_hook("head", after_side=st.head_side)
        ctx.dlogits = ctx.T = None
        if not st.direct_grads or gbuf is not None:
            # the head's weight / bias gradients were written on the side stream (head_side): autograd consumes the views it is
            # handed at once on THIS stream (hooks, AccumulateGrad's += on a second backward, clones) and a fresh ``gbuf`` may be
            # recycled by the caching allocator while the side stream still writes it -- join first.  The trainer's direct
            # delivery (views of the persistent flat buffer, ordered by the DP hook / the end-of-backward join) stays unjoined.
            E.join_side(st)
        return (None, None, dv1.view(B, S, d), dv2.view(B, S, d) if dv2 is not None else None, None) + E.grads_out(st, names, gbuf)


class MultiScaleTemporalDetrLeaveFocal(nn.Module):
    """Drop-in for decoder_leave_focal.py:425-658."""

    def __init__(self, backbone1, backbone2, head, frame_pooler, model_cfg) -> None:
        super().__init__()
        if head is not None:
            raise NotImplementedError("head != None (stage_mlps) is never used by the reference trainers")
        self.backbone1 = backbone1
        self.backbone2 = backbone2
        self.model_cfg = model_cfg
        self.head = head
        self.frame_pooler = frame_pooler
        self.debug = getattr(model_cfg, "debug", 0)
        self.input_type = model_cfg.input_type
        d = model_cfg.d_model
        self.d_model = d
        self.bias_weight = None
        self.bias_bias = None
        self.exposure_prob = model_cfg.exposure_prob
        S_bias = len(self.exposure_prob)
        if model_cfg.learnable_bias:
            self.bias_weight = nn.Parameter(torch.ones(1, S_bias))
            self.bias_bias = nn.Parameter(torch.ones(1, S_bias))
        self.fusion_heads = getattr(model_cfg, "fusion_heads", 2)
        if backbone2 is None:
            self.stage_mlp1 = nn.Linear(d, 1)
            self.stage_mlp1.apply(_xavier_like_kn_util)
        else:
            fh = self.fusion_heads
            if fh in (-2, -3):
                self.stage_mlp1 = nn.Linear(d, 1)
                self.stage_mlp1.apply(_xavier_like_kn_util)
            elif fh == -1:
                self.stage_mlp1 = nn.Linear(2 * d, 1)
                self.stage_mlp1.apply(_xavier_like_kn_util)
            elif fh == 0:
                self.stage_mlp1 = nn.Linear(d, 1)
                self.stage_mlp1.apply(_xavier_like_kn_util)
                self.stage_mlp2 = nn.Linear(d, 1)
                self.stage_mlp2.apply(_xavier_like_kn_util)
            else:
                self.fusion_module = InteractionAggregation(d, d, output_dim=1, num_heads=fh)
        if not isinstance(frame_pooler, nn.Identity):
            raise NotImplementedError("frame_pooler must be nn.Identity (main_for_seq_leave_earlystop_SegMM.py:62)")
        self._store = E.ParamStore(self)
        self._loss_spec = LossSpec(model_cfg) if getattr(model_cfg, "loss_type_list", None) else None
        self._dp_hook = None
        self._dp_stats_bufs = None
        self._stats = None
        self._consts = {}
        backbone1._store, backbone1._prefix = self._store, "backbone1."
        if backbone2 is not None:
            backbone2._store, backbone2._prefix = self._store, "backbone2."

    # ------------------------------------------------------------------ parameter layout (engine.ParamStore)
    def _head_param_names(self):
        names = []
        if self.backbone2 is None or self.fusion_heads in (-1, -2, -3):
            names += ["stage_mlp1.weight", "stage_mlp1.bias"]
        elif self.fusion_heads == 0:
            names += ["stage_mlp1.weight", "stage_mlp1.bias", "stage_mlp2.weight", "stage_mlp2.bias"]
        else:
            names += ["fusion_module.w_x.weight", "fusion_module.w_x.bias", "fusion_module.w_y.weight",
                      "fusion_module.w_y.bias", "fusion_module.w_xy"]
        if self.bias_weight is not None:
            names += ["bias_weight", "bias_bias"]
        return names

    def _param_buckets(self):
        buckets = [("head", [[n] for n in self._head_param_names()])]
        if self.backbone2 is not None:
            buckets += E.backbone_layout("backbone2.", self.backbone2)
        buckets += E.backbone_layout("backbone1.", self.backbone1)
        return buckets

    def _label_stats(self, gt, B, S):
        """(view lengths of all rows, second length vector, 3 normalisers) -- global under data parallelism."""
        bufs = getattr(self, "_dp_stats_bufs", None)
        if self._dp_hook is not None and bufs is not None and gt.is_cuda:
            v, v2s, norms = bufs(B, gt.device)          # data parallel: straight into the record the all-gather sends
        else:
            v = torch.empty(B, device=gt.device)
            v2s = torch.empty(B, device=gt.device)
            norms = torch.empty(3, device=gt.device)
        H.label_stats(gt, B, S, int(self._loss_spec.has_focal), v, v2s, norms)
        if self._dp_hook is not None:
            return self._dp_hook(v, v2s, norms)
        return v, v2s, norms

    def _exposure_tensor(self, S, dev):
        k = ("expo", S, str(dev))
        t = self._consts.get(k)
        if t is None:
            e = list(self._loss_spec.exposure)[:S]
            if len(e) < S:
                raise RuntimeError("exposure_prob has %d entries, S=%d" % (len(e), S))
            t = self._consts[k] = torch.tensor(e, dtype=torch.float32, device=dev)
        return t

    def unit_grad(self, dev):
        """The constant 1.0 the trainer seeds ``backward`` with; never written after creation."""
        t = getattr(self, "_unit_grad", None)
        if t is None or t.device != torch.device(dev):
            t = self._unit_grad = torch.ones((), dtype=torch.float32, device=dev)
        return t

    def _coef_tensor(self, dev):
        k = ("coef", str(dev))
        t = self._consts.get(k)
        if t is None:
            c = list(self._loss_spec.coef)[:9]
            t = self._consts[k] = torch.tensor(c + [0.0] * (12 - len(c)), dtype=torch.float32, device=dev)      # padded to the 12 loss slots
        return t

    # ------------------------------------------------------------------ head kernels
    def _head_fwd(self, v1, v2, raw, M, d):
        st = self._store
        fh = self.fusion_heads
        if v2 is None:
            H.rowdot(v1, d, st.p("stage_mlp1.weight"), st.p("stage_mlp1.bias"), raw, M, d)
            return None
        if fh in (-2, -3):          # Linear(v1 + v2) = v1.w + v2.w + b
            H.rowdot(v1, d, st.p("stage_mlp1.weight"), st.p("stage_mlp1.bias"), raw, M, d)
            H.rowdot(v2, d, st.p("stage_mlp1.weight"), None, raw, M, d, accumulate=True)
        elif fh == -1:              # Linear(cat(v1, v2))
            H.rowdot(v1, d, st.p("stage_mlp1.weight"), st.p("stage_mlp1.bias"), raw, M, d)
            H.rowdot(v2, d, st.p("stage_mlp1.weight"), None, raw, M, d, accumulate=True, w_off=d)
        elif fh == 0:
            H.rowdot(v1, d, st.p("stage_mlp1.weight"), st.p("stage_mlp1.bias"), raw, M, d)
            H.rowdot(v2, d, st.p("stage_mlp2.weight"), st.p("stage_mlp2.bias"), raw, M, d, accumulate=True)
        else:                       # InteractionAggregation: w_x.x + w_y.y + sum_h x_h^T W_h y_h
            hx = d // fh
            H.rowdot(v1, d, st.p("fusion_module.w_x.weight"), st.p("fusion_module.w_x.bias"), raw, M, d)
            H.rowdot(v2, d, st.p("fusion_module.w_y.weight"), st.p("fusion_module.w_y.bias"), raw, M, d, accumulate=True)
            T = torch.empty(M, d, device=v1.device)
            wxy = st.p("fusion_module.w_xy")
            for h in range(fh):
                H.gemm(H.LAYOUT_NN, M, hx, hx, v1, d, wxy, hx, T, d, a_off=h * hx, b_off=h * hx * hx, c_off=h * hx)
            H.rowdot_pair(T, d, v2, d, raw, M, d, accumulate=True)
            return T
        return None

    def _head_bwd(self, v1, v2, dl, dv1, dv2, T, M, d, B, S, gbuf):
        st = self._store
        fh = self.fusion_heads

        def lin_bwd(wname, bname, x, dx, w_off=0, acc_w=False, lazy=False):
            if lazy:          # dx stays unwritten: the backbone's first LayerNorm backward forms dl[row] * w[c] itself (engine.BackboneRun.backward)
                st._lazy_dy = (dx.data_ptr(), dl, st.p(wname).view(-1)[w_off:w_off + d])
            else:
                H.rowscale_bcast(dl, st.p(wname), dx, d, M, d, w_off=w_off)
            # the head's own weight / bias gradients feed nothing in the backward: on the side stream, off the chain
            # loss -> d(features) -> LayerNorm backward -> first input-gradient GEMM that the main stream is waiting on
            with (E.side_work(st) if st.head_side else contextlib.nullcontext()):
                if st.head_side and st.overlap:          # x and dl are autograd-owned tensors: the allocator must not hand their
                    side = st.side_stream()               # memory out again before the side stream has read them
                    x.record_stream(side)
                    dl.record_stream(side)
                E._colsum(st, x, d, M, d, st.g(wname, gbuf).view(-1)[w_off:w_off + d], w=dl, accumulate=acc_w)
                if bname is not None:
                    H.vecsum(dl, M, st.g(bname, gbuf))

        if v2 is None:
            # (only on the trainer's own step: there dv1 goes straight from this Function to BackboneFn.backward, no hook, no
            # accumulation, no other consumer can look at it in between)
            lin_bwd("stage_mlp1.weight", "stage_mlp1.bias", v1, dv1, lazy=bool(st.lazy_head_grad and st.direct_grads and gbuf is None))
        elif fh in (-2, -3):
            lin_bwd("stage_mlp1.weight", "stage_mlp1.bias", v1, dv1)
            lin_bwd("stage_mlp1.weight", None, v2, dv2, acc_w=True)
        elif fh == -1:
            lin_bwd("stage_mlp1.weight", "stage_mlp1.bias", v1, dv1)
            lin_bwd("stage_mlp1.weight", None, v2, dv2, w_off=d)
        elif fh == 0:
            lin_bwd("stage_mlp1.weight", "stage_mlp1.bias", v1, dv1)
            lin_bwd("stage_mlp2.weight", "stage_mlp2.bias", v2, dv2)
        else:
            hx = d // fh
            lin_bwd("fusion_module.w_x.weight", "fusion_module.w_x.bias", v1, dv1)
            lin_bwd("fusion_module.w_y.weight", "fusion_module.w_y.bias", v2, dv2)
            H.rowscale_mat(dl, T, d, dv2, d, M, d, accumulate=True)          # dy += dl * (x_h W_h)
            dT = st.buf("fusion_dT", (M, d))
            H.rowscale_mat(dl, v2, d, dT, d, M, d)                           # dT = dl * y
            wxy = st.p("fusion_module.w_xy")
            gw = st.g("fusion_module.w_xy", gbuf)
            for h in range(fh):
                # dx_h += dT_h . W_h^T ; dW_h = x_h^T . dT_h
                H.gemm(H.LAYOUT_NT, M, hx, hx, dT, d, wxy, hx, dv1, d, accumulate=True, a_off=h * hx, b_off=h * hx * hx, c_off=h * hx)
                splits = E._splits_for(hx, hx, M)
                ws = st.buf("splitk_ws", (max(splits, 1) * hx * hx,)) if splits > 1 else None
                H.gemm(H.LAYOUT_TN, hx, hx, M, v1, d, dT, d, gw, hx, splits=splits, workspace=ws, a_off=h * hx, b_off=h * hx,
                       c_off=h * hx * hx)
        if self.bias_weight is not None:
            Sb = self.bias_weight.shape[1]
            gbb = st.g("bias_bias", gbuf)
            gbw = st.g("bias_weight", gbuf)
            if Sb != S:
                H.fill_zero(gbb)
                H.fill_zero(gbw)
            H.bias_grad(dl, B, S, gbw, gbb)          # d bias_bias[s] = sum_b dl[b, s], d bias_weight[s] = (s + 1) d bias_bias[s] (:497-504)

    # ------------------------------------------------------------------ forward (decoder_leave_focal.py:574-658)
    def forward(self, usr_image, usr_id, usr_mask, vid_image, vid_id, vid_mask, gt=None, mode="train", **kwargs):
        if mode not in ("train", "test", "inference"):
            raise ValueError("mode must be train/test/inference")
        st = self._store
        st.ensure()
        training = self.training
        seed = E.next_seed(st) if training else 0
        it = self.input_type

        def pick(kind, image, ident, which):
            if kind == "both":
                return image if which == 1 else ident
            return image if kind == "image" else ident

        def run(bb, prefix, idx, which):
            names = [n for n in st.live_names if n.startswith(prefix)]
            params = [st._params[n] for n in names]
            u = pick(it["user"], usr_image, usr_id, which)
            v = pick(it["photo"], vid_image, vid_id, which)
            return E.BackboneFn.apply(st, bb, prefix, idx, u, usr_mask, v, vid_mask, training, seed + idx, *params)[0]

        self._stats = None
        pre = self.__dict__.pop("_stats_pre", None)          # Trainer._features: the statistics of THIS batch, launched early on the auxiliary stream
        if mode in ("train", "test"):
            # label statistics depend on gt only: computed (and, data-parallel, all-gathered) BEFORE the backbones, so that the
            # collective and its rank skew hide under the forward instead of stalling every rank between forward and loss
            if self._loss_spec is None:
                self._loss_spec = LossSpec(self.model_cfg)
            gtc = E._as(gt, torch.int64, "the labels")
            self._stats = pre if pre is not None else self._label_stats(gtc, gtc.shape[0], gtc.shape[1])
        self.__dict__.pop("_raw_pre", None)
        st.__dict__.pop("_head_dot", None)
        if (st.head_dot and self.backbone2 is None and vid_mask is not None and getattr(self.backbone1, "ablation_type", "ours") not in E.MLP_VARIANTS
                and self.backbone1.n_layers >= 2):
            # single backbone, Linear(d, 1) head: its logits come out of the backbone's last LayerNorm launch (engine._side_post)
            self._raw_pre = torch.empty(vid_mask.shape[0] * vid_mask.shape[1], dtype=torch.float32, device=st.flat.device)
            st._head_dot = (st.p("stage_mlp1.weight"), st.p("stage_mlp1.bias"), self._raw_pre)
        v1 = run(self.backbone1, "backbone1.", 0, 1)
        v2 = run(self.backbone2, "backbone2.", 1, 2) if self.backbone2 is not None else None
        hp = [st._params[n] for n in self._head_param_names()]
        if mode in ("train", "test"):
            if self._loss_spec is None:
                self._loss_spec = LossSpec(self.model_cfg)
            total, losses, logits = HeadLossFn.apply(self, True, v1, v2, gt, *hp)
            out = {}
            for name in self._loss_spec.loss_list:
                out[name] = losses[_LIDX[name]]
            out["mse"] = losses[_LIDX["mse"]]
            out["mse2"] = losses[_LIDX["mse2"]]
            out["loss"] = total
            out["logits"] = logits
            if self._loss_spec.has_focal:       # the reference rewrites gt in place (:534-535)
                if gt.is_cuda and gt.dtype == torch.int64 and gt.is_contiguous():
                    H.focal_relabel(gt)
                else:
                    gt[gt > 0] = 1
                    gt[gt == -1] = 0
            out["gt"] = gt
            return out
        _, _, logits = HeadLossFn.apply(self, False, v1, v2, gt, *hp)
        return dict(logits=logits, gt=gt)
