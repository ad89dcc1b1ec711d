"""Host-side mirror of ``MMinterest/models/encoder.py`` (the reference's backbone plugin surface).

Same class names, constructor signatures, ``state_dict`` keys and initialisation as the reference
(SURVEY.md §8(b)); the arithmetic is NOT here -- ``SegFormerX.forward`` hands the whole backbone to
the fused HIP engine (``engine.BackboneFn``), which also skips the compute the reference wastes on
the dead last layer (encoder.py:316-319 + output_layers=[-1]).  The sub-modules are parameter
containers: their tensors are re-pointed into one flat fp32 buffer so that fused projections read
concatenated weights in place and the optimizer / gradient all-reduce see one contiguous range.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from . import engine as _engine


def _xavier_like_kn_util(module):
    """kn_util init_module (kn_util/nn_utils/init.py:50-60): xavier-uniform Linear/Embedding, LN (1,0), zero bias."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        nn.init.xavier_uniform_(module.weight.data)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


def clones(module, n):
    """kn_util clones (kn_util/nn_utils/ops.py:7-11): n deep copies, each re-initialised."""
    mods = nn.ModuleList([copy.deepcopy(module) for _ in range(n)])
    for m in mods:
        m.apply(_xavier_like_kn_util)
    return mods


class _FusedOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise NotImplementedError(
            "%s is a parameter container in segmminterest_amd: its arithmetic is fused into the HIP engine; "
            "call SegFormerX.forward / MultiScaleTemporalDetrLeaveFocal.forward" % type(self).__name__)


class MLP(_FusedOnly):
    """kn_util MLP (kn_util/nn_utils/layers/mlp.py:6-23): Linear stack, activation + dropout(0.1) between."""

    def __init__(self, dims, activation="relu", dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.activation_name = activation
        self.inner_dropout = dropout


class SegFormerXAttention(_FusedOnly):
    """Parameter container of encoder.py:12-42."""

    def __init__(self, d_model, num_head, sr_ratio=1, dropout=0.1, ablation_type="ours"):
        super().__init__()
        if sr_ratio != 1:
            raise NotImplementedError("sr_ratio > 1 is unused by the reference trainers (main...SegMM.py:94)")
        self.t2v_proj = clones(nn.Linear(d_model, d_model), 3)
        self.v2v_proj = clones(nn.Linear(d_model, d_model), 3)
        self.t2t_proj = clones(nn.Linear(d_model, d_model), 3)
        self.v2t_proj = clones(nn.Linear(d_model, d_model), 3)
        self.sr_ratio = sr_ratio
        self.d_head = d_model // num_head
        self.num_head = num_head
        self.ff_usr = nn.Linear(d_model, d_model)
        self.ff_vid = nn.Linear(d_model, d_model)
        self.ln_usr = nn.LayerNorm(d_model, 1e-12)
        self.ln_vid = nn.LayerNorm(d_model, 1e-12)
        self.ablation_type = ablation_type
        self.dropout_p = dropout


class SegFormerXEncoderLayer(_FusedOnly):
    """Parameter container of encoder.py:178-187."""

    def __init__(self, d_model, num_head, ff_dim, sr_ratio, dropout, ablation_type="ours"):
        super().__init__()
        self.cross_attn = SegFormerXAttention(d_model, num_head, sr_ratio, dropout, ablation_type)
        self.ff_usr = MLP([d_model, ff_dim, d_model], activation="gelu")
        self.ff_vid = MLP([d_model, ff_dim, d_model], activation="gelu")
        self.ln_usr = nn.LayerNorm(d_model, eps=1e-12)
        self.ln_vid = nn.LayerNorm(d_model, eps=1e-12)
        self.dropout_p = dropout


class SegFormerXEncoder(_FusedOnly):
    """Parameter container of encoder.py:254-285 (pe_lns / txt_lvl_projs / patch_merge are kept for
    checkpoint compatibility; the reference never gives them a gradient either)."""

    def __init__(self, d_model_in, d_model_lvls, num_head_lvls, sr_ratio_lvls, ff_dim_lvls, use_patch_merge, dropout,
                 ablation_type="ours"):
        super().__init__()
        assert len(d_model_lvls) == len(num_head_lvls) == len(sr_ratio_lvls) == len(ff_dim_lvls)
        self.layers = nn.ModuleList([
            SegFormerXEncoderLayer(d, h, ff, sr, dropout, ablation_type)
            for d, h, sr, ff in zip(d_model_lvls, num_head_lvls, sr_ratio_lvls, ff_dim_lvls)])
        dims = [d_model_in] + list(d_model_lvls)
        self.pe_lns = nn.ModuleList([nn.LayerNorm(d, 1e-12) for d in d_model_lvls])
        self.txt_lvl_projs = nn.ModuleList([
            nn.Sequential(nn.Linear(dims[i - 1], dims[i]), nn.LayerNorm(dims[i], eps=1e-12)) for i in range(1, len(dims))])
        self.use_patch_merge = use_patch_merge
        self.patch_merge = nn.ModuleList([
            nn.Conv1d(dims[i - 1], dims[i], kernel_size=3, stride=2, padding=1) for i in range(1, len(dims))])


class MLP_Block(_FusedOnly):
    """Parameter container of the ablation variants' MLP (encoder.py:210-252): same constructor and ``mlp.{i}`` state_dict
    keys; ``SegFormerX`` runs it inside the HIP engine (``engine.BackboneRun._mlp_fwd``)."""

    def __init__(self, input_dim, hidden_units=[], hidden_activations="ReLU", output_dim=None, output_activation=None,
                 dropout_rates=0.0, batch_norm=False, layer_norm=False, norm_before_activation=True, use_bias=True):
        super().__init__()
        if batch_norm or layer_norm or output_activation is not None or output_dim is None:
            raise NotImplementedError("MLP_Block: only the form SegFormerX builds (Linear/ReLU/Dropout + output Linear, "
                                      "encoder.py:392-400) runs in the HIP engine")
        hidden_units = list(hidden_units)
        if not isinstance(dropout_rates, list):
            dropout_rates = [dropout_rates] * len(hidden_units)
        if not isinstance(hidden_activations, list):
            hidden_activations = [hidden_activations] * len(hidden_units)
        if any(a != "ReLU" for a in hidden_activations):
            raise NotImplementedError("MLP_Block: hidden activation must be ReLU")
        acts = [getattr(nn, a)() for a in hidden_activations]
        dims = [input_dim] + hidden_units
        mods = []
        for i in range(len(dims) - 1):
            mods.append(nn.Linear(dims[i], dims[i + 1], bias=use_bias))
            norm = nn.BatchNorm1d(dims[i + 1]) if batch_norm else (nn.LayerNorm(dims[i + 1]) if layer_norm else None)
            if norm is not None and norm_before_activation:
                mods.append(norm)
            mods.append(acts[i])
            if norm is not None and not norm_before_activation:
                mods.append(norm)
            if dropout_rates[i] > 0:
                mods.append(nn.Dropout(dropout_rates[i]))
        if output_dim is not None:
            mods.append(nn.Linear(dims[-1], output_dim, bias=use_bias))
        if output_activation is not None:
            mods.append(getattr(nn, output_activation)())
        self.mlp = nn.Sequential(*mods)


# --ablation_type choices of the trainers (main...SegMM.py:529).  Model side: 'CrossAtt' / 'SelfAtt' restrict the key
# blocks (encoder.py:108-135), 'noPos' shuffles the id-mode segment positions (:428-429), the *MLP / w/oAtt variants replace
# the encoder (:392-400,503-511); 'noUser' is applied by the trainer (random user inputs, main...SegMM.py:275-280).
ABLATION_TYPES = ("ours", "CrossAtt", "SelfAtt", "noPos", "noUser", "SelfMLP", "CrossMLP", "noUser_SelfAtt", "w/oAtt")


class SegFormerX(nn.Module):
    """Drop-in for encoder.py:327-520.  ``forward(usr_feat, usr_mask, vid_feat, vid_mask)`` returns
    ``([vid_state[B,S,d]], usr_embedding[B,Lt,d])`` exactly like the reference with output_layers=[-1]."""

    def __init__(self, d_model_in=128, d_model_lvls=[128, 256, 512, 1024], num_head_lvls=[2, 4, 8, 16],
                 ff_dim_lvls=[256, 512, 1024, 2048], sr_ratio_lvls=[8, 4, 2, 1], input_vid_dim=768, input_usr_dim=768,
                 max_vid_len=256, max_usr_len=20, dropout=0.1, pe_kernel_size=3,
                 use_patch_merge=[True, False, True, False], output_layers=None, model_cfg=None, user_id_max=-1,
                 video_id_max=-1, use_pe=1):
        super().__init__()
        d = d_model_in
        if any(x != d for x in d_model_lvls) or any(x != d for x in ff_dim_lvls):
            raise NotImplementedError("the reference trainers use d_model_lvls = ff_dim_lvls = [d_model]*N (main...SegMM.py:88-96)")
        if any(sr != 1 for sr in sr_ratio_lvls) or any(use_patch_merge):
            raise NotImplementedError("sr_ratio > 1 / patch merging are disabled in the reference trainers (main...SegMM.py:94)")
        if len(set(num_head_lvls)) != 1:
            raise NotImplementedError("one head count per backbone")
        self.id_vid = video_id_max != -1
        self.id_usr = user_id_max != -1
        if self.id_vid:
            self.vid_proj = nn.Embedding(video_id_max + 1, d // 2)
            self.frameid_proj = nn.Linear(1, d // 2)
        else:
            self.vid_proj = nn.Linear(input_vid_dim, d)
        self.usr_proj = nn.Embedding(user_id_max + 1, d) if self.id_usr else nn.Linear(input_usr_dim, d)
        self.debug = getattr(model_cfg, "debug", 0)
        self.num_layers_enc = getattr(model_cfg, "num_layers_enc", len(d_model_lvls))
        self.use_pe = use_pe
        self.vid_pe = nn.Embedding(max_vid_len, d)
        self.usr_pe = nn.Embedding(max_usr_len, d)
        self.vid_ln = nn.LayerNorm(d, eps=1e-12)
        self.usr_ln = nn.LayerNorm(d, eps=1e-12)
        self.dropout_p = dropout
        self.ablation_type = getattr(model_cfg, "ablation_type", "ours") if model_cfg is not None else "ours"
        if self.ablation_type not in ABLATION_TYPES:
            raise ValueError("ablation_type=%r not in %s (main...SegMM.py:529)" % (self.ablation_type, ABLATION_TYPES))
        # encoder.py:392-409: the MLP ablations build encoder_mlp INSTEAD of the encoder (w/oAtt builds it and never calls it)
        lvls = list(d_model_lvls)
        if self.ablation_type == "CrossMLP":
            self.encoder_mlp = MLP_Block(input_dim=lvls[0], output_dim=lvls[0], hidden_units=lvls[2:-2],
                                         hidden_activations="ReLU", dropout_rates=dropout, batch_norm=0)
            self.encoder_pooling = nn.AdaptiveAvgPool1d(40)
        elif self.ablation_type in ("SelfMLP", "w/oAtt"):
            self.encoder_mlp = MLP_Block(input_dim=lvls[0], output_dim=lvls[0], hidden_units=lvls[1:-1],
                                         hidden_activations="ReLU", dropout_rates=dropout, batch_norm=0)
        else:
            self.encoder = SegFormerXEncoder(d, lvls, list(num_head_lvls), list(sr_ratio_lvls),
                                             list(ff_dim_lvls), list(use_patch_merge), dropout, self.ablation_type)
        self.output_layers = list(range(len(sr_ratio_lvls))) if output_layers is None else list(output_layers)
        if self.output_layers != [-1]:
            raise NotImplementedError("output_layers must be [-1] as in the reference trainers (main...SegMM.py:95)")
        self.d_model = d
        self.nhead = num_head_lvls[0]
        self.n_layers = len(d_model_lvls)
        self.max_vid_len = max_vid_len
        self.max_usr_len = max_usr_len
        self.apply(self.init_weight)
        self._store = None          # set by the owning model, or lazily for stand-alone use
        self._prefix = ""

    def init_weight(self, module):
        """N(0, 0.02) weights, zero biases, LayerNorm (1, 0) -- encoder.py:414-423."""
        if isinstance(module, (nn.Linear, nn.Embedding, nn.Conv1d)):
            module.weight.data.normal_(mean=0.0, std=0.02)
        if isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def forward(self, usr_feat, usr_mask, vid_feat, vid_mask):
        if self._store is None:
            self._store = _engine.ParamStore(self, standalone_backbone=True)
        vid, usr = _engine.backbone_apply(self._store, self, self._prefix, usr_feat, usr_mask, vid_feat, vid_mask,
                                          self.training)
        return [vid], usr


class SegFormerXFPN(nn.Module):
    """Unused by every trainer of the reference (encoder.py:523-560); exported for import compatibility."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("SegFormerXFPN is not on the segment-interest path")
