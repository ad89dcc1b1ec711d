"""Import the reference's model / loss / metric modules on CPU (TEST INFRASTRUCTURE ONLY).

This file is part of the parity oracle tooling.  It is used only in the build
container (where /root/reference exists) by ``oracle/gen_golden.py`` to produce
the golden vectors committed under ``tests/golden/``.  Nothing in the product
package (``segmminterest_amd``), ``bench.py`` or the ``-m gpu`` tests imports it,
and it never travels to the GPU box in a usable form (the reference tree is
absent there).

Recipe = SURVEY.md Appendix A: register empty package shells so the broken
``__init__`` files of the reference are skipped, and stub the modules the
reference imports but does not ship (``model.ms_temporal_detr.ms_pooler``,
``misc``, ``models.loss``, ``torchvision.ops``).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("SEGMM_REFERENCE", "/root/reference")
REF = os.path.join(REF_ROOT, "MMinterest")
KN = os.path.join(REF, "models", "kn_util")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models"))


def _ns(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns (encoder_module, decoder_module, evaluation_module) of the reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    _ns("kn_util", KN)
    nn_utils = _ns("kn_util.nn_utils", os.path.join(KN, "nn_utils"))
    basic = _ns("kn_util.basic", os.path.join(KN, "basic"))
    nn_utils.clones = importlib.import_module("kn_util.nn_utils.ops").clones
    for sub in ("init", "math", "layers"):
        importlib.import_module("kn_util.nn_utils." + sub)
    basic.eval_env = importlib.import_module("kn_util.basic.ops").eval_env
    _ns("model")
    _ns("model.ms_temporal_detr")
    _ns("model.ms_temporal_detr.ms_pooler", MultiScaleRoIAlign1D=None)
    _ns("misc", cw2se=None, calc_iou=None)
    if "torchvision" not in sys.modules:
        _ns("torchvision")
        _ns("torchvision.ops", sigmoid_focal_loss=None)
    _ns("models", os.path.join(REF, "models"))
    _ns("models.loss", l1_loss=None, iou_loss=None)
    enc = importlib.import_module("models.encoder")
    dec = importlib.import_module("models.decoder_leave_focal")
    ev = importlib.import_module("models.my_evaluation")
    import torch
    # learnable_bias=1 hits hard-coded .cuda() calls (decoder_leave_focal.py:498,651)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _loaded = (enc, dec, ev)
    return _loaded
