"""segmminterest_amd -- MI355X-native drop-in for the ``MMinterest/models`` plugin surface.

Mirrors ``MMinterest/models/__init__.py:1-5``: the trainers do
``from model import MultiScaleTemporalDetrLeaveFocal, SegFormerX, QueryBasedDecoder, main_eval_batch,
TOP_K_leave, TOP_K_leave_mask`` (main_for_seq_leave_earlystop_SegMM.py:5).
"""
import os as _os

# Data-parallel ranks use more streams (main, side, RCCL's) than HIP's default 4 hardware queues: sharing one lets an all-reduce
# that waits for the side stream stall the main stream behind it (bench.py).  Only effective if this import
# happens before the HIP runtime initialises (import this package, or set the variable, before the first CUDA call).
if _os.environ.get("SEGMM_NO_ENV_DEFAULTS", "0") != "1":          # (opt out: the importing application owns the process environment)
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .encoder import (MLP, MLP_Block, SegFormerX, SegFormerXAttention, SegFormerXEncoder,  # noqa: F401
                      SegFormerXEncoderLayer, SegFormerXFPN, clones)
from .decoder_leave_focal import InteractionAggregation, MultiScaleTemporalDetrLeaveFocal  # noqa: F401
from .my_evaluation import TOP_K_leave, TOP_K_leave_mask, draw_hotmap, main_eval_batch  # noqa: F401


class QueryBasedDecoder:  # named by the trainers' import line, defined nowhere in the reference
    def __init__(self, *a, **k):
        raise NotImplementedError("QueryBasedDecoder does not exist in the reference either (SURVEY.md §8(b))")
