import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("SEGMM_GRAD_ERR_LOG")
    if not path:
        return
    from helpers import GRAD_ERRS
    with open(path, "w") as f:
        f.write("worst |grad - reference| / max|reference| over the live gradients of each whole-model test (tolerance: 3e-4)\n")
        for label, (rel, name, scale) in sorted(GRAD_ERRS.items(), key=lambda kv: -kv[1][0]):
            f.write("%.2e  %-45s max %.2e  %s\n" % (rel, name, scale, label))
