"""Shared test helpers: golden-fixture loading."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

MODEL_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f != "metrics_kat.npz")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    cfg["loss_type_list"] = [x.strip() for x in cfg["loss"].split(",")]
    grp = {"sd": {}, "in": {}, "out": {}, "grad": {}, "adam1": {}, "adam3": {}, "inf": {}}
    for k in z.files:
        if "/" in k:
            g, n = k.split("/", 1)
            grp[g][n] = torch.from_numpy(z[k])
    nograd = json.loads(str(z["nograd"]))
    extra = {"adam_loss3": float(z["adam_loss3"])} if "adam_loss3" in z.files else {}
    return cfg, grp, nograd, extra
