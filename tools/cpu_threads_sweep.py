#!/usr/bin/env python
"""CPU-baseline thread sweep (BASELINE.md §3 asks for "all physical cores"; torch-CPU over-subscribes on the GPU boxes' hosts):
the oracle's train step of bench.py's cpu_baseline leg at several thread counts, once, on a GPU box's host.

    python tools/cpu_threads_sweep.py --threads 16,64,128 > profiles/cpu_baseline_threads.json

bench.py quotes the file in cpu_baseline.sample.  TEST/measurement tooling: it imports the oracle, never the product path's kernels."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="16,64,128")
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--budget-s", type=float, default=60.0, help="per thread count: stop timing after this many seconds (>= 1 timed step)")
    a = ap.parse_args()
    import torch
    import torch.nn.functional as F
    import segmm_oracle as O
    from segmminterest_amd.synth import l1_normalize, make_batch
    from segmminterest_amd.trainer import default_args, init_model
    S, D, Lt, N, h, Bc = 40, 768, 100, 2, 16, a.rows
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    torch.manual_seed(0)
    model = init_model(margs, n_users=1903, n_items=1000, input_dim=D, max_vid_len=S, max_usr_len=Lt)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(Bc, S, Lt, D, n_users=1903, n_items=1000, seed=1234)
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=["interestBPR"], loss_weight=margs.loss_weight, exposure_prob=[1.0] * S)
    drop = lambda t: F.dropout(t, 0.1)

    def one():
        inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
                   vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
        O.train_steps(sd, cfg, inp, 1, skip_dead=False, drop=drop)

    host = os.cpu_count() or 1
    res = {}
    for n in [int(x) for x in a.threads.split(",")]:
        n = min(n, host)
        torch.set_num_threads(n)
        one()                                   # warm-up
        ts, t_start = [], time.perf_counter()
        while len(ts) < 5 and (not ts or time.perf_counter() - t_start < a.budget_s):
            t0 = time.perf_counter()
            one()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        res[str(n)] = round(Bc / ts[len(ts) // 2], 2)
        sys.stderr.write("threads %d: %.2f rows/s (%d timed steps)\n" % (n, res[str(n)], len(ts)))
    try:
        git = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        git = os.environ.get("SEGMM_GIT", "?")
    print(json.dumps({"host_threads": host, "workload": "config 2 shapes, B=%d rows, train mode, dead layers executed" % Bc,
                      "rows_per_s": res, "git": git}))


if __name__ == "__main__":
    main()
