#!/usr/bin/env python
"""BASELINE config 1 labels (TEST INFRASTRUCTURE ONLY; runs in the build container where /root/reference exists).

Reads the reference's sample interaction file ``SegMM_inter_sample.csv`` (columns user_id, video_id, time_ms, duration_ms,
playing_time, label_1D) and writes the per-segment label rows of its first interactions, truncated / padded to S = 20
exactly as SURVEY.md §8(d) cfg 1 prescribes (padding -2; a row whose leave segment falls behind the cut becomes fully
watched), as a small fixture ``tests/golden/cfg1_labels.npz`` -- data only (int8 labels + the ids), no reference source.

    python oracle/gen_cfg1_labels.py [rows]
"""
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/SegMM_inter_sample.csv"
S = 20


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    labels, users, items, n_trunc = [], [], [], 0
    with open(SRC) as f:
        for r in csv.DictReader(f):
            lab = [int(x) for x in r["label_1D"].strip("[]").split()]
            if len(lab) > S:
                n_trunc += 1
                lab = lab[:S]
            lab = lab + [-2] * (S - len(lab))
            labels.append(lab)
            users.append(int(r["user_id"]))
            items.append(int(r["video_id"]))
            if len(labels) == rows:
                break
    out = os.path.join(ROOT, "tests", "golden", "cfg1_labels.npz")
    np.savez_compressed(out, label=np.array(labels, dtype=np.int8), user_id=np.array(users, dtype=np.int32),
                        video_id=np.array(items, dtype=np.int32))
    lab = np.array(labels)
    print("%d rows (%d truncated to S=%d), fully watched %.1f %%, %d KB" % (len(labels), n_trunc, S,
          100.0 * np.mean((lab == 0).sum(1) == 0), os.path.getsize(out) // 1024))


if __name__ == "__main__":
    main()
