#!/usr/bin/env python
"""Split-K count of the plane-operand weight-gradient GEMM (gemm_pl_tn8 + splitk_reduce), shape by shape: stand-alone time of
the launch pair for every split count, with the count engine._splits_for_p picks marked.  The shapes are config 2's.
    python tools/tn_split_table.py > gpurun_out/r5/tn_split_table.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmminterest_amd import engine as E          # noqa: E402
from segmminterest_amd import hipabi as H          # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    shapes = [(768, 768, 20480), (3072, 768, 20480), (768, 3072, 20480), (1536, 768, 51200), (768, 768, 51200), (768, 768, 71680)]
    counts = (1, 2, 4, 7, 9, 14, 28, 56)
    print("TN  M x N x K (dW[M, N] = dY[K, M]^T X[K, N]; 256 x 256 tiles): us of gemm_pl_tn8 + splitk_reduce per split count; * = engine's choice")
    print("%-24s %s" % ("shape", "".join("%10d" % c for c in counts)))
    for (M, N, K) in shapes:
        dY = torch.randn(K, M, device=dev) * 0.01
        X = torch.randn(K, N, device=dev)
        pdy, px = H.to_planes(dY, K, M), H.to_planes(X, K, N)
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(max(counts) * M * N, device=dev)
        pick = E._splits_for_p(M, N, K)
        cells = []
        for c in sorted(set(counts) | {pick}):
            if c > (K + 31) // 32:
                continue
            t = timeit(lambda: H.gemm_p(H.LAYOUT_TN, M, N, K, pdy, px, C, N, splits=c, workspace=ws if c > 1 else None))
            cells.append((c, t * 1e6))
        best = min(cells, key=lambda x: x[1])
        print("%6d x %5d x %6d  " % (M, N, K) + "".join(("%9.1f%s" % (t, "*" if c == pick else " ")) for c, t in cells if c in counts or c == pick)
              + "   tiles %3d  pick %2d  best %2d (%.1f us, %.1f TF)" % (((M + 255) // 256) * ((N + 255) // 256), pick, best[0], best[1], 2.0 * M * N * K / best[1] / 1e6),
              flush=True)
        del dY, X, pdy, px, ws


if __name__ == "__main__":
    main()
