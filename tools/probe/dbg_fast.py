"""Reproducer of the store-data hazard (gemm_planes8.h buf_store4): the NT plane GEMM's plain-store epilogue on a small shape,
row by row against fp64 -- without the explicit wait states rows 12-15 of a tile carry the NEXT float4's .y/.w."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from segmminterest_amd import hipabi as H
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 32), (256, 256, 64), (512, 512, 96)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    C = torch.zeros(M, N, device="cuda")
    H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N)
    ref = (A.double() @ W.double().t()).float()
    bad = ((C - ref).abs() > 1e-4 * ref.abs().max())
    print(M, N, K, "bad elements", int(bad.sum()), "of", M * N)
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print(" bad rows", rows[:40], "...", len(rows)); print(" bad cols", cols[:40], "...", len(cols))
        r, c = bad.nonzero()[0].tolist()
        print(" first bad", r, c, float(C[r, c]), float(ref[r, c]))
        # does the wrong value appear elsewhere in ref?
        w = (ref - C[r, c]).abs() < 1e-6
        print(" value found at", w.nonzero()[:4].tolist())
