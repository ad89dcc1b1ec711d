"""Ad-hoc accuracy check of a GEMM engine against fp64 on all layouts (run on the GPU box)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H
eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bad = 0
for lay, M, N, K, splits in [("NT", 360, 96, 48, 1), ("NT", 20480, 768, 768, 1), ("NT", 63, 32, 40, 1), ("NT", 128, 128, 16, 1),
                             ("NN", 360, 48, 96, 1), ("NN", 20480, 768, 3072, 1), ("NN", 200, 32, 32, 1),
                             ("TN", 96, 48, 360, 1), ("TN", 768, 768, 20480, 16), ("TN", 32, 40, 63, 2), ("TN", 3072, 768, 5120, 8)]:
    L = {"NT": 0, "NN": 1, "TN": 2}[lay]
    g = torch.Generator().manual_seed(M + N + K)
    rnd = lambda *s: (torch.randn(*s, generator=g) * torch.exp(torch.randn(*s, generator=g))).cuda()
    if lay == "NT":
        A, B = rnd(M, K), rnd(N, K); lda, ldb = K, K; ref = A.double() @ B.double().t()
    elif lay == "NN":
        A, B = rnd(M, K), rnd(K, N); lda, ldb = K, N; ref = A.double() @ B.double()
    else:
        A, B = rnd(K, M), rnd(K, N); lda, ldb = M, N; ref = A.double().t() @ B.double()
    ws = torch.empty(max(splits, 1) * M * N, device="cuda")
    errs = []
    for e in (0, eng):
        C = torch.full((M, N), float("nan"), device="cuda")
        H.gemm(L, M, N, K, A, lda, B, ldb, C, N, splits=splits, workspace=ws, engine=e)
        errs.append((C.double() - ref).abs().max().item() / ref.abs().mean().item())
    ok = errs[1] <= 2 * errs[0] + 1e-7
    bad += not ok
    print("%s %5dx%5dx%5d  f32 err %.2e  engine%d err %.2e  %s" % (lay, M, N, K, errs[0], eng, errs[1], "ok" if ok else "BAD"))
sys.exit(1 if bad else 0)
