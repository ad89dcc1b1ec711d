"""Micro-benchmark of segmm_gemm on the shapes of BASELINE config 2 (run on the GPU box).
usage: python tools/gemm_bench.py [iters] [shape filter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda"
shapes = [("NT", 20480, 768, 768), ("NT", 51200, 768, 768), ("NT", 20480, 3072, 768), ("NT", 51200, 1536, 768),
          ("NN", 20480, 768, 768), ("NN", 20480, 768, 3072), ("NN", 51200, 768, 1536),
          ("TN", 768, 768, 20480), ("TN", 3072, 768, 20480), ("TN", 1536, 768, 51200), ("TN", 768, 768, 51200)]
for lay, M, N, K in shapes:
    tag = "%s_%dx%dx%d" % (lay, M, N, K)
    if flt and flt not in tag:
        continue
    L = {"NT": 0, "NN": 1, "TN": 2}[lay]
    if lay == "NT":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); lda, ldb = K, K
    elif lay == "NN":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); lda, ldb = K, N
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); lda, ldb = M, N
    if os.environ.get("ZERO") == "1":       # data-dependent power: all-zero operands toggle no datapath bits
        A.zero_(); B.zero_()
    C = torch.empty(M, N, device=dev)
    env_bn = os.environ.get("SEGMM_GEMM_BN", "")
    bn = 256 if (os.environ.get("SEGMM_GEMM", "f32") == "f16x3" and N > 128 and env_bn != "128" and
                 (env_bn == "256" or (lay == "TN" and ((M + 127) // 128) * ((N + 255) // 256) >= 36))) else 128
    tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
    splits = max(1, min(32, (K + 31) // 32, (1024 + tiles - 1) // tiles)) if lay == "TN" else 1
    ws = torch.empty(splits * M * N, device=dev) if splits > 1 else None
    eng = {"f32": 0, "bf16x6": 1, "f16x3": 2}[os.environ.get("SEGMM_GEMM", "f32")]
    kw = {}
    if eng == 2:        # partial maxima precomputed (the fused producers supply them in the model); weights pre-split for NT
        kw["a_amax"] = H.absmax(A, A.shape[0], A.shape[1], lda)
        kw["b_amax"] = H.absmax(B, B.shape[0], B.shape[1], ldb)
        if lay == "NT" and os.environ.get("SEGMM_PLANES", "1") != "0":
            planes = torch.empty(2, N * K, dtype=torch.float16, device=dev)
            H.split2h(B, planes, N * K, kw["b_amax"])
            kw["b_planes"] = (planes, 0)
    run = lambda: H.gemm(L, M, N, K, A, lda, B, ldb, C, N, splits=splits, workspace=ws, engine=eng, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(os.environ.get("SEGMM_GEMM", "f32"), "%-22s splits %2d  %8.1f us  %6.1f TFLOP/s  (%.1f%% of 157.3)" % (tag, splits, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 1.573))
