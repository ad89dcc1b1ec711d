"""fp16x3 engine: accuracy vs fp64 next to the f32-MFMA and bf16x6 engines, and timing (run on the GPU box).
usage: python tools/gemm_h_check.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
torch.manual_seed(0)


def operands(lay, M, N, K, kind):
    def rnd(*s):
        x = torch.randn(*s, device=dev)
        if kind == "wide":          # magnitudes spread over 2^-20 .. 2^0, like gradients
            x = x * torch.exp2(-20 * torch.rand(*s, device=dev))
        elif kind == "tiny":
            x = x * 1e-9
        return x
    if lay == "NT":
        return rnd(M, K), rnd(N, K), K, K
    if lay == "NN":
        return rnd(M, K), rnd(K, N), K, N
    return rnd(K, M), rnd(K, N), M, N


def ref64(lay, A, B):
    a, b = A.double(), B.double()
    return a @ b.t() if lay == "NT" else (a @ b if lay == "NN" else a.t() @ b)


def timeit(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print("== accuracy: mean |C - C64| / mean |C64|   (max |err| / mean |C64|)")
for lay, M, N, K in [("NT", 2048, 768, 768), ("NN", 2048, 768, 3072), ("TN", 768, 768, 20480), ("NT", 1000, 260, 516)]:
    for kind in ("randn", "wide", "tiny"):
        A, B, lda, ldb = operands(lay, M, N, K, kind)
        R = ref64(lay, A, B)
        L = {"NT": 0, "NN": 1, "TN": 2}[lay]
        line = "%s %5dx%4dx%5d %-5s" % (lay, M, N, K, kind)
        for eng in (0, 1, 2):
            C = torch.empty(M, N, device=dev)
            H.gemm(L, M, N, K, A, lda, B, ldb, C, N, engine=eng)
            e = (C.double() - R).abs()
            line += "  e%d %.2e (%.1e)" % (eng, (e.mean() / R.abs().mean()).item(), (e.max() / R.abs().mean()).item())
        print(line)

print("== timing, amax precomputed (algorithmic TFLOP/s)")
shapes = [("NT", 20480, 768, 768), ("NT", 51200, 768, 768), ("NT", 20480, 3072, 768), ("NT", 51200, 1536, 768),
          ("NN", 20480, 768, 3072), ("TN", 768, 768, 20480), ("TN", 3072, 768, 20480), ("TN", 1536, 768, 51200)]
for lay, M, N, K in shapes:
    A, B, lda, ldb = operands(lay, M, N, K, "randn")
    L = {"NT": 0, "NN": 1, "TN": 2}[lay]
    C = torch.empty(M, N, device=dev)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    splits = max(1, min(32, (K + 31) // 32, (1024 + tiles - 1) // tiles)) if lay == "TN" else 1
    ws = torch.empty(splits * M * N, device=dev) if splits > 1 else None
    am = H.absmax(A, A.shape[0], A.shape[1], lda)
    bm = H.absmax(B, B.shape[0], B.shape[1], ldb)
    out = "%s %5dx%4dx%5d splits %2d" % (lay, M, N, K, splits)
    for eng in (1, 2):
        us = timeit(lambda: H.gemm(L, M, N, K, A, lda, B, ldb, C, N, splits=splits, workspace=ws, engine=eng, a_amax=am, b_amax=bm))
        out += "   e%d %7.1f us %6.1f TF" % (eng, us, 2.0 * M * N * K / us / 1e6)
    if lay == "NT":
        planes = torch.empty(2, N * K, dtype=torch.int16, device=dev)
        H.split2h(B, planes, N * K, bm)
        us = timeit(lambda: H.gemm(L, M, N, K, A, lda, None, ldb, C, N, engine=2, a_amax=am, b_amax=bm, b_planes=(planes, 0)))
        out += "   e2+Bplanes %7.1f us %6.1f TF" % (us, 2.0 * M * N * K / us / 1e6)
        C2 = torch.empty(M, N, device=dev)
        H.gemm(L, M, N, K, A, lda, B, ldb, C2, N, engine=2, a_amax=am, b_amax=bm)
        out += "  planes==fly %s" % bool((C == C2).all().item())
    us = timeit(lambda: H.absmax(A, A.shape[0], A.shape[1], lda, out=am))
    out += "   absmax(A) %6.1f us" % us
    print(out)
