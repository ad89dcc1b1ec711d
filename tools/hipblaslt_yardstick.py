"""YARD-STICK ONLY (tools/, never product): what the vendor library reaches on plain fp16 / bf16 GEMMs at the GEMM shapes of BASELINE
config 2 -- one MFMA product per element where the plane engine issues three (hi hi + lo hi + hi lo).  Tells whether the plane GEMM's
0.45 of the 833 TFLOP/s per-product bound stand-alone is the K = 768 ceiling of this part or short of it.
    python tools/hipblaslt_yardstick.py [iters]        (torch.mm -> hipBLASLt / rocBLAS; output fp16, fp32 accumulate)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmminterest_amd import hipabi as H
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
torch.manual_seed(0)

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

print("shape (C[M,N] = A[M,K] . B[N,K]^T)        vendor fp16      vendor bf16      plane GEMM fp16x3 (per-product TF = 3x)   ratio plane-product-rate / vendor fp16")
for M, N, K in [(20480, 768, 768), (51200, 768, 768), (20480, 3072, 768), (51200, 1536, 768), (20480, 768, 3072)]:
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.02
    out = {}
    for dt in (torch.float16, torch.bfloat16):
        a, w = A.to(dt), W.to(dt).t().contiguous().t()          # B as [N, K] row-major, used transposed (NT, like the plane GEMM)
        c = torch.empty(M, N, dtype=dt, device=dev)
        us = timeit(lambda: torch.mm(a, w.t(), out=c))
        out[dt] = (us, 2.0 * M * N * K / us / 1e6)
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    C = torch.empty(M, N, device=dev)
    us = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N))
    tf = 2.0 * M * N * K / us / 1e6
    print("NT %6d x %5d x %5d   %7.1f us %6.1f TF   %7.1f us %6.1f TF   %7.1f us %6.1f TF (%6.1f)      %.2f" % (
        M, N, K, out[torch.float16][0], out[torch.float16][1], out[torch.bfloat16][0], out[torch.bfloat16][1], us, tf, 3 * tf, 3 * tf / out[torch.float16][1]))
