"""Median HIP-event time of one plane-operand GEMM shape:  python tools/probe/time_one.py nt|tn M N K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmminterest_amd import hipabi as H
from segmminterest_amd import engine as E
lay, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda")
torch.manual_seed(0)
C = torch.empty(M, N, device=dev)
ZERO = os.environ.get("PROBE_ZERO", "0") == "1"          # all-zero operands: same instruction stream and traffic, no bit toggling
if lay == "nt":
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.02
    if ZERO:
        A.zero_(); W.zero_()
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    fn = lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N)
else:
    dY, X = torch.randn(K, M, device=dev) * 0.01, torch.randn(K, N, device=dev)
    if ZERO:
        dY.zero_(); X.zero_()
    pa, pw = H.to_planes(dY, K, M), H.to_planes(X, K, N)
    sp = E._splits_for_p(M, N, K)
    ws = torch.empty(sp * M * N, device=dev)
    fn = lambda: H.gemm_p(H.LAYOUT_TN, M, N, K, pa, pw, C, N, splits=sp, workspace=ws)
for _ in range(5):
    fn()
ts = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3 / 20)
ts.sort()
t = ts[len(ts) // 2]
print("%.1f us  %.1f TF" % (t, 2.0 * M * N * K / t * 1e-6))
