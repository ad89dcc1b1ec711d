"""One plane-operand GEMM shape, a few launches (for rocprofv3 --pmc passes):  python tools/gemm_p_one.py nt|tn M N K [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmminterest_amd import hipabi as H
from segmminterest_amd import engine as E
lay, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda")
torch.manual_seed(0)
C = torch.empty(M, N, device=dev)
if lay == "nt":
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.02
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    fn = lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N)
else:
    dY, X = torch.randn(K, M, device=dev) * 0.01, torch.randn(K, N, device=dev)
    pa, pw = H.to_planes(dY, K, M), H.to_planes(X, K, N)
    sp = E._splits_for_p(M, N, K)
    ws = torch.empty(sp * M * N, device=dev)
    fn = lambda: H.gemm_p(H.LAYOUT_TN, M, N, K, pa, pw, C, N, splits=sp, workspace=ws)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
