"""Development aid: small TN plane GEMMs (split-K, column sums) against fp64, element-wise error map."""
import sys, torch
sys.path.insert(0, "/root/repo")
from segmminterest_amd import hipabi as H
dev = torch.device("cuda")
for (M, N, K) in [(256, 256, 32), (64, 32, 32), (64, 32, 8), (96, 64, 300), (32, 32, 8)]:
    dY = torch.randn(K, M, device=dev); X = torch.randn(K, N, device=dev)
    pdy, px = H.to_planes(dY, K, M), H.to_planes(X, K, N)
    C = torch.empty(M, N, device=dev)
    print("launch", M, N, K, flush=True)
    H.gemm_p(H.LAYOUT_TN, M, N, K, pdy, px, C, N)
    torch.cuda.synchronize()
    ref = dY.double().t() @ X.double()
    print("ok", M, N, K, float((C - ref).abs().max()), flush=True)
