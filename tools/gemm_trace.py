"""Phase timeline of one wave of the fp16x3 GEMM main loop (GPU box).  Debug build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSEGMM_GEMM_TRACE -I include -o segmminterest_amd/libsegmm_trace.so segmminterest_amd/csrc/capi.hip
usage: SEGMM_LIB=.../libsegmm_trace.so python tools/gemm_trace.py [M N K]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from segmminterest_amd import hipabi as H
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20480, 3072, 768)
dev = "cuda"
A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
am, bm = H.absmax(A, M, K, K), H.absmax(B, N, K, K)
planes = torch.empty(2, N * K, dtype=torch.float16, device=dev)
H.split2h(B, planes, N * K, bm)
for _ in range(5):
    H.gemm(0, M, N, K, A, K, B, K, C, N, engine=2, a_amax=am, b_amax=bm, b_planes=(planes, 0))
torch.cuda.synchronize()
L = H.lib()
L.segmm_debug_gemm_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(8 * 64, dtype=np.uint64)
L.segmm_debug_gemm_trace(buf.ctypes.data, buf.size)
t = buf.reshape(64, 8).astype(np.int64)
names = ["gload issue", "mma (ds_read+24 MFMA issue)", "barrier 1", "vmcnt(0) wait", "split + ds_write", "barrier 2"]
nit = (K + 31) // 32
print("phase durations in shader clocks, iterations 2..%d of one wave (workgroup mid-grid)" % (nit - 2))
rows = []
for i in range(2, nit - 1):
    d = [t[i, j + 1] - t[i, j] for j in range(6)]
    rows.append(d + [t[i + 1, 0] - t[i, 0]])
rows = np.array(rows)
for j, n in enumerate(names):
    print("%-30s mean %7.0f  min %6d  max %6d" % (n, rows[:, j].mean(), rows[:, j].min(), rows[:, j].max()))
print("%-30s mean %7.0f  min %6d  max %6d" % ("iteration", rows[:, 6].mean(), rows[:, 6].min(), rows[:, 6].max()))
