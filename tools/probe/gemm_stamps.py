"""Where does a tile of the plane NT GEMM spend its time?  Needs a -DSEGMM_STAMPS build (SEGMM_LIB) and SEGMM_PL_VAR=8.
    SEGMM_LIB=build/probe/libsegmm_stamps.so SEGMM_PL_VAR=8 python tools/probe/gemm_stamps.py M N K [NJ]
The tile width (64 NJ columns) is pinned through SEGMM_PL_NJ (default 4) so that the workgroup count is known here.
Per workgroup: s_memtime at tile start / after the prologue / after the k-loop / after the epilogue (+ s_memrealtime at both ends:
100 MHz).  Prints the median section lengths in shader cycles and in us (cycles / the in-kernel clock)."""
import ctypes, os, sys
NJ = int(sys.argv[4]) if len(sys.argv) > 4 else 4
os.environ["SEGMM_PL_NJ"] = str(NJ)          # read once by the library, on its first plane GEMM
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmminterest_amd import hipabi as H
M, N, K = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda")
torch.manual_seed(0)
A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.02
pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
C = torch.empty(M, N, device=dev)
nwg = ((M + 255) // 256) * ((N + 64 * NJ - 1) // (64 * NJ))
st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
L = H.lib()
L.segmm_debug_set_stamps.argtypes = [ctypes.c_void_p]
fn = lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N)
for _ in range(200):          # clocks settle under load
    fn()
torch.cuda.synchronize()
L.segmm_debug_set_stamps(st.data_ptr())
for _ in range(3):
    fn()
torch.cuda.synchronize()
L.segmm_debug_set_stamps(None)
s = st.view(nwg, 8).cpu().double()
pro, loop, epi, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
real_us = (s[:, 5] - s[:, 4]) / 100.0
clk = (tot / real_us).median()          # MHz
med = lambda t: float(t.median())
print("%d x %d x %d, tile 256 x %d: %d workgroups; in-kernel clock %.0f MHz" % (M, N, K, 64 * NJ, nwg, clk))
for name, t in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("tile", tot)):
    print("  %-9s median %8.0f cycles = %6.2f us   (p10 %6.2f  p90 %6.2f us)" % (name, med(t), med(t) / clk, float(t.quantile(0.1)) / clk, float(t.quantile(0.9)) / clk))
start = s[:, 4] - s[:, 4].min()
print("  kernel span (first start -> last end): %.1f us; tile starts: p50 %.1f us, max %.1f us" % (float((s[:, 5].max() - s[:, 4].min()) / 100.0), float(start.median() / 100.0), float(start.max() / 100.0)))
