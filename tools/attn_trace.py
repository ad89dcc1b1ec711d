"""Phase timeline of every wave of ONE workgroup of the fused attention backward (GPU box).  Debug build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSEGMM_ATT_TRACE -I include -o segmminterest_amd/libsegmm_atrace.so segmminterest_amd/csrc/capi.hip
usage: SEGMM_LIB=.../libsegmm_atrace.so python tools/attn_trace.py [drop_p]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from segmminterest_amd import hipabi as H
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
B, Hh, dh, Lq, La, Lb = 512, 16, 48, 40, 40, 100
d = Hh * dh
dev = "cuda"
Yv = torch.randn(B * La, 4 * d, device=dev); Yu = torch.randn(B * Lb, 2 * d, device=dev)
vm = (torch.rand(B, La, device=dev) < 0.8).to(torch.uint8); um = (torch.rand(B, Lb, device=dev) < 0.8).to(torch.uint8)
O = torch.empty(B * Lq, d, device=dev); lse = torch.empty(2, B, Hh, Lq, device=dev)
dO = torch.randn(B * Lq, d, device=dev); Dv = torch.empty(B, Hh, Lq, device=dev)
dYv = torch.empty_like(Yv); dYu = torch.empty_like(Yu)
H.attn_fwd(B, Hh, dh, Lq, La, Lb, (Yv, 0), (Yv, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d, vm, vm, um, O, d, lse,
           drop_p=p_drop, seed=1, site=3)
for ph in (1, 4, 4, 4):
    H.attn_bwd(B, Hh, dh, Lq, La, Lb, (Yv, 0), (Yv, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d, vm, vm, um, lse, O, d,
               dO, d, Dv, (dYv, 0), (dYv, d), 4 * d, (dYv, 2 * d), (dYv, 3 * d), 4 * d, (dYu, 0), (dYu, d), 2 * d, drop_p=p_drop, seed=1, site=3,
               phase=ph)
torch.cuda.synchronize()
L = H.lib()
L.segmm_debug_attn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(16 * 8, dtype=np.uint64)
L.segmm_debug_attn_trace(buf.ctypes.data, buf.size)
t = buf.reshape(16, 8).astype(np.int64)[:10]
t0 = t[:, 0].min()
names = ["stage loads issued+LDS writes", "barrier", "compute (3 query tiles)", "ordered dQ reduction", "dQ store"]
print("shader clocks since the workgroup's first stamp; one row per wave (= key tile)")
print("wave  start " + " ".join("%10s" % n[:10] for n in names) + "   end")
for w in range(10):
    print("%4d %6d " % (w, t[w, 0] - t0) + " ".join("%10d" % (t[w, i + 1] - t[w, i]) for i in range(5)) + " %6d" % (t[w, 5] - t0))
