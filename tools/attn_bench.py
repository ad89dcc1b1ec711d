"""Micro-benchmark of the attention kernels at BASELINE config 2 (video side: Lq=40, keys 40+100, h=16, dh=48)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
B, Hh, dh, Lq, La, Lb = 512, 16, 48, int(os.environ.get("LQ", 40)), 40, 100
d = Hh * dh
dev = "cuda"
Yv = torch.randn(B * La, 4 * d, device=dev)
Yu = torch.randn(B * Lb, 2 * d, device=dev)
Qsrc = Yv if Lq == La else torch.randn(B * Lq, 4 * d, device=dev)
vm = (torch.rand(B, La, device=dev) < 0.8).view(torch.uint8) if False else (torch.rand(B, La, device=dev) < 0.8).to(torch.uint8)
um = (torch.rand(B, Lb, device=dev) < 0.8).to(torch.uint8)
qm = vm if Lq == La else (torch.rand(B, Lq, device=dev) < 0.8).to(torch.uint8)
O = torch.empty(B * Lq, d, device=dev); lse = torch.empty(2, B, Hh, Lq, device=dev)
dO = torch.randn(B * Lq, d, device=dev); Dv = torch.empty(B, Hh, Lq, device=dev)
dYv = torch.empty_like(Yv); dYu = torch.empty_like(Yu); dQs = dYv if Lq == La else torch.empty_like(Qsrc)
fwd = lambda: H.attn_fwd(B, Hh, dh, Lq, La, Lb, (Qsrc, 0), (Qsrc, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d,
                         qm, vm, um, O, d, lse, drop_p=p_drop, seed=1, site=3)
def _site(x, cols):
    hdr = H.new_site(dev)[0]
    H.absmax(x, x.shape[0], cols, cols, out=hdr[H.SITE_HDR:])
    pl = torch.empty((x.shape[0], 2 * cols), dtype=torch.float16, device=dev)
    H.split_p32(x, x.shape[0], cols, cols, pl, 2 * cols, hdr, mode=0)
    return pl, hdr
plv, hv = _site(Yv, 4 * d); plu, hu = _site(Yu, 2 * d)
plq, hq = (plv, hv) if Qsrc is Yv else _site(Qsrc, 4 * d)
PIN = dict(q=(plq, hq, 8 * d), a=(plv, hv, 8 * d), b=(plu, hu, 4 * d))
plo = torch.empty((B * Lq, 2 * d), dtype=torch.float16, device=dev); ho = H.new_site(dev)[0]; so = torch.tensor([4096.0], device=dev)
PO = H.PO(plo, 2 * d, ho, so.data_ptr())
fwd_pl = lambda: H.attn_fwd(B, Hh, dh, Lq, La, Lb, (Qsrc, 0), (Qsrc, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d,
                            qm, vm, um, O, d, lse, drop_p=p_drop, seed=1, site=3, pin=PIN, po=PO)
fwd_po = lambda: H.attn_fwd(B, Hh, dh, Lq, La, Lb, (Qsrc, 0), (Qsrc, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d,
                            qm, vm, um, O, d, lse, drop_p=p_drop, seed=1, site=3, po=PO)
bwd_ph = lambda ph: H.attn_bwd(B, Hh, dh, Lq, La, Lb, (Qsrc, 0), (Qsrc, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d,
                               qm, vm, um, lse, O, d, dO, d, Dv, (dQs, 0), (dQs, d), 4 * d, (dYv, 2 * d), (dYv, 3 * d), 4 * d, (dYu, 0), (dYu, d), 2 * d,
                               drop_p=p_drop, seed=1, site=3, phase=ph)
bwd = lambda: bwd_ph(0)
bwd_pl = lambda: H.attn_bwd(B, Hh, dh, Lq, La, Lb, (None, 0), (None, d), 4 * d, (None, 2 * d), (None, 3 * d), 4 * d, (None, 0), (None, d), 2 * d,
                            qm, vm, um, lse, O, d, dO, d, Dv, (dQs, 0), (dQs, d), 4 * d, (dYv, 2 * d), (dYv, 3 * d), 4 * d, (dYu, 0), (dYu, d), 2 * d,
                            drop_p=p_drop, seed=1, site=3, phase=4, pin=PIN)
def bwd_fused():
    bwd_ph(4)
T = La + Lb
cases = [("fwd", fwd, 4.0 * dh * Lq * T), ("fwd+planes out", fwd_po, 4.0 * dh * Lq * T), ("fwd planes-in", fwd_pl, 4.0 * dh * Lq * T), ("bwd(dq+dkv)", bwd, 14.0 * dh * Lq * T), ("bwd D only", lambda: bwd_ph(1), 0.0)]
if True:
    cases.append(("bwd(fused)", bwd_fused, 14.0 * dh * Lq * T))
    cases.append(("bwd planes-in", bwd_pl, 14.0 * dh * Lq * T))
for name, fn, flops in cases:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("attn %-12s p=%.2f Lq=%d  %8.1f us  %6.2f TFLOP/s (unpadded algorithmic)" % (name, p_drop, Lq, us, flops * B * Hh / us / 1e6))
