"""Micro-benchmark of the attention kernels at config 3's shapes (B = 1024, h = 16, dh = 32; video queries 20 x (20 + 1), user queries 1 x (20 + 1))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, Hh, dh, La, Lb = 1024, 16, 32, 20, 1
d = Hh * dh
dev = "cuda"
Yv = torch.randn(B * La, 6 * d, device=dev); Yu = torch.randn(B * Lb, 6 * d, device=dev)
vm = (torch.rand(B, La, device=dev) < 0.8).to(torch.uint8); um = torch.ones(B, Lb, device=dev, dtype=torch.uint8)
for Lq, Qs, qm in ((20, Yv, vm), (1, Yu, um)):
    O = torch.empty(B * Lq, d, device=dev); lse = torch.empty(2, B, Hh, Lq, device=dev)
    dO = torch.randn(B * Lq, d, device=dev); Dv = torch.empty(B, Hh, Lq, device=dev)
    dYv, dYu = torch.empty_like(Yv), torch.empty_like(Yu); dQs = dYv if Lq == 20 else dYu
    fwd = lambda: H.attn_fwd(B, Hh, dh, Lq, La, Lb, (Qs, 0), (Qs, d), 6 * d, (Yv, 2 * d), (Yv, 3 * d), 6 * d, (Yu, 4 * d), (Yu, 5 * d), 6 * d,
                             qm, vm, um, O, d, lse, drop_p=0.1, seed=1, site=3)
    bwd = lambda: H.attn_bwd(B, Hh, dh, Lq, La, Lb, (Qs, 0), (Qs, d), 6 * d, (Yv, 2 * d), (Yv, 3 * d), 6 * d, (Yu, 4 * d), (Yu, 5 * d), 6 * d,
                             qm, vm, um, lse, O, d, dO, d, Dv, (dQs, 0), (dQs, d), 6 * d, (dYv, 2 * d), (dYv, 3 * d), 6 * d, (dYu, 4 * d), (dYu, 5 * d), 6 * d,
                             drop_p=0.1, seed=1, site=3, phase=4)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        print("attn %s Lq=%d  %8.1f us" % (name, Lq, e0.elapsed_time(e1) * 1e3 / iters))
