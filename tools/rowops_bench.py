"""Stand-alone timing of the HBM-bound row kernels at the config-2 shapes (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


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


for rows in (20480, 51200):
    d = 768
    x, dy = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    y, dx, dxd = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    am = torch.zeros(H.AMAX_SLOTS, device=dev)
    us = timeit(lambda: H.layernorm_fwd(x, g, b, y, mean, rstd, drop_p=0.1, seed=1, site=2, amax=am))
    print("ln_fwd  rows %6d  %7.1f us  %5.2f TB/s (8 B/elem)" % (rows, us, rows * d * 8 / us / 1e6))
    parts = H.layernorm_bwd_parts(rows, d)
    pg, pb, ps = (torch.empty(parts, d, device=dev) for _ in range(3))
    us = timeit(lambda: H.layernorm_bwd(dy, x, mean, rstd, g, dx, dxd, pg, pb, drop_b_p=0.1, drop_b_site=3, seed=1, amax=am, part_dsum=ps))
    print("ln_bwd  rows %6d  %7.1f us  %5.2f TB/s (16 B/elem)  parts %d" % (rows, us, rows * d * 16 / us / 1e6, parts))
    us = timeit(lambda: H.layernorm_bwd(dy, x, mean, rstd, g, dx, None, pg, pb, amax=am))
    print("ln_bwd- rows %6d  %7.1f us  %5.2f TB/s (12 B/elem)" % (rows, us, rows * d * 12 / us / 1e6))
    for N in (768, 3072):
        X = torch.randn(rows, N, device=dev)
        ws = torch.empty(H.colsum_chunks(rows) * N, device=dev)
        out = torch.empty(N, device=dev)
        us = timeit(lambda: H.colsum(X, N, rows, N, out, ws))
        print("colsum  %6d x %4d  %7.1f us  %5.2f TB/s" % (rows, N, us, rows * N * 4 / us / 1e6))
    o = torch.empty_like(x)
    us = timeit(lambda: H.l1norm(x, out=o))
    print("l1norm  rows %6d  %7.1f us  %5.2f TB/s (8 B/elem)" % (rows, us, rows * d * 8 / us / 1e6))
    us = timeit(lambda: H.absmax(x, rows, d, d))
    print("absmax  rows %6d  %7.1f us  %5.2f TB/s" % (rows, us, rows * d * 4 / us / 1e6))
