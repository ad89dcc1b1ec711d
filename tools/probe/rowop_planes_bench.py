"""Stand-alone timing of the row producers with and without their plane output (development aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmminterest_amd import hipabi as H
dev = torch.device("cuda")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows in (20480, 51200):
    d = 768
    x = torch.rand(rows, d, device=dev)
    y = torch.empty_like(x)
    hdr = H.new_site(dev)[0]; sc = torch.tensor([2.0 ** 12], device=dev)
    pl = torch.empty(rows, 2 * d, dtype=torch.float16, device=dev)
    po = H.PO(pl, 2 * d, hdr, sc.data_ptr())
    t0 = timeit(lambda: H.l1norm(x, y, amax=hdr[H.SITE_HDR:]))
    t1 = timeit(lambda: H.l1norm(x, y, po=po))
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    m, r = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    t2 = timeit(lambda: H.layernorm_fwd(x, g, b, y, m, r, amax=hdr[H.SITE_HDR:]))
    t3 = timeit(lambda: H.layernorm_fwd(x, g, b, y, m, r, po=po))
    t4 = timeit(lambda: H.split_p32(x, rows, d, d, pl, 2 * d, hdr, mode=1))
    t5 = timeit(lambda: y.copy_(x))
    mb = rows * d * 4 / 1e6
    print("rows %6d: l1norm %6.1f us (%.2f TB/s) +planes %6.1f us (%.2f TB/s) | ln_fwd %6.1f -> %6.1f us | split pass %6.1f us (%.2f TB/s) | copy %6.1f us (%.2f TB/s)"
          % (rows, t0, 2 * mb / t0, t1, 3 * mb / t1, t2, t3, t4, 2 * mb / t4, t5, 2 * mb / t5))
