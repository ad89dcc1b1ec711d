"""NT plane GEMM at the two fused-projection shapes of config 2: fp32 output, planes beside fp32, planes only, the repair launch.
   python tools/gemm_out_variants.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmminterest_amd import hipabi as H
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
torch.manual_seed(0)

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for M, N, K in [(20480, 3072, 768), (51200, 1536, 768)]:
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.02
    bias = torch.randn(N, device=dev) * 0.01
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    C = torch.empty(M, N, device=dev)
    pl = torch.empty(M, 2 * N, dtype=torch.float16, device=dev)
    hdr = H.new_site(dev)[0]
    sc = torch.tensor([1024.0], device=dev)
    cpt = H.PT(pl, hdr, M, N, ld2=2 * N, p_off=0, f32=None, ldf=N, f_off=0)
    res = {}
    res["fp32"] = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N, bias=bias))
    res["fp32+amax"] = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N, bias=bias, c_hdr=hdr))
    res["fp32+planes"] = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, C, N, bias=bias, c_pt=cpt, c_scale_ptr=sc.data_ptr()))
    res["planes only"] = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, None, N, bias=bias, c_pt=cpt, c_scale_ptr=sc.data_ptr(), write_c=False))
    res["repair (no-op)"] = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, None, N, bias=bias, c_pt=cpt, c_scale_ptr=sc.data_ptr(), write_c=False, repair=True))
    print("NT %d x %d x %d: " % (M, N, K) + "   ".join("%s %.1f us" % kv for kv in res.items()), " flag", float(hdr[1]))
