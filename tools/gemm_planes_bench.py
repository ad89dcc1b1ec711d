"""bf16x6 engine with pre-split operands / 2-plane wgrad: accuracy vs fp64 and speed (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def planes_of(x):          # [R, C] fp32 -> [3, R*C] bf16 via the library kernel
    p = torch.empty(3, x.numel(), dtype=torch.bfloat16, device="cuda")
    H.split3(x, p, x.numel())
    return p

g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g) * torch.exp(torch.randn(*s, generator=g))).cuda()
for M, N, K in [(20480, 768, 768), (51200, 768, 768), (20480, 3072, 768), (20480, 768, 3072), (360, 96, 48)]:
    A, W = rnd(M, K), rnd(N, K)
    ref = A.double() @ W.double().t()
    sc = ref.abs().mean().item()
    C = torch.empty(M, N, device="cuda")
    Ap, Wp = planes_of(A), planes_of(W)
    assert torch.equal((Ap[0].float() + Ap[1].float() + Ap[2].float()).view(M, K), A), "split not exact"
    res = {}
    for name, kw in (("fly", {}), ("Bpre", dict(b_planes=(Wp, 0))), ("ABpre", dict(a_planes=(Ap, 0), b_planes=(Wp, 0)))):
        fn = lambda: H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, engine=1, **kw)
        us = timeit(fn)
        err = (C.double() - ref).abs().max().item() / sc
        res[name] = (us, err)
    print("NT %5dx%5dx%5d " % (M, N, K) + "  ".join("%s %7.1f us %6.1f TF err %.1e" % (k, v[0], 2.0 * M * N * K / v[0] / 1e6, v[1]) for k, v in res.items()))
# transposed planes: dgrad in NT form:  dX[M, Kin] = dY[M, Nout] . W[Nout, Kin]  ==  NT with B' = W^T planes [Kin, Nout]
M, Nout, Kin = 20480, 3072, 768
dY, W = rnd(M, Nout), rnd(Nout, Kin)
WT = torch.empty(3, W.numel(), dtype=torch.bfloat16, device="cuda")
H.split3_transpose(W, Nout, Kin, Kin, WT)
assert torch.equal((WT[0].float() + WT[1].float() + WT[2].float()).view(Kin, Nout), W.t().contiguous()), "transpose split wrong"
dX = torch.empty(M, Kin, device="cuda")
ref = dY.double() @ W.double()
for name, fn in (("NN fly", lambda: H.gemm(H.LAYOUT_NN, M, Kin, Nout, dY, Nout, W, Kin, dX, Kin, engine=1)),
                 ("NT W^T planes", lambda: H.gemm(H.LAYOUT_NT, M, Kin, Nout, dY, Nout, None, Nout, dX, Kin, engine=1, b_planes=(WT, 0)))):
    us = timeit(fn)
    print("dgrad %-14s %7.1f us %6.1f TF err %.1e" % (name, us, 2.0 * M * Kin * Nout / us / 1e6, (dX.double() - ref).abs().max().item() / ref.abs().mean().item()))
# wgrad with 2 planes (three products)
for Mo, No, Kt, splits in [(768, 768, 20480, 29), (3072, 768, 20480, 8), (1536, 768, 51200, 15)]:
    A, B = rnd(Kt, Mo), rnd(Kt, No)
    ref = A.double().t() @ B.double()
    C = torch.empty(Mo, No, device="cuda"); ws = torch.empty(splits * Mo * No, device="cuda")
    out = []
    for npl in (3, 2):
        fn = lambda: H.gemm(H.LAYOUT_TN, Mo, No, Kt, A, Mo, B, No, C, No, splits=splits, workspace=ws, engine=1, nplanes=npl)
        us = timeit(fn)
        out.append("x%d %7.1f us %6.1f TF err %.1e" % (6 if npl == 3 else 3, us, 2.0 * Mo * No * Kt / us / 1e6, (C.double() - ref).abs().max().item() / ref.abs().mean().item()))
    print("wgrad TN %5dx%5dx%5d  " % (Mo, No, Kt) + "   ".join(out))
