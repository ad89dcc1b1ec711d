"""Debug aid: fused fp16x3 attention backward (phase 4) against the two-kernel exact-fp32 backward (phase 0), output by output."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmminterest_amd import hipabi as H
DEV = "cuda"
def case(B, H_, dh, Lq, La, Lb, p=0.0, masks=True, scale=0.7):
    d = H_ * dh
    g = torch.Generator().manual_seed(77)
    mk = lambda L: (torch.randn(B, L, d, generator=g) * scale).to(DEV)
    Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
    if masks:
        mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV); mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV); mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    else:
        mq = torch.ones(B, Lq, dtype=torch.bool, device=DEV); mka = torch.ones(B, La, dtype=torch.bool, device=DEV); mkb = torch.ones(B, Lb, dtype=torch.bool, device=DEV)
    O = torch.empty(B * Lq, d, device=DEV); lse = torch.empty(2, B, H_, Lq, device=DEV)
    z = lambda t: (t, 0)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, O, d, lse, drop_p=p, seed=5, site=3)
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)
    def run(ph):
        Dv = torch.full((B, H_, Lq), float("nan"), device=DEV)
        outs = [torch.full_like(t, float("nan")) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
        H.attn_bwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, lse, O, d, dO, d, Dv,
                   z(outs[0]), z(outs[1]), d, z(outs[2]), z(outs[3]), d, z(outs[4]), z(outs[5]), d, drop_p=p, seed=5, site=3, phase=ph)
        torch.cuda.synchronize()
        return outs
    ref, got = run(0), run(4)
    print("case", (B, H_, dh, Lq, La, Lb), "p", p, "masks", masks)
    for name, a, b in zip(("dQa", "dQb", "dKa", "dVa", "dKb", "dVb"), ref, got):
        nan = torch.isnan(b).sum().item()
        err = (a - b).nan_to_num(0).abs().max().item()
        print("   %-4s max|ref| %.3e  nan %6d / %d  max err %.3e" % (name, a.abs().max().item(), nan, b.numel(), err))
        if err > 1e-4 and not nan:
            r = (b / a).flatten()
            print("      max|got| %.3e  got/ref: median %.4g  min %.4g max %.4g   got[0,:4,:4] %s\n      ref[0,:4,:4] %s" % (b.abs().max().item(), r.median().item(), r.min().item(), r.max().item(), b[0, :4, :4].tolist(), a[0, :4, :4].tolist()))
        if nan:
            idx = torch.isnan(b).nonzero()
            print("      first nan at", idx[0].tolist(), "last", idx[-1].tolist(), "rows with nan", torch.isnan(b).any(-1).sum().item())
for c in [(1, 1, 16, 16, 16, 16), (1, 1, 32, 16, 16, 16), (1, 1, 48, 16, 16, 16)]:
    case(*c, p=0.0, masks=False)
