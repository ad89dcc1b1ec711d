"""The fused attention backward must be bitwise reproducible run to run (ordered dQ adds through LDS turn counters): 30
repetitions of phase 4 on the config-2 video-side shape and on the user-side shape, every output compared bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import hipabi as H
dev = "cuda"
for (B, Hh, dh, Lq, La, Lb) in ((256, 16, 48, 40, 40, 100), (64, 16, 48, 100, 40, 100)):
    d = Hh * dh
    g = torch.Generator().manual_seed(1)
    mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(dev)
    Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(dev); mka = (torch.rand(B, La, generator=g) < 0.8).to(dev); mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(dev)
    O = torch.empty(B * Lq, d, device=dev); lse = torch.empty(2, B, Hh, Lq, device=dev)
    z = lambda t: (t, 0)
    H.attn_fwd(B, Hh, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, O, d, lse, drop_p=0.1, seed=5, site=3)
    dO = torch.randn(B * Lq, d, generator=g).to(dev)
    ref = None
    for it in range(30):
        Dv = torch.empty(B, Hh, Lq, device=dev)
        outs = [torch.full_like(t, float("nan")) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
        H.attn_bwd(B, Hh, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, lse, O, d, dO, d, Dv,
                   z(outs[0]), z(outs[1]), d, z(outs[2]), z(outs[3]), d, z(outs[4]), z(outs[5]), d, drop_p=0.1, seed=5, site=3, phase=4)
        torch.cuda.synchronize()
        if ref is None:
            ref = [o.clone() for o in outs]
            assert all(torch.isfinite(o).all() for o in outs)
        else:
            for a, b_ in zip(ref, outs):
                assert torch.equal(a, b_), "run %d differs" % it
    print("Lq=%d: 30 runs bitwise identical" % Lq)
