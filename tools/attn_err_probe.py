"""Diagnostic: where does the attention dV error of a one-block case sit? (GPU box)"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmminterest_amd import hipabi as H
DEV = "cuda"
B, H_, dh, Lq, La, Lb = 3, 4, 8, 40, 0, 10
d = H_ * dh
g = torch.Generator().manual_seed(B * 1000 + Lq + La)
mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(DEV)
Q, K, V = mk(Lq), mk(Lb), mk(Lb)
mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
mk1 = (torch.rand(B, Lb, generator=g) < 0.75).to(DEV)
mq[0, 0] = False
O = torch.empty(B * Lq, d, device=DEV); lse = torch.empty(2, B, H_, Lq, device=DEV)
z = lambda t: (t, 0)
blocks = (None, z(Q), d, None, None, 0, z(K), z(V), d); masks = (mq, None, mk1)
H.attn_fwd(B, H_, dh, Lq, La, Lb, *blocks, *masks, O, d, lse)
Qd, Kd, Vd = [t.double().requires_grad_(True) for t in (Q, K, V)]
sp = lambda t: t.view(B, t.shape[1], H_, dh)
lg = torch.einsum("bqhd,bkhd->bhqk", sp(Qd), sp(Kd))
lg = torch.where((mq[:, :, None] & mk1[:, None, :])[:, None], lg, torch.full_like(lg, -10000.0)) / math.sqrt(dh)
P = lg.softmax(-1)
ref = torch.einsum("bhqk,bkhd->bqhd", P, sp(Vd)).reshape(B, Lq, d)
dO = torch.randn(B * Lq, d, generator=g).to(DEV)
ref.backward(dO.view(B, Lq, d).double())
Dv = torch.empty(B, H_, Lq, device=DEV)
dQ, dK, dV = [torch.full_like(t, float("nan")) for t in (Q, K, V)]
grads = (None, z(dQ), d, None, None, 0, z(dK), z(dV), d)
H.attn_bwd(B, H_, dh, Lq, La, Lb, *blocks, *masks, lse, O, d, dO, d, Dv, *grads)
e = (dV.double() - Vd.grad).abs()
print("max err", e.max().item(), "at", (e == e.max()).nonzero().tolist(), "mean err", e.mean().item())
print("per (b,key) max err:\n", e.view(B, Lb, d).amax(-1))
print("key mask:\n", mk1.int())
print("lse max plane for b=0,h=0:", lse[0, 0, 0, :8].tolist(), " inv:", lse[1, 0, 0, :8].tolist())
print("P colsum per key (b,h avg):", P.sum(2).mean(1))
eO = (O.view(B, Lq, d).double() - ref).abs()
print("fwd err max", eO.max().item())

def run(mq_, tag, Lq_=Lq):
    O = torch.empty(B * Lq_, d, device=DEV); lse = torch.empty(2, B, H_, Lq_, device=DEV)
    Q_ = Q[:, :Lq_].contiguous()
    blocks = (None, z(Q_), d, None, None, 0, z(K), z(V), d); masks = (mq_, None, mk1)
    H.attn_fwd(B, H_, dh, Lq_, La, Lb, *blocks, *masks, O, d, lse)
    Qd, Kd, Vd = [t.double().requires_grad_(True) for t in (Q_, K, V)]
    lg = torch.einsum("bqhd,bkhd->bhqk", sp(Qd), sp(Kd))
    lg = torch.where((mq_[:, :, None] & mk1[:, None, :])[:, None], lg, torch.full_like(lg, -10000.0)) / math.sqrt(dh)
    P = lg.softmax(-1)
    ref = torch.einsum("bhqk,bkhd->bqhd", P, sp(Vd)).reshape(B, Lq_, d)
    dO_ = dO.view(B, Lq, d)[:, :Lq_].contiguous().view(B * Lq_, d)
    ref.backward(dO_.view(B, Lq_, d).double())
    Dv = torch.empty(B, H_, Lq_, device=DEV)
    dQ, dK, dV = [torch.full_like(t, float("nan")) for t in (Q_, K, V)]
    grads = (None, z(dQ), d, None, None, 0, z(dK), z(dV), d)
    H.attn_bwd(B, H_, dh, Lq_, La, Lb, *blocks, *masks, lse, O, d, dO_, d, Dv, *grads)
    e = (dV.double() - Vd.grad)
    print(tag, "dV max err %.3e" % e.abs().max().item(), "masked queries per b:", (~mq_).sum(1).tolist(),
          "err/(0.1*sum_masked dO) sample:", (e.view(B, Lb, d)[:, 0, :4] / (0.1 * (dO_.view(B, Lq_, d).double() * (~mq_)[:, :, None]).sum(1)[:, :4] + 1e-30)).tolist())

run(torch.ones_like(mq), "all queries valid:")
run(mq, "orig:")
m2 = torch.ones_like(mq); m2[:, 5] = False
run(m2, "one masked query:")
run(mq[:, :32].contiguous(), "Lq=32 (no pad queries):", 32)
