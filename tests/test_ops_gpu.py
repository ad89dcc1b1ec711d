"""Kernel-level parity on the MI355X: every C-ABI op against a torch reference of the same op
(fp64 where the comparison needs head-room).  Run with ``pytest -m gpu``."""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _abi():
    from segmminterest_amd import hipabi
    hipabi.lib()
    return hipabi


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (360, 96, 48), (63, 32, 40), (1000, 768, 768), (20480, 768, 768), (7, 4, 4)])
def test_gemm_nt_bias(M, N, K):
    H = _abi()
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    C = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, bias=b)
    ref = (A.double() @ W.double().t() + b.double())
    err = (C.double() - ref).abs().max().item()
    assert err < 1e-5 * math.sqrt(K) * 4, err


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C/D register map (guide §3)."""
    H = _abi()
    n = 128
    A = torch.eye(n, device=DEV)
    Bm = (torch.arange(n * n, device=DEV, dtype=torch.float32).view(n, n) % 97) + torch.arange(n, device=DEV)[:, None] * 0.5
    C = torch.empty(n, n, device=DEV)
    H.gemm(H.LAYOUT_NN, n, n, n, A, n, Bm, n, C, n)
    assert torch.equal(C, Bm)
    H.gemm(H.LAYOUT_NT, n, n, n, A, n, Bm, n, C, n)
    assert torch.equal(C, Bm.t())
    H.gemm(H.LAYOUT_TN, n, n, n, Bm, n, A, n, C, n)      # C = Bm^T . I
    assert torch.equal(C, Bm.t())


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (360, 48, 96), (20480, 768, 3072)])
def test_gemm_nn(M, N, K):
    H = _abi()
    A, Bm = _rand(M, K, seed=4), _rand(K, N, seed=5)
    C = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NN, M, N, K, A, K, Bm, N, C, N)
    ref = A.double() @ Bm.double()
    assert (C.double() - ref).abs().max().item() < 1e-5 * math.sqrt(K) * 4


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 256, 1), (96, 48, 360, 1), (768, 768, 20480, 16), (32, 40, 63, 2), (3072, 768, 5120, 8)])
def test_gemm_tn_wgrad_splitk(M, N, K, splits):
    H = _abi()
    A, Bm = _rand(K, M, seed=6), _rand(K, N, seed=7)
    C = torch.empty(M, N, device=DEV)
    ws = torch.empty(max(splits, 1) * M * N, device=DEV)
    H.gemm(H.LAYOUT_TN, M, N, K, A, M, Bm, N, C, N, splits=splits, workspace=ws)
    ref = A.double().t() @ Bm.double()
    assert (C.double() - ref).abs().max().item() < 1e-5 * math.sqrt(K) * 4
    # accumulate into an existing gradient + run-to-run determinism
    C2 = C.clone()
    H.gemm(H.LAYOUT_TN, M, N, K, A, M, Bm, N, C2, N, splits=splits, workspace=ws, accumulate=True)
    assert (C2.double() - 2 * ref).abs().max().item() < 2e-5 * math.sqrt(K) * 4
    C3 = torch.empty_like(C)
    H.gemm(H.LAYOUT_TN, M, N, K, A, M, Bm, N, C3, N, splits=splits, workspace=ws)
    assert torch.equal(C, C3)


def test_gemm_epilogues():
    H = _abi()
    M, N, K = 360, 96, 64
    A, W, b = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.3), _rand(N, seed=10)
    rs = torch.rand(M, device=DEV) + 0.5
    res = _rand(40, N, seed=11)
    C = torch.empty(M, N, device=DEV)
    aux = torch.empty(M, N, device=DEV)
    # row_scale, bias, GELU (aux = pre-activation), periodic residual
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, bias=b, row_scale=rs, residual=res, ldr=N, res_period=40,
           activation=H.ACT_GELU, aux=aux, ldaux=N)
    pre = (A.double() @ W.double().t()) * rs.double()[:, None] + b.double()
    ref = torch.nn.functional.gelu(pre) + res.double().repeat(M // 40, 1)
    assert (aux.double() - pre).abs().max().item() < 1e-5
    assert (C.double() - ref).abs().max().item() < 1e-5
    # dGELU epilogue: C = (A.W^T) * gelu'(aux)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, activation=H.ACT_DGELU, aux=aux, ldaux=N)
    x = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref = (A.double() @ W.double().t()) * x.grad
    assert (C.double() - ref).abs().max().item() < 1e-5
    # residual aliasing the output (accumulate)
    C0 = _rand(M, N, seed=12)
    C = C0.clone()
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, accumulate=True)
    assert (C.double() - (C0.double() + A.double() @ W.double().t())).abs().max().item() < 1e-5


def test_gemm_strided_views():
    """Column slices of a fused projection buffer: lda/ldc larger than the logical width."""
    H = _abi()
    M, d = 200, 32
    X = _rand(M, d, seed=13)
    Wcat = _rand(4 * d, d, seed=14)
    Y = torch.zeros(M, 4 * d, device=DEV)
    H.gemm(H.LAYOUT_NT, M, 4 * d, d, X, d, Wcat, d, Y, 4 * d)
    assert (Y.double() - X.double() @ Wcat.double().t()).abs().max().item() < 1e-5
    # dgrad from the slice [:, d:2d] only
    dX = torch.empty(M, d, device=DEV)
    H.gemm(H.LAYOUT_NN, M, d, d, Y, 4 * d, Wcat, d, dX, d, a_off=d, b_off=d * d)
    ref = Y[:, d:2 * d].double() @ Wcat[d:2 * d].double()
    assert (dX.double() - ref).abs().max().item() < 1e-4


def test_gemm_dropout_epilogue_matches_mask_hook():
    H = _abi()
    M, N, K, p = 256, 64, 32, 0.25
    A, W = _rand(M, K, seed=15), _rand(N, K, seed=16)
    C = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, drop_p=p, seed=77, site=5)
    mult = torch.empty(M * N, device=DEV)
    H.dropout_mult(mult, M * N, p, 77, 5)
    ref = (A.double() @ W.double().t()) * mult.view(M, N).double()
    assert (C.double() - ref).abs().max().item() < 1e-5
    keep = (mult > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.02
    assert torch.allclose(mult[mult > 0], torch.tensor(1 / (1 - p), device=DEV))
    mult2 = torch.empty(M * N, device=DEV)
    H.dropout_mult(mult2, M * N, p, 77, 6)
    assert not torch.equal(mult, mult2)


def test_dropout_hash_rate_and_independence():
    """The stateless (seed, site, element) hash behind every dropout mask (common.h): keep rate within 4 sigma of the binomial
    for several p, and no structure a counter hash could leak -- masks of neighbouring elements, of elements one row (768)
    apart, of the same element under consecutive sites and under consecutive seeds are uncorrelated (|r| < 4 / sqrt(n)),
    and the byte-level population count of packed masks matches the binomial variance (no clumping)."""
    H = _abi()
    n = 1 << 22
    lim = 4.0 / n ** 0.5

    def mask(p, seed, site):
        m = torch.empty(n, device=DEV)
        H.dropout_mult(m, n, p, seed, site)
        return (m > 0).double()

    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).mean() / (a.std() * b.std() + 1e-30))

    for p in (0.1, 0.25, 0.5):
        k = mask(p, 1234, 7)
        sigma = (p * (1 - p) / n) ** 0.5
        assert abs(float(k.mean()) - (1 - p)) < 4 * sigma + 1.0 / 65536, (p, float(k.mean()))      # p is quantised to 1/65536
        assert abs(corr(k[:-1], k[1:])) < lim, ("neighbours", p)
        assert abs(corr(k[:-4], k[4:])) < lim, ("next hash group", p)
        assert abs(corr(k[:-768], k[768:])) < lim, ("row stride", p)
        assert abs(corr(k, mask(p, 1234, 8))) < lim, ("consecutive sites", p)
        assert abs(corr(k, mask(p, 1235, 7))) < lim, ("consecutive seeds", p)
        # population count of 64-element blocks: variance of a binomial(64, 1-p), within 2 %
        blocks = k.view(-1, 64).sum(1)
        var, ref = float(blocks.var()), 64 * p * (1 - p)
        assert abs(var - ref) < 0.02 * ref, (p, var, ref)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 1000, 1024, 4097, 8192, 8193, 12000, 16384, 16385, 40000, 65536])
def test_argsort_ids_is_torch_stable_argsort(n):
    """One workgroup up to 8192 ids (config 3 on 8 ranks sits exactly on the edge); beyond, the multi-workgroup network over a
    workspace (round 6: 8192 + 1, 16384, 16384 + 1, a non-power-of-two and 65536 ids)."""
    H = _abi()
    g = torch.Generator().manual_seed(n)
    ids = torch.randint(0, max(2, n // 3), (n,), generator=g)          # many duplicates: stability matters
    ids[::5] = torch.randint(0, 352495, (len(ids[::5]),), generator=g)
    got = H.argsort_ids(ids.to(DEV))
    ref = torch.argsort(ids, stable=True)
    assert got.dtype == torch.int32 and torch.equal(got.cpu().long(), ref)


def test_gemm_rejects_bad_shapes():
    H = _abi()
    A = torch.zeros(8, 6, device=DEV)
    with pytest.raises(RuntimeError):
        H.gemm(H.LAYOUT_NT, 8, 8, 6, A, 6, A, 6, torch.zeros(8, 8, device=DEV), 8)


# ------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("rows,D", [(37, 48), (20480, 768), (5, 1024)])
def test_l1norm(rows, D):
    H = _abi()
    x = torch.rand(rows, D, device=DEV)
    x[0] = 0
    y = torch.empty_like(x)
    inv = torch.empty(rows, device=DEV)
    H.l1norm(x, y, inv)
    ref = x / (x.norm(p=1, dim=-1, keepdim=True) + 1e-6)
    assert torch.allclose(y, ref, rtol=2e-6, atol=1e-9)
    assert torch.allclose(inv, 1 / (x.abs().sum(-1) + 1e-6), rtol=2e-6)


@pytest.mark.parametrize("rows,d", [(77, 32), (20480, 768), (9, 2048), (301, 512), (130, 1024), (50, 260)])
def test_layernorm_fwd_bwd(rows, d):
    H = _abi()
    x = _rand(rows, d, seed=20) * 2 + 0.3
    gamma, beta = 1 + 0.1 * _rand(d, seed=21), 0.1 * _rand(d, seed=22)
    y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    H.layernorm_fwd(x, gamma, beta, y, mean, rstd)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-12)
    assert (y.double() - ref).abs().max().item() < 5e-6
    dy = _rand(rows, d, seed=23)
    ref.backward(dy.double())
    parts = H.layernorm_bwd_parts(rows, d)
    dx = torch.empty_like(x)
    pg, pb = torch.empty(parts, d, device=DEV), torch.empty(parts, d, device=DEV)
    H.layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, pg, pb)
    assert (dx.double() - xr.grad).abs().max().item() < 2e-5
    ws = torch.empty(H.colsum_chunks(parts) * d, device=DEV)
    dg, db = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    H.colsum(pg, d, parts, d, dg, ws)
    H.colsum(pb, d, parts, d, db, ws)
    assert (dg.double() - gr.grad).abs().max().item() < 1e-4 * max(1.0, math.sqrt(rows) / 10)
    assert (db.double() - br.grad).abs().max().item() < 1e-4 * max(1.0, math.sqrt(rows) / 10)


def test_layernorm_dropout_roundtrip():
    H = _abi()
    rows, d, p = 64, 128, 0.1
    x = _rand(rows, d, seed=24)
    gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    H.layernorm_fwd(x, gamma, beta, y, mean, rstd, drop_p=p, seed=3, site=9)
    mult = torch.empty(rows * d, device=DEV)
    H.dropout_mult(mult, rows * d, p, 3, 9)
    ref = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-12) * mult.view(rows, d)
    assert torch.allclose(y, ref, atol=1e-5)
    # backward: dy masked by the same stream; dx_drop masked by the branch stream
    dy = _rand(rows, d, seed=25)
    parts = H.layernorm_bwd_parts(rows, d)
    dx, dxd = torch.empty_like(x), torch.empty_like(x)
    pg, pb = torch.empty(parts, d, device=DEV), torch.empty(parts, d, device=DEV)
    H.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dxd, pg, pb, drop_y_p=p, drop_y_site=9, drop_b_p=p, drop_b_site=4, seed=3)
    xr = x.double().requires_grad_(True)
    (torch.nn.functional.layer_norm(xr, (d,), gamma.double(), beta.double(), 1e-12) * mult.view(rows, d).double() * dy.double()).sum().backward()
    assert (dx.double() - xr.grad).abs().max().item() < 2e-5
    mult2 = torch.empty(rows * d, device=DEV)
    H.dropout_mult(mult2, rows * d, p, 3, 4)
    assert torch.allclose(dxd, dx * mult2.view(rows, d), atol=1e-6)


@pytest.mark.parametrize("M,N", [(20480, 768), (37, 40), (5120, 3072)])
def test_colsum(M, N):
    H = _abi()
    X = _rand(M, N, seed=26)
    w = _rand(M, seed=27)
    ws = torch.empty(H.colsum_chunks(M) * N, device=DEV)
    out = torch.empty(N, device=DEV)
    H.colsum(X, N, M, N, out, ws)
    assert (out.double() - X.double().sum(0)).abs().max().item() < 1e-4 * math.sqrt(M)
    H.colsum(X, N, M, N, out, ws, w=w, accumulate=True)
    ref = X.double().sum(0) + (X.double() * w.double()[:, None]).sum(0)
    assert (out.double() - ref).abs().max().item() < 2e-4 * math.sqrt(M)


def test_rowdot_head():
    H = _abi()
    rows, d = 20480, 768
    x, w, b = _rand(rows, d, seed=28), _rand(d, seed=29, scale=0.05), _rand(1, seed=30)
    out = torch.empty(rows, device=DEV)
    H.rowdot(x, d, w, b, out, rows, d)
    assert (out.double() - (x.double() @ w.double() + b.double())).abs().max().item() < 1e-4
    g = _rand(rows, seed=31)
    dx = torch.empty_like(x)
    H.rowscale_bcast(g, w, dx, d, rows, d)
    assert torch.allclose(dx, g[:, None] * w[None, :], rtol=1e-6, atol=1e-7)
    s = torch.zeros(1, device=DEV)
    H.vecsum(g, rows, s)
    assert abs(s.item() - g.double().sum().item()) < 1e-2


# ------------------------------------------------------------------ attention
def _attn_ref(Qa, Qb, Ka, Va, Kb, Vb, mq, mka, mkb, H_, mult=None):
    """fp64 torch restatement of encoder.py:44-73,138-161 for one side."""
    B, Lq, d = Qa.shape
    dh = d // H_
    sp = lambda t: t.view(B, t.shape[1], H_, dh)
    la = torch.einsum("bqhd,bkhd->bhqk", sp(Qa), sp(Ka))
    lb = torch.einsum("bqhd,bkhd->bhqk", sp(Qb), sp(Kb))
    la = torch.where((mq[:, :, None] & mka[:, None, :])[:, None], la, torch.full_like(la, -10000.0))
    lb = torch.where((mq[:, :, None] & mkb[:, None, :])[:, None], lb, torch.full_like(lb, -10000.0))
    lg = torch.cat([la, lb], -1)
    if mult is not None:
        lg = lg * mult
    lg = lg / math.sqrt(dh)
    P = lg.softmax(-1)
    V = torch.cat([sp(Va), sp(Vb)], 1)
    return torch.einsum("bhqk,bkhd->bqhd", P, V).reshape(B, Lq, d)


@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb", [(3, 4, 8, 40, 40, 10), (2, 16, 4, 40, 40, 100), (2, 2, 48, 40, 40, 100),
                                              (2, 4, 8, 7, 40, 7), (2, 16, 48, 100, 40, 100), (3, 4, 8, 1, 40, 1),
                                              (2, 2, 32, 20, 20, 10), (1, 2, 64, 40, 40, 10), (2, 2, 16, 40, 40, 10)])
def test_attention_fwd_bwd(B, H_, dh, Lq, La, Lb):
    H = _abi()
    d = H_ * dh
    g = torch.Generator().manual_seed(B * 1000 + Lq)
    mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(DEV)
    Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    mq[0, 0] = False
    mq[-1, -1] = True
    O = torch.empty(B * Lq, d, device=DEV)
    lse = torch.empty(2, B, H_, Lq, device=DEV)
    z = lambda t: (t, 0)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, O, d, lse)
    leaves = [t.double().requires_grad_(True) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
    ref = _attn_ref(*leaves, mq, mka, mkb, H_)
    assert (O.view(B, Lq, d).double() - ref).abs().max().item() < 2e-5
    dO = (torch.randn(B * Lq, d, generator=g)).to(DEV)
    ref.backward(dO.view(B, Lq, d).double())
    Dv = torch.empty(B, H_, Lq, device=DEV)
    outs = [torch.full_like(t, float("nan")) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
    H.attn_bwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, lse, O, d, dO, d, Dv,
               z(outs[0]), z(outs[1]), d, z(outs[2]), z(outs[3]), d, z(outs[4]), z(outs[5]), d)
    for name, got, leaf in zip(("dQa", "dQb", "dKa", "dVa", "dKb", "dVb"), outs, leaves):
        err = (got.double() - leaf.grad).abs().max().item()
        assert err < 5e-5, (name, err)


@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb,p", [(4, 16, 48, 40, 40, 100, 0.1), (3, 4, 16, 40, 40, 10, 0.0), (2, 16, 48, 100, 40, 100, 0.1),
                                                (5, 8, 32, 20, 20, 1, 0.1), (2, 2, 64, 40, 40, 10, 0.0), (3, 4, 48, 40, 0, 100, 0.1),
                                                (3, 4, 48, 40, 40, 0, 0.1), (2, 4, 32, 7, 40, 7, 0.0)])
def test_attention_fwd_lds_staged_form_equals_direct_form(B, H_, dh, Lq, La, Lb, p):
    """The LDS-DMA staged forward (one workgroup per head, K / V staged once; the default where it fits) against the direct-load
    form (knob ATT_FWD_LDS = 0): same arithmetic, masks and dropout stream.  With ONE key group per query tile
    (knob ATT_FWD_KSPLIT = 1) O and the softmax statistics are BIT-IDENTICAL; with the default key split the groups' softmax sums
    are merged in another order: equal to 2e-6 of the maximum.  Operands are column slices of fused projection buffers (the engine's layout), with NaN poison around them: a staging bug
    that reads a neighbouring column or row shows up at once."""
    H = _abi()
    d = H_ * dh
    g = torch.Generator().manual_seed(B * 77 + Lq)
    Yv = (torch.randn(B * max(La, 1), 4 * d, generator=g) * 0.7).to(DEV)
    Yu = (torch.randn(B * max(Lb, 1), 2 * d, generator=g) * 0.7).to(DEV)
    Qs = Yv if Lq == La else (torch.randn(B * Lq, 4 * d, generator=g) * 0.7).to(DEV)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mka = (torch.rand(B, max(La, 1), generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, max(Lb, 1), generator=g) < 0.7).to(DEV)
    outs = {}
    for form in ("0", "1", "k1"):          # direct | staged with the key tiles split over wave groups | staged, one group
        prev_lds = H.config_set("ATT_FWD_LDS", 0 if form == "0" else 2)
        prev_ksp = H.config_set("ATT_FWD_KSPLIT", 1 if form == "k1" else 2)
        try:
            O = torch.full((B * Lq, d), float("nan"), device=DEV)
            lse = torch.full((2, B, H_, Lq), float("nan"), device=DEV)
            am = torch.zeros(H.AMAX_SLOTS, device=DEV)
            H.attn_fwd(B, H_, dh, Lq, La, Lb, (Qs, 0), (Qs, d), 4 * d, (Yv, 2 * d) if La else None, (Yv, 3 * d) if La else None, 4 * d,
                       (Yu, 0) if Lb else None, (Yu, d) if Lb else None, 2 * d, mq, mka[:, :La] if La else None, mkb[:, :Lb] if Lb else None,
                       O, d, lse, drop_p=p, seed=11, site=3, amax_o=am)
            outs[form] = (O, lse, am)
        finally:
            H.config_set("ATT_FWD_LDS", prev_lds)
            H.config_set("ATT_FWD_KSPLIT", prev_ksp)
    assert torch.isfinite(outs["1"][0]).all() and torch.isfinite(outs["1"][1]).all()
    assert torch.equal(outs["0"][0], outs["k1"][0]) and torch.equal(outs["0"][1], outs["k1"][1])
    assert float(outs["0"][2].max()) == float(outs["k1"][2].max()) == float(outs["k1"][0].abs().max())      # (slot positions differ)
    omax = float(outs["0"][0].abs().max())
    assert float((outs["0"][0] - outs["1"][0]).abs().max()) <= 2e-6 * omax
    assert float((outs["0"][1][0] - outs["1"][1][0]).abs().max()) <= 1e-6 * float(outs["0"][1][0].abs().max())                       # row maxima
    assert float(((outs["0"][1][1] - outs["1"][1][1]) / outs["0"][1][1]).abs().max()) <= 2e-6                                       # 1 / row sums
    assert float(outs["1"][2].max()) == float(outs["1"][0].abs().max())


def test_attention_dropout_consistency():
    """Train-mode logits dropout: forward equals the reference with the kernel's own mask; the
    backward regenerates the same mask (checked against autograd through that mask)."""
    H = _abi()
    B, H_, dh, Lq, La, Lb, p = 2, 4, 8, 40, 40, 10, 0.1
    d = H_ * dh
    g = torch.Generator().manual_seed(5)
    mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(DEV)
    Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
    mq = torch.ones(B, Lq, dtype=torch.bool, device=DEV)
    mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV)
    mkb = torch.ones(B, Lb, dtype=torch.bool, device=DEV)
    O = torch.empty(B * Lq, d, device=DEV)
    lse = torch.empty(2, B, H_, Lq, device=DEV)
    z = lambda t: (t, 0)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, O, d, lse,
               drop_p=p, seed=11, site=3)
    La_p, Lb_p = (La + 15) // 16 * 16, (Lb + 15) // 16 * 16
    Tp = La_p + Lb_p
    mult = torch.empty(B * H_ * Lq * Tp, device=DEV)
    H.dropout_mult(mult, mult.numel(), p, 11, 3)
    mult = mult.view(B, H_, Lq, Tp)
    mult = torch.cat([mult[..., :La], mult[..., La_p:La_p + Lb]], -1).double()
    leaves = [t.double().requires_grad_(True) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
    ref = _attn_ref(*leaves, mq, mka, mkb, H_, mult=mult)
    assert (O.view(B, Lq, d).double() - ref).abs().max().item() < 2e-5
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)
    ref.backward(dO.view(B, Lq, d).double())
    Dv = torch.empty(B, H_, Lq, device=DEV)
    outs = [torch.empty_like(t) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
    H.attn_bwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, lse, O, d, dO, d, Dv,
               z(outs[0]), z(outs[1]), d, z(outs[2]), z(outs[3]), d, z(outs[4]), z(outs[5]), d, drop_p=p, seed=11, site=3)
    for got, leaf in zip(outs, leaves):
        assert (got.double() - leaf.grad).abs().max().item() < 5e-5


# ------------------------------------------------------------------ optimiser
def test_adamw_matches_torch():
    H = _abi()
    n = 100003
    p0 = _rand(n, seed=40)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-4)
    p = p0.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = _rand(n, seed=40 + step) * (0.1 if step == 2 else 1.0)
        ref.grad = gr.clone()
        opt.step()
        H.adamw(p, gr, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step)
    assert torch.allclose(p, ref.detach(), rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ id embedding
def test_embed_id():
    H = _abi()
    B, S, d, n_items, n_users = 6, 40, 32, 50, 20
    dh = d // 2
    table, utable = _rand(n_items + 1, dh, seed=50), _rand(n_users + 1, d, seed=51)
    fw, fb = _rand(dh, seed=52), _rand(dh, seed=53)
    vpe, upe = _rand(S, d, seed=54), _rand(1, d, seed=55)
    ids = torch.tensor([3, 7, 3, 50, 1, 7], device=DEV)
    uids = torch.tensor([1, 2, 1, 20, 5, 5], device=DEV)
    out = torch.empty(B * S, d, device=DEV)
    H.embed_id_vid(ids, table, dh, fw, fb, vpe, out, B, S)
    pos = torch.arange(S, device=DEV, dtype=torch.float32)
    ref = torch.cat([table[ids][:, None, :].expand(B, S, dh), (pos[:, None] * fw[None] + fb[None])[None].expand(B, S, dh)], -1) + vpe[None]
    assert torch.allclose(out.view(B, S, d), ref, atol=1e-6)
    uo = torch.empty(B, d, device=DEV)
    H.embed_id_usr(uids, utable, d, upe, uo, B)
    assert torch.allclose(uo, utable[uids] + upe, atol=1e-6)
    # backward: dense table grads
    dpre = _rand(B * S, d, seed=56)
    order = torch.argsort(ids, stable=True).to(torch.int32)
    dtab = torch.zeros_like(table)
    H.embed_id_bwd(dpre, S, d, 0, dh, order, ids, dtab, B)
    ref = torch.zeros_like(table)
    ref.index_add_(0, ids, dpre.view(B, S, d)[:, :, :dh].sum(1))
    assert torch.allclose(dtab, ref, atol=1e-5)
    dpe = torch.empty(S, d, device=DEV)
    H.pe_grad(dpre, d, B, S, d, dpe)
    assert torch.allclose(dpe, dpre.view(B, S, d).sum(0), atol=1e-5)
    # ids outside the table (torch.nn.Embedding raises): no memory is touched, the forward rows are NaN, the
    # backward skips them
    bad = torch.tensor([3, 10 ** 9, -1, 51, 1, 7], device=DEV)
    H.embed_id_vid(bad, table, dh, fw, fb, vpe, out, B, S)
    o = out.view(B, S, d)
    assert torch.isnan(o[[1, 2, 3], :, :dh]).all() and torch.isfinite(o[[0, 4, 5]]).all() and torch.isfinite(o[:, :, dh:]).all()
    ubad = torch.tensor([1, 21, 1, -5, 5, 5], device=DEV)
    H.embed_id_usr(ubad, utable, d, upe, uo, B)
    assert torch.isnan(uo[[1, 3]]).all() and torch.isfinite(uo[[0, 2, 4, 5]]).all()
    dtab.zero_()
    H.embed_id_bwd(dpre, S, d, 0, dh, torch.argsort(bad, stable=True).to(torch.int32), bad, dtab, B)
    ok = torch.tensor([0, 4, 5], device=DEV)
    ref = torch.zeros_like(table)
    ref.index_add_(0, bad[ok], dpre.view(B, S, d)[ok][:, :, :dh].sum(1))
    assert torch.allclose(dtab, ref, atol=1e-5)


# ------------------------------------------------------------------ bf16x6 split-MFMA engine
@pytest.mark.parametrize("lay,M,N,K,splits", [("NT", 360, 96, 48, 1), ("NT", 20480, 768, 768, 1), ("NT", 63, 32, 40, 1),
                                              ("NN", 360, 48, 96, 1), ("NN", 20480, 768, 3072, 1),
                                              ("TN", 96, 48, 360, 1), ("TN", 768, 768, 20480, 16), ("TN", 32, 40, 63, 2)])
def test_gemm_bf16x6_matches_fp64_like_fp32(lay, M, N, K, splits):
    """engine 1: exact 3-way bf16 split, six partial products, fp32 accumulation -- must be at least as close
    to the fp64 product as the f32-MFMA engine (same tolerance), on every operand layout."""
    H = _abi()
    L = {"NT": 0, "NN": 1, "TN": 2}[lay]
    g = torch.Generator().manual_seed(M + N + K)
    rnd = lambda *s: (torch.randn(*s, generator=g) * torch.exp(torch.randn(*s, generator=g))).to(DEV)   # wide dynamic range
    if lay == "NT":
        A, Bm = rnd(M, K), rnd(N, K); lda, ldb = K, K; ref = A.double() @ Bm.double().t()
    elif lay == "NN":
        A, Bm = rnd(M, K), rnd(K, N); lda, ldb = K, N; ref = A.double() @ Bm.double()
    else:
        A, Bm = rnd(K, M), rnd(K, N); lda, ldb = M, N; ref = A.double().t() @ Bm.double()
    ws = torch.empty(max(splits, 1) * M * N, device=DEV)
    outs = {}
    for eng in (H.ENGINE_F32, H.ENGINE_BF16X6):
        C = torch.full((M, N), float("nan"), device=DEV)
        H.gemm(L, M, N, K, A, lda, Bm, ldb, C, N, splits=splits, workspace=ws, engine=eng)
        outs[eng] = C
    scale = ref.abs().mean().item()
    e32 = (outs[H.ENGINE_F32].double() - ref).abs().max().item() / scale
    e6 = (outs[H.ENGINE_BF16X6].double() - ref).abs().max().item() / scale
    assert e6 < 2e-5 * math.sqrt(K), (e6, e32)
    assert e6 <= 2.0 * e32 + 1e-7, ("bf16x6 less accurate than fp32 MFMA", e6, e32)


def test_gemm_bf16x6_identity_exact_and_epilogue():
    H = _abi()
    n = 128
    A = torch.eye(n, device=DEV)
    Bm = (torch.randn(n, n, device=DEV) * 3.7)
    C = torch.empty(n, n, device=DEV)
    H.gemm(H.LAYOUT_NN, n, n, n, A, n, Bm, n, C, n, engine=H.ENGINE_BF16X6)
    assert torch.equal(C, Bm)                      # hi + mid + lo reassembles every fp32 value exactly
    H.gemm(H.LAYOUT_TN, n, n, n, Bm, n, A, n, C, n, engine=H.ENGINE_BF16X6)
    assert torch.equal(C, Bm.t())
    M, N, K = 360, 96, 64
    X, W, b = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.3), _rand(N, seed=10)
    res = _rand(40, N, seed=11)
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, X, K, W, K, out, N, bias=b, residual=res, ldr=N, res_period=40, activation=H.ACT_GELU,
           aux=aux, ldaux=N, engine=H.ENGINE_BF16X6)
    pre = X.double() @ W.double().t() + b.double()
    assert (aux.double() - pre).abs().max().item() < 1e-5
    assert (out.double() - (torch.nn.functional.gelu(pre) + res.double().repeat(M // 40, 1))).abs().max().item() < 1e-5


# ------------------------------------------------------------------ fp16x3 split-MFMA engine (scaled two-term fp16 split)
@pytest.mark.parametrize("lay,M,N,K,splits", [("NT", 360, 96, 48, 1), ("NT", 20480, 768, 768, 1), ("NT", 63, 32, 40, 1),
                                              ("NN", 360, 48, 96, 1), ("NN", 20480, 768, 3072, 1),
                                              ("TN", 96, 48, 360, 1), ("TN", 768, 768, 20480, 16), ("TN", 32, 40, 63, 2)])
@pytest.mark.parametrize("kind", ["wide", "tiny", "huge"])
def test_gemm_f16x3_matches_fp64_like_fp32(lay, M, N, K, splits, kind):
    """engine 2: x*s = hi + lo in fp16 (per-tensor power-of-two scale from max|x|), three partial products, fp32
    accumulation.  Must be as close to the fp64 product as the f32-MFMA engine on every layout, for operands with a
    wide dynamic range and for magnitudes far outside the fp16 range (gradients ~1e-9, activations ~1e6)."""
    H = _abi()
    L = {"NT": 0, "NN": 1, "TN": 2}[lay]
    g = torch.Generator().manual_seed(M + N + K)
    mag = {"wide": 1.0, "tiny": 1e-9, "huge": 1e6}[kind]
    rnd = lambda *s: (torch.randn(*s, generator=g) * torch.exp(torch.randn(*s, generator=g)) * mag).to(DEV)
    if lay == "NT":
        A, Bm = rnd(M, K), rnd(N, K); lda, ldb = K, K; ref = A.double() @ Bm.double().t()
    elif lay == "NN":
        A, Bm = rnd(M, K), rnd(K, N); lda, ldb = K, N; ref = A.double() @ Bm.double()
    else:
        A, Bm = rnd(K, M), rnd(K, N); lda, ldb = M, N; ref = A.double().t() @ Bm.double()
    ws = torch.empty(max(splits, 1) * M * N, device=DEV)
    outs = {}
    for eng in (H.ENGINE_F32, H.ENGINE_F16X3):
        C = torch.full((M, N), float("nan"), device=DEV)
        H.gemm(L, M, N, K, A, lda, Bm, ldb, C, N, splits=splits, workspace=ws, engine=eng)
        outs[eng] = C
    scale = ref.abs().mean().item()
    e32 = (outs[H.ENGINE_F32].double() - ref).abs()
    e3 = (outs[H.ENGINE_F16X3].double() - ref).abs()
    assert e3.max().item() / scale < 2e-5 * math.sqrt(K), (e3.max().item() / scale, e32.max().item() / scale)
    assert e3.max().item() <= 2.0 * e32.max().item() + 1e-7 * scale, ("fp16x3 max error above the fp32 MFMA's", e3.max().item() / scale, e32.max().item() / scale)
    assert e3.mean().item() <= 1.25 * e32.mean().item() + 1e-9 * scale, ("fp16x3 mean error above the fp32 MFMA's", e3.mean().item() / scale, e32.mean().item() / scale)


def test_gemm_f16x3_small_elements_next_to_large_ones():
    """One scale per tensor: elements 2^-30 below the maximum must still contribute with an error far below
    their own size (fp16 subnormals keep an ABSOLUTE error of 2^-40 max)."""
    H = _abi()
    M, N, K = 128, 128, 64
    A = torch.zeros(M, K, device=DEV)
    A[:, 0] = 1.0e3
    A[:, 1:] = torch.rand(M, K - 1, device=DEV) * 1e-5 + 1e-6        # ~2^-27 .. 2^-30 of the maximum
    Bm = torch.zeros(N, K, device=DEV)
    Bm[:, 1:] = torch.rand(N, K - 1, device=DEV) + 0.5                # column 0 of B is zero: only the small terms count
    C = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, Bm, K, C, N, engine=H.ENGINE_F16X3)
    ref = A.double() @ Bm.double().t()
    assert ((C.double() - ref).abs() / ref.abs()).max().item() < 1e-4      # small terms: >= 13 correct bits each
    A[:, 0] = 0.0
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, Bm, K, C, N, engine=H.ENGINE_F16X3)  # same data rescaled by its own maximum
    assert ((C.double() - ref).abs() / ref.abs()).max().item() < 2e-6


def test_gemm_f16x3_planes_epilogue_and_zero():
    H = _abi()
    M, N, K = 360, 96, 64
    X, W, b = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.3), _rand(N, seed=10)
    res = _rand(40, N, seed=11)
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    wam = H.absmax(W, N, K, K)
    assert wam.numel() == H.AMAX_PARTS and abs(wam.max().item() - W.abs().max().item()) == 0.0
    planes = torch.empty(2, N * K, dtype=torch.float16, device=DEV)
    H.split2h(W, planes, N * K, wam)
    s = 2.0 ** (14 - math.floor(math.log2(W.abs().max().item())))
    assert torch.equal(planes[0].view(N, K), (W * s).half())
    assert (planes[0].double() + planes[1].double() - (W * s).double().view(-1)).abs().max().item() <= 2.0 ** -21 * (W * s).abs().max().item()
    cam = torch.zeros(H.AMAX_SLOTS, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, X, K, None, K, out, N, bias=b, residual=res, ldr=N, res_period=40, activation=H.ACT_GELU,
           aux=aux, ldaux=N, engine=H.ENGINE_F16X3, b_planes=(planes, 0), b_amax=wam, c_amax=cam)
    pre = X.double() @ W.double().t() + b.double()
    assert (aux.double() - pre).abs().max().item() < 1e-5
    assert (out.double() - (torch.nn.functional.gelu(pre) + res.double().repeat(M // 40, 1))).abs().max().item() < 1e-5
    assert cam.max().item() == out.abs().max().item()              # fused partial maxima of |C|
    out2 = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, X, K, W, K, out2, N, bias=b, residual=res, ldr=N, res_period=40, activation=H.ACT_GELU,
           aux=aux, ldaux=N, engine=H.ENGINE_F16X3)
    assert torch.equal(out, out2)                                   # pre-split planes == split on the fly
    # W^T planes: dX = dY . W in the NT form
    dY = _rand(M, N, seed=12)
    wT = torch.empty(2, N * K, dtype=torch.float16, device=DEV)
    H.split2h_transpose(W, N, K, K, wT, wam)
    dX = torch.empty(M, K, device=DEV)
    H.gemm(H.LAYOUT_NT, M, K, N, dY, N, None, N, dX, K, engine=H.ENGINE_F16X3, b_planes=(wT, 0), b_amax=wam)
    assert (dX.double() - dY.double() @ W.double()).abs().max().item() < 2e-5
    # all-zero operand: scale falls back to 1, result exactly zero
    Z = torch.zeros(M, K, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, Z, K, W, K, out2, N, engine=H.ENGINE_F16X3)
    assert torch.equal(out2, torch.zeros_like(out2))


def test_fused_amax_producers():
    """LayerNorm forward/backward and attention forward/backward fold max|output| into a zeroed slot array."""
    H = _abi()
    rows, d = 333, 96
    x = _rand(rows, d, seed=30) * 3
    gamma, beta = 1 + 0.1 * _rand(d, seed=31), 0.1 * _rand(d, seed=32)
    y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    am = torch.zeros(H.AMAX_SLOTS, device=DEV)
    H.layernorm_fwd(x, gamma, beta, y, mean, rstd, drop_p=0.1, seed=5, site=3, amax=am)
    assert am.max().item() == y.abs().max().item()
    dy = _rand(rows, d, seed=33) * 1e-6
    parts = H.layernorm_bwd_parts(rows, d)
    dx, dxd = torch.empty_like(x), torch.empty_like(x)
    pg, pb = torch.empty(parts, d, device=DEV), torch.empty(parts, d, device=DEV)
    am.zero_()
    ps = torch.empty(parts, d, device=DEV)
    H.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dxd, pg, pb, drop_b_p=0.1, drop_b_site=4, seed=5, amax=am, part_dsum=ps)
    assert am.max().item() == dxd.abs().max().item()
    # fused column sums of the forwarded gradient (= bias gradient of the Linear in front of the residual add)
    assert (ps.double().sum(0) - dxd.double().sum(0)).abs().max().item() < 1e-5 * dxd.abs().sum(0).max().item() + 1e-12
    am.zero_()
    H.layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, pg, pb, amax=am)
    assert am.max().item() == dx.abs().max().item()
    B, H_, dh, Lq, La, Lb = 2, 4, 8, 40, 40, 10
    dm = H_ * dh
    g = torch.Generator().manual_seed(7)
    mk = lambda L: (torch.randn(B, L, dm, generator=g) * 0.7).to(DEV)
    Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    O = torch.empty(B * Lq, dm, device=DEV)
    lse = torch.empty(2, B, H_, Lq, device=DEV)
    z = lambda t: (t, 0)
    am.zero_()
    H.attn_fwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), dm, z(Ka), z(Va), dm, z(Kb), z(Vb), dm, mq, mka, mkb, O, dm, lse, amax_o=am)
    assert am.max().item() == O.abs().max().item()
    dO = torch.randn(B * Lq, dm, generator=g).to(DEV)
    Dv = torch.empty(B, H_, Lq, device=DEV)
    outs = [torch.empty_like(t) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
    aq, aka, akb = (torch.zeros(H.AMAX_SLOTS, device=DEV) for _ in range(3))
    H.attn_bwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), dm, z(Ka), z(Va), dm, z(Kb), z(Vb), dm, mq, mka, mkb, lse, O, dm, dO, dm, Dv,
               z(outs[0]), z(outs[1]), dm, z(outs[2]), z(outs[3]), dm, z(outs[4]), z(outs[5]), dm, amax_q=aq, amax_ka=aka, amax_kb=akb)
    assert aq.max().item() == max(outs[0].abs().max().item(), outs[1].abs().max().item())
    assert aka.max().item() == max(outs[2].abs().max().item(), outs[3].abs().max().item())
    assert akb.max().item() == max(outs[4].abs().max().item(), outs[5].abs().max().item())


# ------------------------------------------------------------------ ablation variants (encoder.py:108-135,392-400,428-429,503-511)
@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb", [(3, 4, 8, 40, 0, 10), (2, 16, 48, 40, 0, 100), (2, 4, 8, 40, 40, 0), (2, 16, 48, 100, 40, 0),
                                              (3, 4, 8, 1, 40, 0), (2, 2, 16, 40, 0, 1)])
def test_attention_one_empty_key_block(B, H_, dh, Lq, La, Lb):
    """CrossAtt / SelfAtt: the query attends to ONE key block; the other has length 0 and null pointers."""
    H = _abi()
    d = H_ * dh
    g = torch.Generator().manual_seed(B * 1000 + Lq + La)
    mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(DEV)
    L1 = La or Lb
    Q, K, V = mk(Lq), mk(L1), mk(L1)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mk1 = (torch.rand(B, L1, generator=g) < 0.75).to(DEV)
    mq[0, 0] = False
    O = torch.empty(B * Lq, d, device=DEV)
    lse = torch.empty(2, B, H_, Lq, device=DEV)
    z = lambda t: (t, 0)
    if La:
        blocks = (z(Q), None, d, z(K), z(V), d, None, None, 0)
        masks = (mq, mk1, None)
    else:
        blocks = (None, z(Q), d, None, None, 0, z(K), z(V), d)
        masks = (mq, None, mk1)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, *blocks, *masks, O, d, lse)
    Qd, Kd, Vd = [t.double().requires_grad_(True) for t in (Q, K, V)]
    sp = lambda t: t.view(B, t.shape[1], H_, dh)
    lg = torch.einsum("bqhd,bkhd->bhqk", sp(Qd), sp(Kd))
    lg = torch.where((mq[:, :, None] & mk1[:, None, :])[:, None], lg, torch.full_like(lg, -10000.0)) / math.sqrt(dh)
    ref = torch.einsum("bhqk,bkhd->bqhd", lg.softmax(-1), sp(Vd)).reshape(B, Lq, d)
    assert (O.view(B, Lq, d).double() - ref).abs().max().item() < 2e-5
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)
    ref.backward(dO.view(B, Lq, d).double())
    Dv = torch.empty(B, H_, Lq, device=DEV)
    dQ, dK, dV = [torch.full_like(t, float("nan")) for t in (Q, K, V)]
    if La:
        grads = (z(dQ), None, d, z(dK), z(dV), d, None, None, 0)
    else:
        grads = (None, z(dQ), d, None, None, 0, z(dK), z(dV), d)
    H.attn_bwd(B, H_, dh, Lq, La, Lb, *blocks, *masks, lse, O, d, dO, d, Dv, *grads)
    for name, got, leaf in (("dQ", dQ, Qd), ("dK", dK, Kd), ("dV", dV, Vd)):
        err = (got.double() - leaf.grad).abs().max().item()
        assert err < 2e-5 * max(1.0, leaf.grad.abs().max().item()), (name, err, leaf.grad.abs().max().item())
    with pytest.raises(RuntimeError):          # both blocks empty
        H.attn_fwd(B, H_, dh, Lq, 0, 0, *blocks, *masks, O, d, lse)


def test_gemm_relu_epilogues():
    H = _abi()
    M, N, K = 360, 96, 64
    A, W, b = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.3), _rand(N, seed=10)
    C = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, C, N, bias=b, activation=H.ACT_RELU)
    pre = A.double() @ W.double().t() + b.double()
    assert (C.double() - pre.clamp_min(0)).abs().max().item() < 1e-5
    # ReLU + dropout, then the backward epilogue reading the forward OUTPUT: zero where it is <= 0, same dropout mask
    Hd = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Hd, N, bias=b, activation=H.ACT_RELU, drop_p=0.3, seed=77, site=5)
    mult = torch.empty(M * N, device=DEV)
    H.dropout_mult(mult, M * N, 0.3, 77, 5)
    mult = mult.view(M, N)
    assert (Hd.double() - pre.clamp_min(0) * mult.double()).abs().max().item() < 1e-5
    G, W2 = _rand(M, K, seed=12), _rand(N, K, seed=13, scale=0.3)
    dZ = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, G, K, W2, K, dZ, N, activation=H.ACT_DRELU, aux=Hd, ldaux=N, drop_p=0.3, seed=77, site=5)
    ref = (G.double() @ W2.double().t()) * (pre > 0).double() * mult.double()
    assert (dZ.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,Lu,Lv,d,bins", [(3, 10, 40, 32, 40), (2, 100, 40, 768, 40), (2, 1, 40, 64, 40), (2, 7, 20, 32, 40)])
def test_pool_tokens(B, Lu, Lv, d, bins):
    H = _abi()
    U, V = _rand(B, Lu, d, seed=1), _rand(B, Lv, d, seed=2)
    out = torch.empty(B, bins, d, device=DEV)
    H.pool_tokens(U, Lu, V, Lv, out, B, d, bins)
    x = torch.cat((U, V), 1).double().requires_grad_(True)
    ref = torch.nn.functional.adaptive_avg_pool1d(x.permute(0, 2, 1), bins).permute(0, 2, 1)
    assert (out.double() - ref).abs().max().item() < 1e-6
    dOut = _rand(B, bins, d, seed=3)
    ref.backward(dOut.double())
    dU, dV = torch.full_like(U, float("nan")), torch.full_like(V, float("nan"))
    H.pool_tokens_bwd(dOut, dU, Lu, dV, Lv, B, d, bins)
    assert (torch.cat((dU, dV), 1).double() - x.grad).abs().max().item() < 1e-5


def test_embed_id_shuffled_positions():
    """'noPos': frame_pos[b, s] replaces s in the frame-index Linear."""
    H = _abi()
    B, S, d, n_items = 5, 40, 32, 30
    dh = d // 2
    table, fw, fb, vpe = _rand(n_items + 1, dh, seed=50), _rand(dh, seed=52), _rand(dh, seed=53), _rand(S, d, seed=54)
    ids = torch.tensor([3, 7, 3, 30, 1], device=DEV)
    pos = torch.stack([torch.randperm(S) for _ in range(B)]).float().to(DEV)
    out = torch.empty(B * S, d, device=DEV)
    H.embed_id_vid(ids, table, dh, fw, fb, vpe, out, B, S, frame_pos=pos)
    ref = torch.cat([table[ids][:, None, :].expand(B, S, dh), pos[:, :, None] * fw[None, None] + fb[None, None]], -1) + vpe[None]
    assert torch.allclose(out.view(B, S, d), ref, atol=1e-6)


@pytest.mark.parametrize("M,N,nmat", [(2048, 768, 3), (37, 40, 2), (512, 128, 1), (2048, 1024, 3)])
def test_colsum3(M, N, nmat):
    H = _abi()
    Xs = [_rand(M, N, seed=60 + i) for i in range(nmat)]
    outs = [torch.full((N,), float("nan"), device=DEV) for _ in range(nmat)]
    ws = torch.empty(3 * H.colsum_chunks(M) * N, device=DEV)
    H.colsum3(Xs, N, M, N, outs, ws)
    for X, o in zip(Xs, outs):
        assert (o.double() - X.double().sum(0)).abs().max().item() < 1e-4 * max(1.0, X.abs().sum(0).max().item())


@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb", [(2, 16, 48, 40, 40, 100), (2, 16, 48, 100, 40, 100), (2, 4, 8, 7, 40, 7), (3, 4, 8, 1, 40, 1),
                                              (2, 2, 32, 20, 20, 10), (2, 2, 64, 49, 40, 10), (2, 4, 16, 96, 40, 100)])
@pytest.mark.parametrize("mode", [1, 2, 0])
def test_attention_bwd_phases_equal_whole(B, H_, dh, Lq, La, Lb, mode):
    """phase 1 (D) + 2 (dQ) + 3 (dK/dV), and phase 4 (fused, query side in chunks of 48 rows), reproduce the single-call
    backward.  ``mode`` = segmm_attn_mode: the fused kernel in its default mix (fp16x3 products for single-chunk launches), with
    fp16x3 products wherever that form is built (several chunks too), and all exact-fp32."""
    H = _abi()
    prev = H.attn_mode(mode)
    try:
        _attention_bwd_phases(H, B, H_, dh, Lq, La, Lb)
    finally:
        H.attn_mode(prev)


def _attention_bwd_phases(H, B, H_, dh, Lq, La, Lb):
    d = H_ * dh
    g = torch.Generator().manual_seed(77)
    mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(DEV)
    Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    O = torch.empty(B * Lq, d, device=DEV)
    lse = torch.empty(2, B, H_, Lq, device=DEV)
    z = lambda t: (t, 0)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, O, d, lse, drop_p=0.1, seed=5, site=3)
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)

    def run(phases):
        Dv = torch.full((B, H_, Lq), float("nan"), device=DEV)
        outs = [torch.full_like(t, float("nan")) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
        for ph in phases:
            H.attn_bwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, lse, O, d, dO, d, Dv,
                       z(outs[0]), z(outs[1]), d, z(outs[2]), z(outs[3]), d, z(outs[4]), z(outs[5]), d, drop_p=0.1, seed=5, site=3,
                       phase=ph)
        return Dv, outs
    D0, ref = run([0])
    _, fused = run([4])                           # dQ + dK + dV in one kernel per key block (LDS-staged query side, D inside)
    for name, a, b in zip(("dQa", "dQb", "dKa", "dVa", "dKb", "dVb"), ref, fused):
        err = (a - b).abs().max().item()
        assert err < 2e-5 * max(1.0, a.abs().max().item()), (name, err)
    D1, got = run([1, 2, 3])
    assert (D0 - D1).abs().max().item() < 1e-5 * max(1.0, D0.abs().max().item())      # different summation order of the 48 products
    for a, b in zip(ref[:2], got[:2]):
        assert torch.equal(a, b)                  # dQ: the kernel computes its own D either way
    for a, b in zip(ref[2:], got[2:]):
        assert (a - b).abs().max().item() < 2e-5 * max(1.0, a.abs().max().item())        # dK/dV read the D kernel's D


@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb", [(6, 16, 32, 20, 20, 1), (6, 16, 32, 1, 1, 20), (3, 4, 16, 20, 20, 7), (2, 2, 64, 31, 17, 30), (3, 4, 8, 7, 40, 7)])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_bwd_merged_key_blocks_is_bitwise_the_per_block_form(B, H_, dh, Lq, La, Lb, p_drop, monkeypatch):
    """Short heads (config 3: 20 x (20 + 1), 1 x (1 + 20)): the fused backward runs ONE workgroup per head for both key blocks
    (query side staged once, one launch).  Every sum keeps the order of the per-block launches: results are bit-identical
    (SEGMM_ATT_MERGE=0 = the per-block launches; models/encoder.py:138-161 is what both compute)."""
    H = _abi()
    prev = H.attn_mode(0)          # the exact-fp32 fused kernel (what short heads take by default)
    try:
        d = H_ * dh
        g = torch.Generator().manual_seed(91)
        mk = lambda L: (torch.randn(B, L, d, generator=g) * 0.7).to(DEV)
        Qa, Qb, Ka, Va, Kb, Vb = mk(Lq), mk(Lq), mk(La), mk(La), mk(Lb), mk(Lb)
        mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
        mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV)
        mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
        O = torch.empty(B * Lq, d, device=DEV)
        lse = torch.empty(2, B, H_, Lq, device=DEV)
        z = lambda t: (t, 0)
        H.attn_fwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, O, d, lse, drop_p=p_drop, seed=5, site=3)
        dO = torch.randn(B * Lq, d, generator=g).to(DEV)

        def run(merge):
            H.config_set("ATT_MERGE", int(merge))
            Dv = torch.empty((B, H_, Lq), device=DEV)
            outs = [torch.full_like(t, float("nan")) for t in (Qa, Qb, Ka, Va, Kb, Vb)]
            H.attn_bwd(B, H_, dh, Lq, La, Lb, z(Qa), z(Qb), d, z(Ka), z(Va), d, z(Kb), z(Vb), d, mq, mka, mkb, lse, O, d, dO, d, Dv,
                       z(outs[0]), z(outs[1]), d, z(outs[2]), z(outs[3]), d, z(outs[4]), z(outs[5]), d, drop_p=p_drop, seed=5, site=3, phase=4)
            torch.cuda.synchronize()
            return outs
        a, b = run("0"), run("1")
        for name, x, y in zip(("dQa", "dQb", "dKa", "dVa", "dKb", "dVb"), a, b):
            assert not torch.isnan(y).any(), name
            assert torch.equal(x, y), (name, (x - y).abs().max().item())
    finally:
        H.attn_mode(prev)
        H.config_set("ATT_MERGE", 1)


@pytest.mark.parametrize("B,L,d", [(512, 100, 768), (512, 40, 768), (37, 20, 256), (64, 1, 512), (9, 7, 64), (3, 33, 1024)])
def test_layernorm_bwd_per_position_sums(B, L, d):
    """Embedding LayerNorms (encoder.py:450-471): the backward on the per-position grid leaves per-wave sums of dx whose
    segmm_colsum_pos combine is the positional-embedding gradient sum_b dx[b, s, :]; dx and the affine partials are those of
    the plain launch (bit-identical dx; the partial sums only differ by their grouping)."""
    H = _abi()
    rows = B * L
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(rows, d, generator=g).to(DEV)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.5).to(DEV)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    mean = x.mean(-1)
    rstd = 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-12)
    parts0 = H.layernorm_bwd_parts(rows, d)
    dx0 = torch.empty_like(x)
    pg0, pb0 = torch.empty(parts0, d, device=DEV), torch.empty(parts0, d, device=DEV)
    H.layernorm_bwd(dy, x, mean, rstd, gamma, dx0, None, pg0, pb0, drop_y_p=0.1, drop_y_site=3, seed=11)
    parts = H.layernorm_bwd_pos_parts(rows, L, d)
    assert parts > 0 and (4 * parts) % L == 0 and parts <= 1024
    dx = torch.empty_like(x)
    pg, pb = torch.empty(parts, d, device=DEV), torch.empty(parts, d, device=DEV)
    pp = torch.full((4 * parts, d), float("nan"), device=DEV)
    H.layernorm_bwd_pos(dy, x, mean, rstd, gamma, dx, None, pg, pb, pp, L, drop_y_p=0.1, drop_y_site=3, seed=11)
    out = torch.full((L, d), float("nan"), device=DEV)
    H.colsum_pos(pp, L, out)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx0)
    ref = dx0.double().view(B, L, d).sum(0)
    assert (out.double() - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item()) * max(1.0, B ** 0.5 / 4)
    for a, b in ((pg, pg0), (pb, pb0)):
        ra, rb = a.double().sum(0), b.double().sum(0)
        assert (ra - rb).abs().max().item() < 1e-5 * max(1.0, rb.abs().max().item())
    assert H.layernorm_bwd_pos_parts(rows + 1, L, d) == 0 or (rows + 1) % L == 0

