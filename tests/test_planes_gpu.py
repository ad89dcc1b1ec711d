"""The plane-operand GEMM path on the MI355X (csrc/gemm_planes.h, common.h PlaneOut): GEMM kernels against fp64 and the
on-the-fly fp16x3 kernel, producer-written planes against the stand-alone split pass (bit for bit), the overflow fall-back,
the delayed-scale update, and whole-model agreement of delayed scaling with exact scaling.  Run with ``pytest -m gpu``."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import MODEL_CASES, build_model, call_model, load_case      # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _abi():
    from segmminterest_amd import hipabi
    hipabi.lib()
    return hipabi


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _planes_to_f32(planes, rows, cols, scale):
    v = planes.view(rows, cols // 32, 2, 32).float()
    return ((v[:, :, 0] + v[:, :, 1]) / scale).reshape(rows, cols)


def _ref_planes(H, x, rows, cols, scale):
    """planes of x made by the stand-alone pass with a GIVEN scale (mode 1) -- the reference for every fused producer"""
    hdr = H.new_site(x.device)[0]
    hdr[0] = scale
    pl = torch.empty((rows, 2 * cols), dtype=torch.float16, device=x.device)
    H.split_p32(x, rows, cols, cols, pl, 2 * cols, hdr, mode=1)
    return pl, hdr


def _po(H, rows, cols, scale):
    hdr = H.new_site(DEV)[0]
    sc = torch.tensor([scale], dtype=torch.float32, device=DEV)
    pl = torch.zeros((rows, 2 * cols), dtype=torch.float16, device=DEV)
    return pl, hdr, sc, H.PO(pl, 2 * cols, hdr, sc.data_ptr())


@pytest.fixture(autouse=True, params=["default", "round6", "round3"])
def gemm_generation(request):
    """Every test of this module runs on each generation of the plane GEMM kernels: the library's own choice per launch, the round-6
    kernels wherever they fit (gemm_pl_nt4 / gemm_pl_tn4: 128 x 256 tiles, two workgroups per CU; knobs PL_VAR = 44, TN_VAR = 4) and
    the round-3 kernels (gemm_pl_nt8 / gemm_pl_tn8).  The generations give bitwise the same results (tools/probe/gemm4_bench.hip);
    here each is held to the same references."""
    if request.param == "default" or not request.node.name.startswith(("test_gemm_p", "test_low_scale", "test_delayed_scaling_matches")):
        if request.param != "default":
            pytest.skip("not a GEMM-kernel test: runs once")
        yield
        return
    H = _abi()
    pl, tn = (44, 4) if request.param == "round6" else (8, 88)
    prev = H.config_set("PL_VAR", pl), H.config_set("TN_VAR", tn)
    try:
        yield
    finally:
        H.config_set("PL_VAR", prev[0])
        H.config_set("TN_VAR", prev[1])


# ------------------------------------------------------------------ GEMM kernels
# (N > 1024: more than four column tiles of gemm_pl_nt4 -- its grouped tile order, P4_NGROUP, with full, partial and single-tile last groups)
@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (1024, 768, 768), (300, 96, 64), (20480, 768, 768), (37, 32, 160),
                                   (640, 1280, 96), (384, 3072, 64), (300, 2304, 768), (130, 1536, 768)])
def test_gemm_p_nt_matches_fp64_and_on_the_fly(M, N, K):
    H = _abi()
    A, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.02)
    A[::7] *= 3.0
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    Cp, Cl = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, Cp, N)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cl, N, engine=H.ENGINE_F16X3)
    ref = A[:512].double() @ W.double().t()
    e_p = (Cp[:512].double() - ref).abs().mean() / ref.abs().mean()
    e_l = (Cl[:512].double() - ref).abs().mean() / ref.abs().mean()
    assert e_p <= 1.5 * e_l + 1e-7, (float(e_p), float(e_l))
    assert float((Cp - Cl).abs().max() / Cl.abs().max()) < 2e-6


def test_gemm_p_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C/D map or a wrong LDS permutation (guide §3)."""
    H = _abi()
    n = 256
    A = torch.eye(n, device=DEV)
    Bm = (torch.arange(n * n, device=DEV, dtype=torch.float32).view(n, n) % 97) + torch.arange(n, device=DEV)[:, None] * 0.5
    C = torch.empty(n, n, device=DEV)
    H.gemm_p(H.LAYOUT_NT, n, n, n, H.to_planes(A, n, n), H.to_planes(Bm, n, n), C, n)
    assert torch.equal(C, Bm.t())
    H.gemm_p(H.LAYOUT_TN, n, n, n, H.to_planes(Bm, n, n), H.to_planes(A, n, n), C, n)      # C = Bm^T . I
    assert torch.equal(C, Bm.t())


@pytest.mark.parametrize("M,N,K,splits", [(768, 768, 2048, 7), (96, 64, 300, 1), (96, 64, 300, 3), (64, 32, 8, 1), (3072, 768, 4096, 4)])
def test_gemm_p_tn_splitk(M, N, K, splits):
    H = _abi()
    dY, X = _rand(K, M, seed=3, scale=0.01), _rand(K, N, seed=4)
    C = torch.empty(M, N, device=DEV)
    ws = torch.empty(splits * M * N, device=DEV)
    H.gemm_p(H.LAYOUT_TN, M, N, K, H.to_planes(dY, K, M), H.to_planes(X, K, N), C, N, splits=splits, workspace=ws)
    ref = dY.double().t() @ X.double()
    assert float((C.double() - ref).abs().max() / ref.abs().max()) < 3e-6
    # accumulate
    C0 = C.clone()
    H.gemm_p(H.LAYOUT_TN, M, N, K, H.to_planes(dY, K, M), H.to_planes(X, K, N), C, N, splits=splits, workspace=ws, accumulate=True)
    assert float((C - 2 * C0).abs().max() / C0.abs().max()) < 1e-6


@pytest.mark.parametrize("M,N,K,splits", [(768, 768, 2048, 7), (96, 64, 300, 1), (512, 256, 1000, 3), (3072, 768, 4096, 4)])
def test_gemm_p_tn_folded_column_sums(M, N, K, splits):
    """colsum_out = sum_k A[k, :]: the bias gradient formed inside the weight-gradient kernel."""
    H = _abi()
    dY, X = _rand(K, M, seed=40), _rand(K, N, seed=41)
    C, cs = torch.empty(M, N, device=DEV), torch.full((M,), 7.0, device=DEV)
    ws = torch.empty(splits * (M * N + M), device=DEV)
    pdy, px = H.to_planes(dY, K, M), H.to_planes(X, K, N)
    H.gemm_p(H.LAYOUT_TN, M, N, K, pdy, px, C, N, splits=splits, workspace=ws, colsum_out=cs)
    ref = dY.double().sum(0)
    tol = 3e-6 * float(dY.double().abs().sum(0).max())
    assert float((cs.double() - ref).abs().max()) <= tol
    refC = dY.double().t() @ X.double()
    assert float((C.double() - refC).abs().max() / refC.abs().max()) < 3e-6
    H.gemm_p(H.LAYOUT_TN, M, N, K, pdy, px, C, N, splits=splits, workspace=ws, colsum_out=cs, accumulate=True)
    assert float((cs.double() - 2 * ref).abs().max()) <= 2 * tol


def test_gemm_p_column_slices_of_fused_buffers():
    """Operands as column slices (multiples of 32) of wider plane buffers: the fused dY / projection layouts."""
    H = _abi()
    T, d = 640, 64
    dY, X = _rand(T, 3 * d, seed=5), _rand(T, d, seed=6)
    pdy, px = H.to_planes(dY, T, 3 * d), H.to_planes(X, T, d)
    C = torch.empty(d, d, device=DEV)
    H.gemm_p(H.LAYOUT_TN, d, d, T, pdy.cols_slice(d, d), px, C, d)
    ref = dY[:, d:2 * d].double().t() @ X.double()
    assert float((C.double() - ref).abs().max() / ref.abs().max()) < 3e-6
    W = _rand(d, 3 * d, seed=7, scale=0.05)          # dgrad: [T, 3d] . W^T planes [d, 3d]
    Cd = torch.empty(T, d, device=DEV)
    H.gemm_p(H.LAYOUT_NT, T, d, 3 * d, pdy, H.to_planes(W, d, 3 * d), Cd, d)
    refd = dY.double() @ W.double().t()
    assert float((Cd.double() - refd).abs().max() / refd.abs().max()) < 3e-6


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_p_epilogue_matches_on_the_fly(act):
    H = _abi()
    M, N, K = 512, 128, 64
    A, W = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.1)
    bias, res = _rand(N, seed=10), _rand(40, N, seed=11)
    aux0 = _rand(M, N, seed=12)
    Cp, Cl = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    auxp, auxl = aux0.clone(), aux0.clone()
    kw = dict(bias=bias, residual=res, ldr=N, res_period=40, activation=act, drop_p=0.25, seed=77, site=5, ldaux=N)
    pl, hdr, sc, po = _po(H, M, N, 2.0 ** 9)
    H.gemm_p(H.LAYOUT_NT, M, N, K, H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False), Cp, N, c_pt=H.PT(pl, hdr, M, N, f32=Cp),
             c_scale_ptr=sc.data_ptr(), aux=auxp if act in (1, 2) else None, **kw)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cl, N, engine=H.ENGINE_F16X3, aux=auxl if act in (1, 2) else None, **kw)
    assert float((Cp - Cl).abs().max()) <= 1e-6 * float(Cl.abs().max())
    if act == 1:
        assert float((auxp - auxl).abs().max()) <= 1e-6 * float(auxl.abs().max())
    ref_pl, _ = _ref_planes(H, Cp, M, N, 2.0 ** 9)
    assert torch.equal(pl, ref_pl)
    assert float(hdr[0]) == 2.0 ** 9 and float(hdr[1]) == 0.0
    assert float(hdr[H.SITE_HDR:].max()) == float(Cp.abs().max())


@pytest.mark.parametrize("M,N,K", [(600, 448, 96), (16384, 768, 64), (16300, 1024, 64)])
@pytest.mark.parametrize("case", ["plain", "bias_res_full", "res_periodic_drop", "accumulate", "gelu_planes_drop", "dgelu_drop_planes",
                                  "relu", "drelu", "planes_only"])
def test_gemm_p_nt_epilogue_variants(M, N, K, case):
    """Every epilogue form of the NT plane GEMM against the on-the-fly fp16x3 kernel (same arithmetic, independent epilogue
    code), on shapes that are ragged in M and N and that make the launcher pick each tile width (128 / 192 / 256 columns): the
    residual / aux operand staged through LDS, periodic (positional-table) residuals, accumulate, activations with their aux
    tensors, dropout, plane output with and without the fp32 copy."""
    H = _abi()
    A, W = _rand(M, K, seed=81), _rand(N, K, seed=82, scale=0.1)
    pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
    bias = _rand(N, seed=83)
    Cp, Cl = torch.full((M, N), 7.0, device=DEV), torch.full((M, N), 7.0, device=DEV)
    kw, kwp, aux_p, aux_l, want_planes = {}, {}, None, None, False
    if case == "bias_res_full":
        res = _rand(M, N, seed=84)
        kw = dict(bias=bias, residual=res, ldr=N, res_period=M)
    elif case == "res_periodic_drop":
        res = _rand(100, N, seed=85)
        kw = dict(bias=bias, residual=res, ldr=N, res_period=100, drop_p=0.1, seed=5, site=9)
    elif case == "accumulate":
        Cp.copy_(_rand(M, N, seed=86)); Cl.copy_(Cp)
        kw = dict(accumulate=True)
    elif case == "gelu_planes_drop":
        aux_p, aux_l = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
        kw = dict(bias=bias, activation=1, drop_p=0.1, seed=6, site=3, ldaux=N)
        want_planes = N % 32 == 0
    elif case == "dgelu_drop_planes":
        aux_p = _rand(M, N, seed=87); aux_l = aux_p.clone()
        kw = dict(activation=2, drop_p=0.1, seed=6, site=3, ldaux=N)
        want_planes = N % 32 == 0
    elif case == "relu":
        kw = dict(bias=bias, activation=3, drop_p=0.2, seed=7, site=4)
    elif case == "drelu":
        aux_p = _rand(M, N, seed=88); aux_l = aux_p.clone()
        kw = dict(activation=4, ldaux=N)
    elif case == "planes_only":
        if N % 32:
            pytest.skip("plane output needs N % 32 == 0")
        want_planes = True
        kwp = dict(write_c=False)
    if want_planes:
        pl, hdr, sc, po = _po(H, M, N, 2.0 ** 6)
        kwp.update(c_pt=H.PT(pl, hdr, M, N, f32=Cp), c_scale_ptr=sc.data_ptr())
    H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, Cp, N, aux=aux_p, **kw, **kwp)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cl, N, engine=H.ENGINE_F16X3, aux=aux_l, **kw)
    tol = 2e-6 * float(Cl.abs().max())
    if case == "planes_only":
        assert float(Cp.min()) == 7.0 and float(Cp.max()) == 7.0          # the fp32 C was not written
        # (the on-the-fly result differs at rounding level: compare the values the planes reconstruct)
        assert float((_planes_to_f32(pl, M, N, 2.0 ** 6) - Cl).abs().max()) <= tol + 2e-7 * float(Cl.abs().max())
    else:
        assert float((Cp - Cl).abs().max()) <= tol, case
        if case == "gelu_planes_drop":
            assert float((aux_p - aux_l).abs().max()) <= 2e-6 * float(aux_l.abs().max())
        if want_planes:
            ref_pl, _ = _ref_planes(H, Cp, M, N, 2.0 ** 6)
            assert torch.equal(pl, ref_pl)
            assert float(hdr[0]) == 2.0 ** 6 and float(hdr[1]) == 0.0
            assert float(hdr[H.SITE_HDR:].max()) == float(Cp.abs().max())


def test_gemm_p_overflow_flag_takes_fp32_path():
    """A delayed scale that has become too large: the producer raises the flag, the consumer reads the fp32 copy instead of
    the (infinite) planes -- same result as with exact planes."""
    H = _abi()
    M, N, K = 384, 96, 128
    A, W = _rand(M, K, seed=13), _rand(N, K, seed=14, scale=0.05)
    A[5, 7] = 300.0
    good = H.to_planes(A, M, K)
    pw = H.to_planes(W, N, K)
    C0, C1, C2 = (torch.empty(M, N, device=DEV) for _ in range(3))
    H.gemm_p(H.LAYOUT_NT, M, N, K, good, pw, C0, N)
    # planes written with a scale 2^10 too large for the outlier: 300 * 2^10 > 65504
    pl, hdr = _ref_planes(H, A, M, K, 2.0 ** 10)
    assert float(hdr[1]) != 0.0 and not torch.isfinite(pl.float()).all()
    bad = H.PT(pl, hdr, M, K, f32=A)
    H.gemm_p(H.LAYOUT_NT, M, N, K, bad, pw, C1, N)
    assert torch.isfinite(C1).all() and float((C1 - C0).abs().max()) <= 2e-6 * float(C0.abs().max())
    # TN: both operands flagged
    dY, X = _rand(K, M, seed=15), _rand(K, N, seed=16)
    dY[3, 3] = 500.0
    pdy, h1 = _ref_planes(H, dY, K, M, 2.0 ** 9)
    px, h2 = _ref_planes(H, X, K, N, 2.0 ** 14)
    assert float(h1[1]) != 0.0 and float(h2[1]) != 0.0
    H.gemm_p(H.LAYOUT_TN, M, N, K, H.PT(pdy, h1, K, M, f32=dY), H.PT(px, h2, K, N, f32=X), C2, N)
    ref = dY.double().t() @ X.double()
    assert float((C2.double() - ref).abs().max() / ref.abs().max()) < 3e-6
    # never-written planes (scale 0) take the same path
    hz = H.new_site(DEV)[0]
    H.absmax(A, M, K, K, out=hz[H.SITE_HDR:])
    H.gemm_p(H.LAYOUT_NT, M, N, K, H.PT(torch.zeros_like(pl), hz, M, K, f32=A), pw, C1, N)
    assert float((C1 - C0).abs().max()) <= 2e-6 * float(C0.abs().max())


def test_low_scale_keeps_fp32_level_accuracy():
    """Delayed scales sit 2^3 below the exact ones (head-room for growth): the GEMM error vs fp64 must stay at fp32 level."""
    H = _abi()
    M, N, K = 2048, 768, 768
    A, W = _rand(M, K, seed=17), _rand(N, K, seed=18, scale=0.02)
    A[:, ::5] *= 1e-3                                    # wide dynamic range inside rows
    ref = A[:256].double() @ W.double().t()
    pw = H.to_planes(W, N, K)
    errs = []
    for shift in (0, 3, 6):
        exact = H.to_planes(A, M, K)
        s = float(exact.hdr[0]) / 2.0 ** shift
        pl, hdr = _ref_planes(H, A, M, K, s)
        C = torch.empty(M, N, device=DEV)
        H.gemm_p(H.LAYOUT_NT, M, N, K, H.PT(pl, hdr, M, K, f32=A), pw, C, N)
        errs.append(float((C[:256].double() - ref).abs().mean() / ref.abs().mean()))
    Cf = torch.empty(M, N, device=DEV)
    H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cf, N, engine=H.ENGINE_F32)
    e32 = float((Cf[:256].double() - ref).abs().mean() / ref.abs().mean())
    assert errs[1] <= 1.5 * e32 + 1e-8 and errs[2] <= 2.0 * e32 + 1e-8, (errs, e32)


# ------------------------------------------------------------------ producers: fused planes == split pass of their fp32 output
def test_rowop_producers_write_the_split_pass_planes():
    H = _abi()
    rows, d = 300, 96
    x = _rand(rows, d, seed=20).abs() + 0.01
    # L1 normalisation
    y = torch.empty_like(x)
    pl, hdr, sc, po = _po(H, rows, d, 2.0 ** 13)
    H.l1norm(x, y, po=po)
    ref_pl, _ = _ref_planes(H, y, rows, d, 2.0 ** 13)
    assert torch.equal(pl, ref_pl) and float(hdr[0]) == 2.0 ** 13 and float(hdr[1]) == 0.0
    assert float(hdr[H.SITE_HDR:].max()) == float(y.abs().max())
    # LayerNorm forward (+ dropout)
    g, b = _rand(d, seed=21), _rand(d, seed=22)
    out, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    pl, hdr, sc, po = _po(H, rows, d, 2.0 ** 10)
    H.layernorm_fwd(x, g, b, out, mean, rstd, drop_p=0.1, seed=5, site=3, po=po)
    ref_pl, _ = _ref_planes(H, out, rows, d, 2.0 ** 10)
    assert torch.equal(pl, ref_pl) and float(hdr[0]) == 2.0 ** 10
    out2 = torch.empty_like(x)
    H.layernorm_fwd(x, g, b, out2, mean, rstd, drop_p=0.1, seed=5, site=3)
    assert torch.equal(out, out2)                       # the fp32 output does not depend on the plane output
    # LayerNorm backward: planes of the forwarded (dropped) gradient
    dy = _rand(rows, d, seed=23)
    parts = H.layernorm_bwd_parts(rows, d)
    dx, dxd = torch.empty_like(x), torch.empty_like(x)
    pg, pb = torch.empty(parts, d, device=DEV), torch.empty(parts, d, device=DEV)
    pl, hdr, sc, po = _po(H, rows, d, 2.0 ** 11)
    H.layernorm_bwd(dy, x, mean, rstd, g, dx, dxd, pg, pb, drop_b_p=0.1, drop_b_site=9, seed=5, po=po)
    ref_pl, _ = _ref_planes(H, dxd, rows, d, 2.0 ** 11)
    assert torch.equal(pl, ref_pl)
    assert float(hdr[H.SITE_HDR:].max()) == float(dxd.abs().max())
    # no scale yet (first use of a site): no planes, only the maxima
    pl, hdr, sc, po = _po(H, rows, d, 0.0)
    am = hdr[H.SITE_HDR:]
    H.layernorm_fwd(x, g, b, out2, mean, rstd, amax=am, po=po)
    assert float(hdr[0]) == 0.0 and float(pl.float().abs().max()) == 0.0 and float(am.max()) == float(out2.abs().max())


def test_gather_writes_planes_and_maxima():
    H = _abi()
    table = _rand(50, 64, seed=24).abs()
    idx = torch.tensor([[3, 7, -1, 49], [0, 0, 12, -1]], device=DEV)
    out = torch.empty(2, 4, 64, device=DEV)
    mask = torch.empty(2, 4, dtype=torch.uint8, device=DEV)
    pl, hdr, sc, po = _po(H, 8, 64, 2.0 ** 14)
    H.gather_l1(table, idx, out=out, mask=mask, po=po)
    ref_pl, _ = _ref_planes(H, out.view(8, 64), 8, 64, 2.0 ** 14)
    assert torch.equal(pl, ref_pl) and float(hdr[H.SITE_HDR:].max()) == float(out.abs().max())


@pytest.mark.parametrize("Hh,dh,B", [(4, 16, 3), (2, 32, 3), (2, 48, 3), (1, 64, 3), (8, 8, 3), (16, 8, 8)])
def test_attention_producers_write_the_split_pass_planes(Hh, dh, B):
    """(B = 8, H = 16: a grid on which no 'b' workgroup's wave key is a multiple of 1024 -- the user-key header's scale must
    still be written, by the (b, h) = (0, 0) workgroup.)"""
    H = _abi()
    S, Lt = 40, 23
    d = Hh * dh
    Yv, Yu = _rand(B * S, 4 * d, seed=30), _rand(B * Lt, 2 * d, seed=31)
    vm = (torch.rand(B, S, generator=torch.Generator().manual_seed(1)) > 0.2).to(torch.uint8).to(DEV)
    um = (torch.rand(B, Lt, generator=torch.Generator().manual_seed(2)) > 0.2).to(torch.uint8).to(DEV)
    O, lse = torch.empty(B * S, d, device=DEV), torch.empty(2, B, Hh, S, device=DEV)
    args = (B, Hh, dh, S, S, Lt, (Yv, 0), (Yv, d), 4 * d, (Yv, 2 * d), (Yv, 3 * d), 4 * d, (Yu, 0), (Yu, d), 2 * d, vm, vm, um)
    pl, hdr, sc, po = _po(H, B * S, d, 2.0 ** 12)
    H.attn_fwd(*args, O, d, lse, drop_p=0.1, seed=3, site=2, po=po)
    ref_pl, _ = _ref_planes(H, O, B * S, d, 2.0 ** 12)
    assert torch.equal(pl, ref_pl) and float(hdr[0]) == 2.0 ** 12
    assert float(hdr[H.SITE_HDR:].max()) == float(O.abs().max())
    # fused backward: planes of dYv (Qa, Qb, Ka, Va column blocks) and dYu (Kb, Vb)
    dO = _rand(B * S, d, seed=32)
    dYv, dYu = torch.zeros_like(Yv), torch.zeros_like(Yu)
    Dv = torch.empty(B * Hh * S, device=DEV)
    plv, hv, scv, _ = _po(H, B * S, 4 * d, 2.0 ** 9)
    plu, hu, scu, _ = _po(H, B * Lt, 2 * d, 2.0 ** 8)
    pln = H.AttnPlanes()
    base_v, base_u = plv.data_ptr(), plu.data_ptr()
    pln.dqa, pln.dqb, pln.lddq2 = base_v, base_v + 4 * d, 8 * d
    pln.dka, pln.dva, pln.lddka2 = base_v + 8 * d, base_v + 12 * d, 8 * d
    pln.dkb, pln.dvb, pln.lddkb2 = base_u, base_u + 4 * d, 4 * d
    pln.hdr_q = pln.hdr_ka = hv.data_ptr()
    pln.hdr_kb = hu.data_ptr()
    pln.sin_q = pln.sin_ka = scv.data_ptr()
    pln.sin_kb = scu.data_ptr()
    H.attn_bwd(*args, lse, O, d, dO, d, Dv, (dYv, 0), (dYv, d), 4 * d, (dYv, 2 * d), (dYv, 3 * d), 4 * d, (dYu, 0), (dYu, d), 2 * d,
               drop_p=0.1, seed=3, site=2, phase=4, planes=pln)
    rv, _ = _ref_planes(H, dYv, B * S, 4 * d, 2.0 ** 9)
    ru, _ = _ref_planes(H, dYu, B * Lt, 2 * d, 2.0 ** 8)
    assert torch.equal(plv, rv) and torch.equal(plu, ru)
    assert float(hv[0]) == 2.0 ** 9 and float(hu[0]) == 2.0 ** 8 and float(hv[1]) == 0.0 and float(hu[1]) == 0.0
    assert float(hv[H.SITE_HDR:].max()) == float(dYv.abs().max()) and float(hu[H.SITE_HDR:].max()) == float(dYu.abs().max())


def _site_planes(H, x, rows, cols, scale=None):
    """P32 planes + complete site header of x: written with ``scale`` (mode 1: maxima / flag folded in, like a fused producer)
    or with the exact scale of its maxima (mode 0)."""
    hdr = H.new_site(x.device)[0]
    pl = torch.empty((rows, 2 * cols), dtype=torch.float16, device=x.device)
    if scale is None:
        H.absmax(x, rows, cols, cols, out=hdr[H.SITE_HDR:])
        H.split_p32(x, rows, cols, cols, pl, 2 * cols, hdr, mode=0)
    else:
        hdr[0] = scale
        H.split_p32(x, rows, cols, cols, pl, 2 * cols, hdr, mode=1)
    return pl, hdr


@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb,p,case", [
    (4, 16, 48, 40, 40, 100, 0.1, "ok"), (3, 16, 48, 40, 40, 100, 0.0, "ok"), (2, 16, 48, 100, 40, 100, 0.1, "ok"),
    (5, 8, 32, 20, 20, 12, 0.1, "ok"), (2, 2, 64, 40, 40, 8, 0.0, "ok"), (3, 4, 48, 40, 0, 100, 0.1, "ok"),
    (3, 4, 48, 40, 40, 0, 0.1, "ok"), (2, 4, 16, 8, 40, 8, 0.0, "ok"), (3, 4, 48, 20, 20, 100, 0.1, "ok"),
    (3, 16, 48, 40, 40, 100, 0.1, "overflow_a"), (3, 16, 48, 40, 40, 100, 0.1, "tiny_b"), (3, 16, 48, 40, 40, 100, 0.1, "overflow_q"),
    (2, 16, 48, 40, 40, 100, 0.1, "scales_differ")])
def test_attention_fwd_on_input_planes(B, H_, dh, Lq, La, Lb, p, case):
    """The planes-in forward (csrc/attention_pl.h: Q / K / V as the P32 planes their producer GEMMs wrote, K / V staged by
    LDS-DMA, three fp16 MFMAs per product) against the fp32-operand forward on the same column slices, masks and dropout
    stream, and against an fp64 reference: O within 3e-6 of the maximum, softmax statistics to 2e-6.  Unusable sites (overflow
    flag up; a maximum far below the fp16 window; the query site) take the in-kernel fallback from the fp32 views."""
    H = _abi()
    d = H_ * dh
    g = torch.Generator().manual_seed(B * 131 + Lq + La)
    nv, nu = 4, 2
    Yv = (torch.randn(B * max(La, 1), nv * d, generator=g) * 0.7).to(DEV)
    Yu = (torch.randn(B * max(Lb, 1), nu * d, generator=g) * (3.0 if case == "scales_differ" else 0.7)).to(DEV)
    Qs = Yv if Lq == La else (torch.randn(B * Lq, nv * d, generator=g) * 0.7).to(DEV)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mka = (torch.rand(B, max(La, 1), generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, max(Lb, 1), generator=g) < 0.7).to(DEV)
    mq[0, 0] = False
    sc = {"overflow_a": (2.0 ** 15, None, None), "tiny_b": (None, 2.0 ** -9, None), "overflow_q": (None, None, 2.0 ** 15),
          "scales_differ": (2.0 ** 12, 2.0 ** 9, None)}.get(case, (None, None, None))
    plv, hv = _site_planes(H, Yv, Yv.shape[0], nv * d, sc[0])
    plu, hu = _site_planes(H, Yu, Yu.shape[0], nu * d, sc[1])
    if Qs is Yv and sc[2] is None:
        plq, hq = plv, hv
    else:
        plq, hq = _site_planes(H, Qs, Qs.shape[0], nv * d, sc[2])
    if case == "overflow_a":
        assert float(hv[1]) != 0.0          # some |Yv| 2^15 > 65504: the producer raised the flag
    args = (B, H_, dh, Lq, La, Lb, (Qs, 0), (Qs, d), nv * d, (Yv, 2 * d) if La else None, (Yv, 3 * d) if La else None, nv * d,
            (Yu, 0) if Lb else None, (Yu, d) if Lb else None, nu * d, mq, mka[:, :La].contiguous() if La else None,
            mkb[:, :Lb].contiguous() if Lb else None)
    outs = {}
    for form in ("f32", "pl"):
        O = torch.full((B * Lq, d), float("nan"), device=DEV)
        lse = torch.full((2, B, H_, Lq), float("nan"), device=DEV)
        am = torch.zeros(H.AMAX_SLOTS, device=DEV)
        pin = dict(q=(plq, hq, 2 * nv * d), a=(plv, hv, 2 * nv * d), b=(plu, hu, 2 * nu * d)) if form == "pl" else None
        H.attn_fwd(*args, O, d, lse, drop_p=p, seed=11, site=3, amax_o=am, pin=pin)
        outs[form] = (O, lse, am)
    O0, O1 = outs["f32"][0], outs["pl"][0]
    assert torch.isfinite(O1).all() and torch.isfinite(outs["pl"][1]).all()
    omax = float(O0.abs().max())
    assert float((O0 - O1).abs().max()) <= 3e-6 * omax, float((O0 - O1).abs().max()) / omax
    l0, l1 = outs["f32"][1], outs["pl"][1]
    assert float((l0[0] - l1[0]).abs().max()) <= 2e-6 * float(l0[0].abs().max())
    assert float(((l0[1] - l1[1]) / l0[1]).abs().max()) <= 4e-6
    assert float(outs["pl"][2].max()) == float(O1.abs().max())
    if p == 0.0:          # fp64 reference (no dropout): the fp16x3 products are as exact as the fp32 matrix cores'
        sl = lambda Y, k, L, n: Y.view(B, L, n * d)[:, :, k * d:(k + 1) * d].double()
        Qa_, Qb_ = sl(Qs, 0, Lq, nv), sl(Qs, 1, Lq, nv)
        z = torch.zeros(B, 0, d, dtype=torch.float64, device=DEV)
        Ka_, Va_ = (sl(Yv, 2, La, nv), sl(Yv, 3, La, nv)) if La else (z, z)
        Kb_, Vb_ = (sl(Yu, 0, Lb, nu), sl(Yu, 1, Lb, nu)) if Lb else (z, z)
        from test_ops_gpu import _attn_ref
        ref = _attn_ref(Qa_, Qb_, Ka_, Va_, Kb_, Vb_, mq, mka[:, :La], mkb[:, :Lb], H_)
        e0 = float((O0.view(B, Lq, d).double() - ref).abs().max()); e1 = float((O1.view(B, Lq, d).double() - ref).abs().max())
        assert e1 <= max(1.5 * e0, 2e-6 * omax), (e0, e1)
    # plane output of O (the out-projection GEMM's operand) from the planes-in form equals a split pass over its own fp32 O
    pl_o, hdr_o, sc_o, po = _po(H, B * Lq, d, 2.0 ** 12)
    O2 = torch.empty(B * Lq, d, device=DEV); lse2 = torch.empty(2, B, H_, Lq, device=DEV)
    H.attn_fwd(*args, O2, d, lse2, drop_p=p, seed=11, site=3, po=po, pin=dict(q=(plq, hq, 2 * nv * d), a=(plv, hv, 2 * nv * d), b=(plu, hu, 2 * nu * d)))
    ref_pl, _ = _ref_planes(H, O2, B * Lq, d, 2.0 ** 12)
    assert torch.equal(O2, O1) and torch.equal(pl_o, ref_pl) and float(hdr_o[0]) == 2.0 ** 12


@pytest.mark.parametrize("H_,dh,Lq,La,Lb,R", [(16, 48, 40, 40, 100, 13), (16, 48, 40, 40, 100, 18), (4, 32, 20, 20, 12, 13),
                                              (16, 48, 40, 40, 100, -13)])
def test_attention_fwd_planes_only_with_site_scales_far_apart(H_, dh, Lq, La, Lb, R):
    e_all, e_row0, share = _scale_gap_errors(H_, dh, Lq, La, Lb, R)
    assert e_all <= 3e-6, e_all
    assert e_row0 <= (Lb if R > 0 else La) * 2.0 ** (abs(R) - 37) + 3e-6, (e_row0, share)


def _scale_gap_errors(H_, dh, Lq, La, Lb, R):
    """Planes ONLY (no fp32 views: nothing to restage from) with the two key blocks' site scales 2^|R| apart (R > 0: the user
    block's K / V are 2^-R of the video block's; R < 0 the other way round).  One accumulator serves both blocks of O = P V: the
    P terms of the small-magnitude block are split with SP = 2^(14 - |R|), which costs that block's P an ABSOLUTE error of at most
    2^(|R| - 39) per key -- and its V are 2^-|R| of the other block's, so against the output's maximum the loss is ~L 2^-39 whatever
    R is.  Checked against fp64: all rows to 3e-6 of the maximum; the rows of batch row 0, whose large-magnitude
    block is masked out entirely (the output IS the small block's), to L 2^(|R| - 37) + 3e-6 of THEIR maximum.
    Returns (max error / max |O|, the same over batch row 0 alone, max |O[0]| / max |O|)."""
    H = _abi()
    B, d = 3, H_ * dh
    g = torch.Generator().manual_seed(R * 7 + Lq + 1000)
    nv, nu = 4, 2
    fa, fb = (1.0, 2.0 ** -R) if R > 0 else (2.0 ** R, 1.0)
    Yv = (torch.randn(B * La, nv * d, generator=g) * 0.7).to(DEV)
    Yu = (torch.randn(B * Lb, nu * d, generator=g) * 0.7).to(DEV)
    Yv[:, 2 * d:] *= fa          # K / V columns of the video block (the Q columns keep their size: the logits stay O(1) ...
    Yu *= fb
    Qs = (torch.randn(B * Lq, nv * d, generator=g) * 0.7).to(DEV)
    Qs[:, :d] /= fa; Qs[:, d:2 * d] /= fb          # ... because each block's query projection grows as its keys shrink)
    mq = torch.ones(B, Lq, dtype=torch.bool, device=DEV)
    mka = (torch.rand(B, La, generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    (mka if R > 0 else mkb)[0] = False          # batch row 0: the large-magnitude block contributes nothing
    # the Q site holds both projections (one buffer, one scale): give each its own site here so that the scale of Q is not the issue
    plv, hv = _site_planes(H, Yv[:, 2 * d:].contiguous(), B * La, 2 * d)
    plu, hu = _site_planes(H, Yu, B * Lb, nu * d)
    assert max(float(hv[0]) / float(hu[0]), float(hu[0]) / float(hv[0])) >= 2.0 ** (abs(R) - 1)
    # logits O(1) need |Q| ~ 1 / |K|: Q of the two blocks differ by 2^|R| as well; scale Q's site to its maxima (one site)
    plq, hq = _site_planes(H, Qs, B * Lq, nv * d)
    none = lambda off: (None, off)
    O = torch.full((B * Lq, d), float("nan"), device=DEV)
    lse = torch.full((2, B, H_, Lq), float("nan"), device=DEV)
    pin = dict(q=(plq, hq, 2 * nv * d), a=(plv, hv, 2 * 2 * d), b=(plu, hu, 2 * nu * d))
    H.attn_fwd(B, H_, dh, Lq, La, Lb, none(0), none(d), nv * d, none(0), none(d), 2 * d, none(0), none(d), nu * d, mq, mka, mkb,
               O, d, lse, drop_p=0.0, seed=11, site=3, pin=pin)
    assert torch.isfinite(O).all()
    # fp64 reference on the values the planes hold (hi + lo under the site scale: the operands the kernel was given)
    def held(pl, hdr, rows, cols):
        v = pl.view(rows, cols // 32, 2, 32).double()
        return ((v[:, :, 0] + v[:, :, 1]) / float(hdr[0])).reshape(rows, cols)
    Qh, Vh, Uh = held(plq, hq, B * Lq, nv * d), held(plv, hv, B * La, 2 * d), held(plu, hu, B * Lb, nu * d)
    sl = lambda Y, k, L, n: Y.view(B, L, n * d)[:, :, k * d:(k + 1) * d]
    from test_ops_gpu import _attn_ref
    ref = _attn_ref(sl(Qh, 0, Lq, nv), sl(Qh, 1, Lq, nv), sl(Vh, 0, La, 2), sl(Vh, 1, La, 2), sl(Uh, 0, Lb, nu), sl(Uh, 1, Lb, nu),
                    mq, mka, mkb, H_)
    err = (O.view(B, Lq, d).double() - ref).abs()
    omax, o0 = float(ref.abs().max()), float(ref[0].abs().max())
    assert o0 > 0
    return float(err.max()) / omax, float(err[0].max()) / o0, o0 / omax


@pytest.mark.parametrize("B,H_,dh,Lq,La,Lb,p,case", [
    (4, 16, 48, 40, 40, 100, 0.1, "ok"), (3, 16, 48, 40, 40, 100, 0.0, "ok"), (2, 16, 48, 100, 40, 100, 0.1, "ok"),
    (5, 8, 32, 20, 20, 12, 0.1, "ok"), (3, 4, 48, 40, 0, 100, 0.1, "ok"), (3, 4, 48, 40, 40, 0, 0.1, "ok"),
    (2, 4, 16, 8, 40, 8, 0.0, "ok"), (3, 4, 48, 20, 20, 100, 0.1, "ok"), (3, 16, 48, 40, 40, 100, 0.1, "repaired"),
    (3, 16, 48, 40, 40, 100, 0.1, "no_f32")])
def test_attention_bwd_on_input_planes(B, H_, dh, Lq, La, Lb, p, case):
    """The planes-in fused backward (csrc/attention_pl.h: Q / K / V from the projection GEMMs' P32 planes -- the chunk's Q rows
    and each wave's K tile by LDS-DMA, no operand splits) against the fp16x3 fused backward on the fp32 views: same masks,
    dropout stream, softmax statistics; gradients within 3e-6 of each tensor's maximum.  ``repaired``: the site headers carry a
    scale under which the planes overflowed (flag up) while the planes hold the producer's repair (exact scale of the maxima) --
    both sides derive that scale from the header.  ``no_f32``: the fp32 views are not passed at all (forward and backward)."""
    H = _abi()
    d = H_ * dh
    g = torch.Generator().manual_seed(B * 17 + Lq + La)
    nv, nu = 4, 2
    Yv = (torch.randn(B * max(La, 1), nv * d, generator=g) * 0.7).to(DEV)
    Yu = (torch.randn(B * max(Lb, 1), nu * d, generator=g) * 0.7).to(DEV)
    Qs = Yv if Lq == La else (torch.randn(B * Lq, nv * d, generator=g) * 0.7).to(DEV)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mka = (torch.rand(B, max(La, 1), generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, max(Lb, 1), generator=g) < 0.7).to(DEV)
    mq[0, 0] = False
    plv, hv = _site_planes(H, Yv, Yv.shape[0], nv * d)
    plu, hu = _site_planes(H, Yu, Yu.shape[0], nu * d)
    plq, hq = (plv, hv) if Qs is Yv else _site_planes(H, Qs, Qs.shape[0], nv * d)
    if case == "repaired":
        for h_ in {id(hv): hv, id(hu): hu, id(hq): hq}.values():
            h_[0] = float(h_[0]) * 4.0          # max * s in [2^16, 2^17): the delayed write would have overflowed
            h_[1] = 1.0
    mka_, mkb_ = (mka[:, :La].contiguous() if La else None), (mkb[:, :Lb].contiguous() if Lb else None)
    pin = dict(q=(plq, hq, 2 * nv * d), a=(plv, hv, 2 * nv * d), b=(plu, hu, 2 * nu * d))

    def views(with_f32):
        t = (lambda x: x) if with_f32 else (lambda x: None)
        return ((t(Qs), 0), (t(Qs), d), nv * d, (t(Yv), 2 * d) if La else None, (t(Yv), 3 * d) if La else None, nv * d,
                (t(Yu), 0) if Lb else None, (t(Yu), d) if Lb else None, nu * d)
    O = torch.empty(B * Lq, d, device=DEV); lse = torch.empty(2, B, H_, Lq, device=DEV)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, *views(True), mq, mka_, mkb_, O, d, lse, drop_p=p, seed=11, site=3)
    if case == "no_f32":          # the planes-in forward without fp32 views: equal to the one with them
        O2 = torch.empty_like(O); lse2 = torch.empty_like(lse)
        H.attn_fwd(B, H_, dh, Lq, La, Lb, *views(True), mq, mka_, mkb_, O2, d, lse2, drop_p=p, seed=11, site=3, pin=pin)
        O3 = torch.empty_like(O); lse3 = torch.empty_like(lse)
        H.attn_fwd(B, H_, dh, Lq, La, Lb, *views(False), mq, mka_, mkb_, O3, d, lse3, drop_p=p, seed=11, site=3, pin=pin)
        assert torch.equal(O2, O3) and torch.equal(lse2, lse3)
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)
    Dv = torch.empty(B * H_ * Lq, device=DEV)
    outs = {}
    prev = H.attn_mode(2)          # the fp16x3 fused backward wherever it is built: the comparison form
    try:
        for form in ("f32", "pl"):
            dQs = torch.zeros(B * Lq, nv * d, device=DEV) if Qs is not Yv else None
            dYv, dYu = torch.zeros_like(Yv), torch.zeros_like(Yu)
            dq = dYv if Qs is Yv else dQs
            H.attn_bwd(B, H_, dh, Lq, La, Lb, *views(form == "f32" or case != "no_f32"), mq, mka_, mkb_, lse, O, d, dO, d, Dv,
                       (dq, 0), (dq, d), nv * d, (dYv, 2 * d) if La else None, (dYv, 3 * d) if La else None, nv * d,
                       (dYu, 0) if Lb else None, (dYu, d) if Lb else None, nu * d, drop_p=p, seed=11, site=3, phase=4,
                       pin=pin if form == "pl" else None)
            outs[form] = (dq.clone(), dYv.clone(), dYu.clone())
    finally:
        H.attn_mode(prev)
    for name, a_, b_ in zip(("dQ", "dYv", "dYu"), outs["f32"], outs["pl"]):
        assert torch.isfinite(b_).all(), name
        m = float(a_.abs().max())
        assert float((a_ - b_).abs().max()) <= 3e-6 * max(m, 1e-30), (name, float((a_ - b_).abs().max()) / max(m, 1e-30))


def test_attention_bwd_on_input_planes_is_independent_of_the_wave_count(request):
    """Knob ATT_WAVES_PL (waves per workgroup of the single-chunk planes-in backward: a wave walks its key tiles in passes; default 3 =
    four workgroups per CU): dQ is accumulated in key-tile order whatever wave owns a tile, so 1, 2, 3 and 4 waves give BIT-identical
    gradients."""
    H = _abi()
    prev = H.config_set("ATT_WAVES_PL", 4)
    request.addfinalizer(lambda: H.config_set("ATT_WAVES_PL", prev))
    assert prev == 3
    B, H_, dh, Lq, La, Lb, p = 3, 16, 48, 40, 40, 100, 0.1
    d = H_ * dh
    g = torch.Generator().manual_seed(77)
    nv, nu = 4, 2
    Yv = (torch.randn(B * La, nv * d, generator=g) * 0.7).to(DEV)
    Yu = (torch.randn(B * Lb, nu * d, generator=g) * 0.7).to(DEV)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    plv, hv = _site_planes(H, Yv, Yv.shape[0], nv * d)
    plu, hu = _site_planes(H, Yu, Yu.shape[0], nu * d)
    pin = dict(q=(plv, hv, 2 * nv * d), a=(plv, hv, 2 * nv * d), b=(plu, hu, 2 * nu * d))
    views = ((Yv, 0), (Yv, d), nv * d, (Yv, 2 * d), (Yv, 3 * d), nv * d, (Yu, 0), (Yu, d), nu * d)
    O = torch.empty(B * Lq, d, device=DEV); lse = torch.empty(2, B, H_, Lq, device=DEV)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, *views, mq, mq, mkb, O, d, lse, drop_p=p, seed=11, site=3, pin=pin)
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)
    Dv = torch.empty(B * H_ * Lq, device=DEV)
    outs = []
    for w in (4, 3, 2, 1):
        H.config_set("ATT_WAVES_PL", w)
        dYv, dYu = torch.full_like(Yv, float("nan")), torch.full_like(Yu, float("nan"))
        H.attn_bwd(B, H_, dh, Lq, La, Lb, *views, mq, mq, mkb, lse, O, d, dO, d, Dv, (dYv, 0), (dYv, d), nv * d, (dYv, 2 * d), (dYv, 3 * d),
                   nv * d, (dYu, 0), (dYu, d), nu * d, drop_p=p, seed=11, site=3, phase=4, pin=pin)
        assert torch.isfinite(dYv).all() and torch.isfinite(dYu).all()
        outs.append((dYv, dYu))
    for dYv, dYu in outs[1:]:
        assert torch.equal(dYv, outs[0][0]) and torch.equal(dYu, outs[0][1])


@pytest.mark.parametrize("B,H_,bad,walk,La,Lb", [(40, 16, 2.0 ** 30, 512, 40, 100), (40, 16, 2.0 ** -30, 512, 40, 100), (3, 16, 2.0 ** 30, 512, 40, 100),
                                                  (3, 16, 2.0 ** 30, 2, 40, 100), (5, 16, 2.0 ** -30, 6, 40, 100), (36, 16, 2.0 ** 30, 512, 100, 40),
                                                  (2, 16, 2.0 ** -30, 4, 100, 40)])
def test_attention_bwd_repair_launch_walks_the_heads(request, B, H_, bad, walk, La, Lb):
    """The planes-only protocol of the planes-in fused backward at the op level: outputs written with a delayed scale that is
    2^30 off (overflow / below the fp16 window), segmm_site_fixup, then the REPAIR launch -- 512 workgroups wide, each walking
    its share of the B * H = 640 heads (csrc/attention_pl.h) -- must leave exactly the planes of the fp32 gradients under the
    exact scale of their maxima, and that scale in the header.  (B = 3 at the default width: the repair launch at one workgroup per
    head; widths 2 and 6: two / six workgroups walk 24 / 14 heads each.  La = 100: a multi-chunk query side -- the seven- /
    three-wave forms of the kernel, one wave per key tile.)"""
    H = _abi()
    prev_walk = H.config_set("ATT_REPAIR_WALK", walk)
    request.addfinalizer(lambda: H.config_set("ATT_REPAIR_WALK", prev_walk))
    dh, Lq, p = 48, La, 0.1
    d = H_ * dh
    g = torch.Generator().manual_seed(B + 5)
    nv, nu = 4, 2
    Yv = (torch.randn(B * La, nv * d, generator=g) * 0.7).to(DEV)
    Yu = (torch.randn(B * Lb, nu * d, generator=g) * 0.7).to(DEV)
    mq = (torch.rand(B, Lq, generator=g) < 0.8).to(DEV)
    mkb = (torch.rand(B, Lb, generator=g) < 0.7).to(DEV)
    plv, hv = _site_planes(H, Yv, Yv.shape[0], nv * d)
    plu, hu = _site_planes(H, Yu, Yu.shape[0], nu * d)
    pin = dict(q=(plv, hv, 2 * nv * d), a=(plv, hv, 2 * nv * d), b=(plu, hu, 2 * nu * d))
    views = ((Yv, 0), (Yv, d), nv * d, (Yv, 2 * d), (Yv, 3 * d), nv * d, (Yu, 0), (Yu, d), nu * d)
    none_views = ((None, 0), (None, d), nv * d, (None, 2 * d), (None, 3 * d), nv * d, (None, 0), (None, d), nu * d)
    O = torch.empty(B * Lq, d, device=DEV); lse = torch.empty(2, B, H_, Lq, device=DEV)
    H.attn_fwd(B, H_, dh, Lq, La, Lb, *views, mq, mq, mkb, O, d, lse, drop_p=p, seed=11, site=3, pin=pin)
    dO = torch.randn(B * Lq, d, generator=g).to(DEV)
    Dv = torch.empty(B * H_ * Lq, device=DEV)

    def bwd(dYv, dYu, planes=None):
        H.attn_bwd(B, H_, dh, Lq, La, Lb, *(views if planes is None else none_views), mq, mq, mkb, lse, O, d, dO, d, Dv,
                   (dYv, 0), (dYv, d), nv * d, (dYv, 2 * d), (dYv, 3 * d), nv * d, (dYu, 0), (dYu, d), nu * d,
                   drop_p=p, seed=11, site=3, phase=4, pin=pin, planes=planes)
    dYv, dYu = torch.zeros_like(Yv), torch.zeros_like(Yu)
    bwd(dYv, dYu)                                              # fp32 gradients of the same kernel
    import math

    def scale_of(m):          # common.h f16_scale_of: the power of two with m * s in [2^14, 2^15)
        return 2.0 ** (14 - math.floor(math.log2(m)))
    sv, su = scale_of(float(dYv.abs().max())), scale_of(float(dYu.abs().max()))
    ov, _, scv, _ = _po(H, B * La, nv * d, sv * bad)
    ou, _, scu, _ = _po(H, B * Lb, nu * d, su * bad)
    hov, hou = H.new_site(DEV)[0], H.new_site(DEV)[0]
    pln = H.AttnPlanes()
    bv, bu = ov.data_ptr(), ou.data_ptr()
    pln.dqa, pln.dqb, pln.lddq2 = bv, bv + 4 * d, 2 * nv * d
    pln.dka, pln.dva, pln.lddka2 = bv + 8 * d, bv + 12 * d, 2 * nv * d
    pln.dkb, pln.dvb, pln.lddkb2 = bu, bu + 4 * d, 2 * nu * d
    pln.hdr_q = pln.hdr_ka = hov.data_ptr()
    pln.hdr_kb = hou.data_ptr()
    pln.sin_q = pln.sin_ka = scv.data_ptr()
    pln.sin_kb = scu.data_ptr()
    pln.flags = H.ATTN_PLANES_ONLY
    zv, zu = torch.zeros_like(Yv), torch.zeros_like(Yu)          # (no fp32 gradient is written under ATTN_PLANES_ONLY)
    stats = torch.zeros(4, device=DEV)
    bwd(zv, zu, planes=pln)
    H.site_fixup(hov, hou, stats=stats)
    assert float(hov[2]) != 0.0 and float(hou[2]) != 0.0 and float(stats[0]) == 2.0
    pln.flags = H.ATTN_PLANES_ONLY | H.ATTN_REPAIR
    bwd(zv, zu, planes=pln)
    assert float(zv.abs().max()) == 0.0 and float(zu.abs().max()) == 0.0
    rv, _ = _ref_planes(H, dYv, B * La, nv * d, sv)
    ru, _ = _ref_planes(H, dYu, B * Lb, nu * d, su)
    assert float(hov[0]) == sv and float(hou[0]) == su
    assert torch.equal(ov, rv) and torch.equal(ou, ru)


def test_scales_update():
    H = _abi()
    arena = H.new_site(DEV, 4)
    arena[0, H.SITE_HDR + 3] = 5.0            # max 5 -> 2^2 <= 5 < 2^3: scale 2^(11-2) = 512 (5 * 512 = 2560 in [2048, 4096))
    arena[1, H.SITE_HDR + 200] = 0.002        # 2^-9 <= .002 < 2^-8 -> 2^20
    arena[2, H.SITE_HDR] = float("inf")       # non-finite: scale kept
    arena[2, 1] = 1.0                         # flagged
    idx = torch.tensor([0, 1, 2, -1], dtype=torch.int32, device=DEV)
    sc = torch.full((8,), 7.0, device=DEV)
    stats = torch.zeros(8, device=DEV)
    H.scales_update(arena, idx, 4, sc, stats, 12)
    assert sc[:4].tolist() == [512.0, 2.0 ** 20, 7.0, 7.0] and float(stats[0]) == 1.0


# ------------------------------------------------------------------ whole model: delayed scaling == exact scaling
@pytest.mark.parametrize("name", ["img_d32_N3_alllosses", "img_d64_h16_N3_Lt100", "both_fh2", "abl_crossmlp_N2", "abl_selfatt_N3"])
def test_delayed_scaling_matches_exact(name):
    """Second pass of the same eval-mode forward/backward with producer-written planes under DELAYED scales (the first pass
    calibrates them) against the exact-split pass: same logits and gradients to fp32 rounding, and still within the
    reference tolerance of the golden fixture."""
    from segmminterest_amd import hipabi as H
    if H.GEMM_ENGINE != H.ENGINE_F16X3P:
        pytest.skip("plane engine only")
    cfg, g, nograd, _ = load_case(name)
    model = build_model(cfg)
    model.load_state_dict(g["sd"])
    model = model.cuda().eval()
    st = model._store

    def run():
        model.zero_grad(set_to_none=True)
        out = call_model(model, g["in"], "train", DEV)
        out["loss"].backward()
        return out["logits"].detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    st.scaling = "exact"
    lg0, gr0 = run()
    st.scaling = "always"
    run()                                   # calibration pass (sites unknown -> exact split passes), scales recorded
    assert len(st.calibrated) > 0
    lg1, gr1 = run()                        # delayed scales, producer-written planes
    assert st.overflow_count() == 0
    assert float((lg1 - lg0).abs().max()) < 2e-6 * max(1.0, float(lg0.abs().max()))
    for k in gr0:
        sc = max(float(gr0[k].abs().max()), 1e-6)
        assert float((gr1[k] - gr0[k]).abs().max()) <= 2e-5 * sc + 1e-7, k
    assert float((lg1.cpu() - g["out"]["logits"]).abs().max()) < 1e-4
    for k, ref in g["grad"].items():
        sc = max(float(ref.abs().max()), 1e-6)
        assert float((gr1[k].cpu() - ref).abs().max()) <= 3e-4 * sc + 2e-6, k


def test_delayed_scaling_survives_a_magnitude_jump():
    """Inputs 4096x larger than the calibration pass: delayed scales overflow, the flags route the GEMMs to the fp32 copies,
    results stay correct (and the next pass runs on fresh scales again)."""
    from segmminterest_amd import hipabi as H
    if H.GEMM_ENGINE != H.ENGINE_F16X3P:
        pytest.skip("plane engine only")
    cfg, g, _, _ = load_case("img_d32_N2")
    model = build_model(cfg)
    model.load_state_dict(g["sd"])
    model = model.cuda().eval()
    st = model._store
    inp = {k: v.clone() for k, v in g["in"].items()}
    big = dict(model.backbone1.named_parameters())

    def fwd():
        with torch.no_grad():
            return call_model(model, inp, "inference", DEV)["logits"].clone()

    st.scaling = "always"
    fwd()
    fwd()
    assert st.overflow_count() == 0
    with torch.no_grad():
        big["encoder.layers.0.cross_attn.ln_vid.weight"].mul_(4096.0)      # X1 of layer 0 (a GEMM operand) grows 4096x
    got = fwd()                                # stale scales: even 128x head-room is not enough
    assert st.overflow_count() >= 1
    n = st.overflow_count()
    got2 = fwd()                               # rescaled by the end-of-pass update: no new overflow
    assert st.overflow_count() == n
    st.scaling = "exact"
    ref = fwd()
    tol = 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.isfinite(got).all() and float((got - ref).abs().max()) <= tol and float((got2 - ref).abs().max()) <= tol


def test_site_header_ring_wraps_cleanly():
    """The ParamStore hands site headers out of a ring that is cleared a quarter at a time (engine.ParamStore.hdr_rows).  With
    a ring so small that 30 training steps lap it several times, every step must still equal the run on the default ring
    bitwise: a header row that came back dirty would carry stale maxima into a tensor's scale and change the fp16 split."""
    from segmminterest_amd import engine as E
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    B, S, Lt, D, N = 8, 20, 6, 64, 3
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batches = [{k: v.cuda() for k, v in make_batch(B, S, Lt, D, seed=40 + i).items()} for i in range(3)]
    res = []
    saved = E.ParamStore.HDR_RING_ROWS
    try:
        for rows in (saved, 256):
            E.ParamStore.HDR_RING_ROWS = rows
            torch.manual_seed(3)
            model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
            tr = Trainer(model, dropout=False)
            losses = [float(tr.train_step(batches[i % 3])["loss"].detach()) for i in range(30)]
            torch.cuda.synchronize()
            res.append((losses, model._store.flat.detach().cpu().clone(), model._store.overflow_count()))
    finally:
        E.ParamStore.HDR_RING_ROWS = saved
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] == 0


def test_delayed_scaling_survives_a_magnitude_drop():
    """The opposite of the jump: a GEMM operand 2^20 times SMALLER than in the calibration pass.  Its delayed scale would leave
    the values deep in fp16's subnormals (12 bits instead of 22, silently); the consuming GEMMs see the complete maxima, refuse
    the planes and take the fp32 copy, so the result still equals exact scaling."""
    from segmminterest_amd import hipabi as H
    if H.GEMM_ENGINE != H.ENGINE_F16X3P:
        pytest.skip("plane engine only")
    cfg, g, _, _ = load_case("img_d32_N2")
    model = build_model(cfg)
    model.load_state_dict(g["sd"])
    model = model.cuda().eval()
    st = model._store
    inp = {k: v.clone() for k, v in g["in"].items()}
    big = dict(model.backbone1.named_parameters())

    def fwd():
        with torch.no_grad():
            return call_model(model, inp, "inference", DEV)["logits"].clone()

    st.scaling = "always"
    fwd()
    fwd()
    n0 = st.overflow_count()
    with torch.no_grad():
        big["encoder.layers.0.cross_attn.ln_vid.weight"].mul_(2.0 ** -20)      # X1 of layer 0 (a GEMM operand) shrinks 2^20 x
        big["encoder.layers.0.cross_attn.ln_vid.bias"].mul_(2.0 ** -20)
    got = fwd()                                # stale scale: the tensor sits 2^20 below where the scale expects it
    assert st.overflow_count() > n0            # counted as refused planes
    n = st.overflow_count()
    got2 = fwd()                               # rescaled by the end-of-pass update
    assert st.overflow_count() == n
    st.scaling = "exact"
    ref = fwd()
    tol = 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.isfinite(got).all() and float((got - ref).abs().max()) <= tol and float((got2 - ref).abs().max()) <= tol


@pytest.mark.parametrize("towers", [1, 2])
def test_backward_scales_follow_the_loss_gradient(towers):
    """Every backward tensor is linear in d loss / d logits.  Scaling the loss by 1e-6 and back by 1e6 between steps (a stand-in for
    a batch whose BPR loss has collapsed, then a normal one) moves every gradient by the same factor: with the backward sites'
    delayed scales tied to the step's max |d loss / d logits| nothing leaves the fp16 window; with repeat-the-last-maximum scales
    both steps do (under-range, then overflow) and run through the consumers' fallback.  The gradients agree either way."""
    from segmminterest_amd import hipabi as H
    if H.GEMM_ENGINE != H.ENGINE_F16X3P:
        pytest.skip("plane engine only")
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    B, S, Lt, D, N = 8, 20, 6, 64, 3
    # towers == 2: image backbone + id backbone ("both" inputs) -- BOTH backbones' backward sites must get loss-relative scales
    kind = "image" if towers == 1 else "both"
    nu, ni = (1, 1) if towers == 1 else (50, 60)
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)
    batch = {k: v.cuda() for k, v in make_batch(B, S, Lt, D, n_users=nu, n_items=ni, seed=60).items()}
    res = {}
    for rel in (True, False):
        torch.manual_seed(3)
        model = init_model(margs, n_users=nu, n_items=ni, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
        tr = Trainer(model, lr=0.0, weight_decay=0.0, dropout=False)          # lr 0: the parameters (and so the forward) never change
        st = model._store
        st.loss_relative = rel
        st.scaling = "always"          # delayed scales although dropout=False keeps the model in eval mode (identical steps)
        coef0 = list(model._loss_spec.coef)
        grads, exits = [], []
        for factor in (1.0, 1.0, 1.0, 1e-6, 1e6 * 1e-6, 1.0):
            model._loss_spec.coef = [c * factor for c in coef0]
            model._consts.clear()
            tr.train_step(batch)
            grads.append(st.gflat.detach().clone())
            exits.append(st.overflow_count())
        res[rel] = (grads, exits)
    (g_rel, e_rel), (g_plain, e_plain) = res[True], res[False]
    assert e_rel[-1] == 0, e_rel
    assert e_plain[3] > e_plain[2] and e_plain[4] > e_plain[3], e_plain          # the shrunken step AND the step after it
    gmax = float(g_rel[2].abs().max())
    for k, f in enumerate((1.0, 1.0, 1.0, 1e-6, 1.0, 1.0)):
        assert float((g_rel[k] - g_plain[k]).abs().max()) <= 2e-6 * gmax * f, k          # planes and fallback: the same numbers
        assert float((g_rel[k] / f - g_rel[2]).abs().max()) <= 2e-5 * gmax, k            # linear in the loss coefficient


@pytest.mark.parametrize("factor", [2.0 ** 22, 2.0 ** -22])
def test_attention_backward_planes_only_repair_pass(factor):
    """The fused attention backward writes dQ / dK / dV as planes ONLY (no fp32 copy: engine._layer_bwd, SEGMM_ATTN_PLANES_ONLY);
    what used to be the consumers' fp32 fallback is a repair pass of the same launches.  Tamper with the loss-relative gains of
    the two attention-gradient sites so that their delayed scales are 2^22 too large (overflow flag) or too small (maximum
    below the fp16 window): the step must still produce the gradients of an untampered twin, and the repaired sites are counted."""
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    if H.GEMM_ENGINE != H.ENGINE_F16X3P:
        pytest.skip("plane engine only")
    B, S, Lt, D, N = 16, 20, 10, 64, 3          # N = 3: one full layer (user queries too) + one video-only layer
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batches = [{k: v.cuda() for k, v in make_batch(B, S, Lt, D, seed=70 + i).items()} for i in range(2)]
    grads, counts = [], []
    for tamper in (False, True):
        torch.manual_seed(11)
        model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
        tr = Trainer(model, dropout=False, lr=0.0, weight_decay=0.0)          # lr 0: both twins keep identical parameters
        st = model._store
        if not (st.engine_p and st.attn_planes_only and st.attn_fused):
            pytest.skip("planes-only attention backward disabled")
        st.scaling = "always"          # delayed scales although dropout is off (deterministic twins)
        st.attn_planes_only = 2          # also for this test's short query side
        for i in range(3):
            tr.train_step(batches[i % 2])
        n0 = st.overflow_count()
        if tamper:
            sites = [n for n in st.site_index if "dYv" in n or "dYu" in n]
            assert sites
            for n in sites:
                st.gains()[st.site(n)] *= factor
        tr.train_step(batches[1])
        torch.cuda.synchronize()
        grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
        counts.append(st.overflow_count() - n0)
    assert counts[0] == 0 and counts[1] >= 2          # every tampered site was repaired (and counted)
    for k, ref in grads[0].items():
        sc = max(float(ref.abs().max()), 1e-9)
        assert float((grads[1][k] - ref).abs().max()) <= 2e-6 * sc + 1e-12, k


def test_input_features_as_planes_only():
    """Training steps after the first write the L1-normalised input features as planes ONLY, with the fixed scale 2^14 (rows are
    L1-normalised: 1/D <= max <= 1, so neither overflow nor the lower window edge is reachable) -- no fp32 copy, no fallback for
    the consumers (trainer.Trainer._input_act).  Twins with and without the mode must train alike, and a batch with one-hot and
    all-zero feature rows (the extremes of the range) must stay finite and uncounted."""
    from segmminterest_amd import hipabi as H
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import Trainer, default_args, init_model
    if H.GEMM_ENGINE != H.ENGINE_F16X3P:
        pytest.skip("plane engine only")
    B, S, Lt, D, N = 16, 20, 10, 64, 2
    margs = default_args(num_layers_enc=N, d_model=D, nhead=4, input_type={"user": "image", "photo": "image"}, exposure_prob=[1.0] * S)
    batches = [{k: v.cuda() for k, v in make_batch(B, S, Lt, D, seed=90 + i).items()} for i in range(3)]
    with torch.no_grad():          # extremes: a one-hot row (max = 1 after normalisation) and an all-zero row
        batches[2]["photo"][0, 0].zero_()
        batches[2]["photo"][0, 0, 5] = 3.0
        batches[2]["photo"][1, 1].zero_()
        batches[2]["user"][2, 0].zero_()
        batches[2]["user"][2, 0, 7] = -2.0
    res = []
    for only in (True, False):
        torch.manual_seed(21)
        model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).cuda()
        tr = Trainer(model, dropout=False, lr=0.0, weight_decay=0.0)          # lr 0: the twins keep identical parameters
        st = model._store
        st.scaling = "always"
        st.input_planes_only = only
        losses = [float(tr.train_step(batches[i % 3])["loss"].detach()) for i in range(6)]
        torch.cuda.synchronize()
        res.append((losses, st.gflat.detach().clone(), st.overflow_count()))
        if only:          # the mode was really on: the last step's input Acts carry no fp32 copy and the fixed scale
            act = tr._norm[("photo", tr._slot)]._segmm_act
            assert act.no_f32 and float(act.hdr[0]) == 16384.0 and float(act.hdr[1]) == 0.0
    (l1, p1, o1), (l0, p0, o0) = res
    assert o1 == 0 and o0 == 0
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(l1, l0)), (l1, l0)
    assert float((p1 - p0).abs().max()) <= 2e-6 * float(p0.abs().max())          # gradients of the last step (extreme rows in it)
    assert torch.isfinite(p1).all()
