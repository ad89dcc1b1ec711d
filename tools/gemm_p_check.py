#!/usr/bin/env python
"""Plane-operand GEMM (segmm_gemm_p) against the on-the-fly fp16x3 GEMM (segmm_gemm_h) and fp64: accuracy + TFLOP/s.

    python tools/gemm_p_check.py [--quick] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmminterest_amd import hipabi as H  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rel(a, b):
    return float((a.double() - b.double()).abs().mean() / b.double().abs().mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layouts", default="nt,tn")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    H.lib()
    nt_shapes = [(20480, 768, 768), (20480, 3072, 768), (51200, 1536, 768), (51200, 768, 768), (20480, 768, 3072), (71680, 768, 768)]
    tn_shapes = [(768, 768, 20480), (3072, 768, 20480), (1536, 768, 51200), (768, 768, 71680)]
    if args.quick:
        nt_shapes, tn_shapes = [(1024, 768, 768), (300, 96, 64)], [(768, 768, 2048), (96, 64, 300)]
    ok = True
    if "nt" in args.layouts:
        for (M, N, K) in nt_shapes:
            A = torch.randn(M, K, device=dev)
            A[::7] *= 3.0
            W = torch.randn(N, K, device=dev) * 0.02
            pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
            Cp, Cl = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, Cp, N)
            H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cl, N, engine=H.ENGINE_F16X3)
            torch.cuda.synchronize()
            sub = slice(0, min(M, 512))
            ref = A[sub].double() @ W.double().t()
            e_p, e_l = rel(Cp[sub], ref), rel(Cl[sub], ref)
            d = float((Cp - Cl).abs().max() / Cl.abs().max())
            # slow path (flag raised): must agree as well
            pa.hdr[1] = 1.0
            Cs = torch.empty(M, N, device=dev)
            H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, Cs, N)
            pa.hdr[1] = 0.0
            e_s = rel(Cs[sub], ref)
            tp = timeit(lambda: H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, Cp, N), args.iters)
            ama, amw = H.absmax(A, M, K, K), H.absmax(W, N, K, K)
            tl = timeit(lambda: H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cl, N, engine=H.ENGINE_F16X3, a_amax=ama, b_amax=amw), args.iters)
            fl = 2.0 * M * N * K
            good = e_p < 2 * e_l + 1e-7 and e_s < 2 * e_l + 1e-7 and d < 1e-5
            ok &= good
            print("NT %6d x %5d x %5d  planes %7.1f us %6.1f TF | on-the-fly %7.1f us %6.1f TF | err vs fp64: planes %.2e slow %.2e otf %.2e  maxdiff %.1e %s"
                  % (M, N, K, tp * 1e6, fl / tp / 1e12, tl * 1e6, fl / tl / 1e12, e_p, e_s, e_l, d, "ok" if good else "FAIL"), flush=True)
        # epilogue + plane output
        M, N, K = (2048, 768, 768) if not args.quick else (512, 128, 64)
        A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
        bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
        pa, pw = H.to_planes(A, M, K), H.to_planes(W, N, K, keep_f32=False)
        for act in (H.ACT_NONE, H.ACT_GELU):
            Cp, Cl = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            auxp, auxl = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            hdr = H.new_site(dev)[0]
            sc = torch.tensor([2.0 ** 10], device=dev)
            cpl = torch.empty(M, 2 * N, dtype=torch.float16, device=dev)
            cpt = H.PT(cpl, hdr, M, N, f32=Cp)
            kw = dict(bias=bias, residual=res, ldr=N, res_period=M, activation=act, drop_p=0.1, seed=1234, site=7, ldaux=N)
            H.gemm_p(H.LAYOUT_NT, M, N, K, pa, pw, Cp, N, c_pt=cpt, c_scale_ptr=sc.data_ptr(), aux=auxp if act else None, **kw)
            H.gemm(H.LAYOUT_NT, M, N, K, A, K, W, K, Cl, N, engine=H.ENGINE_F16X3, aux=auxl if act else None, **kw)
            torch.cuda.synchronize()
            d = float((Cp - Cl).abs().max() / Cl.abs().max())
            v = cpl.view(M, N // 32, 2, 32).float()
            recon = ((v[:, :, 0] + v[:, :, 1]) / hdr[0]).reshape(M, N)
            dr = float((recon - Cp).abs().max() / Cp.abs().max())
            amax_ok = abs(float(hdr[H.SITE_HDR:].max()) - float(Cp.abs().max())) == 0.0
            good = d < 1e-5 and dr < 1e-6 and amax_ok and float(hdr[1]) == 0.0 and float(hdr[0]) == 2.0 ** 10
            ok &= good
            print("NT epilogue act=%d: maxdiff vs on-the-fly %.1e, plane output reconstruction %.1e, amax exact %s %s" % (act, d, dr, amax_ok, "ok" if good else "FAIL"))
    if "tn" in args.layouts:
        for (M, N, K) in tn_shapes:
            if M % 32 or N % 32:
                continue
            dY = torch.randn(K, M, device=dev) * 0.01
            X = torch.randn(K, N, device=dev)
            pdy, px = H.to_planes(dY, K, M), H.to_planes(X, K, N)
            Cp, Cl = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            from segmminterest_amd import engine as E
            sp_l = E._splits_for(M, N, K)
            sp_p = E._splits_for_p(M, N, K) if hasattr(E, "_splits_for_p") else sp_l
            ws = torch.empty(max(sp_l, sp_p) * M * N, device=dev)
            try:
                H.gemm_p(H.LAYOUT_TN, M, N, K, pdy, px, Cp, N, splits=sp_p, workspace=ws)
            except RuntimeError as e:
                print("TN skipped:", e)
                break
            H.gemm(H.LAYOUT_TN, M, N, K, dY, M, X, N, Cl, N, engine=H.ENGINE_F16X3, splits=sp_l, workspace=ws)
            torch.cuda.synchronize()
            ref = dY[:, :256].double().t() @ X.double()
            e_p, e_l = rel(Cp[:256], ref), rel(Cl[:256], ref)
            d = float((Cp - Cl).abs().max() / Cl.abs().max())
            tp = timeit(lambda: H.gemm_p(H.LAYOUT_TN, M, N, K, pdy, px, Cp, N, splits=sp_p, workspace=ws), args.iters)
            ama, amb = H.absmax(dY, K, M, M), H.absmax(X, K, N, N)
            tl = timeit(lambda: H.gemm(H.LAYOUT_TN, M, N, K, dY, M, X, N, Cl, N, engine=H.ENGINE_F16X3, splits=sp_l, workspace=ws, a_amax=ama, b_amax=amb), args.iters)
            fl = 2.0 * M * N * K
            good = e_p < 2 * e_l + 1e-7 and d < 2e-5
            ok &= good
            print("TN %6d x %5d x %5d  planes(%2d) %7.1f us %6.1f TF | on-the-fly(%2d) %7.1f us %6.1f TF | err vs fp64: planes %.2e otf %.2e  maxdiff %.1e %s"
                  % (M, N, K, sp_p, tp * 1e6, fl / tp / 1e12, sp_l, tl * 1e6, fl / tl / 1e12, e_p, e_l, d, "ok" if good else "FAIL"), flush=True)
    print("ALL OK" if ok else "SOME FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
