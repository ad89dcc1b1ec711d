// Stand-alone timing harness for the planes-in attention forward (csrc/attention_pl.h) at BASELINE config 2's shape
// (B = 512, H = 16, dh = 48, Lq = 40, keys 40 + 100).  Synthetic planes (random fp16 terms), all sites usable.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I segmminterest_amd/csrc [-DSEGMM_ATT_PROBE] -o /tmp/attn_pl_bench tools/probe/attn_pl_bench.hip
//   ./attn_pl_bench [iters] [pflags] [drop_p] [Lq]
// pflags (SEGMM_ATT_PROBE builds): 256 = return after the staging, 512 = no staging.  Results are NOT checked here
// (tests/test_planes_gpu.py::test_attention_fwd_on_input_planes does that through the C ABI).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include <math.h>
thread_local char g_segmm_err[512];
int segmm_fail(int code, const char*, ...) { return code; }
#include "attention_pl.h"
StepState* g_segmm_step = nullptr;
StepState* segmm_step_current() { return g_segmm_step; }
using namespace segmm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; ::memcpy(&u, &h, 2); return u; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20, pflags = argc > 2 ? atoi(argv[2]) : 0;
    const float drop_p = argc > 3 ? atof(argv[3]) : 0.1f;
    const int B = 512, H = 16, Lq = argc > 4 ? atoi(argv[4]) : 40, La = 40, Lb = 100;
    constexpr int DH = 48;
    const int d = H * DH, nv = 4, nu = 2;
    const size_t nYv = (size_t)B * La * nv * d, nYu = (size_t)B * Lb * nu * d, nQ = (size_t)B * Lq * nv * d;
    std::vector<uint16_t> hv(2 * nYv), hu(2 * nYu);
    srand(1);
    auto fill = [&](std::vector<uint16_t>& v) {
        for (size_t i = 0; i < v.size(); i += 64)
            for (int j = 0; j < 64; ++j) v[i + j] = f2h(j < 32 ? (float)((rand() % 8192) - 4096) : (float)((rand() % 2048) - 1024) / 2048.f);
    };
    fill(hv); fill(hu);
    uint16_t *pv, *pu, *pq, *plo;
    float *fv, *fu, *fq, *O, *lse, *hdr;
    uint8_t *mv, *mu, *mq;
    CK(hipMalloc(&pv, 4 * nYv)); CK(hipMalloc(&pu, 4 * nYu));
    CK(hipMemcpy(pv, hv.data(), 4 * nYv, hipMemcpyHostToDevice)); CK(hipMemcpy(pu, hu.data(), 4 * nYu, hipMemcpyHostToDevice));
    if (Lq == La) pq = pv; else { CK(hipMalloc(&pq, 4 * nQ)); CK(hipMemcpy(pq, hv.data(), 4 * (nQ < nYv ? nQ : nYv), hipMemcpyHostToDevice)); }
    CK(hipMalloc(&fv, 4 * nYv)); CK(hipMalloc(&fu, 4 * nYu)); CK(hipMemset(fv, 0, 4 * nYv)); CK(hipMemset(fu, 0, 4 * nYu));
    if (Lq == La) fq = fv; else { CK(hipMalloc(&fq, 4 * nQ)); CK(hipMemset(fq, 0, 4 * nQ)); }
    CK(hipMalloc(&O, 4 * (size_t)B * Lq * d)); CK(hipMalloc(&plo, 4 * (size_t)B * Lq * d)); CK(hipMalloc(&lse, 8 * (size_t)B * H * Lq));
    CK(hipMalloc(&hdr, 4 * 4 * SITE_FLOATS)); CK(hipMemset(hdr, 0, 4 * 4 * SITE_FLOATS));
    {
        std::vector<float> hh(4 * SITE_FLOATS, 0.f);
        for (int s = 0; s < 3; ++s) { hh[s * SITE_FLOATS] = 4096.f; for (int k = 0; k < AMAX_SLOTS; ++k) hh[s * SITE_FLOATS + SITE_HDR + k] = 1.0f; }
        hh[3 * SITE_FLOATS + 7] = 4096.f;          // scale_in of the output site (kept in an unused header word)
        CK(hipMemcpy(hdr, hh.data(), 4 * hh.size(), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&mv, B * La)); CK(hipMalloc(&mu, B * Lb)); CK(hipMalloc(&mq, B * Lq));
    {
        std::vector<uint8_t> m(B * 112);
        for (auto& x : m) x = (rand() % 10) < 8;
        CK(hipMemcpy(mv, m.data(), B * La, hipMemcpyHostToDevice)); CK(hipMemcpy(mu, m.data(), B * Lb, hipMemcpyHostToDevice)); CK(hipMemcpy(mq, m.data(), B * Lq, hipMemcpyHostToDevice));
    }
    AttnArgs a;
    ::memset(&a, 0, sizeof(a));
    a.B = B; a.H = H; a.Lq = Lq; a.La = La; a.Lb = Lb;
    a.Qa = fq; a.Qb = fq + d; a.ldq = nv * d; a.Ka = fv + 2 * d; a.Va = fv + 3 * d; a.ldka = nv * d; a.Kb = fu; a.Vb = fu + d; a.ldkb = nu * d;
    a.mq = mq; a.mka = mv; a.mkb = mu; a.O = O; a.ldo = d; a.lse = lse; a.scale = 1.0f / sqrtf((float)DH);
    a.drop = make_drop(drop_p, 1, 3);
    a.pflags = pflags;
    a.po_o.p = (_Float16*)plo; a.po_o.ld2 = 2 * d; a.po_o.hdr = hdr + 3 * SITE_FLOATS; a.po_o.scale_in = hdr + 3 * SITE_FLOATS + 7;
    AttnInPlanes& in = a.in;
    in.Qa = (const _Float16*)pq; in.Qb = (const _Float16*)pq + 2 * d; in.ldq2 = 2 * nv * d;
    in.baseA = (const _Float16*)pv; in.offKa = 2 * 2 * d * 2; in.offVa = 2 * 3 * d * 2; in.ldka2 = 2 * nv * d; in.bytesA = (uint32_t)(4 * nYv);
    in.baseB = (const _Float16*)pu; in.offKb = 0; in.offVb = 2 * d * 2; in.ldkb2 = 2 * nu * d; in.bytesB = (uint32_t)(4 * nYu);
    in.hdr_q = hdr; in.hdr_ka = hdr + SITE_FLOATS; in.hdr_kb = hdr + 2 * SITE_FLOATS;
    const int nqt = (Lq + 15) / 16;
    const size_t lds = attn_fwd_pl_lds_bytes<DH>(La, Lb);
    auto kern = attn_fwd_pl_kernel<DH, 9, 40, true>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * nqt, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(B * H), dim3(64 * nqt), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(B * H), dim3(64 * nqt), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, fl = 4.0 * DH * Lq * (La + Lb) * B * H;
    const double bytes = 4.0 * (2.0 * Lq + 2.0 * (La + Lb)) * DH * B * H + 8.0 * Lq * DH * B * H;
    if (argc > 5 && !strcmp(argv[5], "bwd")) {          // backward: ./attn_pl_bench iters pflags drop Lq bwd   (planes-in fused backward and the fp16x3 fused backward beside it)
        float *dO, *Dq, *dYv, *dYu;
        uint16_t *pdv, *pdu;
        CK(hipMalloc(&dO, 4 * (size_t)B * Lq * d)); CK(hipMemset(dO, 0, 4 * (size_t)B * Lq * d));
        CK(hipMalloc(&dYv, 4 * nYv)); CK(hipMalloc(&dYu, 4 * nYu)); CK(hipMalloc(&pdv, 4 * nYv)); CK(hipMalloc(&pdu, 4 * nYu));
        {   // dO, O: small random values; lse: max 0, 1/sum 1/140
            std::vector<float> r((size_t)B * Lq * d);
            for (auto& x : r) x = (float)((rand() % 2001) - 1000) * 1e-3f;
            CK(hipMemcpy(dO, r.data(), 4 * r.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(O, r.data(), 4 * r.size(), hipMemcpyHostToDevice));
            std::vector<float> l(2 * (size_t)B * H * Lq, 0.f);
            for (size_t i = l.size() / 2; i < l.size(); ++i) l[i] = 1.f / 140.f;
            CK(hipMemcpy(lse, l.data(), 4 * l.size(), hipMemcpyHostToDevice));
            std::vector<float> f(nYv);
            for (auto& x : f) x = (float)((rand() % 2001) - 1000) * 1e-3f;
            CK(hipMemcpy(fv, f.data(), 4 * nYv, hipMemcpyHostToDevice)); CK(hipMemcpy(fu, f.data(), 4 * nYu, hipMemcpyHostToDevice));
        }
        a.dO = dO; a.lddo = d; a.Dvec = nullptr;
        a.dQa = dYv; a.dQb = dYv + d; a.lddq = nv * d; a.dKa = dYv + 2 * d; a.dVa = dYv + 3 * d; a.lddka = nv * d; a.dKb = dYu; a.dVb = dYu + d; a.lddkb = nu * d;
        a.dQap = (_Float16*)pdv; a.dQbp = (_Float16*)pdv + 2 * d; a.lddq2 = 2 * nv * d;
        a.dKap = (_Float16*)pdv + 4 * d; a.dVap = (_Float16*)pdv + 6 * d; a.lddka2 = 2 * nv * d;
        a.dKbp = (_Float16*)pdu; a.dVbp = (_Float16*)pdu + 2 * d; a.lddkb2 = 2 * nu * d;
        float* hd; CK(hipMalloc(&hd, 4 * 3 * SITE_FLOATS)); CK(hipMemset(hd, 0, 4 * 3 * SITE_FLOATS));
        { float sc = 1024.f; CK(hipMemcpy(hd + 2 * SITE_FLOATS + 7, &sc, 4, hipMemcpyHostToDevice)); }
        a.hdr_q = hd; a.hdr_ka = hd; a.hdr_kb = hd + SITE_FLOATS; a.sin_q = a.sin_ka = a.sin_kb = hd + 2 * SITE_FLOATS + 7;
        a.pflags = ATT_PLANES_ONLY | pflags;
        const int Tp = 48 + 112, QC = ATT_FUSED_QCHUNK;
        for (int form = 0; form < 2; ++form) {
            double tot = 0;
            for (int blk = 0; blk < 2; ++blk) {
                const int nt = blk == 0 ? 3 : 7, nwp = nt > 4 ? 4 : nt;
                a.hpb = blk;
                const size_t l16 = ((size_t)3 * QC * (DH + 4) + 3 * QC + (size_t)nwp * 16 * 20 + 4 + 36 + (size_t)QC * (DH / 4)) * 4 + QC + Tp;
                const size_t lpl = attn_bwd_pl_lds_bytes<DH>(QC, nwp, Tp);
                int oc = 0;
                if (form == 0) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&oc, attn_bwd_fused16_kernel<DH, 4, true>, 64 * nwp, l16));
                else CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&oc, attn_bwd_pl_kernel<DH, 4, true>, 64 * nwp, lpl));
                auto go = [&]() {
                    if (form == 0) hipLaunchKernelGGL((attn_bwd_fused16_kernel<DH, 4, true>), dim3(B * H), dim3(64 * nwp), l16, 0, a);
                    else hipLaunchKernelGGL((attn_bwd_pl_kernel<DH, 4, true>), dim3(B * H), dim3(64 * nwp), lpl, 0, a);
                };
                for (int i = 0; i < 3; ++i) go();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) go();
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  %s block %c: lds %zu occ %d  %8.1f us\n", form ? "attn_bwd_pl     " : "attn_bwd_fused16", blk ? 'b' : 'a', form ? lpl : l16, oc, ms * 1e3 / iters);
                tot += ms * 1e3 / iters;
            }
            printf("%s pflags=%d p=%.2f Lq=%d: %8.1f us  %6.2f TFLOP/s (14 dh Lq T)\n", form ? "attn_bwd_pl     " : "attn_bwd_fused16", pflags, drop_p, Lq, tot,
                   14.0 * DH * Lq * (La + Lb) * B * H / tot / 1e6);
        }
        return 0;
    }
    printf("attn_fwd_pl pflags=%d p=%.2f Lq=%d lds=%zu occ=%d wg/CU: %8.1f us  %6.2f TFLOP/s  %6.0f GB/s (Q+K+V in, O fp32+planes out)\n", pflags, drop_p, Lq, lds, occ, us,
           fl / us / 1e6, bytes / us / 1e3);
    return 0;
}
