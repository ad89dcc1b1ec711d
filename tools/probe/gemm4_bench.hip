// Stand-alone harness for the round-6 plane GEMMs (csrc/gemm_planes4.h) beside the round-3 kernels (gemm_planes8.h): same
// synthetic planes (random fp16 terms, every site usable), results compared BITWISE, each timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I segmminterest_amd/csrc -o build/probe/gemm4_bench tools/probe/gemm4_bench.hip
//   ./gemm4_bench [iters] [which: 0 both, 8, 4] [shape filter substring]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
thread_local char g_segmm_err[512];
int segmm_fail(int code, const char*, ...) { return code; }
#include "gemm_planes4.h"
StepState* g_segmm_step = nullptr;
StepState* segmm_step_current() { return g_segmm_step; }
using namespace segmm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_planes(uint16_t* p, size_t n, uint32_t seed) {          // [.. 32 hi | 32 lo ..]: hi ~ +-[0, 4096), lo ~ +-[0, 0.5)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const bool lo = (i & 63) >= 32;
        const float v = lo ? ((float)(x & 2047) - 1024.f) / 2048.f : (float)((int)(x & 8191) - 4096);
        _Float16 h = (_Float16)v;
        p[i] = __builtin_bit_cast(uint16_t, h);
    }
}

struct Shape { int layout, M, N, K, splits; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20, which = argc > 2 ? atoi(argv[2]) : 0;
    const char* filt = argc > 3 ? argv[3] : "";
    // SEGMM_GEMM_PROBE builds: dbg words for the v4 kernel, each timed in turn (bit 0 no C stores, bit 1 no epilogue,
    // bits 8-15 stagger units, bits 16-17 stagger pattern); the first one is the one compared with v8
    std::vector<int> dbgs;
    for (int i = 4; i < argc; ++i) dbgs.push_back(atoi(argv[i]));
    if (dbgs.empty()) dbgs.push_back(0);
    int dbg = dbgs[0];
    const Shape shapes[] = {
        {0, 20480, 3072, 768, 1}, {0, 20480, 768, 768, 1}, {0, 51200, 1536, 768, 1}, {0, 51200, 768, 768, 1}, {0, 20480, 768, 3072, 1},
        {0, 10240, 768, 768, 1}, {0, 10240, 3072, 768, 1}, {0, 20480, 2048, 512, 1}, {0, 1000, 200, 64, 1},
        {2, 3072, 768, 20480, 7}, {2, 768, 768, 20480, 28}, {2, 1536, 768, 51200, 14}, {2, 768, 768, 10240, 28}, {2, 512, 512, 20480, 28},
    };
    float* hdr;
    CK(hipMalloc(&hdr, 4 * 3 * SITE_FLOATS));
    {
        std::vector<float> hh(3 * SITE_FLOATS, 0.f);
        for (int s = 0; s < 2; ++s) { hh[s * SITE_FLOATS] = 1.f; for (int k = 0; k < AMAX_SLOTS; ++k) hh[s * SITE_FLOATS + SITE_HDR + k] = 1.0f; }
        CK(hipMemcpy(hdr, hh.data(), 4 * hh.size(), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int bad = 0;
    for (const Shape& sh : shapes) {
        char name[64];
        snprintf(name, sizeof name, "%s_%dx%dx%d", sh.layout ? "TN" : "NT", sh.M, sh.N, sh.K);
        if (*filt && !strstr(name, filt)) continue;
        const int M = sh.M, N = sh.N, K = sh.K;
        // NT: A [M][2K], B [N][2K].  TN: A [K][2M], B [K][2N]
        const size_t na = sh.layout ? (size_t)K * 2 * M : (size_t)M * 2 * K, nb = sh.layout ? (size_t)K * 2 * N : (size_t)N * 2 * K;
        uint16_t *pa, *pb;
        float *C8, *C4, *ws;
        CK(hipMalloc(&pa, 2 * na + 4096)); CK(hipMalloc(&pb, 2 * nb + 4096));
        CK(hipMalloc(&C8, 4 * (size_t)M * N)); CK(hipMalloc(&C4, 4 * (size_t)M * N));
        CK(hipMalloc(&ws, 4 * (size_t)sh.splits * ((size_t)M * N + M)));
        hipLaunchKernelGGL(fill_planes, dim3(2048), dim3(256), 0, 0, pa, na, 1u);
        hipLaunchKernelGGL(fill_planes, dim3(2048), dim3(256), 0, 0, pb, nb, 2u);
        CK(hipMemset(C8, 0xff, 4 * (size_t)M * N)); CK(hipMemset(C4, 0xee, 4 * (size_t)M * N));
        GemmArgs g;
        ::memset(&g, 0, sizeof g);
        g.M = M; g.N = N; g.K = K; g.ldc = N; g.res_period = 1;
        g.drop = make_drop(0.f, 0, 0);
        PGemmX q;
        ::memset(&q, 0, sizeof q);
        q.A.p = (const _Float16*)pa; q.A.hdr = hdr; q.B.p = (const _Float16*)pb; q.B.hdr = hdr + SITE_FLOATS; q.write_c = 1; q.dbg = dbg;
        if (sh.layout == 0) { q.A.ld2 = 2 * K; q.B.ld2 = 2 * K; q.A.bytes = (uint32_t)(2 * na); q.B.bytes = (uint32_t)(2 * nb); }
        else { q.A.ld2 = 2 * M; q.B.ld2 = 2 * N; q.A.bytes = (uint32_t)(2 * na); q.B.bytes = (uint32_t)(2 * nb); }
        const double flop = 2.0 * M * N * K;
        float us8 = 0.f, us4 = 0.f;
        auto run = [&](int var, float* C, float& us) -> int {
            GemmArgs a = g;
            a.C = C;
            dim3 grid, block;
            if (sh.layout == 0) {
                if (var == 8) { a.nbm = (M + 255) / 256; a.nbn = (N + 255) / 256; grid = dim3(a.nbm * a.nbn); block = dim3(512); }
                else { a.nbm = (M + 127) / 128; a.nbn = (N + 255) / 256; grid = dim3(a.nbm * a.nbn); block = dim3(256); }
            } else {
                const int ktiles = (K + 31) / 32, tps = (ktiles + sh.splits - 1) / sh.splits, splits = (ktiles + tps - 1) / tps;
                a.k_per_split = tps * 32; a.C = ws; a.ldc = N; a.slab_stride = (long long)M * N;
                if (var == 8) { a.nbm = M / 256; a.nbn = N / 256; grid = dim3(a.nbm * a.nbn, 1, splits); block = dim3(512); }
                else { a.nbm = M / 128; a.nbn = N / 256; grid = dim3(a.nbm * a.nbn, 1, splits); block = dim3(256); }
            }
            auto launch = [&]() {
                if (sh.layout == 0) {
                    if (var == 8) hipLaunchKernelGGL(gemm_pl_nt8<4>, grid, block, 0, 0, a, q);
                    else hipLaunchKernelGGL(gemm_pl_nt4, grid, block, 0, 0, a, q);
                } else {
#if 1
                    if (var == 8) hipLaunchKernelGGL(gemm_pl_tn8, grid, block, 0, 0, a, q);
                    else hipLaunchKernelGGL(gemm_pl_tn4, grid, block, 0, 0, a, q);
#else
                    hipLaunchKernelGGL(gemm_pl_tn8, grid, block, 0, 0, a, q);
#endif
                    if (grid.z > 1) {
                        const long long n4 = (long long)M * (N / 4);
                        int blocks = (int)((n4 + 255) / 256);
                        if (blocks > 2048) blocks = 2048;
                        hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, 0, (const float*)ws, (int)grid.z, (long long)M * N, C, N, M, N, 0,
                                           (const float*)nullptr, (float*)nullptr);
                    }
                }
            };
            for (int i = 0; i < iters; ++i) launch();          // warm-up: as many launches as are timed
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {          // best of three batches
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            us = best * 1e3f / iters;
            return 0;
        };
        if (sh.layout == 0 && (M % 256 || N % 256) && false) {}
        const bool tn8_ok = sh.layout == 0 || (M % 256 == 0 && N % 256 == 0);
        if (which != 4 && tn8_ok) if (run(8, C8, us8)) return 1;
        if (which != 8) { q.dbg = dbg; if (run(4, C4, us4)) return 1; }
        size_t ndiff = 0;
        if (which == 0 && tn8_ok) {
            std::vector<uint32_t> h8((size_t)M * N), h4((size_t)M * N);
            CK(hipMemcpy(h8.data(), C8, 4 * h8.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(h4.data(), C4, 4 * h4.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < h8.size(); ++i) if (h8[i] != h4[i]) { if (ndiff < 4) printf("  diff at (%zu, %zu): %08x vs %08x\n", i / N, i % N, h8[i], h4[i]); ++ndiff; }
            bad += ndiff != 0;
        }
        printf("%-22s  v8 %8.1f us %6.1f TF | v4 %8.1f us %6.1f TF | ratio %.3f | %s\n", name, us8, us8 > 0 ? flop / us8 * 1e-6 : 0.0, us4,
               us4 > 0 ? flop / us4 * 1e-6 : 0.0, us4 > 0 && us8 > 0 ? us8 / us4 : 0.0, which == 0 && tn8_ok ? (ndiff ? "MISMATCH" : "bitwise equal") : "-");
        for (size_t v = 1; v < dbgs.size(); ++v) {
            float us = 0.f;
            q.dbg = dbgs[v];
            if (run(4, C4, us)) return 1;
            printf("      v4 dbg 0x%05x %8.1f us %6.1f TF\n", dbgs[v], us, flop / us * 1e-6);
        }
#ifdef SEGMM_STAMPS
        for (int var : {8, 4}) {          // one stamped launch of each kernel: per-workgroup phases and the timeline of a few CUs
            if ((var == 8 && (which == 4 || !tn8_ok)) || (var == 4 && which == 8) || sh.layout != 0) continue;
            const int nwg = var == 8 ? ((M + 255) / 256) * ((N + 255) / 256) : ((M + 127) / 128) * ((N + 255) / 256);
            unsigned long long* st;
            CK(hipMalloc(&st, (size_t)nwg * 192)); CK(hipMemset(st, 0, (size_t)nwg * 192));
            q.stamps = st; q.dbg = dbg;
            float us = 0.f;
            if (run(var, var == 8 ? C8 : C4, us)) return 1;
            q.stamps = nullptr;
            std::vector<unsigned long long> hs((size_t)nwg * 24);
            CK(hipMemcpy(hs.data(), st, (size_t)nwg * 192, hipMemcpyDeviceToHost));
            if (var == 4) {          // finer stamps of the epilogue: cycles from the end of the k-loop to the end of each row block
                double d[8] = {0};
                for (int w = 0; w < nwg; ++w) for (int k = 0; k < 8; ++k) d[k] += (double)(hs[(size_t)nwg * 8 + (size_t)w * 16 + k] - hs[(size_t)w * 8 + 2]);
                printf("      epilogue row blocks (cycles after loop end):");
                for (int k = 0; k < 8; ++k) printf(" %.0f", d[k] / nwg);
                printf("\n");
            }
            CK(hipFree(st));
            unsigned long long t00 = ~0ull;
            for (int w = 0; w < nwg; ++w) if (hs[w * 8 + 4] < t00) t00 = hs[w * 8 + 4];
            double pro = 0, loop = 0, epi = 0, life = 0, tend = 0;
            struct Ev { double s, le, e; int wg; };
            std::vector<std::vector<Ev>> cus(4096);
            for (int w = 0; w < nwg; ++w) {
                const unsigned long long* x = &hs[w * 8];
                pro += (double)(x[1] - x[0]); loop += (double)(x[2] - x[1]); epi += (double)(x[3] - x[2]);
                const double s0 = (x[4] - t00) * 0.01, le = (x[7] - t00) * 0.01, e0 = (x[5] - t00) * 0.01;          // 100 MHz -> us
                life += e0 - s0; if (e0 > tend) tend = e0;
                const unsigned hw = (unsigned)x[6], xcc = (unsigned)(x[6] >> 32) & 15u;
                cus[(xcc << 8) | ((hw >> 8) & 0xff)].push_back({s0, le, e0, w});
            }
            int ncu = 0; double busy2 = 0, busy1 = 0, gap_sum = 0; int gaps = 0;
            for (auto& c : cus) {
                if (c.empty()) continue;
                ++ncu;
                // coverage: time with >= 1 / 2 workgroups in their k-LOOP on this CU
                std::vector<std::pair<double, int>> ev;
                for (auto& e : c) { ev.push_back({e.s, 0}); }
                std::vector<std::pair<double, int>> lp;
                for (auto& e : c) { lp.push_back({e.s + 0.0, 0}); }
                std::vector<std::pair<double, int>> pts;
                for (auto& e : c) { pts.push_back({e.le - (e.le - e.s) * (loop / (pro + loop)), +1}); pts.push_back({e.le, -1}); }
                std::sort(pts.begin(), pts.end());
                int lvl = 0; double last = 0;
                for (auto& pt : pts) { if (lvl >= 1) busy1 += pt.first - last; if (lvl >= 2) busy2 += pt.first - last; lvl += pt.second; last = pt.first; }
                // slot turnover: gap between a workgroup's end and the next start on the CU
                std::vector<double> ends, starts;
                for (auto& e : c) { ends.push_back(e.e); starts.push_back(e.s); }
                std::sort(ends.begin(), ends.end()); std::sort(starts.begin(), starts.end());
                const size_t first = var == 8 ? 1 : 2;
                for (size_t k = first; k < starts.size(); ++k) { gap_sum += starts[k] - ends[k - first]; ++gaps; }
            }
            printf("      stamps v%d: %d WGs on %d CUs; per WG cycles: prologue %.0f  loop %.0f  epilogue %.0f | life %.1f us | kernel %.1f us | per CU: >=1 WG in loop %.1f us, 2 in loop %.1f us | slot turnover %.2f us\n",
                   var, nwg, ncu, pro / nwg, loop / nwg, epi / nwg, life / nwg, tend, busy1 / ncu, busy2 / ncu, gaps ? gap_sum / gaps : 0.0);
            int shown = 0;
            for (auto& c : cus) {
                if (c.empty() || shown >= 2) continue;
                ++shown;
                std::sort(c.begin(), c.end(), [](const Ev& a, const Ev& b) { return a.s < b.s; });
                printf("        CU timeline (start / loop end / end, us):");
                for (auto& e : c) printf("  [%d: %.1f %.1f %.1f]", e.wg, e.s, e.le, e.e);
                printf("\n");
            }
        }
#endif
        if (which != 4 && tn8_ok && dbgs.size() > 1) { q.dbg = dbg; float us = 0.f; if (run(8, C8, us)) return 1; printf("      v8 again      %8.1f us %6.1f TF\n", us, flop / us * 1e-6); }
        fflush(stdout);
        CK(hipFree(pa)); CK(hipFree(pb)); CK(hipFree(C8)); CK(hipFree(C4)); CK(hipFree(ws));
    }
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad != 0;
}
