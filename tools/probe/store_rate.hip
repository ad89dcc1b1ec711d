// How fast can ONE workgroup (8 waves, one per CU) drain a 256 x 256 fp32 tile (256 KB) to global memory, by store shape?
// Every lane stores float4 (16 B); a wave-instruction (1 KB) covers  R rows x (1024 / R) contiguous bytes  of a row-major
// matrix with a 12 KB row pitch (N = 3072):  R = 16 (64 B per row: the register layout of the 16 x 16 MFMA accumulators),
// 8 (128 B), 4 (256 B), 1 (1 KB contiguous).  Plain and nontemporal stores.  Prints shader cycles per tile (median over
// workgroups) for 64 / 256 workgroups -- per-CU store-path limit vs chip-wide bandwidth.
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, bool NT>
__global__ __launch_bounds__(512) void k(float* C, int ldc, int nbn, unsigned long long* cyc, int reps, int xcd_mask) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) { if (lane == 0) cyc[blockIdx.x * 8 + wave] = 0; return; }      // only workgroups dealt to the chosen XCDs store
    const int m0 = (blockIdx.x / nbn) * 256, n0 = (blockIdx.x % nbn) * 256;
    constexpr int LPR = 64 / R;                 // lanes per row
    const int r = lane / LPR, c4 = (lane % LPR) * 4;
    f32x4 v = {(float)tid, 1.f, 2.f, 3.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        // wave w stores rows [32 w, 32 w + 32) of the tile: 32 rows x 256 cols = 32 KB = 32 instructions
#pragma unroll 8
        for (int it = 0; it < 32; ++it) {
            // instruction `it` covers R rows x (4 LPR) cols:  blocks tile the 32 x 256 region row-block major
            const int blocks_per_rowgroup = 256 / (4 * LPR);      // column blocks per group of R rows
            const int rg = it / blocks_per_rowgroup, cb = it % blocks_per_rowgroup;
            float* dst = C + (size_t)(m0 + 32 * wave + rg * R + r) * ldc + n0 + cb * 4 * LPR + c4;
            if (NT) __builtin_nontemporal_store(v, (f32x4*)dst); else *(f32x4*)dst = v;
        }
        v.x += 1.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int R, bool NT>
void run(const char* name, float* C, unsigned long long* d_cyc, int wgs, int xcd_mask = 0xff) {
    const int reps = 4, nbn = 12, ldc = 3072;
    std::vector<unsigned long long> h(wgs * 8);
    k<R, NT><<<wgs, 512>>>(C, ldc, nbn, d_cyc, reps, xcd_mask);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<R, NT><<<wgs, 512>>>(C, ldc, nbn, d_cyc, reps, xcd_mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (int b = 0; b < wgs; ++b) { unsigned long long m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, h[b * 8 + w]); if (m) per.push_back((double)m / reps); }
    std::sort(per.begin(), per.end());
    const int act = (int)per.size();
    printf("%-28s wgs %4d (active %4d, xcd mask %02x): %7.0f cycles per 256 KB tile (median; %.1f B/clk/CU)   kernel %.1f us  %.2f TB/s\n", name, wgs, act, xcd_mask,
           per[act / 2], 262144.0 / per[act / 2], ms * 1e3, (double)act * 262144.0 * reps / ms * 1e-9);
}


// the GEMM epilogue's exact shape: 8 waves as 2 (m) x 4 (n), lane (l15, lq) stores C[128 grp + 16 i + l15][64 wn + 16 j + 4 lq ..+3],
// i = 0..7 via the scalar offset, j = 0..3 via the immediate; MODE 0: buffer_store (MUBUF, soffset), 1: global_store (64-bit address)
template <int MODE>
__global__ __launch_bounds__(512) void kepi(float* C, int ldc, int nbn, unsigned long long* cyc, int reps, int xcd_mask) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) { if (lane == 0) cyc[blockIdx.x * 8 + wave] = 0; return; }
    const int m0 = (blockIdx.x / nbn) * 256, n0 = (blockIdx.x % nbn) * 256;
    const int grp = wave >> 2, wn = wave & 3, l15 = lane & 15, lq = lane >> 4;
    f32x4 v = {(float)tid, 1.f, 2.f, 3.f};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, 20480 * 3072 * 4, 0x00020000);
    const unsigned base = ((unsigned)(m0 + grp * 128 + l15) * (unsigned)ldc + (unsigned)(n0 + wn * 64 + 4 * lq)) * 4u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(base + 64u * j), (int)((unsigned)i * 16u * (unsigned)ldc * 4u), 0);
                else *(f32x4*)(C + (size_t)(m0 + grp * 128 + l15 + 16 * i) * ldc + n0 + wn * 64 + 4 * lq + 16 * j) = v;
                asm volatile("s_nop 3" :: "v"(v));
            }
        v.x += 1.f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();          // ISSUE time (what the GEMM's epilogue stamp measures)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = ((t1 - t0) << 32) | (t2 - t0);
}
template <int MODE>
void runepi(const char* name, float* C, unsigned long long* d_cyc, int wgs) {
    const int reps = 4, nbn = 12, ldc = 3072;
    std::vector<unsigned long long> h(wgs * 8);
    kepi<MODE><<<wgs, 512>>>(C, ldc, nbn, d_cyc, reps, 0xff);
    hipDeviceSynchronize();
    kepi<MODE><<<wgs, 512>>>(C, ldc, nbn, d_cyc, reps, 0xff);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, tot;
    for (int b = 0; b < wgs; ++b) { unsigned long long mi = 0, mt = 0; for (int w = 0; w < 8; ++w) { mi = std::max(mi, h[b * 8 + w] >> 32); mt = std::max(mt, h[b * 8 + w] & 0xffffffffull); } iss.push_back((double)mi / reps); tot.push_back((double)mt / reps); }
    std::sort(iss.begin(), iss.end()); std::sort(tot.begin(), tot.end());
    printf("%-34s wgs %4d: issue %7.0f cycles, drained %7.0f cycles per 256 KB tile (median)\n", name, wgs, iss[wgs / 2], tot[wgs / 2]);
}

int main() {
    float* C; hipMalloc(&C, (size_t)20480 * 3072 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 1024 * 8 * 8);
    for (int wgs : {64, 256, 960}) { runepi<0>("epilogue shape, buffer_store+soffset", C, cyc, wgs); runepi<1>("epilogue shape, global_store", C, cyc, wgs); }
    // one XCD / two / four XCDs storing, 32 workgroups each: is the burst limit per XCD (fabric link) or chip-wide (HBM)?
    run<8, false>(" 8 rows x 128 B plain", C, cyc, 256, 0x01);
    run<8, false>(" 8 rows x 128 B plain", C, cyc, 256, 0x03);
    run<8, false>(" 8 rows x 128 B plain", C, cyc, 256, 0x0f);
    run<8, false>(" 8 rows x 128 B plain", C, cyc, 256, 0x55);
    for (int wgs : {64, 256, 960}) {
        run<16, false>("16 rows x 64 B  plain", C, cyc, wgs);
        run<8, false>(" 8 rows x 128 B plain", C, cyc, wgs);
        run<4, false>(" 4 rows x 256 B plain", C, cyc, wgs);
        run<1, false>(" 1 row  x 1 KB  plain", C, cyc, wgs);
        run<16, true>("16 rows x 64 B  nontemporal", C, cyc, wgs);
        run<8, true>(" 8 rows x 128 B nontemporal", C, cyc, wgs);
        run<4, true>(" 4 rows x 256 B nontemporal", C, cyc, wgs);
    }
    return 0;
}
