// fp32-accurate GEMM on the 16-bit matrix cores: the fp32 operands are split into a few 16-bit terms, the largest
// partial products run on v_mfma_f32_32x32x16_{f16,bf16} (16-bit x 16-bit products are EXACT in fp32) and are
// accumulated in fp32.  One kernel template, two engines:
//
//  F16 = true, NPL = 2  -- the fp16x3 engine (default of the host facade):
//     x*s = hi + lo with two fp16 terms (11 + 11 mantissa bits, residual <= 2^-22 |x|), s a per-tensor power of two
//     that puts max|x| in [2^14, 2^15) so that neither term leaves the fp16 range (elements more than 2^18 below the
//     maximum keep an ABSOLUTE error of 2^-40 max: fp16 subnormals); three products  hh + hl + lh ; the result is
//     scaled back by the exact 1/(sa sb) in the epilogue.  Per-product error <= 3 * 2^-22, the order of ONE fp32
//     accumulation rounding; measured error vs fp64 at or below the f32-MFMA GEMM's on every layout
//     (tests/test_ops_gpu.py::test_gemm_f16x3_*).  96 matrix-pipe cycles per 32 x 32 x 16 block against 512 for the
//     fp32 MFMA.  max|x| arrives as an array of partial maxima written by the operand's producer (or segmm_absmax):
//     no host sync, no float atomics; every workgroup reduces the (<= 1024) partials itself.
//  F16 = false, NPL = 3 -- the bf16x6 engine: x = hi + mid + lo EXACTLY in three bf16 terms (8 + 8 + 8 bits, bf16 has
//     the fp32 exponent range: no scaling), six products  hh + (hm + mh) + (hl + lh + mm), dropped terms <= 2^-24;
//     192 cycles per block.  NPL = 2 with bf16 (hh + hm + mh, ~2e-5) is an opt-in for weight gradients only.
//
// Operands arrive either as fp32 (split ON THE FLY between the register prefetch and the LDS store: four
// v_fma_mix per element pair for fp16, see splith_pair) or PRE-SPLIT as 16-bit planes [NPL][rows][ld] written once by
// segmm_split* (weights: one split pass per optimizer step, W and W^T) -- then the main loop has no VALU work for
// that operand at all.
//
// Tile 128 x 128 x 32, 256 threads (2 x 2 waves, 64 x 64 each), 3 workgroups per CU for NPL = 2 (166 VGPR, 40 KB
// LDS).  LDS: per operand NPL planes [128 rows][32 k] of 16-bit values with an 80-byte row stride (conflict-free
// ds_read_b128 of the 8-element MFMA fragments).  fp32 operands whose k index is NOT contiguous in memory (B of NN,
// A and B of TN) are transposed in registers: a thread owns a 4(k) x 4(m) micro-block and writes 4-element k runs.
// Global -> register staging uses buffer loads (see the main loop); measurements in profiles/README.md.
#pragma once
#include <type_traits>

#include "gemm.h"

// wave priority during the fragment-read + MFMA phase of a k-tile (0 = off): with three waves per SIMD in different phases
// the arbiter then prefers the wave that can feed the matrix pipe over those issuing loads / split VALU / LDS stores.
// Measured on one box, alternating runs: NT 20480x3072x768 368 -> 358 us, TN 3072x768x20480 466 -> 450 us, step 87.5 k ->
// 88.4 k interactions/s (priority 3: the same).
#ifndef SEGMM_GEMM_SETPRIO
#define SEGMM_GEMM_SETPRIO 1
#endif

namespace segmm {

#ifdef SEGMM_GEMM_TRACE
// debug build only (tools/gemm_trace.py): shader-clock timestamps of the main-loop phases of ONE wave
__device__ unsigned long long g_gemm_trace[8 * 64];
#define TRACE_MARK(it, ph) do { if (trace_on && (it) < 64) { g_gemm_trace[(it) * 8 + (ph)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TRACE_MARK(it, ph) do { } while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XRS = 40;                          // plane row stride in bf16 elements (80 bytes)
constexpr int XPLANE = GBM * XRS;                // 5120 bf16 per plane

struct GemmPlanes {                              // optional pre-split operands (16-bit planes, k-contiguous rows)
    const __bf16* Ap; long long a_pstride;       // plane p of A at Ap + p * a_pstride, row stride = GemmArgs.lda
    const __bf16* Bp; long long b_pstride;
    const float* a_amax; int a_namax;            // fp16x3 engine: partial maxima of |A| and |B| (>= 0, any count <= 1024)
    const float* b_amax; int b_namax;
};

// exact 3-way split of two floats -> packed (hi0,hi1), (mid0,mid1), (lo0,lo1)
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& ph, uint32_t& pm, uint32_t& pl) {
    ph = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
    pm = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
    pl = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
__device__ __forceinline__ void split2_pair(float x0, float x1, uint32_t& ph, uint32_t& pm) {
    ph = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
    pm = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
}
// four consecutive-k floats of one row -> one 8-byte store per plane
template <int NPL, bool F16, int PLANE = XPLANE>
__device__ __forceinline__ void split_store4(__bf16* plane0, int off, f32x4 v, float s) {
    if (F16) {
        uint32_t h0, l0, h1, l1;
        splith_pair(v.x, v.y, s, h0, l0);
        splith_pair(v.z, v.w, s, h1, l1);
        *(uint2*)(plane0 + off) = make_uint2(h0, h1);
        *(uint2*)(plane0 + PLANE + off) = make_uint2(l0, l1);
    } else if (NPL == 3) {
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        *(uint2*)(plane0 + off) = make_uint2(h0, h1);
        *(uint2*)(plane0 + PLANE + off) = make_uint2(m0, m1);
        *(uint2*)(plane0 + 2 * PLANE + off) = make_uint2(l0, l1);
    } else {
        uint32_t h0, m0, h1, m1;
        split2_pair(v.x, v.y, h0, m0);
        split2_pair(v.z, v.w, h1, m1);
        *(uint2*)(plane0 + off) = make_uint2(h0, h1);
        *(uint2*)(plane0 + PLANE + off) = make_uint2(m0, m1);
    }
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma_x(f32x4 a, f32x4 b, f32x16 c) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// WNT = 32-column MFMA tiles per wave along n: 2 -> 128 x 128 workgroup tile (64 x 64 per wave, 3 workgroups per CU);
// 4 -> 128 x 256 (64 x 128 per wave: 12 fragment reads per 24 MFMAs instead of 8 per 12, the A operand -- fp32, split on
// the fly -- loaded, split and stored once per 256 instead of 128 output columns; 60 KB LDS, two workgroups per CU).
template <bool A_KC, bool B_KC, bool A_PRE, bool B_PRE, int NPL, bool F16 = false, int WNT = 2>
__global__ __launch_bounds__(256, 2) void gemm_split_mfma(const GemmArgs p, const GemmPlanes q) {
    static_assert(!A_PRE || A_KC, "pre-split operands are k-contiguous");
    static_assert(!B_PRE || B_KC, "pre-split operands are k-contiguous");
    static_assert(!F16 || NPL == 2, "the fp16 engine has two planes per operand");
    constexpr int BN = 64 * WNT;                   // workgroup tile columns
    constexpr int RB = 2 * WNT;                    // float4 (or 4 x 4 micro-block rows) per thread of a B k-tile
    constexpr int XOPER = NPL * XPLANE;
    constexpr int XPLANE_B = BN * XRS;             // B plane: BN rows
    static_assert(WNT == 2 || WNT == 4, "128 or 256 columns");
    static_assert(WNT == 2 || NPL == 2, "the wide tile is built for two-term engines");
    __shared__ __attribute__((aligned(16))) __bf16 smem[NPL * (XPLANE + XPLANE_B)];    // 128 cols: 61 440 B (NPL 3) / 40 960 B (NPL 2); 256 cols: 61 440 B; the epilogue needs 18 KB of it
    __bf16* As = smem;
    __bf16* Bs = smem + XOPER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    const int m0 = (lb / p.nbn) * GBM, n0 = (lb % p.nbn) * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    f32x16 acc[2][WNT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float sa = 1.f, sb = 1.f;                // fp16 engine: per-tensor power-of-two scales from the partial maxima
    if (F16) {
        float ma = 0.f, mb = 0.f;
        for (int i = tid; i < q.a_namax; i += 256) ma = fmaxf(ma, q.a_amax[i]);
        for (int i = tid; i < q.b_namax; i += 256) mb = fmaxf(mb, q.b_amax[i]);
        ma = wave_max(ma); mb = wave_max(mb);
        float* red = (float*)smem;
        if (lane == 0) { red[wave] = ma; red[4 + wave] = mb; }
        __syncthreads();
        sa = f16_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        sb = f16_scale_of(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        __syncthreads();
    }

    // ---- staging through BUFFER loads: one resource descriptor per operand (plane), a per-thread byte offset
    // computed once, and the k position as the instruction's SCALAR offset -- no per-tile address arithmetic, and
    // reads past the end of an operand return 0 instead of faulting.  Rows >= M (N) of a tile may therefore hold
    // whatever lies behind the operand inside its extent: they only reach accumulator rows that are never stored.
    // Only a partial last k-tile of a k-contiguous operand needs zeroing (uniform branch, tail tile only); a
    // k-strided operand runs off the end of its extent there and reads zeros.
    //  fp32 k-contiguous: 4 float4 per thread, f = tid + 256 r -> (row f>>3, k 4*(f&7))
    //  fp32 k-strided   : one 4(k) x 4(m) micro-block per thread: k4 = tid>>5, m4 = tid&31; load r = k row
    //  pre-split planes : per plane 128 rows x 4 chunks of 8 halves; f = tid + 256 r (r < 2) -> (row f>>2, chunk f&3)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    struct Stage {                           // one k-tile in flight between global memory and LDS
        f32x4 ra[4], rb[RB];                 // fp32 operand registers
        f32x4 qa[A_PRE ? 2 * NPL : 1], qb[B_PRE ? WNT * NPL : 1];   // pre-split operand registers (16 B = 8 x 16 bit each)
    };
    Stage R0;
    __amdgpu_buffer_rsrc_t rsA[A_PRE ? NPL : 1], rsB[B_PRE ? NPL : 1];
    uint32_t voa[4], vob[RB];                // byte offsets of this thread's loads at k = 0
    int ka[4], kb[RB];
    if (A_PRE) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) rsA[pl] = make_rsrc(q.Ap + (size_t)pl * q.a_pstride, p.a_bytes >> 1);
    } else {
        rsA[0] = make_rsrc(p.A, p.a_bytes);
    }
    if (B_PRE) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) rsB[pl] = make_rsrc(q.Bp + (size_t)pl * q.b_pstride, p.b_bytes >> 1);
    } else {
        rsB[0] = make_rsrc(p.B, p.b_bytes);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = tid + 256 * r;
        if (A_PRE) {
            ka[r] = (f & 3) << 3;
            voa[r] = ((uint32_t)(m0 + (f >> 2)) * (uint32_t)p.lda + ka[r]) * 2u;
        } else if (A_KC) {
            ka[r] = (f & 7) << 2;
            voa[r] = ((uint32_t)(m0 + (f >> 3)) * (uint32_t)p.lda + ka[r]) * 4u;
        } else {
            ka[r] = ((tid >> 5) << 2) + r;
            voa[r] = ((uint32_t)ka[r] * (uint32_t)p.lda + m0 + ((tid & 31) << 2)) * 4u;
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {          // B: BN rows (k-contiguous forms) or BN/128 micro-blocks of 128 columns (k-strided)
        const int f = tid + 256 * r;
        if (B_PRE) {
            kb[r] = (f & 3) << 3;
            vob[r] = ((uint32_t)(n0 + (f >> 2)) * (uint32_t)p.ldb + kb[r]) * 2u;
        } else if (B_KC) {
            kb[r] = (f & 7) << 2;
            vob[r] = ((uint32_t)(n0 + (f >> 3)) * (uint32_t)p.ldb + kb[r]) * 4u;
        } else {
            kb[r] = ((tid >> 5) << 2) + (r & 3);
            vob[r] = ((uint32_t)kb[r] * (uint32_t)p.ldb + n0 + 128 * (r >> 2) + ((tid & 31) << 2)) * 4u;
        }
    }
    auto gload = [&](Stage& R, int k0) {
        const uint32_t sa_off = A_PRE ? (uint32_t)k0 * 2u : (A_KC ? (uint32_t)k0 * 4u : (uint32_t)k0 * (uint32_t)p.lda * 4u);
        const uint32_t sb_off = B_PRE ? (uint32_t)k0 * 2u : (B_KC ? (uint32_t)k0 * 4u : (uint32_t)k0 * (uint32_t)p.ldb * 4u);
        if (A_PRE) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) R.qa[r * NPL + pl] = buf_load4(rsA[pl], voa[r], sa_off);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) R.ra[r] = buf_load4(rsA[0], voa[r], sa_off);
        }
        if (B_PRE) {
#pragma unroll
            for (int r = 0; r < WNT; ++r)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) R.qb[r * NPL + pl] = buf_load4(rsB[pl], vob[r], sb_off);
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r) R.rb[r] = buf_load4(rsB[0], vob[r], sb_off);
        }
    };
    // registers -> 16-bit planes in LDS (rows = m or n).  TAIL: the tile crosses kend (last tile, K % 32 != 0)
    auto lstore = [&](const Stage& R, int k0, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        if (A_PRE) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + 256 * r;
                const bool ok = !TAIL || k0 + ka[r] < kend;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *(f32x4*)(As + pl * XPLANE + (f >> 2) * XRS + ((f & 3) << 3)) = ok ? R.qa[r * NPL + pl] : zero4;
            }
        } else if (A_KC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = tid + 256 * r;
                split_store4<NPL, F16>(As, (f >> 3) * XRS + ((f & 7) << 2), (!TAIL || k0 + ka[r] < kend) ? R.ra[r] : zero4, sa);
            }
        } else {        // ra[r] = row k (4*k4 + r), columns m = 4*m4 .. +3  -> transpose 4x4 in registers
            const int mrow = (tid & 31) << 2, kcol = (tid >> 5) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split_store4<NPL, F16>(As, (mrow + j) * XRS + kcol, f32x4{R.ra[0][j], R.ra[1][j], R.ra[2][j], R.ra[3][j]}, sa);
        }
        if (B_PRE) {
#pragma unroll
            for (int r = 0; r < WNT; ++r) {
                const int f = tid + 256 * r;
                const bool ok = !TAIL || k0 + kb[r] < kend;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *(f32x4*)(Bs + pl * XPLANE_B + (f >> 2) * XRS + ((f & 3) << 3)) = ok ? R.qb[r * NPL + pl] : zero4;
            }
        } else if (B_KC) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int f = tid + 256 * r;
                split_store4<NPL, F16, XPLANE_B>(Bs, (f >> 3) * XRS + ((f & 7) << 2), (!TAIL || k0 + kb[r] < kend) ? R.rb[r] : zero4, sb);
            }
        } else {
            const int nrow = (tid & 31) << 2, kcol = (tid >> 5) << 2;
#pragma unroll
            for (int b = 0; b < WNT / 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    split_store4<NPL, F16, XPLANE_B>(Bs, (128 * b + nrow + j) * XRS + kcol,
                                                     f32x4{R.rb[4 * b][j], R.rb[4 * b + 1][j], R.rb[4 * b + 2][j], R.rb[4 * b + 3][j]}, sb);
        }
    };
    auto lstore_at = [&](const Stage& R, int k0) {
        if (k0 + GBK <= kend) lstore(R, k0, std::false_type{});
        else lstore(R, k0, std::true_type{});
    };

    auto mma = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two K=16 steps per tile
            f32x4 fa[2][NPL], fb[WNT][NPL];   // 8 sixteen-bit k values per lane
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    fa[t][pl] = *(const f32x4*)(As + pl * XPLANE + (wm * 64 + t * 32 + li) * XRS + s * 16 + lh * 8);
#pragma unroll
            for (int t = 0; t < WNT; ++t)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    fb[t][pl] = *(const f32x4*)(Bs + pl * XPLANE_B + (wn * 32 * WNT + t * 32 + li) * XRS + s * 16 + lh * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    f32x16 c = acc[i][j];
                    if (F16) {
                        c = mfma_x<true>(fa[i][1], fb[j][0], c);                // lo.hi
                        c = mfma_x<true>(fa[i][0], fb[j][1], c);                // hi.lo
                        c = mfma_x<true>(fa[i][0], fb[j][0], c);                // hi.hi
                    } else {
                        if (NPL == 3) {         // smallest terms first
                            c = mfma_x<false>(fa[i][1], fb[j][1], c);           // mid.mid
                            c = mfma_x<false>(fa[i][NPL - 1], fb[j][0], c);     // lo.hi
                            c = mfma_x<false>(fa[i][0], fb[j][NPL - 1], c);     // hi.lo
                        }
                        c = mfma_x<false>(fa[i][1], fb[j][0], c);               // mid.hi
                        c = mfma_x<false>(fa[i][0], fb[j][1], c);               // hi.mid
                        c = mfma_x<false>(fa[i][0], fb[j][0], c);               // hi.hi
                    }
                    acc[i][j] = c;
                }
        }
    };
#ifdef SEGMM_GEMM_TRACE
    const bool trace_on = blockIdx.x == gridDim.x / 2 + 3 && blockIdx.z == 0 && tid == 0;
    int it = 0;
#endif
    gload(R0, kbeg);
    lstore_at(R0, kbeg);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        // One tile in flight, three workgroups per CU (NPL 2: 166 registers, 40 KB LDS).  Measured alternatives
        // (profiles/README.md): two tiles in flight at two workgroups per CU -10 %; the split moved under the MFMAs of the
        // current tile (packed registers, one or two LDS buffers) -3...-10 %; loads/VALU/LDS-refill removed one at a time
        // (timing ablations) bound what any staging change can win at +16 % (B loads), +7 % (A loads), +19 % (split + refill).
        TRACE_MARK(it, 0);
        gload(R0, k0 + GBK);                   // next tile L2/HBM -> registers, lands under the MFMAs
        __builtin_amdgcn_sched_barrier(0);     // (the scheduler would otherwise sink the loads below the MFMAs to save registers)
        TRACE_MARK(it, 1);
#if SEGMM_GEMM_SETPRIO
        __builtin_amdgcn_s_setprio(SEGMM_GEMM_SETPRIO);
#endif
        mma();
#if SEGMM_GEMM_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef SEGMM_GEMM_TRACE
        __builtin_amdgcn_sched_barrier(0);
        TRACE_MARK(it, 2);
#endif
        __syncthreads();
        TRACE_MARK(it, 3);
        if (k0 + GBK < kend) {
#ifdef SEGMM_GEMM_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TRACE_MARK(it, 4);
#endif
            lstore_at(R0, k0 + GBK);
#ifdef SEGMM_GEMM_TRACE
            __builtin_amdgcn_sched_barrier(0);
            TRACE_MARK(it, 5);
#endif
            __syncthreads();
            TRACE_MARK(it, 6);
        }
#ifdef SEGMM_GEMM_TRACE
        ++it;
#endif
    }

    // ---- epilogue: identical to gemm_f32_mfma (the 32x32 C/D register map does not depend on the input type)
    float* Cs = (float*)smem + wave * (32 * 36);
    const bool split = gridDim.z > 1;
    float am = 0.f;
    float* Cout = p.C + (size_t)blockIdx.z * (size_t)p.slab_stride;
    const float inv_a = 1.f / sa, inv_b = 1.f / sb;      // exact powers of two
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int idx = lane + 64 * r4;
                const int row = idx >> 3, c4 = (idx & 7) << 2;
                const int gm = m0 + wm * 64 + i * 32 + row, gn = n0 + wn * 32 * WNT + j * 32 + c4;
                if (gm < p.M && gn < p.N) {
                    f32x4 v = *(const f32x4*)(Cs + row * 36 + c4);
                    if (F16) v = (v * inv_a) * inv_b;
                    if (!split) {
                        if (p.row_scale) v *= p.row_scale[gm];
                        if (p.bias) v += *(const f32x4*)(p.bias + gn);
                        if (p.epi == EPI_GELU) {
                            *(f32x4*)(p.aux + (size_t)gm * p.ldaux + gn) = v;
                            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                        } else if (p.epi == EPI_DGELU) {
                            const f32x4 g = *(const f32x4*)(p.aux + (size_t)gm * p.ldaux + gn);
                            v.x *= gelu_erf_grad(g.x); v.y *= gelu_erf_grad(g.y);
                            v.z *= gelu_erf_grad(g.z); v.w *= gelu_erf_grad(g.w);
                        } else if (p.epi == EPI_RELU) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        } else if (p.epi == EPI_DRELU) {      // aux = the forward OUTPUT (post-ReLU, post-dropout): > 0 iff z > 0 and kept
                            const f32x4 g = *(const f32x4*)(p.aux + (size_t)gm * p.ldaux + gn);
                            v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f;
                            v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
                        }
                        if (p.drop.p > 0.f) v = drop_apply4(drop_live(p.drop), ((uint64_t)gm * (uint64_t)p.N + gn) >> 2, v);
                        if (p.residual) v += *(const f32x4*)(p.residual + (size_t)(gm % p.res_period) * p.ldr + gn);
                    }
                    *(f32x4*)(Cout + (size_t)gm * p.ldc + gn) = v;
                    am = absmax4(am, v);
                }
            }
            __syncthreads();
        }
    }
    if (p.amax_out) amax_commit(p.amax_out, am, blockIdx.x * 4 + wave);
}

// ---------------------------------------------------------------- producers of pre-split planes
// planes[p][i] = p-th bf16 term of x[i]   (flat, e.g. the whole parameter buffer once per optimizer step)
__global__ __launch_bounds__(256) void split3_flat_kernel(const float* __restrict__ x, __bf16* __restrict__ planes,
                                                          long long n, long long pstride) {
    for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *(const f32x4*)(x + i);
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        *(uint2*)(planes + i) = make_uint2(h0, h1);
        *(uint2*)(planes + pstride + i) = make_uint2(m0, m1);
        *(uint2*)(planes + 2 * pstride + i) = make_uint2(l0, l1);
    }
}
// fp16 engine: planes[0/1][i] = hi / lo of x[i] * s, s from the partial maxima (one scale for the whole buffer)
__global__ __launch_bounds__(256) void splith_flat_kernel(const float* __restrict__ x, __bf16* __restrict__ planes,
                                                          long long n, long long pstride, const float* __restrict__ amax, int namax) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < namax; i += 256) m = fmaxf(m, amax[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float s = f16_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *(const f32x4*)(x + i);
        uint32_t h0, l0, h1, l1;
        splith_pair(v.x, v.y, s, h0, l0);
        splith_pair(v.z, v.w, s, h1, l1);
        *(uint2*)(planes + i) = make_uint2(h0, h1);
        *(uint2*)(planes + pstride + i) = make_uint2(l0, l1);
    }
}
// partial maxima of |x| over an [rows, cols] view (row stride ld): out[blockIdx.x], gridDim.x partials
__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ x, long long rows, int cols, int ld,
                                                             float* __restrict__ out) {
    __shared__ float red[4];
    const int c4n = cols >> 2;
    const long long n4 = rows * c4n;
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4n;
        const int c = (int)(i - r * c4n) << 2;
        const f32x4 v = *(const f32x4*)(x + r * ld + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// planes[p][c * R + r] = p-th term of x[r * ld + c]   (transposed copy of an [R, C] matrix; 32 x 32 tiles through LDS)
template <bool F16>
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ x, int R, int Cc, int ld,
                                                               __bf16* __restrict__ planes, long long pstride,
                                                               const float* __restrict__ amax, int namax) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    float s = 1.f;
    if (F16) {
        float m = 0.f;
        for (int i = threadIdx.x; i < namax; i += 256) m = fmaxf(m, amax[i]);
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        s = f16_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    }
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < Cc) ? x[(size_t)r * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;                // output row = c, column = r (contiguous over tx)
        if (c < Cc && r < R) {
            const float v = tile[tx][ty + 8 * k];
            const size_t o = (size_t)c * R + r;
            if (F16) {
                const _Float16 h = (_Float16)(v * s);
                const _Float16 l = (_Float16)__builtin_fmaf(v, s, -(float)h);
                ((_Float16*)planes)[o] = h; ((_Float16*)planes)[pstride + o] = l;
            } else {
                const __bf16 h = (__bf16)v;
                const float r1 = v - (float)h;
                const __bf16 m = (__bf16)r1;
                const __bf16 l = (__bf16)(r1 - (float)m);
                planes[o] = h; planes[pstride + o] = m; planes[2 * pstride + o] = l;
            }
        }
    }
}

}  // namespace segmm
