// fp32-accurate GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16
// terms (x = hi + mid + lo, 8 + 8 + 8 mantissa bits) and the product is formed from the six largest
// partial products, accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//
//     a.b ~= ah.bh + (ah.bm + am.bh) + (ah.bl + al.bh + am.bm)        dropped terms <= 2^-24 |a||b|
//
// bf16 x bf16 products are exact in fp32, so the only errors are the dropped 2^-24 terms and the fp32
// accumulation itself: measured error vs fp64 is BELOW that of the f32-MFMA GEMM on every layout
// (tests/test_ops_gpu.py::test_gemm_bf16x6_*).  Cost: 6 bf16 MFMAs (32 cycles each, K = 16) replace 8 f32
// MFMAs (64 cycles each, K = 2): 192 vs 512 matrix-pipe cycles per 32 x 32 x 16 block = 2.67x the fp32-MFMA
// roofline (SURVEY.md §7 "hard parts": split-bf16, decided by measurement).
//
// Operands arrive either as fp32 (split ON THE FLY while a k-tile moves registers -> LDS, 6 VALU ops per
// element) or PRE-SPLIT as bf16 planes [NPL][rows][ld] written once by their producer (weights: one split
// pass per optimizer step) -- then the main loop has no VALU work for that operand at all.
// NPL = 3 gives the six-product form above; NPL = 2 (hi, mid only; products hh + hm + mh, error ~2e-5) is
// offered for weight gradients, which are leaves of the backward graph (nothing propagates their error).
//
// Tile 128 x 128 x 32, 256 threads (2 x 2 waves, 64 x 64 each), 2 workgroups per CU.  LDS: per operand NPL
// planes [128 rows][32 k] of bf16 with an 80-byte row stride (conflict-free ds_read_b128 of the 8-element
// MFMA fragments).  fp32 operands whose k index is NOT contiguous in memory (B of NN, A and B of TN) are
// transposed in registers: a thread owns a 4(k) x 4(m) micro-block and writes 4-element k runs.
#pragma once
#include "gemm.h"

namespace segmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XRS = 40;                          // plane row stride in bf16 elements (80 bytes)
constexpr int XPLANE = GBM * XRS;                // 5120 bf16 per plane

struct GemmPlanes {                              // optional pre-split operands (bf16 planes, k-contiguous rows)
    const __bf16* Ap; long long a_pstride;       // plane p of A at Ap + p * a_pstride, row stride = GemmArgs.lda
    const __bf16* Bp; long long b_pstride;
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
template <int NPL>
__device__ __forceinline__ void split_store4(__bf16* plane0, int off, f32x4 v) {
    if (NPL == 3) {
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        *(uint2*)(plane0 + off) = make_uint2(h0, h1);
        *(uint2*)(plane0 + XPLANE + off) = make_uint2(m0, m1);
        *(uint2*)(plane0 + 2 * XPLANE + off) = make_uint2(l0, l1);
    } else {
        uint32_t h0, m0, h1, m1;
        split2_pair(v.x, v.y, h0, m0);
        split2_pair(v.z, v.w, h1, m1);
        *(uint2*)(plane0 + off) = make_uint2(h0, h1);
        *(uint2*)(plane0 + XPLANE + off) = make_uint2(m0, m1);
    }
}

template <bool A_KC, bool B_KC, bool A_PRE, bool B_PRE, int NPL>
__global__ __launch_bounds__(256) void gemm_bf16x6_mfma(const GemmArgs p, const GemmPlanes q) {
    static_assert(!A_PRE || A_KC, "pre-split operands are k-contiguous");
    static_assert(!B_PRE || B_KC, "pre-split operands are k-contiguous");
    constexpr int XOPER = NPL * XPLANE;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 3 * XPLANE];      // 61 440 B (epilogue needs 18 KB of it)
    __bf16* As = smem;
    __bf16* Bs = smem + XOPER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    const int m0 = (lb / p.nbn) * GBM, n0 = (lb % p.nbn) * GBN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging coordinates (clamped addresses + zeroing select, branch-free like gemm_f32_mfma)
    //  fp32 k-contiguous: 4 float4 per thread, f = tid + 256 r -> (row f>>3, k 4*(f&7))
    //  fp32 k-strided   : one 4(k) x 4(m) micro-block per thread: k4 = tid>>5, m4 = tid&31; load r = k row
    //  pre-split planes : per plane 128 rows x 4 chunks of 8 bf16; f = tid + 256 r (r < 2) -> (row f>>2, chunk f&3)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[4], rb[4];                      // fp32 operand registers
    f32x4 qa[A_PRE ? 2 * NPL : 1], qb[B_PRE ? 2 * NPL : 1];     // pre-split operand registers (16 B = 8 bf16 each)
    const float* pa[4];
    const float* pb[4];
    const __bf16* pqa[2];
    const __bf16* pqb[2];
    int ka[4], kb[4];
    bool va[4], vb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = tid + 256 * r;
        if (A_PRE) {
            if (r < 2) {
                const int gm = m0 + (f >> 2);
                ka[r] = (f & 3) << 3; va[r] = gm < p.M;
                pqa[r] = q.Ap + (size_t)min(gm, p.M - 1) * p.lda;
            }
        } else if (A_KC) {
            const int gm = m0 + (f >> 3);
            ka[r] = (f & 7) << 2; va[r] = gm < p.M;
            pa[r] = p.A + (size_t)min(gm, p.M - 1) * p.lda;
        } else {
            const int gm = m0 + ((tid & 31) << 2);
            ka[r] = ((tid >> 5) << 2) + r; va[r] = gm < p.M;
            pa[r] = p.A + min(gm, p.M - 4);
        }
        if (B_PRE) {
            if (r < 2) {
                const int gn = n0 + (f >> 2);
                kb[r] = (f & 3) << 3; vb[r] = gn < p.N;
                pqb[r] = q.Bp + (size_t)min(gn, p.N - 1) * p.ldb;
            }
        } else if (B_KC) {
            const int gn = n0 + (f >> 3);
            kb[r] = (f & 7) << 2; vb[r] = gn < p.N;
            pb[r] = p.B + (size_t)min(gn, p.N - 1) * p.ldb;
        } else {
            const int gn = n0 + ((tid & 31) << 2);
            kb[r] = ((tid >> 5) << 2) + r; vb[r] = gn < p.N;
            pb[r] = p.B + min(gn, p.N - 4);
        }
    }
    const int kclampA = A_PRE ? kend - 8 : (A_KC ? kend - 4 : kend - 1);
    const int kclampB = B_PRE ? kend - 8 : (B_KC ? kend - 4 : kend - 1);
    auto gload = [&](int k0) {
        if (A_PRE) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    qa[r * NPL + pl] = *(const f32x4*)(pqa[r] + (size_t)pl * q.a_pstride + min(k0 + ka[r], kclampA));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gk = min(k0 + ka[r], kclampA);
                ra[r] = A_KC ? *(const f32x4*)(pa[r] + gk) : *(const f32x4*)(pa[r] + (size_t)gk * p.lda);
            }
        }
        if (B_PRE) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    qb[r * NPL + pl] = *(const f32x4*)(pqb[r] + (size_t)pl * q.b_pstride + min(k0 + kb[r], kclampB));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gk = min(k0 + kb[r], kclampB);
                rb[r] = B_KC ? *(const f32x4*)(pb[r] + gk) : *(const f32x4*)(pb[r] + (size_t)gk * p.ldb);
            }
        }
    };
    // registers -> bf16 planes in LDS (rows = m or n)
    auto lstore = [&](int k0) {
        if (A_PRE) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + 256 * r;
                const bool ok = va[r] && k0 + ka[r] < kend;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *(f32x4*)(As + pl * XPLANE + (f >> 2) * XRS + ((f & 3) << 3)) = ok ? qa[r * NPL + pl] : zero4;
            }
        } else if (A_KC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = tid + 256 * r;
                split_store4<NPL>(As, (f >> 3) * XRS + ((f & 7) << 2), (va[r] && k0 + ka[r] < kend) ? ra[r] : zero4);
            }
        } else {        // ra[r] = row k (4*k4 + r), columns m = 4*m4 .. +3  -> transpose 4x4 in registers
            f32x4 x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (va[r] && k0 + ka[r] < kend) ? ra[r] : zero4;
            const int mrow = (tid & 31) << 2, kcol = (tid >> 5) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) split_store4<NPL>(As, (mrow + j) * XRS + kcol, f32x4{x[0][j], x[1][j], x[2][j], x[3][j]});
        }
        if (B_PRE) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + 256 * r;
                const bool ok = vb[r] && k0 + kb[r] < kend;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *(f32x4*)(Bs + pl * XPLANE + (f >> 2) * XRS + ((f & 3) << 3)) = ok ? qb[r * NPL + pl] : zero4;
            }
        } else if (B_KC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = tid + 256 * r;
                split_store4<NPL>(Bs, (f >> 3) * XRS + ((f & 7) << 2), (vb[r] && k0 + kb[r] < kend) ? rb[r] : zero4);
            }
        } else {
            f32x4 x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (vb[r] && k0 + kb[r] < kend) ? rb[r] : zero4;
            const int nrow = (tid & 31) << 2, kcol = (tid >> 5) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) split_store4<NPL>(Bs, (nrow + j) * XRS + kcol, f32x4{x[0][j], x[1][j], x[2][j], x[3][j]});
        }
    };

    gload(kbeg);
    lstore(kbeg);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        gload(k0 + GBK);                       // next tile HBM/L2 -> registers (clamped past the end), lands under the MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two K=16 steps per tile
            bf16x8 fa[2][NPL], fb[2][NPL];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    fa[t][pl] = *(const bf16x8*)(As + pl * XPLANE + (wm * 64 + t * 32 + li) * XRS + s * 16 + lh * 8);
                    fb[t][pl] = *(const bf16x8*)(Bs + pl * XPLANE + (wn * 64 + t * 32 + li) * XRS + s * 16 + lh * 8);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    if (NPL == 3) {             // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);           // mid.mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][NPL - 1], fb[j][0], c, 0, 0, 0);     // lo.hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NPL - 1], c, 0, 0, 0);     // hi.lo
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);               // mid.hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);               // hi.mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);               // hi.hi
                    acc[i][j] = c;
                }
        }
        __syncthreads();
        if (k0 + GBK < kend) {
            lstore(k0 + GBK);
            __syncthreads();
        }
    }

    // ---- epilogue: identical to gemm_f32_mfma (the 32x32 C/D register map does not depend on the input type)
    float* Cs = (float*)smem + wave * (32 * 36);
    const bool split = gridDim.z > 1;
    float* Cout = p.C + (size_t)blockIdx.z * (size_t)p.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int idx = lane + 64 * r4;
                const int row = idx >> 3, c4 = (idx & 7) << 2;
                const int gm = m0 + wm * 64 + i * 32 + row, gn = n0 + wn * 64 + j * 32 + c4;
                if (gm < p.M && gn < p.N) {
                    f32x4 v = *(const f32x4*)(Cs + row * 36 + c4);
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
                        }
                        if (p.drop.p > 0.f) v = drop_apply4(p.drop, ((uint64_t)gm * (uint64_t)p.N + gn) >> 2, v);
                        if (p.residual) v += *(const f32x4*)(p.residual + (size_t)(gm % p.res_period) * p.ldr + gn);
                    }
                    *(f32x4*)(Cout + (size_t)gm * p.ldc + gn) = v;
                }
            }
            __syncthreads();
        }
    }
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
// planes[p][c * R + r] = p-th term of x[r * ld + c]   (transposed copy of an [R, C] matrix; 32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ x, int R, int Cc, int ld,
                                                               __bf16* __restrict__ planes, long long pstride) {
    __shared__ float tile[32][33];
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
            const __bf16 h = (__bf16)v;
            const float r1 = v - (float)h;
            const __bf16 m = (__bf16)r1;
            const __bf16 l = (__bf16)(r1 - (float)m);
            const size_t o = (size_t)c * R + r;
            planes[o] = h; planes[pstride + o] = m; planes[2 * pstride + o] = l;
        }
    }
}

}  // namespace segmm
