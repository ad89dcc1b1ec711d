// fp32-accurate GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16
// terms (x = hi + mid + lo, 8 + 8 + 8 mantissa bits) and the product is formed from the six largest
// partial products, accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//
//     a.b ~= ah.bh + (ah.bm + am.bh) + (ah.bl + al.bh + am.bm)        dropped terms <= 2^-24 |a||b|
//
// bf16 x bf16 products are exact in fp32, so the only errors are the dropped 2^-24 terms and the fp32
// accumulation itself: measured error vs fp64 is BELOW that of a plain fp32 GEMM (oracle emulation:
// 1.7e-6 vs 3.6e-6 of the mean |C| at K = 768).  Cost: 6 bf16 MFMAs (32 cycles each, K = 16) replace
// 8 f32 MFMAs (64 cycles each, K = 2): 192 vs 512 matrix-pipe cycles per 32 x 32 x 16 block = 2.67x
// the fp32-MFMA roofline (SURVEY.md §7 "hard parts": split-bf16, decided by measurement).
//
// The split is done ON THE FLY while a k-tile moves registers -> LDS (6 VALU ops per element, issued
// in the shadow of the other workgroup's MFMAs), so the kernel has exactly the interface, operand
// layouts (NT / NN / TN), epilogue and split-K behaviour of gemm_f32_mfma and no tensor changes format.
//
// Tile 128 x 128 x 32, 256 threads (2 x 2 waves, 64 x 64 each), 2 workgroups per CU.  LDS: per operand
// three planes [128 rows][32 k] of bf16 with an 80-byte row stride (conflict-free ds_read_b128 of the
// 8-element MFMA fragments).  Operands whose k index is NOT contiguous in memory (B of NN, A and B of
// TN) are transposed in registers: a thread owns a 4(k) x 4(m) micro-block and writes 4-element k runs.
#pragma once
#include "gemm.h"

namespace segmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XRS = 40;                          // plane row stride in bf16 elements (80 bytes)
constexpr int XPLANE = GBM * XRS;                // 5120 bf16 per plane
constexpr int XOPER = 3 * XPLANE;                // one operand: hi | mid | lo

// exact 3-way split of two floats -> packed (hi0,hi1), (mid0,mid1), (lo0,lo1)
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& ph, uint32_t& pm, uint32_t& pl) {
    ph = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
    pm = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
    pl = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
// four consecutive-k floats of one row -> one 8-byte store per plane
__device__ __forceinline__ void split3_store4(__bf16* plane0, int off, f32x4 v) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3_pair(v.x, v.y, h0, m0, l0);
    split3_pair(v.z, v.w, h1, m1, l1);
    *(uint2*)(plane0 + off) = make_uint2(h0, h1);
    *(uint2*)(plane0 + XPLANE + off) = make_uint2(m0, m1);
    *(uint2*)(plane0 + 2 * XPLANE + off) = make_uint2(l0, l1);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_bf16x6_mfma(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * XOPER];       // 61 440 B
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
    //  k-contiguous operand: 4 float4 per thread, f = tid + 256 r -> (row f>>3, k 4*(f&7))
    //  k-strided operand   : one 4(k) x 4(m) micro-block per thread: k4 = tid>>5, m4 = tid&31; load r = k row
    f32x4 ra[4], rb[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* pa[4];
    const float* pb[4];
    int ka[4], kb[4];
    bool va[4], vb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = tid + 256 * r;
        if (A_KC) {
            const int gm = m0 + (f >> 3);
            ka[r] = (f & 7) << 2; va[r] = gm < p.M;
            pa[r] = p.A + (size_t)min(gm, p.M - 1) * p.lda;
        } else {
            const int gm = m0 + ((tid & 31) << 2);
            ka[r] = ((tid >> 5) << 2) + r; va[r] = gm < p.M;
            pa[r] = p.A + min(gm, p.M - 4);
        }
        if (B_KC) {
            const int gn = n0 + (f >> 3);
            kb[r] = (f & 7) << 2; vb[r] = gn < p.N;
            pb[r] = p.B + (size_t)min(gn, p.N - 1) * p.ldb;
        } else {
            const int gn = n0 + ((tid & 31) << 2);
            kb[r] = ((tid >> 5) << 2) + r; vb[r] = gn < p.N;
            pb[r] = p.B + min(gn, p.N - 4);
        }
    }
    const int kclampA = A_KC ? kend - 4 : kend - 1, kclampB = B_KC ? kend - 4 : kend - 1;
    auto gload = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gka = k0 + ka[r], gkb = k0 + kb[r];
            ra[r] = A_KC ? *(const f32x4*)(pa[r] + min(gka, kclampA)) : *(const f32x4*)(pa[r] + (size_t)min(gka, kclampA) * p.lda);
            rb[r] = B_KC ? *(const f32x4*)(pb[r] + min(gkb, kclampB)) : *(const f32x4*)(pb[r] + (size_t)min(gkb, kclampB) * p.ldb);
        }
    };
    // registers -> three bf16 planes in LDS (rows = m or n, 4-element k runs)
    auto lstore = [&](int k0) {
        f32x4 xa[4], xb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xa[r] = (va[r] && k0 + ka[r] < kend) ? ra[r] : zero4;
            xb[r] = (vb[r] && k0 + kb[r] < kend) ? rb[r] : zero4;
        }
        if (A_KC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = tid + 256 * r;
                split3_store4(As, (f >> 3) * XRS + ((f & 7) << 2), xa[r]);
            }
        } else {        // xa[r] = row k (4*k4 + r), columns m = 4*m4 .. +3  -> transpose 4x4 in registers
            const int mrow = (tid & 31) << 2, kcol = (tid >> 5) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split3_store4(As, (mrow + j) * XRS + kcol, f32x4{xa[0][j], xa[1][j], xa[2][j], xa[3][j]});
        }
        if (B_KC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = tid + 256 * r;
                split3_store4(Bs, (f >> 3) * XRS + ((f & 7) << 2), xb[r]);
            }
        } else {
            const int nrow = (tid & 31) << 2, kcol = (tid >> 5) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split3_store4(Bs, (nrow + j) * XRS + kcol, f32x4{xb[0][j], xb[1][j], xb[2][j], xb[3][j]});
        }
    };

    gload(kbeg);
    lstore(kbeg);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        gload(k0 + GBK);                       // next tile HBM/L2 -> registers (clamped past the end), lands under the MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two K=16 steps per tile
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    fa[t][pl] = *(const bf16x8*)(As + pl * XPLANE + (wm * 64 + t * 32 + li) * XRS + s * 16 + lh * 8);
                    fb[t][pl] = *(const bf16x8*)(Bs + pl * XPLANE + (wn * 64 + t * 32 + li) * XRS + s * 16 + lh * 8);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);   // mid.mid   (smallest first)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);   // lo.hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);   // hi.lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);   // mid.hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);   // hi.mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);   // hi.hi
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

}  // namespace segmm
