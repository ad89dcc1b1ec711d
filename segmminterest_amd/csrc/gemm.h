// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// The segment-interest path is twelve d x d Linear projections per encoder layer;
// the 1e-4 logit tolerance rules out plain bf16 MFMA, so every contraction runs on the f32-input
// MFMA.  One kernel template covers the three operand layouts the forward / dgrad / wgrad need:
//
//   NT  C[M,N] = A[M,K] . B[N,K]^T      forward of nn.Linear (weight is [out,in])
//   NN  C[M,N] = A[M,K] . B[K,N]        dgrad: dX = dY . W
//   TN  C[M,N] = A[K,M]^T . B[K,N]      wgrad: dW = dY^T . X   (split-K over the token dimension)
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles,
// 64 accumulator VGPRs), 2 workgroups per CU.  Two-stage LDS ring with one barrier per k-tile; tile
// t+2 travels HBM -> registers while tile t+1 moves registers -> LDS and tile t is multiplied; LDS
// rows are padded (+4 floats) so the ds_read_b128 fragment reads are bank-conflict free.  Because the
// f32 MFMA takes ONE scalar per lane per operand, the k order inside an 8-wide chunk is permuted
// (lane half h supplies k = 4h..4h+3) so a lane fetches its four k values with one ds_read_b128.
// The epilogue restages each 32 x 32 accumulator tile through LDS so that global traffic is
// row-major float4 and one dropout-hash call serves 4 consecutive elements.
#pragma once
#include "common.h"

namespace segmm {

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int GLDK = GBK + 4;    // row stride of a k-contiguous LDS tile  [128][36]
constexpr int GLDM = GBM + 4;    // row stride of an m-contiguous LDS tile [32][132]
constexpr int GTILE = (GBM * GLDK > GBK * GLDM) ? GBM * GLDK : GBK * GLDM;   // 4608 floats

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_RELU = 3, EPI_DRELU = 4 };

struct GemmArgs {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;        // [N] or null
    const float* row_scale;   // [M] or null: acc *= row_scale[m] (fused L1 normalisation)
    const float* residual; int ldr; int res_period;   // v += residual[(m % res_period)*ldr + n]
    float* aux; int ldaux;    // EPI_GELU: pre-activation written; EPI_DGELU: pre-activation read
    int epi;
    DropCfg drop;             // applied after the activation, before the residual
    int k_per_split;          // K range handled by one blockIdx.z (multiple of GBK)
    long long slab_stride;    // split-K: partial C of split z at C + z*slab_stride, plain store
    int nbm, nbn;
    uint32_t a_bytes, b_bytes;  // extents of the A / B views in bytes of fp32 (split engines: buffer-load range check)
    float* amax_out;          // optional: AMAX_SLOTS partial maxima of |C| (after the epilogue), see common.h
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * GTILE];      // 2 stages x (A tile, B tile) = 73.7 KB
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

    // ---- staging coordinates of this thread (4 float4 of A and of B per k-tile), hoisted out of the loop.
    // Out-of-range rows / k are CLAMPED to a valid address and zeroed by a select, so the loop body is
    // branch-free straight-line code (one scheduling region per phase).
    f32x4 ra[4], rb[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* pa[4];
    const float* pb[4];
    int ka[4], kb[4];            // k offset of the element inside a k-tile
    bool va[4], vb[4];           // row (m / n) in range
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = tid + 256 * r;
        if (A_KC) {
            const int gm = m0 + (f >> 3);
            ka[r] = (f & 7) << 2; va[r] = gm < p.M;
            pa[r] = p.A + (size_t)min(gm, p.M - 1) * p.lda;
        } else {
            const int gm = m0 + ((f & 31) << 2);
            ka[r] = f >> 5; va[r] = gm < p.M;
            pa[r] = p.A + min(gm, p.M - 4);
        }
        if (B_KC) {
            const int gn = n0 + (f >> 3);
            kb[r] = (f & 7) << 2; vb[r] = gn < p.N;
            pb[r] = p.B + (size_t)min(gn, p.N - 1) * p.ldb;
        } else {
            const int gn = n0 + ((f & 31) << 2);
            kb[r] = f >> 5; vb[r] = gn < p.N;
            pb[r] = p.B + min(gn, p.N - 4);
        }
    }
    const int kclampA = A_KC ? kend - 4 : kend - 1, kclampB = B_KC ? kend - 4 : kend - 1;
    auto gload = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gka = k0 + ka[r], gkb = k0 + kb[r];
            const f32x4 xa = A_KC ? *(const f32x4*)(pa[r] + min(gka, kclampA))
                                  : *(const f32x4*)(pa[r] + (size_t)min(gka, kclampA) * p.lda);
            const f32x4 xb = B_KC ? *(const f32x4*)(pb[r] + min(gkb, kclampB))
                                  : *(const f32x4*)(pb[r] + (size_t)min(gkb, kclampB) * p.ldb);
            ra[r] = xa;          // raw: the zeroing select happens at lstore time, so nothing consumes
            rb[r] = xb;          // the loaded registers (=> no s_waitcnt vmcnt) until a whole tile later
        }
    };
    auto lstore = [&](int k0, float* As_, float* Bs_) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = tid + 256 * r;
            const f32x4 xa = (va[r] && k0 + ka[r] < kend) ? ra[r] : zero4;
            const f32x4 xb = (vb[r] && k0 + kb[r] < kend) ? rb[r] : zero4;
            if (A_KC) *(f32x4*)(As_ + (f >> 3) * GLDK + ((f & 7) << 2)) = xa;
            else      *(f32x4*)(As_ + (f >> 5) * GLDM + ((f & 31) << 2)) = xa;
            if (B_KC) *(f32x4*)(Bs_ + (f >> 3) * GLDK + ((f & 7) << 2)) = xb;
            else      *(f32x4*)(Bs_ + (f >> 5) * GLDM + ((f & 31) << 2)) = xb;
        }
    };
    auto read_frag = [&](const float* As_, const float* Bs_, int c, float (&af)[2][4], float (&bf)[2][4]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (A_KC) {
                const f32x4 v = *(const f32x4*)(As_ + (wm * 64 + t * 32 + li) * GLDK + c * 8 + lh * 4);
                af[t][0] = v.x; af[t][1] = v.y; af[t][2] = v.z; af[t][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) af[t][s] = As_[(c * 8 + lh * 4 + s) * GLDM + wm * 64 + t * 32 + li];
            }
            if (B_KC) {
                const f32x4 v = *(const f32x4*)(Bs_ + (wn * 64 + t * 32 + li) * GLDK + c * 8 + lh * 4);
                bf[t][0] = v.x; bf[t][1] = v.y; bf[t][2] = v.z; bf[t][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[t][s] = Bs_[(c * 8 + lh * 4 + s) * GLDM + wn * 64 + t * 32 + li];
            }
        }
    };

    // ---- main loop: 2-stage LDS ring, ONE barrier per k-tile.  Inside a tile the four 8-deep chunks are
    // software-pipelined: the LDS reads of chunk c+1, the LDS stores of tile t+1 (chunk 1) and the global
    // loads of tile t+2 (chunk 2) are issued BEFORE the 16 MFMAs of chunk c, so they fly under them
    // (an f32 MFMA occupies the matrix pipe for 64 cycles; everything else issues in its shadow).
    const int ntiles = (kend - kbeg + GBK - 1) / GBK;
    gload(kbeg);
    lstore(kbeg, smem, smem + GTILE);
    gload(kbeg + GBK);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const float* Ac = smem + (t & 1) * (2 * GTILE);
        const float* Bc = Ac + GTILE;
        float* An = smem + ((t + 1) & 1) * (2 * GTILE);
        float* Bn = An + GTILE;
        float af[2][2][4], bf[2][2][4];
        read_frag(Ac, Bc, 0, af[0], bf[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) read_frag(Ac, Bc, c + 1, af[(c + 1) & 1], bf[(c + 1) & 1]);
            if (c == 1) lstore(kbeg + (t + 1) * GBK, An, Bn);        // tile t+1: registers -> other stage
            if (c == 2) gload(kbeg + (t + 2) * GBK);                 // tile t+2: HBM/L2 -> registers
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][i][s], bf[c & 1][j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulator tile -> LDS (per-wave region) -> row-major float4
    float* Cs = smem + wave * (32 * 36);
    const bool split = gridDim.z > 1;
    float am = 0.f;
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

// C[m,n] (+)= sum_z slabs[z][m,n]   (deterministic split-K combine)
// cs_out != null: also cs_out[m] (+)= sum_z cs_ws[z][m], the split-K combine of the column sums folded into the plane TN GEMM
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ slabs, int splits, long long slab_stride, float* C, int ldc,
                              int M, int N, int accumulate, const float* __restrict__ cs_ws, float* cs_out) {
    if (cs_out) {
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int z = 0; z < splits; ++z) s += cs_ws[(size_t)z * M + m];
            cs_out[m] = accumulate ? cs_out[m] + s : s;
        }
    }
    const int n4 = N >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)M * n4;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i % n4) << 2;
        const float* p0 = slabs + (size_t)m * N + n;
        f32x4 s = ld_row4b(p0);
        int z = 1;
        for (; z + 4 <= splits; z += 4) {          // four slab loads in flight, summed in slab order
            const f32x4 a = ld_row4b(p0 + (size_t)z * slab_stride), b = ld_row4b(p0 + (size_t)(z + 1) * slab_stride);
            const f32x4 c = ld_row4b(p0 + (size_t)(z + 2) * slab_stride), d = ld_row4b(p0 + (size_t)(z + 3) * slab_stride);
            s += a; s += b; s += c; s += d;
        }
        for (; z < splits; ++z) s += ld_row4b(p0 + (size_t)z * slab_stride);
        float* c = C + (size_t)m * ldc + n;
        if (accumulate) s += *(const f32x4*)c;
        *(f32x4*)c = s;
    }
}

}  // namespace segmm
