// fp32-accurate GEMM on the fp16 matrix cores with operands that arrive PRE-SPLIT ("plane" operands).
//
// Same arithmetic as the fp16x3 engine of gemm_split.h -- x * s = hi + lo in two fp16 terms, s a per-tensor power of
// two, three exact partial products hh + hl + lh on v_mfma_f32_32x32x16_f16 accumulated in fp32, result scaled back
// by the exact 1/(sa sb) -- but the split is NOT done in the main loop any more: every producer of a GEMM operand
// (LayerNorm, GEMM epilogue, attention, the L1 normalisation, the per-step weight split) writes the two fp16 terms
// itself, and this kernel moves them global -> LDS with LDS-DMA (buffer_load ... lds, no VGPR round trip, no VALU).
//
// P32 plane format of a matrix X[R][C] (C % 32 == 0): one fp16 array, per row and per block of 32 columns
// [32 hi | 32 lo] = 128 contiguous bytes -- element (r, c): hi at r * ld2 + (c >> 5) * 64 + (c & 31), lo 32 further.
// A k-tile of 32 of one row is therefore ONE 128-byte line holding both terms (the NT form), and a row of 256 features
// one contiguous KB (the TN form).  Column slices of fused buffers stay contiguous slices (offsets are multiples of 32).
//
// Site header (hdr, SITE_HDR floats followed by AMAX_SLOTS partial maxima; common.h): hdr[0] = the scale s the planes
// were written with, hdr[1] != 0 = some element left the fp16 range under that scale (delayed scaling: s comes from an
// earlier step) -- or hdr[0] == 0: no planes were written at all (first use of a tensor site).  In both cases the
// kernel takes the operand from its fp32 copy instead and splits it on the fly with the exact scale of the partial
// maxima (slow path, staged through registers; results identical in accuracy, never silently wrong).
//
// NT kernel: 256 x 256 x 32 tile, 512 threads = 8 waves as 2 (m) x 4 (n), 128 x 64 per wave = 4 x 2 MFMA tiles x 3
// products, 128 accumulator registers; two 64 KB LDS stages (A 256 rows x 128 B, B 256 rows x 128 B); one barrier per
// k-tile.  LDS rows are 128 B = 8 chunks of 16 B; chunk c of row r sits at physical chunk c ^ ((r >> 1) & 7) -- the
// ds_read_b128 fragment reads (16 distinct rows per lane group, same logical chunk) are conflict-free; because LDS-DMA
// writes lane-linear, the permutation is applied to each lane's SOURCE address.
#pragma once
#include <type_traits>
#include "gemm_split.h"

namespace segmm {

struct PlaneOperand {
    const _Float16* p; int ld2;        // P32 planes, row stride in fp16 elements
    uint32_t bytes;                    // extent of the plane view (buffer range check)
    const float* hdr;                  // site header (scale, overflow flag, partial maxima)
    const float* f32; int ldf;         // fp32 copy for the slow path (may be null: the planes are then trusted)
};
struct PGemmX {
    PlaneOperand A, B;
    _Float16* Cp; int ldc2; float* c_hdr;      // optional plane output; amax / flag / the scale used are folded into c_hdr
    const float* c_scale_in;                   // device scalar: scale to write the output planes with (null / 0: none)
    int write_c;                               // 0: the fp32 C is not stored (planes only)
    int repair;                                // gemm_pl_nt8 only: REPAIR launch of a planes-only output (see segmm_gemm_p)
    float* colsum_out; float* colsum_ws;       // TN only: optional [M] column sums of A over k (= bias gradient), split-K partials [splits][M]
    int dbg;                                   // timing ablations (SEGMM_PL_FLAGS; results are wrong when set): 1 no C stores, 2 no epilogue
    unsigned long long* stamps;                // SEGMM_STAMPS builds only (tools/probe/gemm_stamps.py): 8 x u64 per workgroup
};

// timing ablations of the plane GEMMs (PGemmX::dbg; results wrong): compiled in by -DSEGMM_GEMM_PROBE only
#ifdef SEGMM_GEMM_PROBE
#define SEGMM_GEMM_DBG(q) ((q).dbg)
#else
#define SEGMM_GEMM_DBG(q) 0
#endif

constexpr int PBM = 256, PBN = 256, PBK = 32;
constexpr int PSTAGE = (PBM + PBN) * 128;      // bytes per stage

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
#ifndef SEGMM_TN_AUX
#define SEGMM_TN_AUX 0          // probe: cache policy of the weight-gradient GEMMs' operand loads (2 nt)
#endif
__device__ __forceinline__ void lds_dma16t(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, SEGMM_TN_AUX);
}
__device__ __forceinline__ void lds_dma16e(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, uint32_t voff, uint32_t soff) {          // the epilogue's extra operand: read once
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, SEGMM_E_AUX);
}
__device__ __forceinline__ void dma_wait_barrier() {
    __builtin_amdgcn_sched_barrier(0);          // MFMAs are register-only: without this the scheduler sinks them below the barrier
#ifndef SEGMM_PROBE_BAR          // timing probes only (results are wrong): 1 no s_barrier, 2 no DMA wait, 3 neither
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#else
    if (SEGMM_PROBE_BAR & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(SEGMM_PROBE_BAR & 1)) __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// exact scale of a site from its partial maxima (slow path only): every thread of the block gets the same value
__device__ __forceinline__ float site_exact_scale(const float* hdr, float* red, int tid, int nthreads) {
    float m = 0.f;
    for (int i = tid; i < AMAX_SLOTS; i += nthreads) m = fmaxf(m, hdr[SITE_HDR + i]);
    m = wave_max(m);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (nthreads >> 6); ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    return f16_scale_of(r);
}

// ---- shared epilogue.  A wave owns a [32 i] x [32 j] grid of accumulator tiles; it moves one 32-row strip of NJ tiles
// (32 x 32 NJ floats) at a time through its own LDS patch [32][32 NJ] (256-byte rows for NJ = 2: the ds_write_b32 of the
// accumulator layout and the ds_read_b128 of the row layout are both conflict-free) and then walks the strip in a ROLLED
// loop: the element-wise epilogue is emitted once, not once per tile (the unrolled form was 24 k instructions and spent
// 24 us per 256 x 256 tile missing the instruction cache).
template <int NJ>
__device__ __forceinline__ void epi_strip_write(const f32x16 (&c)[NJ], float* Cs, int lane) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * (32 * NJ) + j * 32 + li] = c[j][r];
}
template <bool SPLITK, int NJ>
__device__ __forceinline__ void epi_strip_emit(const GemmArgs& p, const PGemmX& q, const float* Cs, int lane, int gm0, int gn0,
                                               float inv_ab, float c_scale, float* Cout, float& am) {
    constexpr int LPR = 8 * NJ;                    // lanes per row (float4 each)
    constexpr int RPI = 64 / LPR;                  // rows per iteration
    const int c4 = (lane % LPR) << 2, r0 = lane / LPR;
    const int gn = gn0 + c4;
    const bool col_ok = gn < p.N;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (!SPLITK && p.bias && col_ok) bias4 = *(const f32x4*)(p.bias + gn);
#pragma unroll 2
    for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + r0;
        const int gm = gm0 + row;
        if (gm < p.M && col_ok) {
            f32x4 v = *(const f32x4*)(Cs + row * (32 * NJ) + c4);
            v = v * inv_ab;
            if (!SPLITK) {
                if (p.row_scale) v *= p.row_scale[gm];
                v += bias4;
                if (p.epi == EPI_GELU) {
                    *(f32x4*)(p.aux + (size_t)gm * p.ldaux + gn) = v;
                    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                } else if (p.epi == EPI_DGELU) {
                    const f32x4 g = *(const f32x4*)(p.aux + (size_t)gm * p.ldaux + gn);
                    v.x *= gelu_erf_grad(g.x); v.y *= gelu_erf_grad(g.y);
                    v.z *= gelu_erf_grad(g.z); v.w *= gelu_erf_grad(g.w);
                } else if (p.epi == EPI_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else if (p.epi == EPI_DRELU) {
                    const f32x4 g = *(const f32x4*)(p.aux + (size_t)gm * p.ldaux + gn);
                    v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f;
                    v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
                }
                if (p.drop.p > 0.f) v = drop_apply4(drop_live(p.drop), ((uint64_t)gm * (uint64_t)p.N + gn) >> 2, v);
                if (p.residual) {
                    const int rr = p.res_period >= p.M ? gm : gm % p.res_period;
                    v += *(const f32x4*)(p.residual + (size_t)rr * p.ldr + gn);
                }
            }
            if ((SPLITK || q.write_c) && !(SEGMM_GEMM_DBG(q) & 1)) {
#if SEGMM_NT_STORES
                if (!SPLITK) __builtin_nontemporal_store(v, (f32x4*)(Cout + (size_t)gm * p.ldc + gn));
                else
#endif
                *(f32x4*)(Cout + (size_t)gm * p.ldc + gn) = v;
            }
            if (!SPLITK) {
                am = absmax4(am, v);
                if (c_scale > 0.f && q.Cp) plane_store4_pair(q.Cp, q.ldc2, gm, gn, v, c_scale);
            }
        }
    }
}

// =============================================================================== NT: C[M,N] = A[M,K] . B[N,K]^T
// (round 2's form, kept as the fallback for launches gemm_pl_nt8 does not take: row-scaled outputs, a residual together with a
// d-activation, extents of 2 GiB and more.  One DMA piece of the next k-tile between every six MFMAs; its A/B variants -- DMA in
// front of the MFMA block, software-pipelining across the barrier, the four-wave 128 x 256 form -- were retired in round 5.)
__global__ __launch_bounds__(512, 2) void gemm_pl_nt(const GemmArgs p, const PGemmX q) {
    __shared__ __attribute__((aligned(16))) char smem[2 * PSTAGE];      // 128 KB: two stages x (A 32 KB | B 32 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    const int m0 = (lb / p.nbn) * PBM, n0 = (lb % p.nbn) * PBN;
    const int nkt = p.K >> 5;

    // ---- operand state: planes usable?  (block-uniform)
    const float sa_hdr = q.A.hdr[0], sb = q.B.hdr[0];
    const bool slowA = q.A.f32 != nullptr && !site_planes_ok(q.A.hdr, sa_hdr, lane);
    float sa = sa_hdr;
    if (slowA) sa = site_exact_scale(q.A.hdr, (float*)smem, tid, 512);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- LDS-DMA addressing: a wave-instruction moves 8 rows x 128 B; wave w, piece i covers tile rows (4 w + i) * 8 .. + 7
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(q.A.p, q.A.bytes), rsB = make_rsrc(q.B.p, q.B.bytes);
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);                    // logical chunk that lands in this lane's physical slot
        voa[i] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)q.A.ld2 * 2u + (uint32_t)c * 16u;
        vob[i] = (uint32_t)min(n0 + row, p.N - 1) * (uint32_t)q.B.ld2 * 2u + (uint32_t)c * 16u;
    }
    auto dmaB = [&](int kt, char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsB, st + PBM * 128 + (wave * 4 + i) * 1024, vob[i], (uint32_t)kt * 128u);
    };
    auto dmaA = [&](int kt, char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, st + (wave * 4 + i) * 1024, voa[i], (uint32_t)kt * 128u);
    };
    // slow path for A: fp32 copy -> exact split -> the same LDS image (thread: 2 jobs of 8 consecutive k)
    auto slow_stage_A = [&](int kt, char* st) {
#pragma unroll 1
        for (int jj = 0; jj < 2; ++jj) {
            const int j = tid + 512 * jj;
            const int row = j >> 2, kc = j & 3;
            const float* src = q.A.f32 + (size_t)min(m0 + row, p.M - 1) * q.A.ldf + kt * 32 + kc * 8;
            const f32x4 x0 = *(const f32x4*)src, x1 = *(const f32x4*)(src + 4);
            uint32_t h0, l0, h1, l1, h2, l2, h3, l3;
            splith_pair(x0.x, x0.y, sa, h0, l0); splith_pair(x0.z, x0.w, sa, h1, l1);
            splith_pair(x1.x, x1.y, sa, h2, l2); splith_pair(x1.z, x1.w, sa, h3, l3);
            const int sw = (row >> 1) & 7;
            *(uint4*)(st + row * 128 + ((kc ^ sw) << 4)) = make_uint4(h0, h1, h2, h3);
            *(uint4*)(st + row * 128 + (((4 + kc) ^ sw) << 4)) = make_uint4(l0, l1, l2, l3);
        }
    };
    auto stage = [&](int kt, char* st) {
        if (slowA) slow_stage_A(kt, st); else dmaA(kt, st);
        dmaB(kt, st);
    };

    // ---- fragment read addressing: lane (li, lh); logical chunk 4 p + 2 s + lh; physical = logical ^ ((li >> 1) & 7)
    const int swz = (li >> 1) & 7;
    uint32_t fa[2][2], fb[2][2];          // [plane][k16 step] byte offsets inside a stage (row block 0)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ch = ((4 * pl + 2 * s + lh) ^ swz) << 4;
            fa[pl][s] = (uint32_t)((wm * 128 + li) * 128 + ch);
            fb[pl][s] = (uint32_t)(PBM * 128 + (wn * 64 + li) * 128 + ch);
        }

    auto compute = [&](const char* st) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *(const f32x4*)(st + fb[0][s] + j * 4096);
                bl[j] = *(const f32x4*)(st + fb[1][s] + j * 4096);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 ah = *(const f32x4*)(st + fa[0][s] + i * 4096);
                const f32x4 al = *(const f32x4*)(st + fa[1][s] + i * 4096);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = mfma_x<true>(al, bh[j], c);
                    c = mfma_x<true>(ah, bl[j], c);
                    c = mfma_x<true>(ah, bh[j], c);
                    acc[i][j] = c;
                }
            }
        }
    };

    // compute with the DMA of k-tile kt_next woven in: piece g (4 of A, 4 of B) goes out after the fragment reads of MFMA group g
    auto compute_dma = [&](const char* st, char* nx, int kt_next) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *(const f32x4*)(st + fb[0][s] + j * 4096);
                bl[j] = *(const f32x4*)(st + fb[1][s] + j * 4096);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 ah = *(const f32x4*)(st + fa[0][s] + i * 4096);
#if defined(SEGMM_PROBE_BAR) && (SEGMM_PROBE_BAR & 4)          // timing probe: half the LDS fragment traffic
                const f32x4 al = ah;
#else
                const f32x4 al = *(const f32x4*)(st + fa[1][s] + i * 4096);
#endif
                // (no run-time condition here: a branch would end the scheduling region after every 6 MFMAs)
                if (s == 0) lds_dma16(rsA, nx + (wave * 4 + i) * 1024, voa[i], (uint32_t)kt_next * 128u);
                else lds_dma16(rsB, nx + PBM * 128 + (wave * 4 + i) * 1024, vob[i], (uint32_t)kt_next * 128u);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = mfma_x<true>(al, bh[j], c);
                    c = mfma_x<true>(ah, bl[j], c);
                    c = mfma_x<true>(ah, bh[j], c);
                    acc[i][j] = c;
                }
            }
        }
    };

    stage(0, smem);
    dma_wait_barrier();
    if (!slowA) {
        for (int kt = 0; kt < nkt - 1; ++kt) {
            char* cur = smem + (kt & 1) * PSTAGE;
            char* nxt = smem + ((kt + 1) & 1) * PSTAGE;
            compute_dma(cur, nxt, kt + 1);
            dma_wait_barrier();
        }
        compute(smem + ((nkt - 1) & 1) * PSTAGE);
        dma_wait_barrier();
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            char* cur = smem + (kt & 1) * PSTAGE;
            char* nxt = smem + ((kt + 1) & 1) * PSTAGE;
            if (kt + 1 < nkt) stage(kt + 1, nxt);
#if SEGMM_GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(SEGMM_GEMM_SETPRIO);
#endif
            compute(cur);
#if SEGMM_GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            dma_wait_barrier();
        }
    }

    // ---- epilogue (all stages are free: the loop ended with a barrier)
    float* Cs = (float*)smem + wave * (32 * 64);          // 8 KB per wave
    const float inv_ab = (1.f / sa) * (1.f / sb);          // exact powers of two
    const float c_scale = (q.Cp && q.c_scale_in) ? *q.c_scale_in : 0.f;
    float am = 0.f;
    if (SEGMM_GEMM_DBG(q) & 2) {
        float t = 0.f;          // keeps every accumulator alive (no dead-code elimination of MFMAs)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 1.2345f) p.C[0] = 1.f;
        return;
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        f32x16 c[2];
        switch (i) {
            case 0: c[0] = acc[0][0]; c[1] = acc[0][1]; break;
            case 1: c[0] = acc[1][0]; c[1] = acc[1][1]; break;
            case 2: c[0] = acc[2][0]; c[1] = acc[2][1]; break;
            default: c[0] = acc[3][0]; c[1] = acc[3][1]; break;
        }
        epi_strip_write<2>(c, Cs, lane);
        epi_strip_emit<false, 2>(p, q, Cs, lane, m0 + wm * 128 + i * 32, n0 + wn * 64, inv_ab, c_scale, p.C, am);
    }
    if (q.c_hdr) {
        site_commit(q.c_hdr, am, blockIdx.x * 8 + wave, c_scale);
        if (c_scale > 0.f && scale_writer(blockIdx.x * 8 + wave)) q.c_hdr[0] = c_scale;
    } else if (p.amax_out) amax_commit(p.amax_out, am, blockIdx.x * 8 + wave);
}

// =============================================================================== TN: C[M,N] = A[K,M]^T . B[K,N]
// Weight gradients: dW[n_out, n_in] = dY[tokens, n_out]^T . X[tokens, n_in]; the contraction runs over the token axis, the
// SLOW axis of both operands.  A k-tile = 32 tokens x 256 features per operand; a token's 256 features are one contiguous
// KB of P32 planes (8 blocks of [32 hi | 32 lo]) = one LDS-DMA wave-instruction.  MFMA fragments (8 consecutive tokens of
// one feature per lane) come out of LDS through ds_read_b64_tr_b16 (hardware transpose: per 16-lane group a 4 token x 16
// feature block, lane i receives feature i).  LDS row = 16 pieces of 64 B (piece = 2 * feature block + plane); piece c of
// token t sits at physical piece c ^ (t & 3), so the four token rows of a transposed read fall into four different
// 64-byte bank windows (conflict-free per 32-lane half); the permutation is applied to the DMA source address.
// Split-K over blockIdx.z (partial slabs + splitk_reduce), 256 x 256 tile, 8 waves as 2 (m) x 4 (n).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lds_tr8(const char* a) {          // 8 tokens (two 4-token blocks, 4 KB apart) of this lane's feature
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4096));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f32x4, v);
}

__global__ __launch_bounds__(512, 2) void gemm_pl_tn(const GemmArgs p, const PGemmX q) {
    __shared__ __attribute__((aligned(16))) char smem[2 * PSTAGE];      // two stages x (A: 32 tokens x 1 KB | B: 32 tokens x 1 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest, then z): give each XCD a run of
    // consecutive (k-slab, tile) pairs, so the tiles sharing a token slab of A or B meet in one L2 (hit rate 44 % -> see DESIGN.md)
    const int ntile = p.nbm * p.nbn;
    const int lg = xcd_remap(blockIdx.x + ntile * blockIdx.z, ntile * gridDim.z);
    const int kz = lg / ntile, lb = lg - kz * ntile;
    const int m0 = (lb / p.nbn) * PBM, n0 = (lb % p.nbn) * PBN;
    const int kbeg = kz * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg + 31) >> 5;

    // bias gradient folded in: the workgroups of the first column tile also form sum_k A[k, m] -- one more MFMA pair per
    // k16 step and wave against an all-ones B fragment (wave (wm, wn) takes A block i = wn of its 128 rows), instead of a
    // separate column-sum pass over the whole fp32 dY tensor
    const bool do_colsum = q.colsum_out != nullptr && (lb % p.nbn) == 0;
    const float sa_hdr = q.A.hdr[0], sb_hdr = q.B.hdr[0];
    const bool slowA = q.A.f32 != nullptr && !site_planes_ok(q.A.hdr, sa_hdr, lane);
    const bool slowB = q.B.f32 != nullptr && !site_planes_ok(q.B.hdr, sb_hdr, lane);
    float sa = sa_hdr, sb = sb_hdr;
    if (slowA) sa = site_exact_scale(q.A.hdr, (float*)smem, tid, 512);
    if (slowB) sb = site_exact_scale(q.B.hdr, (float*)smem, tid, 512);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const f32x4 ones = __builtin_bit_cast(f32x4, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));      // 8 x fp16 1.0

    // ---- LDS-DMA: wave w, piece i = token row 4 w + i of the k-tile (1 KB); lane = physical 16-byte chunk of the row
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(q.A.p, q.A.bytes), rsB = make_rsrc(q.B.p, q.B.bytes);
    uint32_t voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave * 4 + i;
        const uint32_t inrow = (uint32_t)((((lane >> 2) ^ (t & 3)) << 6) + ((lane & 3) << 4));
        voa[i] = (uint32_t)t * (uint32_t)q.A.ld2 * 2u + (uint32_t)m0 * 4u + inrow;
        vob[i] = (uint32_t)t * (uint32_t)q.B.ld2 * 2u + (uint32_t)n0 * 4u + inrow;
    }
    // slow path: fp32 [token][feature] -> exact split -> the same LDS image; a thread converts 8 consecutive features of one token
    auto slow_stage = [&](const PlaneOperand& op, float sc, int k0, char* dst, int f0, int nfeat) {
#pragma unroll 1
        for (int jj = 0; jj < 2; ++jj) {
            const int j = tid + 512 * jj;
            const int t = j >> 5, f8 = j & 31;
            const int gk = k0 + t, gf = f0 + f8 * 8;
            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (gk < kend && gf < nfeat) {
                const float* src = op.f32 + (size_t)gk * op.ldf + gf;
                x0 = *(const f32x4*)src;
                x1 = *(const f32x4*)(src + 4);
            }
            uint32_t h0, l0, h1, l1, h2, l2, h3, l3;
            splith_pair(x0.x, x0.y, sc, h0, l0); splith_pair(x0.z, x0.w, sc, h1, l1);
            splith_pair(x1.x, x1.y, sc, h2, l2); splith_pair(x1.z, x1.w, sc, h3, l3);
            const int b = f8 >> 2, cp = f8 & 3, sw = t & 3;
            *(uint4*)(dst + t * 1024 + (((2 * b) ^ sw) << 6) + (cp << 4)) = make_uint4(h0, h1, h2, h3);
            *(uint4*)(dst + t * 1024 + (((2 * b + 1) ^ sw) << 6) + (cp << 4)) = make_uint4(l0, l1, l2, l3);
        }
    };
    auto stage = [&](int kt, char* st) {
        const int k0 = kbeg + kt * 32;
        if (slowA) slow_stage(q.A, sa, k0, st, m0, p.M);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_dma16(rsA, st + (wave * 4 + i) * 1024, voa[i], (uint32_t)k0 * (uint32_t)q.A.ld2 * 2u);
        }
        if (slowB) slow_stage(q.B, sb, k0, st + 32768, n0, p.N);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_dma16(rsB, st + 32768 + (wave * 4 + i) * 1024, vob[i], (uint32_t)k0 * (uint32_t)q.B.ld2 * 2u);
        }
    };

    // ---- transposed fragment reads: lane = (g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3) supplies token row qq of its
    // group's 4 x 16 block, features 4 pp ..; group g covers features 16 (g & 1) .. and tokens 8 (g >> 1) ..
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const uint32_t lane_base = (uint32_t)((8 * (g >> 1) + qq) * 1024 + (16 * (g & 1) + 4 * pp) * 2);
    uint32_t fa[4], fb[4];          // [x = 2 (block & 1) + plane]
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        fa[x] = lane_base + (uint32_t)(8 * wm * 64) + (uint32_t)((x ^ qq) << 6);
        fb[x] = lane_base + 32768u + (uint32_t)(4 * wn * 64) + (uint32_t)((x ^ qq) << 6);
    }
    // the column-sum fragments (A block i = wn) are fetched by address, not by "if (i == wn)": a branch inside the MFMA
    // block would cut it into eight scheduling regions, each opening with a full LDS wait
    const uint32_t fcs_h = fa[2 * (wn & 1)] + (uint32_t)((wn >> 1) * 256), fcs_l = fa[2 * (wn & 1) + 1] + (uint32_t)((wn >> 1) * 256);
    auto compute = [&](const char* st, auto cs_tag) {
        constexpr bool CS = decltype(cs_tag)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = lds_tr8(st + fb[2 * j] + s * 16384);
                bl[j] = lds_tr8(st + fb[2 * j + 1] + s * 16384);
            }
            if constexpr (CS) {
                const f32x4 ch = lds_tr8(st + fcs_h + s * 16384);
                const f32x4 cl = lds_tr8(st + fcs_l + s * 16384);
                accb = mfma_x<true>(cl, ones, accb);
                accb = mfma_x<true>(ch, ones, accb);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 ah = lds_tr8(st + fa[2 * (i & 1)] + (i >> 1) * 256 + s * 16384);
                const f32x4 al = lds_tr8(st + fa[2 * (i & 1) + 1] + (i >> 1) * 256 + s * 16384);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = mfma_x<true>(al, bh[j], c);
                    c = mfma_x<true>(ah, bl[j], c);
                    c = mfma_x<true>(ah, bh[j], c);
                    acc[i][j] = c;
                }
            }
        }
    };
    auto k_loop = [&](auto cs_tag) {
        for (int kt = 0; kt < nkt; ++kt) {
            char* cur = smem + (kt & 1) * PSTAGE;
            char* nxt = smem + ((kt + 1) & 1) * PSTAGE;
            if (kt + 1 < nkt) stage(kt + 1, nxt);
#if SEGMM_GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(SEGMM_GEMM_SETPRIO);
#endif
            compute(cur, cs_tag);
#if SEGMM_GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            dma_wait_barrier();
        }
    };

    stage(0, smem);
    dma_wait_barrier();
    if (do_colsum) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    const bool split = gridDim.z > 1;
    if (do_colsum && (lane & 31) == 0) {          // every column of accb holds the row sums: lanes 0 and 32 own 16 rows each
        float* dst = split ? q.colsum_ws + (size_t)kz * p.M : q.colsum_out;
        const float inv_a = 1.f / sa;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 128 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < p.M) dst[m] = (!split && p.residual) ? dst[m] + accb[r] * inv_a : accb[r] * inv_a;      // residual set = accumulate
        }
    }
    float* Cs = (float*)smem + wave * (32 * 64);
    const float inv_ab = (1.f / sa) * (1.f / sb);
    float* Cout = split ? p.C + (size_t)kz * (size_t)p.slab_stride : p.C;
    float am = 0.f;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        f32x16 c[2];
        switch (i) {
            case 0: c[0] = acc[0][0]; c[1] = acc[0][1]; break;
            case 1: c[0] = acc[1][0]; c[1] = acc[1][1]; break;
            case 2: c[0] = acc[2][0]; c[1] = acc[2][1]; break;
            default: c[0] = acc[3][0]; c[1] = acc[3][1]; break;
        }
        epi_strip_write<2>(c, Cs, lane);
        if (split) epi_strip_emit<true, 2>(p, q, Cs, lane, m0 + wm * 128 + i * 32, n0 + wn * 64, inv_ab, 0.f, Cout, am);
        else epi_strip_emit<false, 2>(p, q, Cs, lane, m0 + wm * 128 + i * 32, n0 + wn * 64, inv_ab, 0.f, Cout, am);
    }
}

// =============================================================================== fp32 -> P32 planes (stand-alone pass)
// mode 0: the site's partial maxima are complete (producer or segmm_absmax): s = exact scale, written to hdr[0], flag cleared.
// mode 1: s = hdr[0] as it stands (delayed scale); partial maxima and the overflow flag are folded into hdr.
__global__ __launch_bounds__(256) void split_p32_kernel(const float* __restrict__ x, long long rows, int cols, int ld,
                                                        _Float16* __restrict__ planes, int ld2, float* hdr, int mode) {
    __shared__ float red[8];
    float s;
    if (mode == 0) s = site_exact_scale(hdr, red, threadIdx.x, 256);
    else s = hdr[0];
    const int c4n = cols >> 2;
    const long long n4 = rows * c4n;
    float am = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4n;
        const int c = (int)(i - r * c4n) << 2;
        const f32x4 v = *(const f32x4*)(x + r * ld + c);
        plane_store4_pair(planes, ld2, r, c, v, s);      // i even <-> c % 8 == 0: lane pairs share a row (cols % 8 == 0)
        am = absmax4(am, v);
    }
    if (mode == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = s; hdr[1] = 0.f; }
    } else {
        site_commit(hdr, am, blockIdx.x * 4 + (threadIdx.x >> 6), s);
    }
}
// planes of the TRANSPOSE: out row c (of Cc), column r (of R) = x[r * ld + c]; P32 over the R axis (R % 32 == 0)
__global__ __launch_bounds__(256) void split_p32_transpose_kernel(const float* __restrict__ x, int R, int Cc, int ld,
                                                                  _Float16* __restrict__ planes, int ld2, const float* hdr) {
    __shared__ float tile[32][33];
    const float s = hdr[0];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < Cc) ? x[(size_t)r * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < Cc && r < R) {
            const float v = tile[tx][ty + 8 * k];
            const _Float16 h = (_Float16)(v * s);
            const _Float16 l = (_Float16)__builtin_fmaf(v, s, -(float)h);
            _Float16* o = planes + (size_t)c * ld2 + ((r >> 5) << 6) + (r & 31);
            o[0] = h; o[32] = l;
        }
    }
}

// ---------------------------------------------------------------- all weight matrices of a model in TWO launches per step
// desc[i] = {flat offset of matrix i (floats), rows R, cols C, needs transpose, first 32 x 32 tile index, tile columns}
// (tiles cover [ceil(R/32)][C/32]); hdr = [n][SITE_FLOATS] site headers (zeroed once at allocation; wabsmax rewrites all slots).
struct WMat { long long off; int R, Cc, tr, tile0, tcols; };
__global__ __launch_bounds__(256) void wabsmax_kernel(const float* __restrict__ flat, const WMat* __restrict__ desc, float* hdr) {
    const WMat m = desc[blockIdx.y];
    const float* x = flat + m.off;
    const long long n4 = (long long)m.R * m.Cc / 4;
    float am = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        am = absmax4(am, *(const f32x4*)(x + 4 * i));
    // the launch is (64 workgroups x 4 waves) per matrix = exactly AMAX_SLOTS waves: every slot has ONE writer, so a plain
    // store replaces the atomic max and the header needs no zero-fill between optimizer steps
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) hdr[(size_t)blockIdx.y * SITE_FLOATS + SITE_HDR + blockIdx.x * 4 + (threadIdx.x >> 6)] = am;
}
// one 32 x 32 tile per workgroup: P32 planes of W (wpl, at element offset 2 * off, ld2 = 2 C) and -- if asked -- of W^T
// (wTpl, same offset, ld2 = 2 R), both with the exact scale of the matrix' own maxima, which is also stored in its header
__global__ __launch_bounds__(256) void wsplit_kernel(const float* __restrict__ flat, const WMat* __restrict__ desc, int n_mats, float* hdr,
                                                     _Float16* __restrict__ wpl, _Float16* __restrict__ wTpl) {
    __shared__ float tile[32][33];
    __shared__ float red[8];
    int mi = 0;
    while (mi + 1 < n_mats && (int)blockIdx.x >= desc[mi + 1].tile0) ++mi;
    const WMat m = desc[mi];
    float* h = hdr + (size_t)mi * SITE_FLOATS;
    const float s = site_exact_scale(h, red, threadIdx.x, 256);
    const int t = blockIdx.x - m.tile0;
    const int r0 = (t / m.tcols) * 32, c0 = (t % m.tcols) * 32;
    if (t == 0 && threadIdx.x == 0) { h[0] = s; h[1] = 0.f; }
    const float* x = flat + m.off;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        const float v = (r < m.R && c < m.Cc) ? x[(size_t)r * m.Cc + c] : 0.f;
        tile[ty + 8 * k][tx] = v;
        if (r < m.R && c < m.Cc) {
            const _Float16 hi = (_Float16)(v * s);
            const _Float16 lo = (_Float16)__builtin_fmaf(v, s, -(float)hi);
            _Float16* o = wpl + 2 * m.off + (size_t)r * (2 * m.Cc) + ((c >> 5) << 6) + (c & 31);
            o[0] = hi; o[32] = lo;
        }
    }
    if (!m.tr) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;          // plane row c of the transpose, column r
        if (c < m.Cc && r < m.R) {
            const float v = tile[tx][ty + 8 * k];
            const _Float16 hi = (_Float16)(v * s);
            const _Float16 lo = (_Float16)__builtin_fmaf(v, s, -(float)hi);
            _Float16* o = wTpl + 2 * m.off + (size_t)c * (2 * m.R) + ((r >> 5) << 6) + (r & 31);
            o[0] = hi; o[32] = lo;
        }
    }
}

}  // namespace segmm
