// NT plane GEMM, round-3 form: C[M,N] = A[M,K] . B[N,K]^T on v_mfma_f32_16x16x32_f16 with the two wave groups of a
// workgroup running OUT OF PHASE ("ping-pong").
//
// Same arithmetic, operand format (P32 planes), LDS image and fallback protocol as gemm_pl_nt (gemm_planes.h); tile
// 256 x (64 NJ) x 32 with NJ = 2, 3 or 4 chosen per launch (the host picks the width that wastes the fewest CU-rounds),
// 8 waves as 2 (m) x 4 (n), 128 x 16 NJ per wave.  What changed against gemm_pl_nt, and why:
//
//  * MFMA shape 16x16x32 instead of 32x32x16: same cycles per FLOP, same LDS fragment traffic -- but the chip, which is
//    power-limited in a dense fp16 MFMA loop on real data, holds a ~13 % higher clock on it (tools/probe/mfma_shape.hip on this
//    pool: 1 683 vs 1 492 TFLOP/s with every operand re-read from LDS; MI355X_MICROARCH.md "DVFS give-back" item 7).
//  * The operands of an MFMA are swapped (a := B fragment, b := A fragment), so an accumulator tile is C^T: a lane holds
//    FOUR CONSECUTIVE COLUMNS of one row of C.  The epilogue's arithmetic works on float4s straight from the accumulators;
//    only the finished values pass through a small per-wave LDS transpose patch so that every store instruction covers
//    4 rows x 256 B instead of 16 rows x 64 B (tools/probe/store_rate.hip: 3.5x fewer cycles to drain a tile).
//  * The epilogue's stores are buffer stores (32-bit offsets, rows beyond M dropped by the range check, no 64-bit address
//    arithmetic per store); its extra operand (residual / aux) comes through LDS by LDS-DMA, see the epilogue.
//  * Wave groups g0 = waves 0-3 (rows 0-127 of the tile) and g1 = waves 4-7 (rows 128-255) -- one wave of each per SIMD --
//    alternate LOAD and COMPUTE segments separated by s_barrier, g1 one segment behind g0: while one wave of a SIMD issues
//    its MFMAs its partner reads the next fragments out of LDS and issues its share of the LDS-DMA.
//    A k-tile is two phases per wave (P0: upper 64 rows of its 128, P1: lower 64), four segments:
//        L(t,P0): 8 A + 2 NJ B fragment reads of tile t; DMA of this group's share of A(t+1)   | lgkmcnt(0), barrier
//        C(t,P0): 12 NJ MFMAs                                                                  | vmcnt(4),   barrier
//        L(t,P1): 8 A fragment reads; DMA of this group's share of B(t+2)                      | lgkmcnt(0), barrier
//        C(t,P1): 12 NJ MFMAs                                                                  | vmcnt(NJ),  barrier
//    LDS-DMA stays in flight ACROSS barriers: the counted wait at the end of a compute segment only asks for the pieces
//    issued TWO load segments earlier (every piece has ~3 segments, > 1 us, to land), instead of draining vmcnt(0) before one
//    barrier per k-tile.  Hazards (global segment s; g0 loads in even, g1 in odd segments):
//      RAW  a piece is waited for by its issuing wave at the end of a compute segment and read, by any wave, in a LATER
//           segment (one barrier in between at least);
//      WAR  a region of a stage is refilled at least one full segment after the last segment that read it, and every
//           load segment retires its ds_reads (lgkmcnt(0)) BEFORE its closing barrier.
//    Who reads / refills what (stage = t & 1; A rows 0-63 / 64-127 = g0's P0 / P1 rows, 128-191 / 192-255 = g1's):
//      B(t) all rows : read in L(t,P0) of g0 (s = 4t) and g1 (4t+1);  refilled with B(t+2) in L(t,P1): lower half by g0
//                      (4t+2), upper half by g1 (4t+3)
//      A(t+1) rows 0-63, 128-191  : issued by g0 in L(t,P0) (s = 4t; last read of that stage region: 4t-4, 4t-3)
//      A(t+1) rows 64-127, 192-255: issued by g1 in L(t,P0) (s = 4t+1; last read: 4t-2, 4t-1)
//  * The first DMA pieces leave before the site headers are read (their addresses depend on blockIdx only); the headers
//    (two dependent global reads under load) are fetched while the first k-tiles fly.
#pragma once
#include "gemm_planes.h"

namespace segmm {

__device__ __forceinline__ f32x4 mfma16(f32x4 a, f32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// 16-byte buffer store.  The data registers of a store wider than 8 bytes must not be overwritten for a few cycles after it
// issues; hipcc pads that hazard only when the scalar-offset field is an immediate (GCNHazardRecognizer assumes it does not
// exist with a register there) -- on gfx950 it does: with soffset in an SGPR and the next v_pk_fma_f32 reusing the data
// registers, lanes 12-15 stored the NEXT float4's .y/.w (tools/probe/dbg_fast.py).  The wait states are written out, in an asm
// statement that READS the data registers, so no write to them can be scheduled in front of it.
// Cache policy of the NT kernels' epilogue stores: nt (aux bit 1).  The outputs of a GEMM are written once and read by a LATER kernel;
// kept in the XCD's L2 like ordinary stores they push out the operand panels the other workgroups of the launch are still reading
// (tools/probe/gemm4_bench.hip, -DSEGMM_STORE_AUX=0 / 2 / 16: gemm_pl_nt4 20480 x 3072 x 768 228.4 -> 215.7 us, 51200 x 768 x 768
// 150.0 -> 133.8 us; gemm_pl_nt8 +2 .. 7 %; sc1 alone +1.5 %; in the step +0.4 % -- the consumers find the data in the Infinity
// Cache).  The split-K slabs of the TN kernels are read back by splitk_reduce at once: they keep the default policy (buf_store4k).
#ifndef SEGMM_STORE_AUX
#define SEGMM_STORE_AUX 2          // 0 default policy, 1 sc0, 2 nt, 16 sc1
#endif
#ifndef SEGMM_PLANE_AUX
#define SEGMM_PLANE_AUX SEGMM_STORE_AUX          // probe: cache policy of the plane OUTPUT stores (the next GEMM / attention reads them at once)
#endif
template <int AUX>
__device__ __forceinline__ void buf_store4u_aux(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, u32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, AUX);
    asm volatile("s_nop 3" :: "v"(v));
}
// (same-box check of small outputs, 20480 x 768 x 768 = 63 MB: 60.3 us default, 57.9 us nt -- no size threshold needed)
__device__ __forceinline__ void buf_store4u(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, u32x4_t v) { buf_store4u_aux<SEGMM_STORE_AUX>(r, voff, soff, v); }
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v) {
    buf_store4u(r, voff, soff, __builtin_bit_cast(u32x4_t, v));
}
__device__ __forceinline__ void buf_store4k(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v) {          // "keep": default cache policy
    buf_store4u_aux<0>(r, voff, soff, __builtin_bit_cast(u32x4_t, v));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
constexpr uint32_t BUF_OOB = 0x80000000u;          // a byte offset no descriptor of ours covers: the access is dropped / reads 0

#ifdef SEGMM_STAMPS
// diagnostic build only: shader-clock / real-time stamps of the kernel's sections, 8 x u64 per workgroup
#define STAMP(k) do { if (q.stamps && threadIdx.x == 0) { q.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                                                           if ((k) == 0 || (k) == 3) q.stamps[(size_t)blockIdx.x * 8 + 4 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
                                                           if ((k) == 0) q.stamps[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32); \
                                                           if ((k) == 2) q.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
// second region (16 x u64 per workgroup behind the 8 x u64 records): finer stamps inside a section
#define STAMPX(k) do { if (q.stamps && threadIdx.x == 0) q.stamps[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMPX(k) do { } while (0)
#endif

#ifndef SEGMM_SETPRIO
#define SEGMM_SETPRIO 0          // probe: raise the wave priority for the compute segments (s_setprio 1 ... 0)
#endif
template <int NOUT>          // number of LDS-DMA pieces this wave may leave in flight (0 .. 4)
__device__ __forceinline__ void end_compute_segment() {
    __builtin_amdgcn_sched_barrier(0);
    if (SEGMM_SETPRIO) __builtin_amdgcn_s_setprio(0);
    if (NOUT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (NOUT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (NOUT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void end_load_segment() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (SEGMM_SETPRIO) __builtin_amdgcn_s_setprio(1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int NJ>
__global__ __launch_bounds__(512, 2) void gemm_pl_nt8(const GemmArgs p, const PGemmX q) {
    static_assert(NJ >= 2 && NJ <= 4, "tile widths 128, 192, 256");
    constexpr int BNW = 64 * NJ;                                         // tile columns
    // 128 KB: two stages x (A 32 KB | B 16 NJ KB); + 32 KB: a 16-row x 256-byte transpose patch per wave for the epilogue's stores
    __shared__ __attribute__((aligned(16))) char smem[2 * PSTAGE + 8 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;          // grp = wm: rows [128 grp, +128) of the tile; columns [16 NJ wn, +16 NJ)
    const int l15 = lane & 15, lq = lane >> 4;
    const int nkt = p.K >> 5;
    const int lb = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    const int m0 = (lb / p.nbn) * PBM, n0 = (lb % p.nbn) * BNW;
    STAMP(0);
    // ---- REPAIR launch of a planes-only output (no fp32 C to fall back on): the first launch wrote the planes with the site's
    // delayed scale and recorded maxima + overflow flag in c_hdr.  Judge the site like a consumer would: usable -> every workgroup
    // leaves at once (one header read); unusable (flag up, maximum below the fp16 window, no scale yet) -> recompute the tile and
    // rewrite the planes with the EXACT scale of the recorded maxima.  The header is left as it is: consumers derive the same
    // scale from the same maxima (attention_pl.h), and segmm_scales_update counts the site as refused.
    float c_repair = 0.f;
    if (q.repair) {
        const float hc0 = q.c_hdr[0];
        const uint32_t hc1 = __float_as_uint(q.c_hdr[1]);
        const f32x4 amc = *(const f32x4*)(q.c_hdr + SITE_HDR + lane * 4);
        const float m = wave_max(fmaxf(fmaxf(amc.x, amc.y), fmaxf(amc.z, amc.w)));
        if (hc0 > 0.f && hc1 == 0u && (!(m > 0.f) || ((m * hc0 >= 0.25f || hc0 >= 0x1p60f) && m * hc0 < 65504.f))) return;
        c_repair = f16_scale_of(m);
    }

    // ---- LDS-DMA: a wave-instruction moves 8 rows x 128 B.  Per load segment a wave issues
    //   A share of its group (4 pieces): piece pi = 4 wn + i of 16; tile rows 128 (pi >> 3) + 64 grp + 8 (pi & 7) .. + 7
    //   B share of its group (NJ pieces): tile rows 32 NJ grp + 8 (NJ wn + i) .. + 7
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(q.A.p, q.A.bytes), rsB = make_rsrc(q.B.p, q.B.bytes);
    const int r8 = lane >> 3;
    uint32_t voa[4], vob[NJ];
    uint32_t lda_off[4], ldb_off[NJ];          // wave-uniform LDS byte offsets of the pieces inside a stage
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = 4 * wn + i;
        const int ra0 = 128 * (pi >> 3) + 64 * grp + 8 * (pi & 7);
        const int ra = ra0 + r8;
        voa[i] = (uint32_t)min(m0 + ra, p.M - 1) * (uint32_t)q.A.ld2 * 2u + (uint32_t)(((lane & 7) ^ ((ra >> 1) & 7)) * 16);
        lda_off[i] = (uint32_t)ra0 * 128u;
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int rb0 = 32 * NJ * grp + 8 * (NJ * wn + i);
        const int rb = rb0 + r8;
        vob[i] = (uint32_t)min(n0 + rb, p.N - 1) * (uint32_t)q.B.ld2 * 2u + (uint32_t)(((lane & 7) ^ ((rb >> 1) & 7)) * 16);
        ldb_off[i] = (uint32_t)(PBM * 128) + (uint32_t)rb0 * 128u;
    }
    auto dmaA = [&](int kt) {
        char* st = smem + (kt & 1) * PSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, st + lda_off[i], voa[i], (uint32_t)kt * 128u);
    };
    auto dmaB = [&](int kt) {
        char* st = smem + (kt & 1) * PSTAGE;
#pragma unroll
        for (int i = 0; i < NJ; ++i) lds_dma16(rsB, st + ldb_off[i], vob[i], (uint32_t)kt * 128u);
    };
    // ---- the first k-tiles leave NOW (harmless if the slow path is taken below: it restages synchronously)
    dmaA(0);
    dmaB(0);
    if (nkt > 1) dmaB(1);

    // ---- operand state (block-uniform): planes usable?
    // every header word is requested at once (one round trip under load instead of three dependent ones), then judged like
    // site_planes_ok (gemm_planes.h): scale > 0, flag down, max * s inside the fp16 window
    auto uni = [](float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };      // same in every lane: keep it scalar
    const float ha0 = q.A.hdr[0], ha1 = q.A.hdr[1], hb0 = q.B.hdr[0], hb1 = q.B.hdr[1];
    const f32x4 ama = *(const f32x4*)(q.A.hdr + SITE_HDR + lane * 4), amb = *(const f32x4*)(q.B.hdr + SITE_HDR + lane * 4);
    const float cs_in = (q.Cp && q.c_scale_in) ? *q.c_scale_in : 0.f;
    const float sa_hdr = uni(ha0), sb_hdr = uni(hb0);
    auto planes_ok = [&](float s, float flag, f32x4 v) {
        const float m = wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        if (!(s > 0.f) || __float_as_uint(flag) != 0u) return false;
        return !(m > 0.f) || ((m * s >= 0.25f || s >= 0x1p60f) && m * s < 65504.f);
    };
    const bool slowA = q.A.f32 != nullptr && !planes_ok(sa_hdr, uni(ha1), ama);
    const bool slowB = q.B.f32 != nullptr && !planes_ok(sb_hdr, uni(hb1), amb);
    const float c_scale = q.repair ? uni(c_repair) : uni(cs_in);

    // ---- fragment read addressing (lane: row l15 of a 16-row block, logical chunk 4 plane + lq; physical = logical ^ swz)
    const int swz = (l15 >> 1) & 7;
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int ch = ((4 * pl + lq) ^ swz) << 4;
        fa[pl] = (uint32_t)((grp * 128 + l15) * 128 + ch);
        fb[pl] = (uint32_t)(PBM * 128 + (wn * 16 * NJ + l15) * 128 + ch);
    }

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ah[4], al[4], bh[NJ], bl[NJ];
    auto readA = [&](const char* st, int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *(const f32x4*)(st + fa[0] + (mh * 4 + i) * 2048);
            al[i] = *(const f32x4*)(st + fa[1] + (mh * 4 + i) * 2048);
        }
    };
    auto readB = [&](const char* st) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[j] = *(const f32x4*)(st + fb[0] + j * 2048);
            bl[j] = *(const f32x4*)(st + fb[1] + j * 2048);
        }
    };
    auto mma = [&](auto mh_tag) {
        constexpr int mh = decltype(mh_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 c = acc[mh * 4 + i][j];
                c = mfma16(bh[j], al[i], c);          // (B fragment, A fragment): the accumulator tile is C^T
                c = mfma16(bl[j], ah[i], c);
                c = mfma16(bh[j], ah[i], c);
                acc[mh * 4 + i][j] = c;
            }
        __builtin_amdgcn_s_setprio(0);
    };

    float sa = sa_hdr, sb = sb_hdr;
    if (!(slowA || slowB)) {
        // ---- everyone waits for A(0), B(0); B(1) may still fly
        if (nkt > 1) end_compute_segment<NJ>(); else end_compute_segment<0>();
        if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }          // g1 runs one segment behind g0
        STAMP(1);
#pragma unroll 1
        for (int t = 0; t < nkt; ++t) {
            const char* st = smem + (t & 1) * PSTAGE;
            // L(t, P0)
            readB(st);
            readA(st, 0);
            if (t + 1 < nkt) dmaA(t + 1);
            end_load_segment();
            // C(t, P0)
            mma(std::integral_constant<int, 0>{});
            if (t + 1 < nkt) end_compute_segment<4>(); else end_compute_segment<0>();
            // L(t, P1)
            readA(st, 1);
            if (t + 2 < nkt) dmaB(t + 2);
            end_load_segment();
            // C(t, P1)
            mma(std::integral_constant<int, 1>{});
            if (t + 2 < nkt) end_compute_segment<NJ>(); else end_compute_segment<0>();
        }
        if (grp == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }          // barrier counts of g0 and g1 match
    } else {
        // ---- rare path (a delayed scale left its window): synchronous, stage 0 only, operands split from the fp32 copies
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the early pieces have landed before anything is restaged
        if (slowA) sa = site_exact_scale(q.A.hdr, (float*)(smem + 2 * PSTAGE - 64), tid, 512);
        if (slowB) sb = site_exact_scale(q.B.hdr, (float*)(smem + 2 * PSTAGE - 64), tid, 512);
        auto slow_stage = [&](const PlaneOperand& op, float sc, int row0, int nrows, int ntrows, int kt, char* dst) {
#pragma unroll 1
            for (int j = tid; j < ntrows * 4; j += 512) {
                const int row = j >> 2, kc = j & 3;
                const float* src = op.f32 + (size_t)min(row0 + row, nrows - 1) * op.ldf + kt * 32 + kc * 8;
                const f32x4 x0 = *(const f32x4*)src, x1 = *(const f32x4*)(src + 4);
                uint32_t h0, l0, h1, l1, h2, l2, h3, l3;
                splith_pair(x0.x, x0.y, sc, h0, l0); splith_pair(x0.z, x0.w, sc, h1, l1);
                splith_pair(x1.x, x1.y, sc, h2, l2); splith_pair(x1.z, x1.w, sc, h3, l3);
                const int sw = (row >> 1) & 7;
                *(uint4*)(dst + row * 128 + ((kc ^ sw) << 4)) = make_uint4(h0, h1, h2, h3);
                *(uint4*)(dst + row * 128 + (((4 + kc) ^ sw) << 4)) = make_uint4(l0, l1, l2, l3);
            }
        };
        auto dma_rows = [&](__amdgpu_buffer_rsrc_t rs, const PlaneOperand& op, int row0, int nrows, int ntrows, int kt, char* dst) {
#pragma unroll 1
            for (int pc = wave; pc < ntrows / 8; pc += 8) {
                const int row = pc * 8 + r8;
                lds_dma16(rs, dst + pc * 1024, (uint32_t)min(row0 + row, nrows - 1) * (uint32_t)op.ld2 * 2u +
                          (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16), (uint32_t)kt * 128u);
            }
        };
#pragma unroll 1
        for (int t = 0; t < nkt; ++t) {
            __syncthreads();
            if (slowA) slow_stage(q.A, sa, m0, p.M, PBM, t, smem); else dma_rows(rsA, q.A, m0, p.M, PBM, t, smem);
            if (slowB) slow_stage(q.B, sb, n0, p.N, BNW, t, smem + PBM * 128); else dma_rows(rsB, q.B, n0, p.N, BNW, t, smem + PBM * 128);
            dma_wait_barrier();
            readB(smem);
            readA(smem, 0);
            end_load_segment();
            mma(std::integral_constant<int, 0>{});
            readA(smem, 1);
            end_load_segment();
            mma(std::integral_constant<int, 1>{});
        }
        __syncthreads();
    }
    STAMP(2);

    // ================================================================ epilogue
    // lane holds C[gm = m0 + 128 grp + 16 i + l15][gn = n0 + 16 NJ wn + 16 j + 4 lq .. + 3] of tile (i, j).  Row blocks i are walked
    // in a rolled loop (the element-wise body is emitted NJ times, not 8 NJ).
    //
    // The "extra operand" E of an element -- the residual, or the aux tensor an activation gradient reads (the host routes
    // launches that would need both to gemm_pl_nt) -- is staged through LDS by LDS-DMA, a quarter of the tile (64 rows x 1 KB,
    // both wave groups' row blocks 2 q, 2 q + 1) at a time into the two 64 KB halves the k-loop has left free.  Why not plain
    // loads: vmcnt returns in ISSUE ORDER on gfx9, stores included -- a load queued behind the stores of the previous row block
    // cannot come back before those stores are acknowledged (~2 us under load), which serialised the whole epilogue (20 us per
    // tile).  Quarters 0 and 1 are requested before the first store, quarter q + 2 after the stores of quarter q: every wait
    // is for pieces that sit in front of stores issued a quarter earlier at least.  No extra operand: no loads, no waits, no
    // barriers.  Registers: none (the ring of prefetched rows it replaces spilled).
    float am = 0.f;
    if (SEGMM_GEMM_DBG(q) & 2) {
        float t = 0.f;          // timing ablation: keep every accumulator alive, skip the epilogue
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) t += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
        if (t == 1.2345f) p.C[0] = 1.f;
    } else {
        const float inv_ab = (1.f / sa) * (1.f / sb);          // exact powers of two
        const int epi = p.epi;
        const bool has_res = p.residual != nullptr, has_drop = p.drop.p > 0.f;
        const DropCfg drop_e = drop_live(p.drop);
        const bool aux_r = epi == EPI_DGELU || epi == EPI_DRELU, aux_w = epi == EPI_GELU;
        const bool planes = c_scale > 0.f && q.Cp != nullptr;
        const bool store_c = q.write_c && !(SEGMM_GEMM_DBG(q) & 1);
        const bool periodic = has_res && p.res_period < p.M;
        const int res_rows = has_res ? min(p.res_period, p.M) : 0;
        const bool has_e = has_res || aux_r;
        auto ext = [&](bool on, long long rows, long long ld, long long elt) -> uint32_t {      // view extent in bytes (0: absent)
            if (!on || rows <= 0) return 0u;
            return (uint32_t)(((rows - 1) * ld + p.N) * elt);          // < 2^31 (checked by the host)
        };
        const __amdgpu_buffer_rsrc_t rsC = make_rsrc(p.C, ext(store_c, p.M, p.ldc, 4));
        const __amdgpu_buffer_rsrc_t rsAuxW = make_rsrc(p.aux, ext(aux_w, p.M, p.ldaux, 4));
        const __amdgpu_buffer_rsrc_t rsE = aux_r ? make_rsrc(p.aux, ext(true, p.M, p.ldaux, 4)) : make_rsrc(p.residual, ext(has_res, res_rows, p.ldr, 4));
        const int ldE = aux_r ? p.ldaux : p.ldr;
        // planes: [M][ldc2] halves, a row holds 2 N halves
        const __amdgpu_buffer_rsrc_t rsPl = make_rsrc(q.Cp, planes ? (uint32_t)((((long long)p.M - 1) * q.ldc2 + 2ll * p.N) * 2) : 0u);
        const int ns = (store_c ? 1 : 0) + (planes ? 1 : 0) + (aux_w ? 1 : 0);          // store instructions per float4

        const int gm0 = m0 + grp * 128 + l15;
        const int gn0 = n0 + wn * 16 * NJ + 4 * lq;
        uint32_t colmask[NJ];          // 0 or BUF_OOB: columns beyond N are pushed out of every descriptor's range
        f32x4 bias4[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gn = gn0 + 16 * j;
            colmask[j] = gn < p.N ? 0u : BUF_OOB;
            bias4[j] = (p.bias && gn < p.N) ? *(const f32x4*)(p.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const bool full_tile = m0 + PBM <= p.M && n0 + BNW <= p.N;
        // E quarter qq -> LDS half (qq & 1): slot s = 32 g + r (g = wave group, r = row inside the group's 32 rows of the quarter) at
        // byte s * 1024; 16-byte chunk c of the row at physical chunk c ^ (row & 15) (conflict-free ds_read_b128 of the accumulator
        // layout: 16 rows x 4 chunks per instruction); the permutation is applied to the DMA source address
        auto dmaE = [&](int qq) {
            char* dst = smem + (qq & 1) * 65536;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int slot = wave * 8 + k;
                const int R = (slot >> 5) * 128 + qq * 32 + (slot & 31);          // tile row (wave-uniform)
                const int gmR = m0 + R;
                const int er = aux_r ? gmR : (periodic ? gmR % p.res_period : gmR);
                const int ch = lane ^ (R & 15);
                const uint32_t vo = (ch < 16 * NJ && n0 + 4 * ch < p.N) ? (uint32_t)ch * 16u : BUF_OOB;
                lds_dma16e(rsE, dst + slot * 1024, vo, ((uint32_t)er * (uint32_t)ldE + (uint32_t)n0) * 4u);
            }
        };
        auto vmwait = [&](int kind) {          // kind 0: 8 newer ops; 1: S + 8; 2: S newer ops, S = 8 ns store instructions of the last quarter
            __builtin_amdgcn_sched_barrier(0);          // (2 row blocks x 4 row-segment stores per output tensor)
            if (kind == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (ns == 1) { if (kind == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else if (ns == 2) { if (kind == 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
            else if (ns == 3) { if (kind == 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- stores.  In the accumulator layout a lane owns 16 bytes of ITS row: the 16 lanes of a quarter-wave touch 16
        // different cache lines, 64 sixteen-byte transactions per store instruction -- 18.8 k cycles to drain a 256 KB tile on an
        // otherwise idle chip against 5.4 k for stores that cover 256 contiguous bytes per quarter-wave (tools/probe/store_rate.hip).
        // Each 16-row x 16 NJ-column strip therefore takes a detour through a wave-private 4 KB LDS patch (chunk c of row r at
        // physical chunk c ^ r: conflict-free both ways) and is stored as whole 64 NJ-byte row segments: lane (lq, l15) of pass t
        // stores row 4 t + lq, columns 4 l15 .. 4 l15 + 3 of the strip.
        char* trp = smem + 2 * PSTAGE + wave * 4096;
        const uint32_t tr_w = (uint32_t)(l15 * 256);                                   // + ((lq + 4 j) ^ l15) << 4
        const int gnT = n0 + wn * 16 * NJ + 4 * l15;
        const uint32_t tmask = (l15 < 4 * NJ && gnT < p.N) ? 0u : BUF_OOB;
        const uint32_t oCT = (((uint32_t)(m0 + grp * 128 + lq) * (uint32_t)p.ldc + (uint32_t)gnT) * 4u) | tmask;
        const uint32_t oAuxT = (((uint32_t)(m0 + grp * 128 + lq) * (uint32_t)p.ldaux + (uint32_t)gnT) * 4u) | tmask;
        auto tr_put = [&](int j, f32x4 v) { *(f32x4*)(trp + tr_w + (((lq + 4 * j) ^ l15) << 4)) = v; };
        auto tr_get = [&](int t) { const int r = 4 * t + lq; return *(const f32x4*)(trp + r * 256 + (((l15 ^ r) & 15) << 4)); };
        // planes of a transposed float4: adjacent lanes hold adjacent column groups of one row (plane_store4_pair's pattern)
        const uint32_t oPlT = (((uint32_t)(m0 + grp * 128 + lq) * (uint32_t)q.ldc2 + (uint32_t)((((gnT & ~7) >> 5) << 6) + ((gnT & ~7) & 31) + ((l15 & 1) ? 32 : 0))) * 2u) | tmask;

        // The row-block loop exists in twelve copies -- activation class (none / ReLU-type / GELU-type) x dropout x plane output
        // fixed at compile time -- picked once per tile: with every option tested inside ONE body, the dozen taken branches per
        // float4 (each hopping over an inlined erf) and ~60 VALU instructions cost more than the stores (21 k cycles per tile,
        // the same on an idle chip).
        if (has_e) { dmaE(0); dmaE(1); }
        const uint32_t e_lane = (uint32_t)((grp * 32 + l15) * 1024);          // + 16384 for odd row blocks; chunk ((4 NJ wn + 4 j + lq) ^ l15) * 16
        auto row_loop = [&](auto act_tag, auto drop_tag, auto pl_tag) {
            constexpr int ACT = decltype(act_tag)::value;          // 0 none, 1 ReLU / ReLU', 2 GELU / GELU'
            constexpr bool DROP = decltype(drop_tag)::value, PLANES = decltype(pl_tag)::value;
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                if (has_e && (i & 1) == 0) vmwait(i == 0 ? 0 : (i == 6 ? 2 : 1));          // quarter i / 2 has landed (all waves: barrier)
                f32x4 c[NJ];
                switch (i) {
                    case 0: for (int j = 0; j < NJ; ++j) c[j] = acc[0][j]; break;
                    case 1: for (int j = 0; j < NJ; ++j) c[j] = acc[1][j]; break;
                    case 2: for (int j = 0; j < NJ; ++j) c[j] = acc[2][j]; break;
                    case 3: for (int j = 0; j < NJ; ++j) c[j] = acc[3][j]; break;
                    case 4: for (int j = 0; j < NJ; ++j) c[j] = acc[4][j]; break;
                    case 5: for (int j = 0; j < NJ; ++j) c[j] = acc[5][j]; break;
                    case 6: for (int j = 0; j < NJ; ++j) c[j] = acc[6][j]; break;
                    default: for (int j = 0; j < NJ; ++j) c[j] = acc[7][j]; break;
                }
                const int gm = gm0 + 16 * i;
                const uint32_t rowmask = gm < p.M ? 0xffffffffu : 0u;
                const uint32_t soC = (uint32_t)i * 16u * (uint32_t)p.ldc * 4u, soAux = (uint32_t)i * 16u * (uint32_t)p.ldaux * 4u;
                const char* ebuf = smem + ((i >> 1) & 1) * 65536 + e_lane + (i & 1) * 16384;
                f32x4 e[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) e[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_e) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) e[j] = *(const f32x4*)(ebuf + (((4 * NJ * wn + 4 * j + lq) ^ l15) << 4));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = c[j] * inv_ab + bias4[j];
                    if (ACT == 2) {
                        if (epi == EPI_GELU) {
                            tr_put(j, v);          // the pre-activation leaves through the transpose patch below
                            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                        } else {
                            v.x *= gelu_erf_grad(e[j].x); v.y *= gelu_erf_grad(e[j].y); v.z *= gelu_erf_grad(e[j].z); v.w *= gelu_erf_grad(e[j].w);
                        }
                    } else if (ACT == 1) {
                        if (epi == EPI_RELU) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        } else {
                            v.x = e[j].x > 0.f ? v.x : 0.f; v.y = e[j].y > 0.f ? v.y : 0.f; v.z = e[j].z > 0.f ? v.z : 0.f; v.w = e[j].w > 0.f ? v.w : 0.f;
                        }
                    }
                    if (DROP) v = drop_apply4(drop_e, ((uint64_t)gm * (uint64_t)p.N + (uint64_t)(gn0 + 16 * j)) >> 2, v);
                    if (ACT == 0) v += e[j];                    // e = 0 without a residual
                    else if (has_res) v += e[j];                // (e is the aux tensor of an activation gradient otherwise)
                    c[j] = v;
                    {          // running max |v| over the elements that exist (branch-free: rows / columns beyond the matrix are masked to 0)
                        const uint32_t mk = rowmask & ~((int32_t)colmask[j] >> 31);
                        const float mx = __uint_as_float(__float_as_uint(v.x) & mk), my = __uint_as_float(__float_as_uint(v.y) & mk);
                        const float mz = __uint_as_float(__float_as_uint(v.z) & mk), mw = __uint_as_float(__float_as_uint(v.w) & mk);
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(am) : "v"(mx), "v"(my));
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(am) : "v"(mz), "v"(mw));
                    }
                }
                if (ACT == 2 && epi == EPI_GELU) {          // the pre-activations (put above), as whole row segments
#pragma unroll
                    for (int t = 0; t < 4; ++t) buf_store4(rsAuxW, oAuxT, soAux + (uint32_t)(4 * t) * (uint32_t)p.ldaux * 4u, tr_get(t));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) tr_put(j, c[j]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 v = tr_get(t);
                    buf_store4(rsC, oCT, soC + (uint32_t)(4 * t) * (uint32_t)p.ldc * 4u, v);
                    if (PLANES) {
                        // adjacent lanes hold adjacent float4 column groups of one row: the pair trades half of its terms through
                        // DPP -- the even lane stores the 8 hi terms, the odd lane the 8 lo terms of the aligned 8 columns
                        uint32_t h0, l0, h1, l1;
                        splith_pair(v.x, v.y, c_scale, h0, l0);
                        splith_pair(v.z, v.w, c_scale, h1, l1);
                        const bool oddl = (l15 & 1) != 0;
                        const uint32_t r0 = dpp_swap1(oddl ? h0 : l0), r1 = dpp_swap1(oddl ? h1 : l1);
                        const u32x4_t w = oddl ? u32x4_t{r0, r1, l0, l1} : u32x4_t{h0, h1, r0, r1};
                        buf_store4u_aux<SEGMM_PLANE_AUX>(rsPl, oPlT, (uint32_t)(16 * i + 4 * t) * (uint32_t)q.ldc2 * 2u, w);
                    }
                }
                if (has_e && (i & 1) == 1 && i < 5) {          // both row blocks of the quarter are read: refill its half with quarter + 2
                    end_load_segment();
                    dmaE((i >> 1) + 2);
                }
            }
        };
        // The common case -- a whole tile, no activation, no dropout, no plane output (the fused projections, the input-gradient
        // GEMMs): fully unrolled, ~6 instructions per float4 (the rolled loop spends ~1 200 cycles per row block on its 8-way
        // accumulator switch and scalar bookkeeping: 9.5 k cycles per tile before the first byte is stored)
        auto fast_loop = [&](auto e_tag) {
            constexpr bool HAS_E = decltype(e_tag)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (HAS_E && (i & 1) == 0) vmwait(i == 0 ? 0 : (i == 6 ? 2 : 1));
                const uint32_t soC = (uint32_t)i * 16u * (uint32_t)p.ldc * 4u;
                const char* ebuf = smem + ((i >> 1) & 1) * 65536 + e_lane + (i & 1) * 16384;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = acc[i][j] * inv_ab + bias4[j];
                    if (HAS_E) v += *(const f32x4*)(ebuf + (((4 * NJ * wn + 4 * j + lq) ^ l15) << 4));
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(am) : "v"(v.x), "v"(v.y));
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(am) : "v"(v.z), "v"(v.w));
                    tr_put(j, v);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) buf_store4(rsC, oCT, soC + (uint32_t)(4 * t) * (uint32_t)p.ldc * 4u, tr_get(t));
                if (HAS_E && (i & 1) == 1 && i < 5) {
                    end_load_segment();
                    dmaE((i >> 1) + 2);
                }
            }
        };
        const bool fast = full_tile && epi == EPI_NONE && !has_drop && !planes;
        if (fast) {
            if (has_e) fast_loop(std::true_type{}); else fast_loop(std::false_type{});
        } else {
        using A0 = std::integral_constant<int, 0>; using A1 = std::integral_constant<int, 1>; using A2 = std::integral_constant<int, 2>;
        using T = std::true_type; using F = std::false_type;
        auto pick = [&](auto act_tag) {
            if (has_drop) { if (planes) row_loop(act_tag, T{}, T{}); else row_loop(act_tag, T{}, F{}); }
            else { if (planes) row_loop(act_tag, F{}, T{}); else row_loop(act_tag, F{}, F{}); }
        };
        if (epi == EPI_GELU || epi == EPI_DGELU) pick(A2{});
        else if (epi == EPI_RELU || epi == EPI_DRELU) pick(A1{});
        else pick(A0{});
        }
    }
    STAMP(3);
    if (q.repair) return;          // (the header keeps the first launch's verdict)
    if (q.c_hdr) {
        site_commit(q.c_hdr, am, blockIdx.x * 8 + wave, c_scale);
        if (c_scale > 0.f && scale_writer(blockIdx.x * 8 + wave)) q.c_hdr[0] = c_scale;
    } else if (p.amax_out) amax_commit(p.amax_out, am, blockIdx.x * 8 + wave);
}

}  // namespace segmm

namespace segmm {

// =============================================================================== TN, round-3 form
// Weight gradients gW[M, N] = A[K, M]^T . B[K, N] over the token axis K (the SLOW axis of both operands), split-K over
// blockIdx.z -- the same ping-pong structure as gemm_pl_nt8 on v_mfma_f32_16x16x32_f16, with the operand handling of
// gemm_pl_tn (gemm_planes.h): a k-tile is 32 token rows of 1 KB per operand (256 features x [hi | lo]), one LDS-DMA
// wave-instruction per row, fragments by ds_read_b64_tr_b16 (hardware transpose: per 16-lane group 4 tokens x 16 features,
// a lane ends up with 8 consecutive tokens of ITS feature -- the 16 x 16 x 32 operand form directly).
//
// LDS image of a token row (1 KB): 16 pieces of 64 B (piece = 2 * feature block + plane); piece c of token t sits at
// physical piece c ^ (t & 3) -- the four token rows of a transposed read fall into four different 64-byte bank windows -- and
// inside a piece the two 32-byte halves (16 features each) are swapped for tokens with bit 3 set: the two 16-lane groups of a
// 32-lane half read the SAME 16 features of tokens 8 apart (16 x 16 x 32: lane group = token octet), which would otherwise hit
// the same banks twice.  Both permutations are applied to the DMA source address.
//
// Schedule per k-tile t (stage t & 1; g0 = waves 0-3 = features 0-127 of A, g1 one segment behind):
//     L(t,P0): tr-reads of B (all 64 columns of the wave) and A sub-tile 0; DMA of the group's 16 token rows of A(t+1)
//     C(t,P0): 48 MFMAs                                   | vmcnt(4): B(t+1) has landed
//     L(t,P1): tr-reads of A sub-tile 1; DMA of the group's 16 rows of B(t+2)   | vmcnt(4): A(t+1) has landed
//     C(t,P1): 48 MFMAs
// Every wave reads ALL 32 token rows of a stage (its own feature columns), so a row of A(t+1) may be requested only after both
// groups' L(t-1,P1) (true from L(t,P0) on) and must have landed before g0's L(t+1,P0): the issuing wave waits for it at the
// end of its own L(t,P1).  B(t+2) replaces B(t), last read in L(t,P0) of both groups.
//
// The bias gradient (column sums of A over k) rides along in the workgroups of the first column tile: wave (wm, wn) adds
// one MFMA pair per phase against an all-ones fragment for its m-tile wn.  Output: split-K slabs (plain float4 stores from the
// accumulators, C^T trick as in gemm_pl_nt8) combined by splitk_reduce, or C itself for a single split.
__device__ __forceinline__ f32x4 lds_tr8b(const char* a) {          // 8 tokens (two 4-token blocks, 4 KB apart) of this lane's feature
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4096));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f32x4, v);
}

__global__ __launch_bounds__(512, 2) void gemm_pl_tn8(const GemmArgs p, const PGemmX q) {
    // two stages x (A: 32 tokens x 1 KB | B: 32 tokens x 1 KB) + a 4 KB transpose patch per wave for the output stores
    __shared__ __attribute__((aligned(16))) char smem[2 * PSTAGE + 8 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ntile = p.nbm * p.nbn;
    const int lg = xcd_remap(blockIdx.x + ntile * blockIdx.z, ntile * gridDim.z);          // tiles of one token slab meet in one L2
    const int kz = lg / ntile, lb = lg - kz * ntile;
    const int m0 = (lb / p.nbn) * PBM, n0 = (lb % p.nbn) * PBN;
    const int kbeg = kz * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg + 31) >> 5;
    const bool do_colsum = q.colsum_out != nullptr && (lb % p.nbn) == 0;

    // ---- LDS-DMA: piece = one token row (1 KB); group g fetches rows t = 16 g + 4 wn + i (i < 4) of both operands.  The row
    // and the tile's feature offset are wave-uniform (scalar offset of the instruction); per lane only the position inside the row:
    // physical 16-byte chunk `lane` of row t holds logical piece (lane >> 2) ^ (t & 3), chunk (lane & 3) ^ (2 * bit 3 of t) --
    // t & 3 = i and bit 3 of t = wn >> 1 here, so the offset of piece i is inrow0 ^ (i << 6)
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(q.A.p, q.A.bytes), rsB = make_rsrc(q.B.p, q.B.bytes);
    const int t0 = 16 * grp + 4 * wn;
    const uint32_t inrow0 = (uint32_t)(((lane >> 2) << 6) + (((lane & 3) ^ ((wn >> 1) << 1)) << 4));
    auto dmaA_pl = [&](int kt) {
        char* st = smem + (kt & 1) * PSTAGE + t0 * 1024;
        const uint32_t so = (uint32_t)(kbeg + kt * 32 + t0) * (uint32_t)q.A.ld2 * 2u + (uint32_t)m0 * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16t(rsA, st + i * 1024, inrow0 ^ (uint32_t)(i << 6), so + (uint32_t)i * (uint32_t)q.A.ld2 * 2u);
    };
    auto dmaB_pl = [&](int kt) {
        char* st = smem + (kt & 1) * PSTAGE + 32768 + t0 * 1024;
        const uint32_t so = (uint32_t)(kbeg + kt * 32 + t0) * (uint32_t)q.B.ld2 * 2u + (uint32_t)n0 * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16t(rsB, st + i * 1024, inrow0 ^ (uint32_t)(i << 6), so + (uint32_t)i * (uint32_t)q.B.ld2 * 2u);
    };
    dmaA_pl(0);
    dmaB_pl(0);
    if (nkt > 1) dmaB_pl(1);

    // ---- operand state (all header words requested at once)
    auto uni = [](float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
    const float ha0 = q.A.hdr[0], ha1 = q.A.hdr[1], hb0 = q.B.hdr[0], hb1 = q.B.hdr[1];
    const f32x4 ama = *(const f32x4*)(q.A.hdr + SITE_HDR + lane * 4), amb = *(const f32x4*)(q.B.hdr + SITE_HDR + lane * 4);
    const float sa0 = uni(ha0), sb0 = uni(hb0);
    auto planes_ok = [&](float s, float flag, f32x4 v) {
        const float m = wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        if (!(s > 0.f) || __float_as_uint(flag) != 0u) return false;
        return !(m > 0.f) || ((m * s >= 0.25f || s >= 0x1p60f) && m * s < 65504.f);
    };
    const bool slowA = q.A.f32 != nullptr && !planes_ok(sa0, uni(ha1), ama);          // delayed scale outside its window: fp32 fallback
    const bool slowB = q.B.f32 != nullptr && !planes_ok(sb0, uni(hb1), amb);
    const bool slow = slowA || slowB;
    // Fallback (rare): the SAME schedule, with the group's share of a stage written by ds_write from the operand's fp32 copy --
    // split with the exact scale of its recorded maxima -- instead of by LDS-DMA; every counted vmcnt wait becomes vmcnt(0) then
    // (the counts assume four pieces of the other operand behind them).  A wave converts its 4 token rows, 4 features per lane and row.
    float sa = sa0, sb = sb0;
    if (slow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the early pieces have landed before anything is restaged
        if (slowA) sa = site_exact_scale(q.A.hdr, (float*)(smem + 2 * PSTAGE - 64), tid, 512);
        if (slowB) sb = site_exact_scale(q.B.hdr, (float*)(smem + 2 * PSTAGE - 64), tid, 512);
    }
    auto stage_f32 = [&](const PlaneOperand& op, float sc, int kt, int f0, int nfeat, char* dst) {
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + i, gk = kbeg + kt * 32 + t, gf = f0 + lane * 4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (gk < kend && gf < nfeat) x = *(const f32x4*)(op.f32 + (size_t)gk * op.ldf + gf);
            uint32_t hh0, l0, hh1, l1;
            splith_pair(x.x, x.y, sc, hh0, l0); splith_pair(x.z, x.w, sc, hh1, l1);
            // 8-byte group g8 = lane & 7 of feature block b = lane >> 3: 16-byte chunk g8 >> 1 (halves swapped for tokens with bit 3 set)
            const int b = lane >> 3, g8 = lane & 7, cp = (g8 >> 1) ^ (((t >> 3) & 1) << 1), sw = t & 3;
            *(uint2*)(dst + t * 1024 + (((2 * b) ^ sw) << 6) + (cp << 4) + ((g8 & 1) << 3)) = make_uint2(hh0, hh1);
            *(uint2*)(dst + t * 1024 + (((2 * b + 1) ^ sw) << 6) + (cp << 4) + ((g8 & 1) << 3)) = make_uint2(l0, l1);
        }
    };
    auto dmaA = [&](int kt) { if (slowA) stage_f32(q.A, sa, kt, m0, p.M, smem + (kt & 1) * PSTAGE); else dmaA_pl(kt); };
    auto dmaB = [&](int kt) { if (slowB) stage_f32(q.B, sb, kt, n0, p.N, smem + (kt & 1) * PSTAGE + 32768); else dmaB_pl(kt); };
    if (slow) {          // restage what the early pieces brought
        __syncthreads();
        dmaA(0);
        dmaB(0);
        if (nkt > 1) dmaB(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the ds_writes of the fallback are retired before the first barrier)
    }

    // ---- transposed fragment reads: lane = (g = lq: token octet, qq = (lane >> 2) & 3: token inside a 4-block, pp = lane & 3)
    const int qq = (lane >> 2) & 3, pp = lane & 3;
    const uint32_t lane_base = (uint32_t)((8 * lq + qq) * 1024 + 4 * pp * 2);
    const uint32_t hsw = (uint32_t)((lq & 1) << 5);          // tokens 8 .. 15 and 24 .. 31 (bit 3 set): the 32-byte halves of a piece are swapped
    // A tile i (16 features) of phase mh, plane pl: lane part fr[i >> 1][pl] + (32 (i & 1)) ^ hsw, uniform part (2 wm + mh) * 256;
    // B tile j: the same lane part, uniform part 32768 + 256 wn
    uint32_t fr[2][2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fr[hb][pl] = lane_base + (uint32_t)(((2 * hb + pl) ^ qq) << 6);
    const uint32_t h0 = hsw, h1 = 32u ^ hsw;          // byte offset of the even / odd 16-feature half inside the piece
    const int ua = grp * 2 * 256, ub = 32768 + wn * 256;          // wave-uniform parts

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 ah[4], al[4], bh[4], bl[4];
    auto readA = [&](const char* st, int mh) {
        const char* sa_ = st + ua + mh * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = lds_tr8b(sa_ + fr[i >> 1][0] + ((i & 1) ? h1 : h0));
            al[i] = lds_tr8b(sa_ + fr[i >> 1][1] + ((i & 1) ? h1 : h0));
        }
    };
    auto readB = [&](const char* st) {
        const char* sb_ = st + ub;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = lds_tr8b(sb_ + fr[j >> 1][0] + ((j & 1) ? h1 : h0));
            bl[j] = lds_tr8b(sb_ + fr[j >> 1][1] + ((j & 1) ? h1 : h0));
        }
    };
    auto mma = [&](auto mh_tag, auto cs_tag) {
        constexpr int mh = decltype(mh_tag)::value;
        constexpr bool CS = decltype(cs_tag)::value;
        __builtin_amdgcn_s_setprio(1);
        if (CS) {
            const f32x4 ones = __builtin_bit_cast(f32x4, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));      // 8 x fp16 1.0
            // the column sums of this wave's A tile i = wn of the phase (a wave-uniform choice among fragments it holds anyway)
            f32x4 cb = accb[mh];
            switch (wn) {
                case 0: cb = mfma16(ones, al[0], cb); cb = mfma16(ones, ah[0], cb); break;
                case 1: cb = mfma16(ones, al[1], cb); cb = mfma16(ones, ah[1], cb); break;
                case 2: cb = mfma16(ones, al[2], cb); cb = mfma16(ones, ah[2], cb); break;
                default: cb = mfma16(ones, al[3], cb); cb = mfma16(ones, ah[3], cb); break;
            }
            accb[mh] = cb;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 c = acc[mh * 4 + i][j];
                c = mfma16(bh[j], al[i], c);
                c = mfma16(bl[j], ah[i], c);
                c = mfma16(bh[j], ah[i], c);
                acc[mh * 4 + i][j] = c;
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto end_load_vm4 = [&](bool wait4, bool wait0) {          // close a load segment; A(t+1) must have landed when asked
        __builtin_amdgcn_sched_barrier(0);
        if (wait4 && !slow) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (wait0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto k_loop = [&](auto cs_tag) {
        constexpr bool CS = decltype(cs_tag)::value;
        if (nkt > 1 && !slow) end_compute_segment<4>(); else end_compute_segment<0>();          // A(0), B(0) landed; B(1) may fly
        if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll 1
        for (int t = 0; t < nkt; ++t) {
            const char* st = smem + (t & 1) * PSTAGE;
            readB(st);
            readA(st, 0);
            if (t + 1 < nkt) dmaA(t + 1);
            end_load_segment();
            mma(std::integral_constant<int, 0>{}, cs_tag);
            if (t + 1 < nkt && !slow) end_compute_segment<4>(); else end_compute_segment<0>();          // B(t+1) landed (A(t+1) may fly)
            readA(st, 1);
            if (t + 2 < nkt) dmaB(t + 2);
            end_load_vm4(t + 2 < nkt, t + 1 < nkt);                                             // A(t+1) landed (B(t+2) may fly)
            mma(std::integral_constant<int, 1>{}, cs_tag);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    };
    if (do_colsum) k_loop(std::true_type{}); else k_loop(std::false_type{});

    // ---- outputs
    const bool split = gridDim.z > 1;
    const float inv_a = 1.f / sa;
    const float nanv = 0.f;
    if (do_colsum && lq == 0) {          // every row of accb holds the column sums: lanes of column group 0 own 16 features each
        float* dst = split ? q.colsum_ws + (size_t)kz * p.M : q.colsum_out;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int m = m0 + grp * 128 + mh * 64 + 16 * wn + l15;
            if (m < p.M) dst[m] = accb[mh].x * inv_a + nanv;
        }
    }
    const float inv_ab = inv_a * (1.f / sb);
    float* Cout = split ? p.C + (size_t)kz * (size_t)p.slab_stride : p.C;
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cout, (uint32_t)((((long long)p.M - 1) * p.ldc + p.N) * 4));
    // whole 256-byte row segments through the wave's transpose patch (see gemm_pl_nt8: 16-byte-per-row stores drain 3.5x slower)
    char* trp = smem + 2 * PSTAGE + wave * 4096;
    const int gnT = n0 + wn * 64 + 4 * l15;
    const uint32_t oCT = (((uint32_t)(m0 + grp * 128 + lq) * (uint32_t)p.ldc + (uint32_t)gnT) * 4u) | (gnT < p.N ? 0u : BUF_OOB);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(trp + l15 * 256 + (((lq + 4 * j) ^ l15) << 4)) = acc[i][j] * inv_ab + nanv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * t + lq;
            buf_store4k(rsC, oCT, (uint32_t)(16 * i + 4 * t) * (uint32_t)p.ldc * 4u, *(const f32x4*)(trp + r * 256 + (((l15 ^ r) & 15) << 4)));
        }
    }
}

}  // namespace segmm
