// Segment attention forward on PRODUCER-WRITTEN P32 planes (round 5).
//
// Reference semantics: MMinterest/models/encoder.py:44-73,138-161 (see attention.h).  Same masks, dropout stream, softmax
// statistics and outputs as attn_fwd_kernel; what changes is where the operands come from and which matrix cores multiply them:
//   * Q, K, V arrive as the P32 fp16 planes the fused projection GEMMs wrote beside their fp32 output (x s = hi + lo with the
//     delayed power-of-two scale s of the tensor SITE, gemm_planes8.h) -- the kernel does no operand conversion at all;
//   * ONE workgroup owns one (b, h).  Its waves issue the head's whole K / V working set (both key blocks, hi and lo terms:
//     52.5 KB at config 2) as LDS-DMA (`buffer_load ... lds`, 16 B per lane, no registers), one round of memory latency;
//   * every product is three v_mfma_f32_16x16x16_f16 (hi hi + lo hi + hi lo, fp32 accumulate: 22-bit operands, the GEMM
//     engine's arithmetic) instead of four v_mfma_f32_16x16x4_f32 per k = 16 block: 57 instead of 128 matrix-pipe cycles.
// LDS image (nothing else lives in LDS: three heads fit a CU at config 2): rows of DH/4 16-byte chunks -- the DH/8 chunks of
// 8 hi terms, then the DH/8 chunks of 8 lo terms of the head's columns -- in the order [Ka rows][Va rows][Kb rows][Vb rows],
// NO pad rows: keys are walked as ONE flat list of La + Lb keys in tiles of 16 (the tile that straddles the two blocks is
// multiplied against both query projections and merged per key; reads of the keys behind the last one are clamped to it and
// get probability 0).  A row's chunks are rotated by a function of the key index so that both read patterns are conflict-free
// at the 12-chunk pitch of dh = 48: the S^T product reads K by rows (16 keys x one chunk per 32-lane half: rotation (key >> 2)
// & 3), the O^T product reads V transposed through ds_read_b64_tr_b16 (8 keys x two adjacent chunks per half: rotation
// 2 ((key >> 2) & 1)).  LDS-DMA writes lane-linearly, so the rotation is applied to the SOURCE address of each lane.
// A site whose planes are unusable (overflow flag up, maximum below the fp16 window, no scale) is staged by the same workgroup
// from the fp32 copy with the exact scale of the site's recorded maxima (ds_write; slower, rare, same arithmetic).
#pragma once
#include "attention16.h"

namespace segmm {

constexpr int ATT_PL_MAXW = 7;            // query tiles (= waves) per head

template <int DH> __host__ __device__ constexpr int att_pl_cpr() { return DH / 4; }          // 16-byte chunks per staged row (hi + lo)
template <int DH>
inline size_t attn_fwd_pl_lds_bytes(int La, int Lb) {
    return (size_t)2 * (La + Lb) * att_pl_cpr<DH>() * 16;
}

__device__ __forceinline__ u32x2a lds_b64(const char* a) { return *(const u32x2a*)a; }
// byte offset inside a plane row of the 16-byte chunk holding the hi terms of head-relative columns [c, c + 8) (c % 8 == 0)
__device__ __forceinline__ uint32_t p32_chunk_off(int c) { return (uint32_t)(((c >> 5) << 6) + (c & 31)) * 2u; }

// LA >= 0: the length of key block a is a compile-time constant (tile classes resolved at compile time: the hot shapes run a
// branch-free unrolled body the scheduler can software-pipeline); LA = -1: run-time length (wave-uniform branches per tile).
// EXACT: all NT key tiles exist (16 (NT - 1) < La + Lb <= 16 NT).
template <int DH, int NT, int LA, bool EXACT>
__global__ __launch_bounds__(64 * ATT_PL_MAXW) void attn_fwd_pl_kernel(const AttnArgs p) {
    static_assert(DH % 16 == 0, "planes-in attention: head dim must be a multiple of 16");
    constexpr int CPR = DH / 4, HC = DH / 8, NCH = DH / 16;
    constexpr int ROWB = CPR * 16;
    const DropCfg drop_ = drop_live(p.drop);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_pl[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / p.H, h = bh % p.H;
    const int La = LA >= 0 ? LA : p.La, Lb = p.Lb, T = La + Lb;
    const int La_p = round16(La), Lb_p = round16(Lb), Tp = La_p + Lb_p;
    const int col0 = h * DH;
    const AttnInPlanes& in = p.in;
    auto has_tile = [&](int t) { return EXACT || 16 * t < T; };

    // ================= prologue: EVERY load of the head is requested before anything is waited for (one round of latency) ==========
    // ---- site headers: scale, overflow flag, four partial maxima per lane (judged below)
    const float hq0 = in.hdr_q[0], ha0 = in.hdr_ka[0], hb0 = in.hdr_kb[0];
    const uint32_t hq1 = __float_as_uint(in.hdr_q[1]), ha1 = __float_as_uint(in.hdr_ka[1]), hb1 = __float_as_uint(in.hdr_kb[1]);
    const f32x4 mq4 = *(const f32x4*)(in.hdr_q + SITE_HDR + lane * 4), ma4 = *(const f32x4*)(in.hdr_ka + SITE_HDR + lane * 4),
                mb4 = *(const f32x4*)(in.hdr_kb + SITE_HDR + lane * 4);
    // ---- this wave's query tile; Q fragments straight to registers: lane (query l15, g) holds columns 16 i + 4 g .. + 3
    const int qt = wave;
    const int qi = 16 * qt + l15;
    const bool q_in = qi < p.Lq;
    const size_t qrow = (size_t)b * p.Lq + min(qi, p.Lq - 1);
    HL qa[NCH], qb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = col0 + 16 * i + 4 * g;
        const uint32_t eo = (uint32_t)(((c >> 5) << 6) + (c & 31));
        const _Float16* ra = in.Qa + qrow * in.ldq2 + eo;
        const _Float16* rb = in.Qb + qrow * in.ldq2 + eo;
        const uint2 ah = *(const uint2*)ra, al = *(const uint2*)(ra + 32), bh_ = *(const uint2*)rb, bl = *(const uint2*)(rb + 32);
        qa[i] = HL{ah.x, ah.y, al.x, al.y};
        qb[i] = HL{bh_.x, bh_.y, bl.x, bl.y};
    }
    const uint8_t mq_byte = p.mq[qrow];
    // ---- key flags of this lane's 4 keys per tile (flat keys 16 t + 4 g .. + 3; La, Lb % 4 == 0: a quad never straddles): one
    // dword per tile from the batch row's mask bytes, through a descriptor that covers exactly that row -- a quad behind the end of
    // its block reads 0, so "block a word | block b word" needs no address select
    uint32_t kfl[NT];
    {
        const __amdgpu_buffer_rsrc_t rma = make_rsrc(p.mka + (size_t)b * La, (uint32_t)La), rmb = make_rsrc(p.mkb + (size_t)b * Lb, (uint32_t)Lb);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int f0 = 16 * t + 4 * g;
            uint32_t w = 0u;
            if (LA < 0 || 16 * t < La) w = __builtin_amdgcn_raw_buffer_load_b32(rma, f0, 0, 0);
            if (LA < 0 || 16 * t + 16 > La) w |= __builtin_amdgcn_raw_buffer_load_b32(rmb, f0 - La, 0, 0);
            kfl[t] = w;
        }
    }

    // ---- stage K and V of both key blocks (speculatively from the planes; judged afterwards).  Image = [Ka rows | Va rows | Kb rows |
    // Vb rows], a row = CPR chunks, position pos of a row holds its chunk (pos - rot) mod CPR.  The arrays are staged in blocks of
    // 16 rows = CPR / 4 instructions of 64 chunks: the (row in block, position) of a lane, the rotation of its row and with them
    // the lane's source offset are THE SAME for every block of an array, so they are computed once (4 x CPR / 4 registers) and a
    // block costs no vector arithmetic: its first row travels in the scalar offset.  Lanes whose row lies behind the end of the
    // array are masked off (an LDS-DMA instruction only writes the slots of its active lanes).
    constexpr int IPB = CPR / 4;                        // instructions per 16-row block
    const int nba = (La + 15) >> 4, nbb = (Lb + 15) >> 4;
#ifdef SEGMM_ATT_PROBE
    if (!(p.pflags & 512))               // timing probe: no staging
#endif
    {
        const __amdgpu_buffer_rsrc_t rsA = make_rsrc(in.baseA, in.bytesA), rsB = make_rsrc(in.baseB, in.bytesB);
        uint32_t vK[2][IPB], vV[2][IPB];                // [region][instruction of the block]: this lane's source offset in the block
        int rr_[IPB];
#pragma unroll
        for (int i = 0; i < IPB; ++i) {
            const int sl = 64 * i + lane, rr = sl / CPR, pos = sl - rr * CPR;
            rr_[i] = rr;
#pragma unroll
            for (int reg = 0; reg < 2; ++reg) {
                const int q4 = (rr >> 2) + (reg ? (La >> 2) : 0);          // (flat key >> 2) up to a multiple of 4
                int jk = pos - (q4 & 3); if (jk < 0) jk += CPR;
                int jv = pos - 2 * (q4 & 1); if (jv < 0) jv += CPR;
                const uint32_t rowb = (uint32_t)rr * (uint32_t)(reg ? in.ldkb2 : in.ldka2) * 2u;
                vK[reg][i] = rowb + p32_chunk_off(col0 + 8 * (jk >= HC ? jk - HC : jk)) + (jk >= HC ? 64u : 0u);
                vV[reg][i] = rowb + p32_chunk_off(col0 + 8 * (jv >= HC ? jv - HC : jv)) + (jv >= HC ? 64u : 0u);
            }
        }
        // the blocks of the four arrays are dealt round-robin over the waves (K first: the S^T product needs it first)
        auto stage_array = [&](const __amdgpu_buffer_rsrc_t rs, const uint32_t (&vo)[IPB], int L, uint32_t so0, uint32_t ld2b, char* dst0, int first) {
            for (int m = first; 16 * m < L; m += nw) {          // wave-uniform
                const uint32_t so = so0 + (uint32_t)(16 * m) * ld2b;
                char* dst = dst0 + (size_t)(16 * m) * ROWB;
                if (16 * m + 16 <= L) {
#pragma unroll
                    for (int i = 0; i < IPB; ++i) att_lds_dma16(rs, dst + 1024 * i, vo[i], so);
                } else {                                        // the array's last, partial block
                    const int left = L - 16 * m;
#pragma unroll
                    for (int i = 0; i < IPB; ++i)
                        if (rr_[i] < left) att_lds_dma16(rs, dst + 1024 * i, vo[i], so);
                }
            }
        };
        const uint32_t ldA = (uint32_t)in.ldka2 * 2u, ldB = (uint32_t)in.ldkb2 * 2u;
        const uint32_t soA = (uint32_t)(b * La) * ldA, soB = (uint32_t)(b * Lb) * ldB;
        char* sm0 = (char*)smem_pl;
        auto first_of = [&](int k) { const int r = k % nw; return wave >= r ? wave - r : wave + nw - r; };
        stage_array(rsA, vK[0], La, soA + in.offKa, ldA, sm0, first_of(0));
        stage_array(rsB, vK[1], Lb, soB + in.offKb, ldB, sm0 + (size_t)(2 * La) * ROWB, first_of(nba));
        stage_array(rsA, vV[0], La, soA + in.offVa, ldA, sm0 + (size_t)La * ROWB, first_of(nba + nbb));
        stage_array(rsB, vV[1], Lb, soB + in.offVb, ldB, sm0 + (size_t)(2 * La + Lb) * ROWB, first_of(2 * nba + nbb));
    }

    // ================= verdicts and scales (block-uniform; the sites' maxima are complete: their producers ran before us) ==========
    float s_q = hq0, s_a = ha0, s_b = hb0;
    bool ok_q, ok_a, ok_b;
    {
        // common.h site_planes_ok, on votes instead of wave reductions (the maximum itself is only needed on the fallback path)
        auto okf = [](float s, uint32_t flag, f32x4 m4) {
            const float m = fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w));          // this lane's four slots
            const bool any_pos = __any(m > 0.f), any_big = __any(m * s >= 0.25f), any_over = __any(!(m * s < 65504.f));
            return s > 0.f && flag == 0u && (!any_pos || ((any_big || s >= 0x1p60f) && !any_over));
        };
        auto mx4 = [](f32x4 m4) { return wave_max(fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w))); };
        ok_q = okf(s_q, hq1, mq4); ok_a = okf(s_a, ha1, ma4); ok_b = okf(s_b, hb1, mb4);
        if (!ok_q) s_q = f16_scale_of(mx4(mq4));
        if (!ok_a) s_a = f16_scale_of(mx4(ma4));
        if (!ok_b) s_b = f16_scale_of(mx4(mb4));
        // one accumulator serves both key blocks of O = P V: P of a block is split with the scale SP_x = 2^14 min(1, s_y / s_x), so that
        // SP_a s_a = SP_b s_b; scales further apart than 2^10 (a block's P terms would sink): both blocks restaged at the smaller scale
        if (La > 0 && Lb > 0 && (s_a > 1024.f * s_b || s_b > 1024.f * s_a)) { ok_a = ok_b = false; s_a = s_b = fminf(s_a, s_b); }
        if (La == 0) { ok_a = true; s_a = s_b; }
        if (Lb == 0) { ok_b = true; s_b = s_a; }
    }
    if (!ok_q) {          // the query site's planes are unusable: split the fp32 rows here (exact site scale)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const f32x4 va = *(const f32x4*)(p.Qa + qrow * p.ldq + col0 + 16 * i + 4 * g);
            const f32x4 vb = *(const f32x4*)(p.Qb + qrow * p.ldq + col0 + 16 * i + 4 * g);
            qa[i] = split4c(va, s_q);
            qb[i] = split4c(vb, s_q);
        }
    }
    if (!(ok_a && ok_b)) {          // a key block's planes are unusable: its fp32 rows are split over what the DMA staged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();            // every wave's DMA writes have landed
        for (int it = threadIdx.x; it < 2 * T * HC; it += blockDim.x) {
            // item = (region, array, row, hi-chunk jj): 8 consecutive columns -> one hi chunk + one lo chunk
            const bool ib = it >= 2 * La * HC;
            if (ib ? ok_b : ok_a) continue;
            const int i2 = ib ? it - 2 * La * HC : it, L = ib ? Lb : La;
            const int arr = i2 >= L * HC ? 1 : 0, i3 = i2 - arr * L * HC, r = i3 / HC, jj = i3 - r * HC;
            const float* src = (ib ? (arr ? p.Vb : p.Kb) + (size_t)(b * Lb + r) * p.ldkb : (arr ? p.Va : p.Ka) + (size_t)(b * La + r) * p.ldka) + col0 + 8 * jj;
            const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
            const float s = ib ? s_b : s_a;
            const HL h0 = split4c(v0, s), h1 = split4c(v1, s);
            const int f = ib ? La + r : r;
            const int rot = arr ? 2 * ((f >> 2) & 1) : ((f >> 2) & 3);
            const int base = (ib ? 2 * La * CPR : 0) + arr * L * CPR + r * CPR;
            int ph = jj + rot; if (ph >= CPR) ph -= CPR;
            int pl = jj + HC + rot; if (pl >= CPR) pl -= CPR;
            *(uint4*)((char*)smem_pl + (size_t)(base + ph) * 16) = make_uint4(h0.h0, h0.h1, h1.h0, h1.h1);
            *(uint4*)((char*)smem_pl + (size_t)(base + pl) * 16) = make_uint4(h0.l0, h0.l1, h1.l0, h1.l1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SEGMM_ATT_PROBE
    if (p.pflags & 256) return;          // timing probe: staging only
#endif
    const bool q_ok = q_in && mq_byte != 0;

    // ---- per-lane read offsets inside a staged row (constant over the tiles: the rotations depend on key bits 2..3 only, and a
    // tile starts at a multiple of 16)
    uint32_t offKh[NCH], offKl[NCH], offVh[NCH], offVl[NCH];
    {
        const int rotK = (l15 >> 2) & 3;                 // this lane's K row: key 16 t + l15
        const int rotV = 2 * (g & 1);                    // this lane's V row: key 16 t + 4 g + (l15 >> 2)
        const int pq = l15 & 3;                          // transposed read: piece (columns 4 pq .. + 3) of the 16-column tile
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int jh = 2 * i + (g >> 1) + rotK; if (jh >= CPR) jh -= CPR;
            int jl = 2 * i + (g >> 1) + HC + rotK; if (jl >= CPR) jl -= CPR;
            offKh[i] = (uint32_t)jh * 16u + (uint32_t)(g & 1) * 8u;
            offKl[i] = (uint32_t)jl * 16u + (uint32_t)(g & 1) * 8u;
            int vh = 2 * i + (pq >> 1) + rotV; if (vh >= CPR) vh -= CPR;
            int vl = 2 * i + (pq >> 1) + HC + rotV; if (vl >= CPR) vl -= CPR;
            offVh[i] = (uint32_t)vh * 16u + (uint32_t)(pq & 1) * 8u;
            offVl[i] = (uint32_t)vl * 16u + (uint32_t)(pq & 1) * 8u;
        }
    }
    const char* sm = (const char*)smem_pl;
    const float inv_sa = 1.0f / (s_q * s_a), inv_sb = 1.0f / (s_q * s_b);

    // ---- S^T tiles: acc[t][r] = sum_c K[key 16 t + 4 g + r][c] Q[query][c] (in units of s_q s_k)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_tile(t)) {
            const int f = min(16 * t + l15, T - 1);
            const char* kr = sm + (size_t)(f < La ? f : La + f) * ROWB;
            HL kf[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const u32x2a hh = lds_b64(kr + offKh[i]), ll = lds_b64(kr + offKl[i]);
                kf[i] = HL{hh.x, hh.y, ll.x, ll.y};
            }
            const bool ta = 16 * t + 16 <= La, tb = 16 * t >= La;          // (compile-time when LA >= 0, else wave-uniform)
            if (ta || tb) {               // the whole tile in one block
#pragma unroll
                for (int i = 0; i < NCH; ++i) acc[t] = mfma_hl(kf[i], ta ? qa[i] : qb[i], acc[t]);
                acc[t] *= ta ? inv_sa : inv_sb;
            } else {                      // the straddling tile: both projections, merged per key quad
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NCH; ++i) { sa = mfma_hl(kf[i], qa[i], sa); sb = mfma_hl(kf[i], qb[i], sb); }
                acc[t] = (16 * t + 4 * g < La) ? sa * inv_sa : sb * inv_sb;
            }
        }
    }
    // ---- mask fill, dropout, scale; acc[t][r] is flat key 16 t + 4 g + r of query qi
    float mx = -INFINITY;
    const uint64_t drow = ((uint64_t)bh * p.Lq + (q_in ? qi : 0)) * Tp;          // the query's row in the dropout stream (padded key index)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (has_tile(t)) {
            const int f0 = 16 * t + 4 * g;
            const int jp0 = f0 < La ? f0 : La_p + (f0 - La);          // padded key index (the dropout stream's)
            f32x4 mult = {1.f, 1.f, 1.f, 1.f};
            if (drop_.p > 0.f) mult = drop_apply4(drop_, (drow + jp0) >> 2, f32x4{1.f, 1.f, 1.f, 1.f});
            const bool pad = (!EXACT || t == NT - 1) && f0 >= T;          // (only the last tile can reach behind the keys)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kv = ((kfl[t] >> (8 * r)) & 0xff) != 0;
                float v = logit_xform(acc[t][r], q_ok && kv, mult[r], p.scale);
                if (pad) v = -INFINITY;
                acc[t][r] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (has_tile(t)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fast_exp(acc[t][r] - mx);
                acc[t][r] = e;
                sum += e;
            }
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (g == 0 && q_in) {
        p.lse[(size_t)bh * p.Lq + qi] = mx;
        p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] = inv;
    }

    // ---- O^T[c][query] = sum_key V[key][c] e[key][query]: e in [0, 1] split with the block's P scale (SP_a s_a = SP_b s_b: one
    // accumulator), V^T fragments by transposed reads; normalised at the end
    const float SPa = 16384.f * fminf(1.f, s_b / s_a), SPb = 16384.f * fminf(1.f, s_a / s_b);
    f32x4 o[NCH];
#pragma unroll
    for (int ct = 0; ct < NCH; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (has_tile(t)) {
            const int f = min(16 * t + 4 * g + (l15 >> 2), T - 1);
            const char* vr = sm + (size_t)(f < La ? La + f : La + Lb + f) * ROWB;
            const HL Ph = split4c(acc[t], (16 * t + 4 * g < La) ? SPa : SPb);
#pragma unroll
            for (int ct = 0; ct < NCH; ++ct) {
                const u32x2a vh = lds_tr4(vr + offVh[ct]), vl = lds_tr4(vr + offVl[ct]);
                o[ct] = mfma_hl(HL{vh.x, vh.y, vl.x, vl.y}, Ph, o[ct]);
            }
        }
    }
    // ---- lane (query l15, g) register r of tile ct = head column 16 ct + 4 g + r
    const float nrm = inv / (SPa * s_a);
    float am = 0.f;
    const float ps = plane_scale(p.po_o);
    if (q_in) {
        float* orow = p.O + qrow * p.ldo + col0;
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct) {
            o[ct] *= nrm;
            *(f32x4*)(orow + 16 * ct + 4 * g) = o[ct];
            am = absmax4(am, o[ct]);
        }
    }
    if (ps > 0.f) {          // lane (query, g) and lane (query, g ^ 1) hold the two halves of an aligned 8 columns (col0 % 8 == 0)
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct) {
            const HL hl = split4(o[ct], ps);
            if (q_in) plane_store4_x16(p.po_o.p, p.po_o.ld2, (long long)qrow, col0 + 16 * ct + 4 * g, hl);
        }
    }
    plane_finish(p.po_o, p.amax_o, am, blockIdx.x * nw + wave, ps, blockIdx.x == 0 && threadIdx.x == 0);
}

}  // namespace segmm
