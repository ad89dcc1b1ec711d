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
        const bool have_f32 = p.Qa != nullptr;
        if (have_f32 && La > 0 && Lb > 0 && (s_a > 1024.f * s_b || s_b > 1024.f * s_a)) { ok_a = ok_b = false; s_a = s_b = fminf(s_a, s_b); }
        // no fp32 views (the projection GEMMs write planes only): an unusable site was REPAIRED by its producer -- rewritten with
        // the exact scale of its maxima, the one derived above -- so the staged planes are the operands in every case
        if (!have_f32) ok_q = ok_a = ok_b = true;
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


// LDS bytes of attn_bwd_pl_kernel for a query chunk of QC rows, nw waves and Tp padded keys (keep in step with the kernel's layout)
template <int DH>
inline size_t attn_bwd_pl_lds_bytes(int QC, int nw, int /*Tp*/) {
    return ((size_t)3 * QC * (DH + 4) + 3 * QC + 4 + 36) * 4 + (size_t)nw * 2 * 16 * (DH * 2 + 8) + QC;
}

// ------------------------------------------------------------------------------------------ backward: fused dQ + dK + dV on input planes
// attn_bwd_fused16_kernel (attention16.h: same workgroup = (b, h, key block), same wave = key tile in passes, same ordered dQ
// accumulation, dropout stream, plane outputs and repair protocol) with Q, K and V read from the P32 planes of the projection
// GEMMs instead of their fp32 views -- which then need not exist (the GEMMs write planes only):
//   * a lane's 4 consecutive reduction elements of a row fragment are 8 contiguous bytes of hi terms and 8 of lo terms in the
//     planes: the K / V row fragments of a wave's key tile come STRAIGHT TO REGISTERS (12 eight-byte loads per lane) with the
//     site's scale -- no per-tile maxima, no operand splits (a third of the fp16x3 kernel's vector work);
//   * the K column fragments of the dQ^T = K^T dS^T product (4 consecutive keys of one head column) are read back transposed
//     (ds_read_b64_tr_b16) from a [key][column] image the wave writes into its own LDS from those registers -- no second,
//     fragment-shaped fetch of K (12 more loads per lane in the fp32 kernels);
//   * the chunk's Q rows are staged like before, but their two 8-byte pieces go into the [4 hi | 4 lo] image AS THEY ARE: no
//     maxima exchange, no in-place conversion pass for Q;
//   * operand scales are the sites' (a site that is unusable under its header's scale was rewritten by the producer's repair
//     launch with the exact scale of its maxima: both sides derive that scale from the same header).
// dO and O still arrive as fp32 (staged and converted like before).  (A first version staged Q and the K tile by LDS-DMA: an
// LDS-DMA tile on the critical path -- issue, land, ds_read -- was 24 us slower per launch than fragments straight to registers,
// profiles/r5/attention_planes_in.txt.)
template <int DH, int NW, bool ONE, bool WALK = false>
__global__ __launch_bounds__(64 * NW, NW <= 4 ? SEGMM_ATT16_WPS : 4) void attn_bwd_pl_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    const DropCfg drop_ = drop_live(p.drop);
    static_assert(DH % 16 == 0, "fp16 attention: head dim must be a multiple of 16");
    constexpr int RS = DH + 4;                 // LDS row stride (floats): 16-byte aligned rows, conflict-free row-fragment reads
    constexpr int RSB = RS * 4;                // ... in bytes
    constexpr int TS = 20;                     // row stride of the 16 x 16 transpose scratch
    constexpr int QC = ATT_FUSED_QCHUNK;       // queries staged at a time (3 query tiles)
    constexpr int MAXQT = QC / 16;
    constexpr int NCH = DH / 16;               // k = 16 blocks of a product over the head dim (= C::CT)
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // A launch has one workgroup per (b, h, key block) -- except the REPAIR launch (WALK), which the host makes 512 workgroups
    // wide: each walks its share of the heads if there is anything to repair, and normally all of them leave after two scalar
    // loads.  (A full-width launch of workgroups that leave at once still has to be PLACED with 44 KB of LDS each beside the
    // weight-gradient GEMM that holds 120 of a CU's 160 KB: 59 us on the main stream at config 2 for nothing.)  The stride is
    // even, so a workgroup keeps its key block, its wave count and its verdict over all its heads: the `return`s below end it
    // for good.  A separate instantiation: the loop around the body cost the one-head form 100 us (389 -> 495 us).
    const int n_wg = (p.hpb == 2 ? 2 : 1) * p.B * p.H;
    int wg_it = blockIdx.x;
  do {
    const int wg = WALK ? wg_it : xcd_remap(wg_it, n_wg), bh = p.hpb == 2 ? wg >> 1 : wg, b = bh / p.H, h = bh % p.H;
    const bool isa = p.hpb == 2 ? (wg & 1) == 0 : p.hpb == 0;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, ntb = Lb_p >> 4;
    const int ntk = isa ? nta : ntb;                       // key tiles of the block
    // A wave owns key tile `wave` -- and, in a single-chunk launch with fewer waves than tiles, tiles wave + nwv, wave + 2 nwv ...
    // one PASS after the other over the same staged query side.  The kernel is bound by the latency of its staging loads, i.e. by
    // the number of workgroups a CU holds (tools/attn_bench.py with SEGMM_ATT_LDS_PAD: 476 us at 4 + 2 workgroups per CU for the
    // two key blocks, 700 us at 2 + 1): four-wave workgroups fit four to a CU where the seven-wave ones of a 100-key block fit two.
    const int nwv = min(ntk, nw);
    if (wave >= nwv) return;                               // surplus wave, or empty block (CrossAtt / SelfAtt ablations)
    const int npass = ONE ? (ntk + nwv - 1) / nwv : 1;     // (several chunks: the host launches one wave per tile)
    const int nthr = 64 * nwv;                             // surviving threads
    const int col0 = h * DH;
    const float* Qg = isa ? p.Qa : p.Qb;
    float* dQg = isa ? p.dQa : p.dQb;
    _Float16* dQgp = isa ? p.dQap : p.dQbp;
    float s_q = (dQgp && p.sin_q) ? *p.sin_q : 0.f;
    const float* sin_k = isa ? p.sin_ka : p.sin_kb;
    float s_k = ((isa ? p.dKap : p.dKbp) && sin_k) ? *sin_k : 0.f;
    const bool repair = (p.pflags & ATT_REPAIR) != 0;
    const bool want_q = dQgp && p.sin_q, want_k = (isa ? p.dKap : p.dKbp) && sin_k;          // sites with plane outputs
    if (repair) {
        // segmm_site_fixup has judged the sites between the producers and this launch: hdr[2] != 0 = the planes were unusable
        // (written with no scale at all, overflow flag up, or the maximum below the fp16 window) and hdr[0] now holds the exact
        // scale of the recorded maxima, with which this pass rewrites them.  Two scalar loads and out, normally.
        const float* hk_ = isa ? p.hdr_ka : p.hdr_kb;
        const bool need_q = want_q && p.hdr_q[2] != 0.f, need_k = want_k && hk_[2] != 0.f;
        if (!need_q && !need_k) return;
        s_q = need_q ? p.hdr_q[0] : 0.f;
        s_k = need_k ? hk_[0] : 0.f;
    }
    const bool f32_q = !repair && !((p.pflags & ATT_PLANES_ONLY) && want_q);          // fp32 copies of dQ / of dK, dV
    const bool f32_k = !repair && !((p.pflags & ATT_PLANES_ONLY) && want_k);
    constexpr int KTP = DH * 2 + 8;                                        // row pitch (bytes) of a wave's K tile image: [16 keys][DH fp16] + pad
    char* sQ = (char*)smem_f;                              // [QC][RSB]: [4 hi | 4 lo] groups of the Q planes (written as staged)
    char* sdO = sQ + QC * RSB;                             // [QC][RSB]: fp32 rows while staging, then [4 hi | 4 lo] groups
    float* sdQ = (float*)(sdO + QC * RSB);                 // [QC][RS] dQ accumulators (zeroed once D is formed)
    float* s_Dp = sdQ;                                     // [QC][DH/4] partial products dO . O: alive from the staging to D only
    float* s_mx = sdQ + QC * RS;
    float* s_inv = s_mx + QC;
    float* s_D = s_inv + QC;
    char* sKt = (char*)(s_D + QC) + wave * (2 * 16 * KTP);                  // this wave's K tile, hi image then lo image (row fragments in,
    float* s_tr = (float*)sKt;                                              // transposed fragments out) ... and then its transpose scratch
    int* s_turn = (int*)((char*)(s_D + QC) + nw * (2 * 16 * KTP));          // [4] whose turn it is to add dQ of query tile qt
    float* s_wm = (float*)(s_turn + 4);                                     // [3][12] per-wave maxima: (unused), |dO|, |D|
    uint8_t* qm = (uint8_t*)(s_wm + 36);                                    // [QC] 1 valid query, 0 masked, 2 pad
    static_assert(16 * TS * 4 <= 2 * 16 * KTP, "transpose scratch inside the K image");
    // ---- input planes (the P32 planes of the projection GEMMs; see attn_fwd_pl_kernel): a lane's 4 consecutive reduction elements
    // of a row fragment ARE 8 contiguous bytes of the hi terms and 8 of the lo terms -- fragments come straight to registers with
    // the site's scale, no per-tile maxima, no splits
    const AttnInPlanes& in = p.in;
    const float* hdr_kx = isa ? in.hdr_ka : in.hdr_kb;
    float sQs = 1.f, sK = 1.f, sV = 1.f, maxV = 0.f;                         // (set from the site headers after the first staging round)
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(isa ? in.Qa : in.Qb, in.bytesQ);
    const __amdgpu_buffer_rsrc_t rsKV = make_rsrc(isa ? in.baseA : in.baseB, isa ? in.bytesA : in.bytesB);
    const int Lk = isa ? p.La : p.Lb;                                        // keys of this block
    const uint32_t ldq2b = (uint32_t)in.ldq2 * 2u, ldk2b = (uint32_t)(isa ? in.ldka2 : in.ldkb2) * 2u;
    const uint32_t soK0 = (uint32_t)(b * Lk) * ldk2b + (isa ? in.offKa : in.offKb), soV0 = (uint32_t)(b * Lk) * ldk2b + (isa ? in.offVa : in.offVb);
    uint32_t fco[NCH];                                                       // byte offset of this lane's fragment piece i inside a plane row
#pragma unroll
    for (int i = 0; i < NCH; ++i) fco[i] = p32_chunk_off(col0 + 16 * i + 4 * g);
    const uint8_t* mkx = isa ? p.mka + (size_t)b * p.La : p.mkb + (size_t)b * p.Lb;
    // ---- this wave's key tile
    int jt = (isa ? 0 : nta) + wave;                                        // padded key tile of this wave (first pass)
    HL kfh[NCH], vfh[NCH], kch[C::CT];                                      // K / V row fragments, K column fragments (hi, lo as staged)
    u32x2a kraw_[NCH][2], vraw_[NCH][2];
    uint8_t kraw = 0;
    // issue / finish are separate so that a single-chunk launch requests its first tile BEFORE the query-side staging
    auto issue_frags = [&]() {
        const int tl = jt - (isa ? 0 : nta);                                // tile inside the block
        const int kr = min(16 * tl + l15, Lk - 1);                          // (keys behind the block: clamped, finite; their P and dS are 0)
        const uint32_t ko = soK0 + (uint32_t)kr * ldk2b, vo = soV0 + (uint32_t)kr * ldk2b;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            kraw_[i][0] = __builtin_bit_cast(u32x2a, __builtin_amdgcn_raw_buffer_load_b64(rsKV, (int)(ko + fco[i]), 0, SEGMM_ATT_AUX));
            kraw_[i][1] = __builtin_bit_cast(u32x2a, __builtin_amdgcn_raw_buffer_load_b64(rsKV, (int)(ko + fco[i] + 64u), 0, SEGMM_ATT_AUX));
            vraw_[i][0] = __builtin_bit_cast(u32x2a, __builtin_amdgcn_raw_buffer_load_b64(rsKV, (int)(vo + fco[i]), 0, SEGMM_ATT_AUX));
            vraw_[i][1] = __builtin_bit_cast(u32x2a, __builtin_amdgcn_raw_buffer_load_b64(rsKV, (int)(vo + fco[i] + 64u), 0, SEGMM_ATT_AUX));
        }
        kraw = mkx[kr];                                                     // this lane's key flag (key 16 tl + l15 of the block)
    };
    auto finish_frags = [&]() {
        const bool live_key = 16 * (jt - (isa ? 0 : nta)) + l15 < Lk;       // (a clamped row must not enter dQ = dS K: its dS is 0, its K finite)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            kfh[i] = HL{kraw_[i][0].x, kraw_[i][0].y, kraw_[i][1].x, kraw_[i][1].y};
            vfh[i] = HL{vraw_[i][0].x, vraw_[i][0].y, vraw_[i][1].x, vraw_[i][1].y};
            // the tile as a [key][column] image in this wave's LDS: lane (key l15, g) owns columns 16 i + 4 g .. + 3
            *(u32x2a*)(sKt + l15 * KTP + 32 * i + 8 * g) = kraw_[i][0];
            *(u32x2a*)(sKt + 16 * KTP + l15 * KTP + 32 * i + 8 * g) = kraw_[i][1];
        }
        (void)live_key;
        __builtin_amdgcn_s_waitcnt(0xc07f);                                 // lgkmcnt(0): this wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // column fragments for dQ^T = K^T dS^T: lane 4 q + pq of a 16-lane group supplies key row 4 g + q, columns 16 ct + 4 pq .. + 3
        const char* kc = sKt + (4 * g + (l15 >> 2)) * KTP + 8 * (l15 & 3);
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const u32x2a hh = lds_tr4(kc + 32 * ct), ll = lds_tr4(kc + 16 * KTP + 32 * ct);
            kch[ct] = HL{hh.x, hh.y, ll.x, ll.y};
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                                 // the fragments are in registers before the image becomes the scratch
        __builtin_amdgcn_wave_barrier();
    };
    auto load_frags = [&]() { issue_frags(); finish_frags(); };
    if (!ONE) load_frags();
    const float fscale = p.scale;
    f32x4 dk[C::CT], dv[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) { dk[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float am_q = 0.f, am_k = 0.f;
    float unit_dk = 1.f, unit_dv = 1.f;                    // product of the operand scales the dK / dV accumulators are in
    constexpr float SP = 16384.f;                          // scale of P (<= 1)
    // transposed-read address of this lane inside a (query tile, column tile) block: row 4 g + (l15 >> 2), group (l15 & 3)
    const uint32_t tr_lane = (uint32_t)(4 * g + (l15 >> 2)) * RSB + (uint32_t)(l15 & 3) * 16u;

    // dK / dV rows of the current tile: lane (key l15, g), tile ct register r = head column 16 ct + 4 g + r
    auto emit_dkdv = [&]() {
        const int jp = 16 * jt + l15;
        const float inv_dk = 1.0f / unit_dk, inv_dv = 1.0f / unit_dv;
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) { dk[ct] *= inv_dk; dv[ct] *= inv_dv; }
        const bool ka = jp < La_p;
        const int jloc = ka ? jp : jp - La_p;
        const bool real = ka ? (jloc < p.La) : (jloc < p.Lb);
        if (real) {
            float* dKp = (ka ? p.dKa + (size_t)(b * p.La + jloc) * p.lddka : p.dKb + (size_t)(b * p.Lb + jloc) * p.lddkb) + col0;
            float* dVp = (ka ? p.dVa + (size_t)(b * p.La + jloc) * p.lddka : p.dVb + (size_t)(b * p.Lb + jloc) * p.lddkb) + col0;
            const long long krow = ka ? (long long)b * p.La + jloc : (long long)b * p.Lb + jloc;
            _Float16* dKpp = ka ? p.dKap : p.dKbp;
            _Float16* dVpp = ka ? p.dVap : p.dVbp;
            const int ldk2 = ka ? p.lddka2 : p.lddkb2;
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) {
                if (f32_k) {
                    *(f32x4*)(dKp + 16 * ct + 4 * g) = dk[ct];
                    *(f32x4*)(dVp + 16 * ct + 4 * g) = dv[ct];
                }
                if (s_k > 0.f) {          // lane (key, g) and lane (key, g ^ 1) hold the two halves of an aligned 8
                    if ((col0 & 7) == 0) {
                        plane_store4_x16(dKpp, ldk2, krow, col0 + 16 * ct + 4 * g, split4(dk[ct], s_k));
                        plane_store4_x16(dVpp, ldk2, krow, col0 + 16 * ct + 4 * g, split4(dv[ct], s_k));
                    } else {
                        plane_store4(dKpp, ldk2, krow, col0 + 16 * ct + 4 * g, dk[ct], s_k);
                        plane_store4(dVpp, ldk2, krow, col0 + 16 * ct + 4 * g, dv[ct], s_k);
                    }
                }
                am_k = absmax4(absmax4(am_k, dk[ct]), dv[ct]);
            }
        }
    };

    for (int q0 = 0; ONE ? q0 < 1 : q0 < p.Lq; q0 += QC) {
        const int nq = min(QC, p.Lq - q0);                 // real queries of the chunk
        const int nqt = (nq + 15) >> 4;
        float mdo_ = 0.f;
        // ===== ONE round of memory latency: every load of the chunk is requested before the first result is used =====
        if (ONE) issue_frags();                            // first tile of this wave: K / V fragments, key flag
        // softmax statistics and query flags (one query per thread; QC <= 64 <= nthr)
        float r_mx = 0.f, r_inv = 0.f;
        uint8_t r_qm = 2;
        if ((int)threadIdx.x < nq) {
            r_mx = p.lse[(size_t)bh * p.Lq + q0 + threadIdx.x];
            r_inv = p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + q0 + threadIdx.x];
            r_qm = p.mq[(size_t)b * p.Lq + q0 + threadIdx.x] ? 1 : 0;
        }
        // the site headers of the input planes (first chunk; judged below): scale, flag, four partial maxima per lane
        float hq0 = 0.f, hk0 = 0.f, hq1 = 0.f, hk1 = 0.f;
        f32x4 hq4 = {0.f, 0.f, 0.f, 0.f}, hk4 = hq4;
        if (q0 == 0) {
            hq0 = in.hdr_q[0]; hq1 = in.hdr_q[1]; hk0 = hdr_kx[0]; hk1 = hdr_kx[1];
            hq4 = *(const f32x4*)(in.hdr_q + SITE_HDR + lane * 4); hk4 = *(const f32x4*)(hdr_kx + SITE_HDR + lane * 4);
        }
        // Q (hi and lo terms of 4 columns: two 8-byte pieces of the planes -> one [4 hi | 4 lo] group of the image, as is), dO and O:
        // three items per thread and round, all twelve loads requested before the first LDS store
        for (int i0 = threadIdx.x; i0 < QC * (DH / 4); i0 += 3 * nthr) {
            f32x4 vo[3], oo[3];
            u32x2a qh[3], ql[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u * nthr;
                const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
                vo[u] = f32x4{0.f, 0.f, 0.f, 0.f}; oo[u] = vo[u];
                qh[u] = u32x2a{0u, 0u}; ql[u] = qh[u];
                if (i < QC * (DH / 4) && q < nq) {
                    const size_t row = (size_t)b * p.Lq + q0 + q;
                    const uint32_t qo = (uint32_t)row * ldq2b + p32_chunk_off(col0 + c);
                    qh[u] = __builtin_bit_cast(u32x2a, __builtin_amdgcn_raw_buffer_load_b64(rsQ, (int)qo, 0, SEGMM_ATT_AUX));
                    ql[u] = __builtin_bit_cast(u32x2a, __builtin_amdgcn_raw_buffer_load_b64(rsQ, (int)(qo + 64u), 0, SEGMM_ATT_AUX));
                    vo[u] = *(const f32x4*)(p.dO + row * p.lddo + col0 + c);
                    oo[u] = *(const f32x4*)(p.O + row * p.ldo + col0 + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u * nthr;
                if (i < QC * (DH / 4)) {
                    const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
                    *(uint4*)(sQ + q * RSB + c * 4) = make_uint4(qh[u].x, qh[u].y, ql[u].x, ql[u].y);
                    *(f32x4*)(sdO + q * RSB + c * 4) = vo[u];
                    s_Dp[i] = (vo[u].x * oo[u].x + vo[u].y * oo[u].y) + (vo[u].z * oo[u].z + vo[u].w * oo[u].w);
                    mdo_ = absmax4(mdo_, vo[u]);
                }
            }
        }
        mdo_ = wave_max(mdo_);
        if (lane == 0) s_wm[12 + wave] = mdo_;
        if (threadIdx.x < QC) { s_mx[threadIdx.x] = r_mx; s_inv[threadIdx.x] = r_inv; qm[threadIdx.x] = r_qm; }
        if (threadIdx.x < 4) s_turn[threadIdx.x] = 0;
        if (q0 == 0) {
            // the input sites: a site that is not usable under its header's scale was REPAIRED by its producer (the projection
            // GEMM's repair launch rewrote the planes with the exact scale of the recorded maxima): take that scale
            const float amax_q = wave_max(fmaxf(fmaxf(hq4.x, hq4.y), fmaxf(hq4.z, hq4.w)));
            const float amax_k = wave_max(fmaxf(fmaxf(hk4.x, hk4.y), fmaxf(hk4.z, hk4.w)));
            auto okf = [](float s, uint32_t flag, float m) { return s > 0.f && flag == 0u && (!(m > 0.f) || ((m * s >= 0.25f || s >= 0x1p60f) && m * s < 65504.f)); };
            sQs = okf(hq0, __float_as_uint(hq1), amax_q) ? hq0 : f16_scale_of(amax_q);
            sK = okf(hk0, __float_as_uint(hk1), amax_k) ? hk0 : f16_scale_of(amax_k);
            sV = sK;                                       // K and V of a block are columns of one buffer
            maxV = amax_k;                                 // bound of |V| (the site's maximum)
        }
        if (ONE) finish_frags();
        __syncthreads();
#ifdef SEGMM_ATT_PROBE
        if (p.pflags & 1024) return;          // timing probe: staging loads + K / V fragments only
#endif
        // chunk maxima -> scales; D; every thread converts its own groups in place
        float mdO = 0.f;
        for (int w = 0; w < nwv; ++w) mdO = fmaxf(mdO, s_wm[12 + w]);
        const float sdOs = f16_scale_of(mdO);
        float mD = 0.f;
        for (int q = threadIdx.x; q < QC; q += nthr) {     // D[q]: the DH/4 partials of the row in index order (deterministic)
            float d_ = 0.f;
#pragma unroll
            for (int j = 0; j < DH / 4; ++j) d_ += s_Dp[q * (DH / 4) + j];
            s_D[q] = d_;
            mD = fmaxf(mD, fabsf(d_));
        }
        mD = wave_max(mD);
        if (lane == 0) s_wm[24 + wave] = mD;
        for (int i = threadIdx.x; i < QC * (DH / 4); i += nthr) {
            const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
            char* ad = sdO + q * RSB + c * 4;
            const f32x4 vo = *(const f32x4*)ad;
            const HL hd = split4c(vo, sdOs);
            *(uint4*)ad = make_uint4(hd.h0, hd.h1, hd.l0, hd.l1);
        }
        __syncthreads();                                   // D is formed: the partial products are dead, their space becomes the dQ accumulators
        for (int i = threadIdx.x; i < QC * (DH / 4); i += nthr) {
            const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
            *(f32x4*)(sdQ + q * RS + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
#ifdef SEGMM_ATT_PROBE
        if (p.pflags & 2048) return;          // timing probe: ... + the in-place conversion
#endif
        float mDc = 0.f;
        for (int w = 0; w < nwv; ++w) mDc = fmaxf(mDc, s_wm[24 + w]);
      for (int pass = 0; pass < npass; ++pass) {
        const int tile = wave + pass * nwv;                // this wave's tile of the pass, 0 .. ntk-1 (the dQ turn order)
        if (tile >= ntk) break;
        if (pass > 0) {                                    // next tile of this wave: its K / V fragments, fresh dK / dV sums
            jt += nwv;
            load_frags();
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) { dk[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        const int jp = 16 * jt + l15;                      // this lane's key (padded index)
        // |dS| <= P (|dP| + |D|) mult scale, |dP| <= DH max|dO| max|V tile|
        const float sdS = f16_scale_of(((float)DH * mdO * maxV + mDc) * drop_.scale * fscale);
        const float inv_s = 1.0f / (sQs * sK), inv_dp = 1.0f / (sdOs * sV), inv_dq = 1.0f / (sK * sdS);
        const uint8_t kflag = (16 * (jt - (isa ? 0 : nta)) + l15 < Lk) ? (kraw ? 1 : 0) : 2;          // 1 valid key, 0 masked, 2 pad
        {   // dK / dV accumulate in the units of the CURRENT chunk's operand scales: moving on to a chunk with other scales
            // multiplies what has been summed so far by the ratio -- a power of two, exact
            const float u_dk = sQs * sdS, u_dv = sdOs * SP;
            if (!ONE && q0 > 0) {          // (one pass per chunk here)
                const float rk = u_dk / unit_dk, rv = u_dv / unit_dv;
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) { dk[ct] *= rk; dv[ct] *= rv; }
            }
            unit_dk = u_dk; unit_dv = u_dv;
        }
#pragma unroll
        for (int qt = 0; qt < MAXQT; ++qt) {
#ifdef SEGMM_ATT_PROBE
            if (p.pflags & 4096) break;          // timing probe: no pair loop
#endif
            if (qt < nqt) {
                // row fragments (lane&15 = query): group 4 i + g of the row = elements 16 i + 4 g .. + 3, [hi | lo]
                f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                const char* rq = sQ + (16 * qt + l15) * RSB + g * 16;
                const char* rd = sdO + (16 * qt + l15) * RSB + g * 16;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    sv = mfma_hl(lds_hl(rq + 64 * i), kfh[i], sv);
                    dp = mfma_hl(lds_hl(rd + 64 * i), vfh[i], dp);
                }
                const f32x4 mxq = *(const f32x4*)(s_mx + 16 * qt + 4 * g), invq = *(const f32x4*)(s_inv + 16 * qt + 4 * g);
                const f32x4 Dq = *(const f32x4*)(s_D + 16 * qt + 4 * g);
                const uint32_t qfl = *(const uint32_t*)(qm + 16 * qt + 4 * g);
                f32x4 Pv, dSv;
                uint32_t dw[4] = {0u, 0u, 0u, 0u};
                if (drop_.p > 0.f) {
                    const int rr = l15 & 3;
                    const uint2 hw = drop_rand_quad(drop_, (((uint64_t)bh * p.Lq + (q0 + 16 * qt + 4 * g + rr)) * Tp + jp) >> 2);
                    const uint32_t a0 = quad_bcast<0>(hw.x), a1 = quad_bcast<1>(hw.x), a2 = quad_bcast<2>(hw.x), a3 = quad_bcast<3>(hw.x);
                    const uint32_t b0 = quad_bcast<0>(hw.y), b1 = quad_bcast<1>(hw.y), b2 = quad_bcast<2>(hw.y), b3 = quad_bcast<3>(hw.y);
                    const bool lo_word = rr < 2, hi_half = rr & 1;
                    const uint32_t w0 = lo_word ? a0 : b0, w1 = lo_word ? a1 : b1, w2 = lo_word ? a2 : b2, w3 = lo_word ? a3 : b3;
                    dw[0] = hi_half ? (w0 >> 16) : (w0 & 0xffffu); dw[1] = hi_half ? (w1 >> 16) : (w1 & 0xffffu);
                    dw[2] = hi_half ? (w2 >> 16) : (w2 & 0xffffu); dw[3] = hi_half ? (w3 >> 16) : (w3 & 0xffffu);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t qf_ = (qfl >> (8 * r)) & 0xff;
                    const bool valid = (qf_ == 1) && (kflag == 1);
                    float mult = 1.f;
                    if (drop_.p > 0.f && qf_ != 2) mult = (dw[r] >= drop_.thresh) ? drop_.scale : 0.f;
                    const float v = logit_xform(sv[r] * inv_s, valid, mult, fscale);
                    const float pr = (kflag == 2 || qf_ == 2) ? 0.f : fast_exp(v - mxq[r]) * invq[r];
                    Pv[r] = pr;
                    dSv[r] = valid ? pr * (dp[r] * inv_dp - Dq[r]) * mult * fscale : 0.f;
                }
                const HL Ph = split4c(Pv, SP), dSh = split4c(dSv, sdS);
                // column fragments (4 consecutive queries 16 qt + 4 g .. + 3 of head column 16 ct + l15) by transposed reads
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) {
                    const uint32_t o = (uint32_t)(16 * qt) * RSB + (uint32_t)ct * 64u + tr_lane;
                    const u32x2a dh_ = lds_tr4(sdO + o), dl_ = lds_tr4(sdO + o + 8);
                    const u32x2a qh_ = lds_tr4(sQ + o), ql_ = lds_tr4(sQ + o + 8);
                    dv[ct] = mfma_hl(HL{dh_.x, dh_.y, dl_.x, dl_.y}, Ph, dv[ct]);
                    dk[ct] = mfma_hl(HL{qh_.x, qh_.y, ql_.x, ql_.y}, dSh, dk[ct]);
                }
                // dS[query 4g+r][key l15] -> dS^T fragments (lane&15 = query, registers = keys 4g..4g+3) through the scratch
#pragma unroll
                for (int r = 0; r < 4; ++r) s_tr[(4 * g + r) * TS + l15] = dSv[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's own LDS writes have landed
                __builtin_amdgcn_wave_barrier();
                const f32x4 dST = *(const f32x4*)(s_tr + l15 * TS + 4 * g);
                __builtin_amdgcn_wave_barrier();
                const HL dSTh = split4c(dST, sdS);
                f32x4 dqt[C::CT];
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) dqt[ct] = mfma_hl(kch[ct], dSTh, f32x4{0.f, 0.f, 0.f, 0.f});
                if (tile > 0)
                    while (__hip_atomic_load(s_turn + qt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != tile) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                {   // lane (query l15, g), tile ct, register r = head column 16 ct + 4 g + r (the K column fragments are in natural order)
                    float* row = sdQ + (16 * qt + l15) * RS + 4 * g;
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) {
                        f32x4 a = *(f32x4*)(row + 16 * ct);
                        a += dqt[ct] * inv_dq;
                        *(f32x4*)(row + 16 * ct) = a;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(s_turn + qt, tile + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (ONE) emit_dkdv();                              // this tile is complete
      }
        __syncthreads();                                   // every wave has added its dQ partials of this chunk
        for (int i = threadIdx.x; i < nq * (DH / 4); i += nthr) {
            const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
            const size_t row = (size_t)b * p.Lq + q0 + q;
            const f32x4 v = *(const f32x4*)(sdQ + q * RS + c);
            if (f32_q) *(f32x4*)(dQg + row * p.lddq + col0 + c) = v;
            if (s_q > 0.f) {
                if ((col0 & 7) == 0) plane_store4_pair(dQgp, p.lddq2, (long long)row, col0 + c, v, s_q);
                else plane_store4(dQgp, p.lddq2, (long long)row, col0 + c, v, s_q);
            }
            am_q = absmax4(am_q, v);
        }
        if (!ONE && q0 + QC < p.Lq) __syncthreads();       // the next chunk's staging overwrites what was just read
    }
    {
        if (!ONE) emit_dkdv();
        const float am = am_k;
        float* hk = isa ? p.hdr_ka : p.hdr_kb;
        float* slot = isa ? p.amax_ka : p.amax_kb;
        const bool hdr_writer = bh == 0 && wave == 0 && lane == 0;
        if (!repair) {
            if (s_k > 0.f) { site_commit(hk, am, blockIdx.x * nw + wave, s_k); if (hdr_writer) hk[0] = s_k; }
            else if (slot) amax_commit(slot, am, blockIdx.x * nw + wave);
            if (s_q > 0.f) { site_commit(p.hdr_q, am_q, blockIdx.x * nw + wave, s_q); if (hdr_writer) p.hdr_q[0] = s_q; }
            else if (p.amax_q) amax_commit(p.amax_q, am_q, blockIdx.x * nw + wave);
        }
    }
    if constexpr (!WALK) break;
    wg_it += gridDim.x;
    if (wg_it >= n_wg) break;
    __syncthreads();          // the next head's staging overwrites what was just read
  } while (true);
}

}  // namespace segmm
