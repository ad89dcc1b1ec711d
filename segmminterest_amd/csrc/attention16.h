// Segment attention on the fp16 matrix cores with the GEMM engine's arithmetic: every operand is split x s = hi + lo (two
// fp16 terms, s a power of two taken from the maximum of the TILE the operand comes from) and a product is the three partial
// products hi hi + lo hi + hi lo accumulated in fp32 -- 22-bit operands, error 2^-22 of the tile maximum per element, the same
// as gemm_planes8.h.  Same kernels, lane maps, masks, dropout stream, outputs and determinism as attention.h (the exact-fp32
// v_mfma_f32_16x16x4_f32 kernels, which stay for head dims that are not a multiple of 16, as the C ABI's phases 0-3 and
// under SEGMM_ATTN=f32); what changes is the matrix-pipe time: tools/probe/mfma_small.hip measures 32 cycles for a
// 16x16x4 fp32 MFMA (k = 4) and 19 for both fp16 forms (k = 16 and k = 32), and a k = 16 block of a product is
//     ONE 16x16x32:  A slots [hi_a | lo_a] . B slots [hi_b | hi_b]   (= hi_a hi_b + lo_a hi_b; the reduction index of an MFMA
//                    is free to permute as long as both operands use the same map, so "k" 0..15 and 16..31 may be the same
//                    16 elements twice)
//   + ONE 16x16x16:  hi_a . lo_b
// = 38 cycles instead of 4 x 32 = 128.  A lane's 4 consecutive reduction elements of attention.h's fragments (one float4 per
// 16-float segment of a row; 4 consecutive keys / queries of a column) ARE the fp16 MFMAs' 4-element k groups, so the
// fragment maps carry over unchanged: each float4 becomes {h01, h23, l01, l23} (common.h splith_pair, 2 VALU per element).
//
// Who splits what, and with which scale (all powers of two from f16_scale_of: tile max * s in [2^14, 2^15)):
//   K, V tiles (16 keys x DH): by the wave that owns the tile, from the tile's own maximum (one wave reduction), once;
//   Q, dO of a query chunk (fused backward): staged in LDS as fp32 like before, the per-wave maxima meet at the first barrier,
//       then every thread converts ITS OWN 16-byte groups in place to [4 hi | 4 lo] -- the LDS image keeps its size and its
//       conflict-free row-fragment reads (ds_read_b128 = the A operand {hi | lo} of the k = 32 MFMA, ready made), and the
//       column fragments (4 consecutive queries of one head column) come out of the SAME image through ds_read_b64_tr_b16
//       (no second, transposed copy: LDS stays at 41 KB);
//   P (in [0, 1]): fixed scale 2^14;  dS: scale from the BOUND (DH max|dO| max|V_tile| + max|D|) drop_scale / sqrt(dh) >= |dS| --
//       a bound may be loose by many binades before anything is lost: an element x of a tensor scaled to 2^15 keeps
//       max(2^-22 |x|, 2^-40 * 2^15) absolute accuracy;
//   forward: Q tile, K / V tiles per wave as above.
// Powers of two commute with the split (hi and lo of x s are s times those of x while nothing leaves the normal range), so
// the forward's and the backward's recomputed logits agree to the last bits although their Q scales differ.
#pragma once
#include "attention.h"

namespace segmm {

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2a __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));

// one k = 16 block of a split product: acc += a.hi . b.hi + a.lo . b.hi + a.hi . b.lo -- three MFMAs of ONE shape.
// (The cheaper pairing -- a k = 32 MFMA with slots [hi_a | lo_a] . [hi_b | hi_b] followed by one k = 16 MFMA for hi_a . lo_b,
// 38 cycles instead of 57 -- chains two different MFMA shapes through the same accumulator, and hipcc 7.2 does not put the
// wait states between them that gfx950 needs: two of every four accumulator registers came out stale whenever fewer than
// about five instructions separated the pair, tools/probe/dbg_attn16.py.  The matrix pipe is not what bounds these kernels
// once the products are fp16, so the plain form stays.)
__device__ __forceinline__ f32x4 mfma_hl(const HL& a, const HL& b, f32x4 c) {
    const u32x2a ah = {a.h0, a.h1}, al = {a.l0, a.l1}, bh = {b.h0, b.h1}, bl = {b.l0, b.l1};
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, ah), __builtin_bit_cast(f16x4_t, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, al), __builtin_bit_cast(f16x4_t, bh), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, ah), __builtin_bit_cast(f16x4_t, bl), c, 0, 0, 0);
}
// x s = hi + lo in plain C++ (3 VALU per element: packed multiply, packed converts, packed fma).  NOT common.h's splith_pair
// (2 per element, inline asm): the compiler's hazard recogniser does not see VALU writes inside an asm statement, and here
// the split registers feed MFMAs directly -- with the asm form the MFMAs that follow a split read stale operands (measured:
// dQ wrong in two of every four columns, tools/probe/dbg_attn16.py).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_c(float x0, float x1, float s, uint32_t& ph, uint32_t& pl) {
    const _Float16 a = (_Float16)(x0 * s), b = (_Float16)(x1 * s);
    const _Float16 c = (_Float16)__builtin_fmaf(x0, s, -(float)a), d = (_Float16)__builtin_fmaf(x1, s, -(float)b);
    ph = __builtin_bit_cast(uint32_t, f16x2_t{a, b});
    pl = __builtin_bit_cast(uint32_t, f16x2_t{c, d});
}
__device__ __forceinline__ HL split4f(float x, float y, float z, float w, float s) {
    HL r;
    split_pair_c(x, y, s, r.h0, r.l0);
    split_pair_c(z, w, s, r.h1, r.l1);
    return r;
}
__device__ __forceinline__ HL split4c(f32x4 v, float s) { return split4f(v.x, v.y, v.z, v.w, s); }
__device__ __forceinline__ HL lds_hl(const char* a) {          // a 16-byte group [4 hi | 4 lo]
    const uint4 v = *(const uint4*)a;
    return HL{v.x, v.y, v.z, v.w};
}
typedef short s16x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2a lds_tr4(const char* a) {     // ds_read_b64_tr_b16: 4 rows of this lane's column
    const s16x4a v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4a*)a);
    return __builtin_bit_cast(u32x2a, v);
}
// max |x| over the DH/4 floats of a row fragment, all lanes of the wave
template <int N>
__device__ __forceinline__ float frag_absmax(const float (&f)[N]) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) m = fmaxf(m, fabsf(f[i]));
    return wave_max(m);
}

// ------------------------------------------------------------------------------------------ backward: fused dQ + dK + dV
#ifndef SEGMM_ATT16_WPS
#define SEGMM_ATT16_WPS 3          // waves per SIMD the four-wave form is compiled for (probe: 4 = 128 registers, spills)
#endif
template <int DH, int NW, bool ONE>
__global__ __launch_bounds__(64 * NW, NW <= 4 ? SEGMM_ATT16_WPS : 4) void attn_bwd_fused16_kernel(const AttnArgs p) {
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
    const int wg = xcd_remap(blockIdx.x, gridDim.x), bh = p.hpb == 2 ? wg >> 1 : wg, b = bh / p.H, h = bh % p.H;
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
    char* sQ = (char*)smem_f;                              // [QC][RSB]: fp32 rows while staging, then [4 hi | 4 lo] groups
    char* sdO = sQ + QC * RSB;
    float* sdQ = (float*)(sdO + QC * RSB);
    float* s_mx = sdQ + QC * RS;
    float* s_inv = s_mx + QC;
    float* s_D = s_inv + QC;
    float* s_tr = s_D + QC + wave * (16 * TS);                              // this wave's transpose scratch
    int* s_turn = (int*)(s_D + QC + nw * (16 * TS));                        // [4] whose turn it is to add dQ of query tile qt
    float* s_wm = (float*)(s_turn + 4);                                     // [3][12] per-wave maxima: |Q|, |dO|, |D|
    float* s_Dp = s_wm + 36;                                                // [QC][DH/4] partial products dO . O
    uint8_t* qm = (uint8_t*)(s_Dp + QC * (DH / 4));                         // [QC] 1 valid query, 0 masked, 2 pad
    uint8_t* km = qm + QC;                                                  // [Tp]
    // ---- this wave's key tile
    int jt = (isa ? 0 : nta) + wave;                                        // padded key tile of this wave (first pass)
    KeyBlocks<DH> kbk;
    kbk.init(p, b, col0, l15, g);
    HL kfh[NCH], vfh[NCH], kch[C::CT];                                      // split K / V row fragments, K column fragments
    float sK = 1.f, sV = 1.f;
    float maxV = 0.f;
    // K / V fragments of the current tile: the loads (issue_frags: 18 buffer loads in flight, raw values in kf_ / vf_ / kc_) and
    // the scale derivation + split (finish_frags) are separate, so that a single-chunk launch requests its first tile BEFORE the
    // query-side staging and splits it after -- one round of memory latency for both instead of two
    float kf_[C::KS], vf_[C::KS], kc_[4][C::CT];
    auto issue_frags = [&]() {
        if (isa) {
            const uint32_t so = (uint32_t)(16 * jt) * kbk.pitch_a;
            frag_load<DH>(kf_, kbk.ka, kbk.row_a, so);
            frag_load<DH>(vf_, kbk.va, kbk.row_a, so);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) col_load<DH>(kc_[s4], kbk.ka, kbk.col_a, (uint32_t)(16 * jt + s4) * kbk.pitch_a, l15);
        } else {
            const uint32_t so = (uint32_t)(16 * (jt - nta)) * kbk.pitch_b;
            frag_load<DH>(kf_, kbk.kb, kbk.row_b, so);
            frag_load<DH>(vf_, kbk.vb, kbk.row_b, so);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) col_load<DH>(kc_[s4], kbk.kb, kbk.col_b, (uint32_t)(16 * (jt - nta) + s4) * kbk.pitch_b, l15);
        }
    };
    auto finish_frags = [&]() {
        const float mK = frag_absmax(kf_);
        maxV = frag_absmax(vf_);
        sK = f16_scale_of(mK); sV = f16_scale_of(maxV);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            kfh[i] = split4f(kf_[4 * i], kf_[4 * i + 1], kf_[4 * i + 2], kf_[4 * i + 3], sK);
            vfh[i] = split4f(vf_[4 * i], vf_[4 * i + 1], vf_[4 * i + 2], vf_[4 * i + 3], sV);
        }
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) kch[ct] = split4f(kc_[0][ct], kc_[1][ct], kc_[2][ct], kc_[3][ct], sK);
    };
    auto load_frags = [&]() { issue_frags(); finish_frags(); };
    if (!ONE) load_frags();
    for (int j = threadIdx.x; j < Tp; j += nthr) {         // key flags
        uint8_t v;
        if (j < La_p) v = (j < p.La) ? (p.mka[(size_t)b * p.La + j] ? 1 : 0) : 2;
        else { const int jb = j - La_p; v = (jb < p.Lb) ? (p.mkb[(size_t)b * p.Lb + jb] ? 1 : 0) : 2; }
        km[j] = v;
    }
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
        float mq_ = 0.f, mdo_ = 0.f;
        if (ONE) issue_frags();                            // first tile of this wave: in flight under the staging
        // three items per thread and round: all nine loads are requested before the first LDS store (one round of latency)
        for (int i0 = threadIdx.x; i0 < QC * (DH / 4); i0 += 3 * nthr) {
            f32x4 va[3], vo[3], oo[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u * nthr;
                const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
                va[u] = f32x4{0.f, 0.f, 0.f, 0.f}; vo[u] = va[u]; oo[u] = va[u];
                if (i < QC * (DH / 4) && q < nq) {
                    const size_t row = (size_t)b * p.Lq + q0 + q;
                    va[u] = *(const f32x4*)(Qg + row * p.ldq + col0 + c);
                    vo[u] = *(const f32x4*)(p.dO + row * p.lddo + col0 + c);
                    oo[u] = *(const f32x4*)(p.O + row * p.ldo + col0 + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u * nthr;
                if (i < QC * (DH / 4)) {
                    const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
                    *(f32x4*)(sQ + q * RSB + c * 4) = va[u];
                    *(f32x4*)(sdO + q * RSB + c * 4) = vo[u];
                    *(f32x4*)(sdQ + q * RS + c) = f32x4{0.f, 0.f, 0.f, 0.f};
                    s_Dp[i] = (vo[u].x * oo[u].x + vo[u].y * oo[u].y) + (vo[u].z * oo[u].z + vo[u].w * oo[u].w);
                    mq_ = absmax4(mq_, va[u]);
                    mdo_ = absmax4(mdo_, vo[u]);
                }
            }
        }
        mq_ = wave_max(mq_); mdo_ = wave_max(mdo_);
        if (lane == 0) { s_wm[wave] = mq_; s_wm[12 + wave] = mdo_; }
        for (int q = threadIdx.x; q < QC; q += nthr) {
            const bool in = q < nq;
            s_mx[q] = in ? p.lse[(size_t)bh * p.Lq + q0 + q] : 0.f;
            s_inv[q] = in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + q0 + q] : 0.f;
            qm[q] = in ? (p.mq[(size_t)b * p.Lq + q0 + q] ? 1 : 0) : 2;
        }
        if (threadIdx.x < 4) s_turn[threadIdx.x] = 0;
        if (ONE) finish_frags();
        __syncthreads();
#ifdef SEGMM_ATT_PROBE
        if (p.pflags & 1024) return;          // timing probe: staging loads + K / V fragments only
#endif
        // chunk maxima -> scales; D; every thread converts its own groups in place
        float mQ = 0.f, mdO = 0.f;
        for (int w = 0; w < nwv; ++w) { mQ = fmaxf(mQ, s_wm[w]); mdO = fmaxf(mdO, s_wm[12 + w]); }
        const float sQs = f16_scale_of(mQ), sdOs = f16_scale_of(mdO);
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
            char* aq = sQ + q * RSB + c * 4;
            char* ad = sdO + q * RSB + c * 4;
            const f32x4 va = *(const f32x4*)aq, vo = *(const f32x4*)ad;
            const HL hq = split4c(va, sQs), hd = split4c(vo, sdOs);
            *(uint4*)aq = make_uint4(hq.h0, hq.h1, hq.l0, hq.l1);
            *(uint4*)ad = make_uint4(hd.h0, hd.h1, hd.l0, hd.l1);
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
        const uint8_t kflag = km[jp];
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
                {
                    float* row = sdQ + (16 * qt + l15) * RS + 4 * C::CT * g;
                    float t[4 * C::CT];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ct = 0; ct < C::CT; ++ct) t[C::CT * r + ct] = dqt[ct][r] * inv_dq;
#pragma unroll
                    for (int i = 0; i < C::CT; ++i) {
                        f32x4 a = *(f32x4*)(row + 4 * i);
                        a += f32x4{t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
                        *(f32x4*)(row + 4 * i) = a;
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
}

}  // namespace segmm
