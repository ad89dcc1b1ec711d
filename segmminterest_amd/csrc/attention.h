// Joint self+cross segment attention on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Reference semantics (MMinterest/models/encoder.py:44-73,138-161): for one side (video queries or
// user queries) the keys are the concatenation of two blocks, a (video tokens) and b (user tokens),
// and each block has its OWN query projection:
//     logits = [ Qa.Ka^T | Qb.Kb^T ]             raw, unscaled
//     logits[~(mask_q (x) mask_k)] = -10000      finite fill: a padded query row attends uniformly
//     logits = dropout(logits) / sqrt(dh)        dropout BEFORE the scale, fills included
//     out    = softmax(logits) . [Va ; Vb]
//
// One workgroup (always 4 waves) = one (batch row, head) x up to 64 queries; each wave owns 16 queries
// and all keys.  The product is computed TRANSPOSED (S^T = K.Q^T) so the MFMA result has the query on
// the lane (lane&15) and 4 consecutive keys in the 4 result registers: the softmax row-reduce is
// register-local plus two wave shuffles, and P^T feeds the second product (O^T = V^T.P^T) straight from
// registers -- the reduction index of an f32 MFMA operand is free to permute, so
// "key = 4*lanegroup + step" needs no cross-lane move.  The backward is two kernels, both recomputing S
// from Q, K and the saved softmax row statistics: a query-major one (dQ, writes D = rowsum(P*dP)) and a
// key-major one (dK, dV) whose reductions over queries stay inside one wave => no atomics, bitwise
// reproducible.
//
// Staging: these kernels are latency-bound (a head is only 40 x 140 x 48), so every global->LDS phase
// issues ALL of its loads into registers first (one memory round trip per phase), and the next phase's
// operand (V after K, K again after V) is prefetched into registers while the current MFMAs run.
//
// Key blocks are padded separately to multiples of 16 (pad keys get probability 0); the dropout
// stream is indexed by (b, h, query, padded key) so 4 consecutive keys share one hash call.
#pragma once
#include "common.h"

namespace segmm {

struct AttnArgs {
    int B, H, Lq, La, Lb;
    const float *Qa, *Qb; int ldq;      // [B*Lq, ldq], head h at column h*DH
    const float *Ka, *Va; int ldka;     // [B*La, ldka]
    const float *Kb, *Vb; int ldkb;     // [B*Lb, ldkb]
    const uint8_t *mq, *mka, *mkb;      // [B,Lq] [B,La] [B,Lb]; nonzero = valid token
    float* O; int ldo;                  // [B*Lq, ldo]
    float* lse;                         // [2,B,H,Lq] softmax row statistics: plane 0 = row max, plane 1 = 1/sum.
                                        // Kept as the PAIR (not max+log(sum)): a padded query row has every
                                        // logit at -10000*scale ~ -1e3, where one fp32 ulp of a merged
                                        // log-sum-exp is ~1e-4 and would put a 1e-4 relative error on P.
    float scale;
    DropCfg drop;
    // backward
    const float* dO; int lddo;
    float* Dvec;                        // [B,H,Lq]  rowsum(P * dP)
    float *dQa, *dQb; int lddq;
    float *dKa, *dVa; int lddka;
    float *dKb, *dVb; int lddkb;
    // optional partial maxima (AMAX_SLOTS each, common.h) of what the kernels write, for the fp16x3 GEMM engine:
    float* amax_o;                      // forward: |O|
    float *amax_q, *amax_ka, *amax_kb;  // backward: |dQa|,|dQb| ; |dKa|,|dVa| ; |dKb|,|dVb|
};

constexpr int ATT_THREADS = 256;        // 4 waves, fixed: the register-batched staging sizes depend on it
constexpr int ATT_QB = 64;              // queries (fwd, dQ) / keys (dK,dV) per workgroup

__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }

template <int DH> struct AttnCfg {
    static constexpr int KS = DH / 4;                       // k-steps over the head dim
    static constexpr int CT = (DH + 15) / 16;               // 16-wide column tiles of the head dim
    static constexpr int LDR = DH + 2;                      // "row on lane&15" reads: stride = 2 mod 4
    static constexpr int LDC = DH + ((DH % 8 == 0) ? 4 : 0);  // "column on lane&15" reads: stride = 4 mod 8
    static constexpr int LDMAX = LDR > LDC ? LDR : LDC;
    static constexpr int UQ = (ATT_QB * KS + ATT_THREADS - 1) / ATT_THREADS;    // float4 per thread for 64 rows
};

// ---- register-batched staging: U float4 per thread cover nrows x DH floats; all loads first, stores later
template <int DH, int U>
__device__ __forceinline__ void rows_load(f32x4 (&v)[U], const float* src, size_t ld, int col0, int row_base, int nrows,
                                          int nvalid, int tid) {
    constexpr int KS = DH / 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = tid + u * ATT_THREADS;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < nrows * KS) {
            const int r = idx / KS, c4 = (idx % KS) * 4;
            if (r < nvalid) v[u] = *(const f32x4*)(src + (size_t)(row_base + r) * ld + col0 + c4);
        }
    }
}
// padded key index space: rows [j0, j0+nrows): block a occupies [0, La_p), block b [La_p, La_p+Lb_p)
template <int DH, int U>
__device__ __forceinline__ void keys_load(f32x4 (&v)[U], const float* A, int lda, const float* Bm, int ldb, int b, int La,
                                          int Lb, int La_p, int j0, int nrows, int col0, int tid) {
    constexpr int KS = DH / 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = tid + u * ATT_THREADS;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < nrows * KS) {
            const int j = j0 + idx / KS, c4 = (idx % KS) * 4;
            if (j < La_p) {
                if (j < La) v[u] = *(const f32x4*)(A + (size_t)(b * La + j) * lda + col0 + c4);
            } else {
                const int jb = j - La_p;
                if (jb < Lb) v[u] = *(const f32x4*)(Bm + (size_t)(b * Lb + jb) * ldb + col0 + c4);
            }
        }
    }
}
template <int DH, int U>
__device__ __forceinline__ void rows_store(const f32x4 (&v)[U], float* dst, int lds, int nrows, int tid) {
    constexpr int KS = DH / 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = tid + u * ATT_THREADS;
        if (idx < nrows * KS) {
            const int r = idx / KS, c4 = (idx % KS) * 4;
            float2* d = (float2*)(dst + r * lds + c4);      // row strides are even => 8-byte aligned
            d[0] = make_float2(v[u].x, v[u].y);
            d[1] = make_float2(v[u].z, v[u].w);
        }
    }
}

// kmask[jp] : 1 valid, 0 masked token (-10000 fill), 2 alignment pad (probability 0)
__device__ __forceinline__ void stage_kmask(uint8_t* km, const uint8_t* mka, const uint8_t* mkb, int b, int La, int Lb,
                                            int La_p, int Lb_p, int tid) {
    for (int j = tid; j < La_p + Lb_p; j += ATT_THREADS) {
        uint8_t v;
        if (j < La_p) v = (j < La) ? (mka[(size_t)b * La + j] ? 1 : 0) : 2;
        else { const int jb = j - La_p; v = (jb < Lb) ? (mkb[(size_t)b * Lb + jb] ? 1 : 0) : 2; }
        km[j] = v;
    }
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// exp via v_exp_f32 (2 instructions instead of ~25 for expf): arguments are <= 0 here (x - rowmax), the
// relative error is ~1e-6 at |x| < 20 and anything below -87 flushes to 0 exactly like expf.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// scaled/masked/dropped logit
__device__ __forceinline__ float logit_xform(float s, bool valid, float mult, float scale) {
    return (valid ? s : -10000.0f) * mult * scale;
}

// S^T tiles of one wave: acc[t][r] = sum_c K[16t + 4g + r][c] Q[query][c]  (query = lane&15).
// Two key tiles are interleaved so that consecutive MFMAs hit different accumulators (the 16x16x4 f32
// MFMA has a 40-cycle dependent latency against a 32-cycle issue).
template <int DH, int NT>
__device__ __forceinline__ void qk_tiles(f32x4 (&acc)[NT], const float* Qsa, const float* Qsb, const float* Ks, int nt,
                                         int nta, int wave, int l15, int g) {
    using C = AttnCfg<DH>;
    float qa[C::KS], qb[C::KS];
#pragma unroll
    for (int c = 0; c < C::KS; ++c) {
        qa[c] = Qsa[(wave * 16 + l15) * C::LDR + 4 * c + g];
        qb[c] = Qsb[(wave * 16 + l15) * C::LDR + 4 * c + g];
    }
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        if (t < nt) {
            const bool two = (t + 1 < nt) && (t + 1 < NT);
            const bool isa0 = t < nta, isa1 = (t + 1) < nta;
#pragma unroll
            for (int c = 0; c < C::KS; ++c) {
                const float a0 = Ks[(16 * t + l15) * C::LDR + 4 * c + g];
                acc[t] = MFMA16(a0, isa0 ? qa[c] : qb[c], acc[t]);
                if (t + 1 < NT) {
                    if (two) {
                        const float a1 = Ks[(16 * (t + 1) + l15) * C::LDR + 4 * c + g];
                        acc[t + 1] = MFMA16(a1, isa1 ? qa[c] : qb[c], acc[t + 1]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ forward
template <int DH, int NT>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    constexpr int UK = (NT * 16 * C::KS + ATT_THREADS - 1) / ATT_THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-aware: the heads of one batch row read interleaved 4*DH-byte slices of the same token rows, so
    // consecutive (b,h) must share an L2 or every shared cache line is fetched from HBM once per XCD
    const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / p.H, h = bh % p.H;
    const int q_blk = blockIdx.y * ATT_QB;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    float* Qsa = smem;
    float* Qsb = Qsa + ATT_QB * C::LDR;
    float* KVs = Qsb + ATT_QB * C::LDR;
    uint8_t* km = (uint8_t*)(KVs + Tp * C::LDMAX);
    const int col0 = h * DH;
    const int nq_valid = max(0, min(ATT_QB, p.Lq - q_blk));

    f32x4 rqa[C::UQ], rqb[C::UQ], rk[UK];
    rows_load<DH, C::UQ>(rqa, p.Qa, p.ldq, col0, b * p.Lq + q_blk, ATT_QB, nq_valid, tid);
    rows_load<DH, C::UQ>(rqb, p.Qb, p.ldq, col0, b * p.Lq + q_blk, ATT_QB, nq_valid, tid);
    keys_load<DH, UK>(rk, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, 0, Tp, col0, tid);
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p, tid);
    rows_store<DH, C::UQ>(rqa, Qsa, C::LDR, ATT_QB, tid);
    rows_store<DH, C::UQ>(rqb, Qsb, C::LDR, ATT_QB, tid);
    rows_store<DH, UK>(rk, KVs, C::LDR, Tp, tid);
    __syncthreads();
    keys_load<DH, UK>(rk, p.Va, p.ldka, p.Vb, p.ldkb, b, p.La, p.Lb, La_p, 0, Tp, col0, tid);     // V prefetch, lands under QK^T + softmax

    const int qi = q_blk + wave * 16 + l15;          // this lane's query
    const bool q_in = qi < p.Lq;
    const bool q_ok = q_in && p.mq[(size_t)b * p.Lq + (q_in ? qi : 0)] != 0;
    const bool wave_on = q_blk + wave * 16 < p.Lq;   // wave-uniform: waves past Lq only help staging

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float inv = 0.f;
    if (wave_on) {
        qk_tiles<DH, NT>(acc, Qsa, Qsb, KVs, nt, nta, wave, l15, g);
        // mask fill, dropout, scale; acc[t][r] is key jp = 16t + 4g + r of query qi
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
                const uint32_t kb = *(const uint32_t*)(km + 16 * t + 4 * g);
                f32x4 mult = {1.f, 1.f, 1.f, 1.f};
                if (p.drop.p > 0.f)
                    mult = drop_apply4(p.drop, (((uint64_t)bh * p.Lq + (q_in ? qi : 0)) * Tp + 16 * t + 4 * g) >> 2,
                                       f32x4{1.f, 1.f, 1.f, 1.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t k = (kb >> (8 * r)) & 0xff;
                    float v = logit_xform(acc[t][r], q_ok && k == 1, mult[r], p.scale);
                    if (k == 2) v = -INFINITY;
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
            if (t < nt) {
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
        inv = 1.0f / sum;
        if (g == 0 && q_in) {
            p.lse[(size_t)bh * p.Lq + qi] = mx;
            p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] = inv;
        }
    }
    __syncthreads();      // every wave is done reading K
    rows_store<DH, UK>(rk, KVs, C::LDC, Tp, tid);
    __syncthreads();
    if (!wave_on) return;

    f32x4 o[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float pb = acc[t][s] * inv;            // P^T[key 16t+4g+s][query]
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) {
                    const int c = 16 * ct + l15;
                    const float a = (c < DH) ? KVs[(16 * t + 4 * g + s) * C::LDC + c] : 0.f;
                    o[ct] = MFMA16(a, pb, o[ct]);
                }
            }
        }
    }
    float am = 0.f;
    if (q_in) {
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const int c = 16 * ct + 4 * g;
            if (c < DH) {
                *(f32x4*)(p.O + (size_t)(b * p.Lq + qi) * p.ldo + col0 + c) = o[ct];
                am = absmax4(am, o[ct]);
            }
        }
    }
    if (p.amax_o) amax_commit(p.amax_o, am, (blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave);
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ D)
template <int DH, int NT>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    constexpr int UK = (NT * 16 * C::KS + ATT_THREADS - 1) / ATT_THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-aware: the heads of one batch row read interleaved 4*DH-byte slices of the same token rows, so
    // consecutive (b,h) must share an L2 or every shared cache line is fetched from HBM once per XCD
    const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / p.H, h = bh % p.H;
    const int q_blk = blockIdx.y * ATT_QB;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    float* Qsa = smem;
    float* Qsb = Qsa + ATT_QB * C::LDR;
    float* dOs = Qsb + ATT_QB * C::LDR;
    float* KVs = dOs + ATT_QB * C::LDR;
    uint8_t* km = (uint8_t*)(KVs + Tp * C::LDMAX);
    const int col0 = h * DH;
    const int nq_valid = max(0, min(ATT_QB, p.Lq - q_blk));

    f32x4 rk[UK];
    {
        f32x4 rqa[C::UQ], rqb[C::UQ], rdo[C::UQ];
        rows_load<DH, C::UQ>(rqa, p.Qa, p.ldq, col0, b * p.Lq + q_blk, ATT_QB, nq_valid, tid);
        rows_load<DH, C::UQ>(rqb, p.Qb, p.ldq, col0, b * p.Lq + q_blk, ATT_QB, nq_valid, tid);
        rows_load<DH, C::UQ>(rdo, p.dO, p.lddo, col0, b * p.Lq + q_blk, ATT_QB, nq_valid, tid);
        keys_load<DH, UK>(rk, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, 0, Tp, col0, tid);
        stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p, tid);
        rows_store<DH, C::UQ>(rqa, Qsa, C::LDR, ATT_QB, tid);
        rows_store<DH, C::UQ>(rqb, Qsb, C::LDR, ATT_QB, tid);
        rows_store<DH, C::UQ>(rdo, dOs, C::LDR, ATT_QB, tid);
        rows_store<DH, UK>(rk, KVs, C::LDR, Tp, tid);
    }
    __syncthreads();
    keys_load<DH, UK>(rk, p.Va, p.ldka, p.Vb, p.ldkb, b, p.La, p.Lb, La_p, 0, Tp, col0, tid);      // V prefetch

    const int qi = q_blk + wave * 16 + l15;
    const bool q_in = qi < p.Lq;
    const bool q_ok = q_in && p.mq[(size_t)b * p.Lq + (q_in ? qi : 0)] != 0;
    const bool wave_on = q_blk + wave * 16 < p.Lq;
    const float row_mx = q_in ? p.lse[(size_t)bh * p.Lq + qi] : 0.f;
    const float row_inv = q_in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] : 0.f;

    f32x4 P[NT], dS[NT];              // P^T, dP^T -> dS^T
    uint64_t live = 0;                // bit 4t+r: d(logit)/d(raw) != 0 (valid pair AND kept by dropout); the factor
                                      // itself is the constant drop.scale * scale, so no per-element array is kept
#pragma unroll
    for (int t = 0; t < NT; ++t) { P[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dS[t] = P[t]; }
    if (wave_on) {
        qk_tiles<DH, NT>(P, Qsa, Qsb, KVs, nt, nta, wave, l15, g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
                const uint32_t kb = *(const uint32_t*)(km + 16 * t + 4 * g);
                f32x4 mult = {1.f, 1.f, 1.f, 1.f};
                if (p.drop.p > 0.f)
                    mult = drop_apply4(p.drop, (((uint64_t)bh * p.Lq + (q_in ? qi : 0)) * Tp + 16 * t + 4 * g) >> 2,
                                       f32x4{1.f, 1.f, 1.f, 1.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t k = (kb >> (8 * r)) & 0xff;
                    const bool valid = q_ok && k == 1;
                    const float v = logit_xform(P[t][r], valid, mult[r], p.scale);
                    P[t][r] = (k == 2) ? 0.f : fast_exp(v - row_mx) * row_inv;
                    if (valid && mult[r] != 0.f) live |= 1ull << (4 * t + r);
                }
            }
        }
    }
    __syncthreads();
    rows_store<DH, UK>(rk, KVs, C::LDR, Tp, tid);                                                   // V (row-on-lane layout)
    __syncthreads();
    keys_load<DH, UK>(rk, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, 0, Tp, col0, tid);      // K again, for dQ
    float Dq = 0.f;
    if (wave_on) {
        // dP^T = V . dO^T ; D = sum_j P dP ; dS^T = P (dP - D) * fac
        float dof[C::KS];
#pragma unroll
        for (int c = 0; c < C::KS; ++c) dof[c] = dOs[(wave * 16 + l15) * C::LDR + 4 * c + g];
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
            if (t < nt) {
                const bool two = (t + 1 < nt) && (t + 1 < NT);
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    dS[t] = MFMA16(KVs[(16 * t + l15) * C::LDR + 4 * c + g], dof[c], dS[t]);
                    if (t + 1 < NT) {
                        if (two) dS[t + 1] = MFMA16(KVs[(16 * (t + 1) + l15) * C::LDR + 4 * c + g], dof[c], dS[t + 1]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Dq += P[t][r] * dS[t][r];
        Dq += __shfl_xor(Dq, 16, 64);
        Dq += __shfl_xor(Dq, 32, 64);
        if (g == 0 && q_in) p.Dvec[(size_t)bh * p.Lq + qi] = Dq;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dS[t][r] = ((live >> (4 * t + r)) & 1ull) ? P[t][r] * (dS[t][r] - Dq) * (p.drop.scale * p.scale) : 0.f;
    }
    __syncthreads();
    rows_store<DH, UK>(rk, KVs, C::LDC, Tp, tid);                                                   // K (column-on-lane layout)
    __syncthreads();
    if (!wave_on) return;
    // dQ^T[c][query] = sum_key K[key][c] dS^T[key][query]; block a keys -> dQa, block b keys -> dQb
    f32x4 da[C::CT], db[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) { da[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; db[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            const bool isa = t < nta;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) {
                    const int c = 16 * ct + l15;
                    const float a = (c < DH) ? KVs[(16 * t + 4 * g + s) * C::LDC + c] : 0.f;
                    if (isa) da[ct] = MFMA16(a, dS[t][s], da[ct]);
                    else     db[ct] = MFMA16(a, dS[t][s], db[ct]);
                }
            }
        }
    }
    float am = 0.f;
    if (q_in) {
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const int c = 16 * ct + 4 * g;
            if (c < DH) {
                *(f32x4*)(p.dQa + (size_t)(b * p.Lq + qi) * p.lddq + col0 + c) = da[ct];
                *(f32x4*)(p.dQb + (size_t)(b * p.Lq + qi) * p.lddq + col0 + c) = db[ct];
                am = absmax4(absmax4(am, da[ct]), db[ct]);
            }
        }
    }
    if (p.amax_q) amax_commit(p.amax_q, am, (blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave);
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// One wave owns 16 keys (one padded key tile) and walks all query tiles.  NQT = upper bound of query tiles
// (Lq <= 16*NQT); every operand of the workgroup is fetched in ONE batch of loads.
template <int DH, int NQT>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    constexpr int ULQ = (NQT * 16 * C::KS + ATT_THREADS - 1) / ATT_THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-aware: the heads of one batch row read interleaved 4*DH-byte slices of the same token rows, so
    // consecutive (b,h) must share an L2 or every shared cache line is fetched from HBM once per XCD
    const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    const int Lq_p = round16(p.Lq), nqt = Lq_p >> 4;
    const int k_blk = blockIdx.y * ATT_QB;          // first padded key of this workgroup
    float* Qsa = smem;
    float* Qsb = Qsa + Lq_p * C::LDR;
    float* dOs = Qsb + Lq_p * C::LDR;
    float* Ks = dOs + Lq_p * C::LDR;
    float* Vs = Ks + ATT_QB * C::LDR;
    float* lses = Vs + ATT_QB * C::LDR;
    float* invs = lses + Lq_p;
    float* Ds = invs + Lq_p;
    uint8_t* qm = (uint8_t*)(Ds + Lq_p);
    uint8_t* km = qm + Lq_p;
    const int col0 = h * DH;
    {
        f32x4 rqa[ULQ], rqb[ULQ], rdo[ULQ], rkk[C::UQ], rvv[C::UQ];
        rows_load<DH, ULQ>(rqa, p.Qa, p.ldq, col0, b * p.Lq, Lq_p, p.Lq, tid);
        rows_load<DH, ULQ>(rqb, p.Qb, p.ldq, col0, b * p.Lq, Lq_p, p.Lq, tid);
        rows_load<DH, ULQ>(rdo, p.dO, p.lddo, col0, b * p.Lq, Lq_p, p.Lq, tid);
        keys_load<DH, C::UQ>(rkk, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, k_blk, ATT_QB, col0, tid);
        keys_load<DH, C::UQ>(rvv, p.Va, p.ldka, p.Vb, p.ldkb, b, p.La, p.Lb, La_p, k_blk, ATT_QB, col0, tid);
        for (int i = tid; i < Lq_p; i += ATT_THREADS) {
            const bool in = i < p.Lq;
            lses[i] = in ? p.lse[(size_t)bh * p.Lq + i] : 0.f;
            invs[i] = in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + i] : 0.f;
            Ds[i] = in ? p.Dvec[(size_t)bh * p.Lq + i] : 0.f;
            qm[i] = in ? (p.mq[(size_t)b * p.Lq + i] ? 1 : 0) : 2;
        }
        stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p, tid);
        rows_store<DH, ULQ>(rqa, Qsa, C::LDR, Lq_p, tid);
        rows_store<DH, ULQ>(rqb, Qsb, C::LDR, Lq_p, tid);
        rows_store<DH, ULQ>(rdo, dOs, C::LDR, Lq_p, tid);
        rows_store<DH, C::UQ>(rkk, Ks, C::LDR, ATT_QB, tid);
        rows_store<DH, C::UQ>(rvv, Vs, C::LDR, ATT_QB, tid);
    }
    __syncthreads();

    const int jt = (k_blk >> 4) + wave;             // this wave's key tile
    if (jt >= nt) return;
    const bool isa = jt < nta;
    const int jp = 16 * jt + l15;                   // this lane's key (padded index)
    const uint8_t kflag = km[jp];
    const float* Qs = isa ? Qsa : Qsb;

    float kf[C::KS], vf[C::KS];
#pragma unroll
    for (int c = 0; c < C::KS; ++c) {
        kf[c] = Ks[(wave * 16 + l15) * C::LDR + 4 * c + g];
        vf[c] = Vs[(wave * 16 + l15) * C::LDR + 4 * c + g];
    }
    f32x4 dk[C::CT], dv[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) { dk[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int qt = 0; qt < nqt; ++qt) {
        // S[query 16qt+4g+r][key jp] and dP likewise (two independent accumulators, interleaved)
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C::KS; ++c) {
            const float qa = Qs[(16 * qt + l15) * C::LDR + 4 * c + g];
            const float da = dOs[(16 * qt + l15) * C::LDR + 4 * c + g];
            s = MFMA16(qa, kf[c], s);
            dp = MFMA16(da, vf[c], dp);
        }
        f32x4 Pv, dSv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = 16 * qt + 4 * g + r;
            const uint8_t qf = qm[qi];
            const bool valid = (qf == 1) && (kflag == 1);
            float mult = 1.f;
            if (p.drop.p > 0.f && qf != 2) mult = drop_mult1(p.drop, ((uint64_t)bh * p.Lq + qi) * Tp + jp);
            const float v = logit_xform(s[r], valid, mult, p.scale);
            const float pr = (kflag == 2 || qf == 2) ? 0.f : fast_exp(v - lses[qi]) * invs[qi];
            Pv[r] = pr;
            dSv[r] = valid ? pr * (dp[r] - Ds[qi]) * mult * p.scale : 0.f;
        }
        // dV^T[c][key] += dO[query][c] P[query][key];  dK^T[c][key] += Q[query][c] dS[query][key]
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) {
                const int c = 16 * ct + l15;
                const float ao = (c < DH) ? dOs[(16 * qt + 4 * g + s4) * C::LDR + c] : 0.f;
                const float aq = (c < DH) ? Qs[(16 * qt + 4 * g + s4) * C::LDR + c] : 0.f;
                dv[ct] = MFMA16(ao, Pv[s4], dv[ct]);
                dk[ct] = MFMA16(aq, dSv[s4], dk[ct]);
            }
        }
    }
    // store: rows = key jp, columns col0 + 16ct + 4g + r
    const bool ka = jp < La_p;
    const int jloc = ka ? jp : jp - La_p;
    const bool real = ka ? (jloc < p.La) : (jloc < p.Lb);
    float am = 0.f;
    if (real) {
        float* dKp = ka ? p.dKa + (size_t)(b * p.La + jloc) * p.lddka : p.dKb + (size_t)(b * p.Lb + jloc) * p.lddkb;
        float* dVp = ka ? p.dVa + (size_t)(b * p.La + jloc) * p.lddka : p.dVb + (size_t)(b * p.Lb + jloc) * p.lddkb;
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const int c = 16 * ct + 4 * g;
            if (c < DH) {
                *(f32x4*)(dKp + col0 + c) = dk[ct];
                *(f32x4*)(dVp + col0 + c) = dv[ct];
                am = absmax4(absmax4(am, dk[ct]), dv[ct]);
            }
        }
    }
    float* slot = isa ? p.amax_ka : p.amax_kb;          // wave-uniform: a key tile lies in one block
    if (slot) amax_commit(slot, am, (blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave);
}

}  // namespace segmm
