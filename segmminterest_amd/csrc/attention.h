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
// One workgroup = one (batch row, head) x up to 64 queries; each wave owns 16 queries and all keys.
// The product is computed TRANSPOSED (S^T = K.Q^T) so the MFMA result has the query on the lane
// (lane&15) and 4 consecutive keys in the 4 result registers: the softmax row-reduce is register-local
// plus two wave shuffles, and P^T feeds the second product (O^T = V^T.P^T) straight from registers --
// the reduction index of an f32 MFMA operand is free to permute, so "key = 4*lanegroup + step" needs no
// cross-lane move.  The backward is two kernels, both recomputing S from Q,K and the saved
// log-sum-exp: a query-major one (dQ, writes D = rowsum(P*dP)) and a key-major one (dK, dV) whose
// reductions over queries stay inside one wave => no atomics, bitwise reproducible.
//
// Key blocks are padded separately to multiples of 16 (pad keys get probability 0); the dropout
// stream is indexed by (b, h, query, padded key) so 4 consecutive keys share one Philox call.
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
};

__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }

template <int DH> struct AttnCfg {
    static constexpr int KS = DH / 4;                       // k-steps over the head dim
    static constexpr int CT = (DH + 15) / 16;               // 16-wide column tiles of the head dim
    static constexpr int LDR = DH + 2;                      // "row on lane&15" reads: stride = 2 mod 4
    static constexpr int LDC = DH + ((DH % 8 == 0) ? 4 : 0);  // "column on lane&15" reads: stride = 4 mod 8
    static constexpr int LDMAX = LDR > LDC ? LDR : LDC;
};

// rows [r0, r0+nrows) of a [*, ld] matrix (columns col0..col0+DH) -> LDS [nrows][lds]; rows >= nvalid are zero
template <int DH>
__device__ __forceinline__ void stage_rows(float* dst, int lds, const float* src, size_t ld, int col0, int row_base,
                                           int nrows, int nvalid, int tid, int nthr) {
    // U loads are issued back to back before the first LDS store, so their latencies overlap (a
    // one-load-per-iteration loop was latency-serialised: 3 waves per workgroup cannot hide it).
    constexpr int KS = DH / 4, U = 8;
    const int total = nrows * KS;
    for (int base = tid; base < total; base += nthr * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * nthr;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < total) {
                const int r = idx / KS, c4 = (idx % KS) * 4;
                if (r < nvalid) v[u] = *(const f32x4*)(src + (size_t)(row_base + r) * ld + col0 + c4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * nthr;
            if (idx < total) {
                const int r = idx / KS, c4 = (idx % KS) * 4;
                float2* d = (float2*)(dst + r * lds + c4);      // row strides are even => 8-byte aligned
                d[0] = make_float2(v[u].x, v[u].y);
                d[1] = make_float2(v[u].z, v[u].w);
            }
        }
    }
}

// key-block staging: padded block a then padded block b
template <int DH>
__device__ __forceinline__ void stage_keys(float* dst, int lds, const float* A, int lda, const float* Bm, int ldb,
                                           int b, int La, int Lb, int La_p, int Lb_p, int col0, int tid, int nthr) {
    stage_rows<DH>(dst, lds, A, lda, col0, b * La, La_p, La, tid, nthr);
    stage_rows<DH>(dst + La_p * lds, lds, Bm, ldb, col0, b * Lb, Lb_p, Lb, tid, nthr);
}

// kmask[jp] : 1 valid, 0 masked token (-10000 fill), 2 alignment pad (probability 0)
__device__ __forceinline__ void stage_kmask(uint8_t* km, const uint8_t* mka, const uint8_t* mkb, int b, int La, int Lb,
                                            int La_p, int Lb_p, int tid, int nthr) {
    for (int j = tid; j < La_p + Lb_p; j += nthr) {
        uint8_t v;
        if (j < La_p) v = (j < La) ? (mka[(size_t)b * La + j] ? 1 : 0) : 2;
        else { const int jb = j - La_p; v = (jb < Lb) ? (mkb[(size_t)b * Lb + jb] ? 1 : 0) : 2; }
        km[j] = v;
    }
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// scaled/masked/dropped logit and its derivative factor wrt the raw QK^T value
__device__ __forceinline__ float logit_xform(float s, bool valid, float mult, float scale) {
    return (valid ? s : -10000.0f) * mult * scale;
}

// ------------------------------------------------------------------------------------------ forward
template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int QB = nw * 16, q_blk = blockIdx.y * QB;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    float* Qsa = smem;
    float* Qsb = Qsa + QB * C::LDR;
    float* KVs = Qsb + QB * C::LDR;
    uint8_t* km = (uint8_t*)(KVs + Tp * C::LDMAX);
    const int col0 = h * DH;
    const int nq_valid = max(0, min(QB, p.Lq - q_blk));

    stage_rows<DH>(Qsa, C::LDR, p.Qa, p.ldq, col0, b * p.Lq + q_blk, QB, nq_valid, tid, nthr);
    stage_rows<DH>(Qsb, C::LDR, p.Qb, p.ldq, col0, b * p.Lq + q_blk, QB, nq_valid, tid, nthr);
    stage_keys<DH>(KVs, C::LDR, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, Lb_p, col0, tid, nthr);
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p, tid, nthr);
    __syncthreads();

    const int qi = q_blk + wave * 16 + l15;          // this lane's query
    const bool q_in = qi < p.Lq;
    const bool q_ok = q_in && p.mq[(size_t)b * p.Lq + (q_in ? qi : 0)] != 0;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        float qa[C::KS], qb[C::KS];
#pragma unroll
        for (int c = 0; c < C::KS; ++c) {
            qa[c] = Qsa[(wave * 16 + l15) * C::LDR + 4 * c + g];
            qb[c] = Qsb[(wave * 16 + l15) * C::LDR + 4 * c + g];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
                const bool isa = t < nta;
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    const float a = KVs[(16 * t + l15) * C::LDR + 4 * c + g];
                    acc[t] = MFMA16(a, isa ? qa[c] : qb[c], acc[t]);
                }
            }
        }
    }
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
                const float e = expf(acc[t][r] - mx);
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

    __syncthreads();      // every wave is done reading K
    stage_keys<DH>(KVs, C::LDC, p.Va, p.ldka, p.Vb, p.ldkb, b, p.La, p.Lb, La_p, Lb_p, col0, tid, nthr);
    __syncthreads();

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
    if (q_in) {
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const int c = 16 * ct + 4 * g;
            if (c < DH) *(f32x4*)(p.O + (size_t)(b * p.Lq + qi) * p.ldo + col0 + c) = o[ct];
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ D)
template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int QB = nw * 16, q_blk = blockIdx.y * QB;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    float* Qsa = smem;
    float* Qsb = Qsa + QB * C::LDR;
    float* dOs = Qsb + QB * C::LDR;
    float* KVs = dOs + QB * C::LDR;
    uint8_t* km = (uint8_t*)(KVs + Tp * C::LDMAX);
    const int col0 = h * DH;
    const int nq_valid = max(0, min(QB, p.Lq - q_blk));

    stage_rows<DH>(Qsa, C::LDR, p.Qa, p.ldq, col0, b * p.Lq + q_blk, QB, nq_valid, tid, nthr);
    stage_rows<DH>(Qsb, C::LDR, p.Qb, p.ldq, col0, b * p.Lq + q_blk, QB, nq_valid, tid, nthr);
    stage_rows<DH>(dOs, C::LDR, p.dO, p.lddo, col0, b * p.Lq + q_blk, QB, nq_valid, tid, nthr);
    stage_keys<DH>(KVs, C::LDR, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, Lb_p, col0, tid, nthr);
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p, tid, nthr);
    __syncthreads();

    const int qi = q_blk + wave * 16 + l15;
    const bool q_in = qi < p.Lq;
    const bool q_ok = q_in && p.mq[(size_t)b * p.Lq + (q_in ? qi : 0)] != 0;
    const float row_mx = q_in ? p.lse[(size_t)bh * p.Lq + qi] : 0.f;
    const float row_inv = q_in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] : 0.f;

    f32x4 P[NT], fac[NT];     // P^T and d(logit)/d(raw) factor
#pragma unroll
    for (int t = 0; t < NT; ++t) P[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        float qa[C::KS], qb[C::KS];
#pragma unroll
        for (int c = 0; c < C::KS; ++c) {
            qa[c] = Qsa[(wave * 16 + l15) * C::LDR + 4 * c + g];
            qb[c] = Qsb[(wave * 16 + l15) * C::LDR + 4 * c + g];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
                const bool isa = t < nta;
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    const float a = KVs[(16 * t + l15) * C::LDR + 4 * c + g];
                    P[t] = MFMA16(a, isa ? qa[c] : qb[c], P[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        fac[t] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                P[t][r] = (k == 2) ? 0.f : expf(v - row_mx) * row_inv;
                fac[t][r] = valid ? mult[r] * p.scale : 0.f;
            }
        }
    }
    __syncthreads();
    stage_keys<DH>(KVs, C::LDR, p.Va, p.ldka, p.Vb, p.ldkb, b, p.La, p.Lb, La_p, Lb_p, col0, tid, nthr);
    __syncthreads();
    // dP^T = V . dO^T ; D = sum_j P dP ; dS^T = P (dP - D) * fac
    f32x4 dS[NT];
    float Dq = 0.f;
    {
        float dof[C::KS];
#pragma unroll
        for (int c = 0; c < C::KS; ++c) dof[c] = dOs[(wave * 16 + l15) * C::LDR + 4 * c + g];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            dS[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < nt) {
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    const float a = KVs[(16 * t + l15) * C::LDR + 4 * c + g];
                    dS[t] = MFMA16(a, dof[c], dS[t]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Dq += P[t][r] * dS[t][r];
            }
        }
    }
    Dq += __shfl_xor(Dq, 16, 64);
    Dq += __shfl_xor(Dq, 32, 64);
    if (g == 0 && q_in) p.Dvec[(size_t)bh * p.Lq + qi] = Dq;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dS[t][r] = P[t][r] * (dS[t][r] - Dq) * fac[t][r];

    __syncthreads();
    stage_keys<DH>(KVs, C::LDC, p.Ka, p.ldka, p.Kb, p.ldkb, b, p.La, p.Lb, La_p, Lb_p, col0, tid, nthr);
    __syncthreads();
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
    if (q_in) {
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const int c = 16 * ct + 4 * g;
            if (c < DH) {
                *(f32x4*)(p.dQa + (size_t)(b * p.Lq + qi) * p.lddq + col0 + c) = da[ct];
                *(f32x4*)(p.dQb + (size_t)(b * p.Lq + qi) * p.lddq + col0 + c) = db[ct];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// One wave owns 16 keys (one padded key tile) and walks all query tiles.
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    const int Lq_p = round16(p.Lq), nqt = Lq_p >> 4;
    const int KB = nw * 16;                         // keys per workgroup
    const int k_blk = blockIdx.y * KB;              // first padded key of this workgroup
    float* Qsa = smem;
    float* Qsb = Qsa + Lq_p * C::LDR;
    float* dOs = Qsb + Lq_p * C::LDR;
    float* Ks = dOs + Lq_p * C::LDR;
    float* Vs = Ks + KB * C::LDR;
    float* lses = Vs + KB * C::LDR;
    float* invs = lses + Lq_p;
    float* Ds = invs + Lq_p;
    uint8_t* qm = (uint8_t*)(Ds + Lq_p);
    uint8_t* km = qm + Lq_p;
    const int col0 = h * DH;

    stage_rows<DH>(Qsa, C::LDR, p.Qa, p.ldq, col0, b * p.Lq, Lq_p, p.Lq, tid, nthr);
    stage_rows<DH>(Qsb, C::LDR, p.Qb, p.ldq, col0, b * p.Lq, Lq_p, p.Lq, tid, nthr);
    stage_rows<DH>(dOs, C::LDR, p.dO, p.lddo, col0, b * p.Lq, Lq_p, p.Lq, tid, nthr);
    // this workgroup's keys (padded index space)
    for (int idx = tid; idx < KB * C::KS; idx += nthr) {
        const int r = idx / C::KS, c4 = (idx % C::KS) * 4, jp = k_blk + r;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (jp < La_p) {
            if (jp < p.La) {
                kv = *(const f32x4*)(p.Ka + (size_t)(b * p.La + jp) * p.ldka + col0 + c4);
                vv = *(const f32x4*)(p.Va + (size_t)(b * p.La + jp) * p.ldka + col0 + c4);
            }
        } else if (jp < Tp) {
            const int jb = jp - La_p;
            if (jb < p.Lb) {
                kv = *(const f32x4*)(p.Kb + (size_t)(b * p.Lb + jb) * p.ldkb + col0 + c4);
                vv = *(const f32x4*)(p.Vb + (size_t)(b * p.Lb + jb) * p.ldkb + col0 + c4);
            }
        }
        float* dk = Ks + r * C::LDR + c4; float* dv = Vs + r * C::LDR + c4;
        dk[0] = kv.x; dk[1] = kv.y; dk[2] = kv.z; dk[3] = kv.w;
        dv[0] = vv.x; dv[1] = vv.y; dv[2] = vv.z; dv[3] = vv.w;
    }
    for (int i = tid; i < Lq_p; i += nthr) {
        const bool in = i < p.Lq;
        lses[i] = in ? p.lse[(size_t)bh * p.Lq + i] : 0.f;
        invs[i] = in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + i] : 0.f;
        Ds[i] = in ? p.Dvec[(size_t)bh * p.Lq + i] : 0.f;
        qm[i] = in ? (p.mq[(size_t)b * p.Lq + i] ? 1 : 0) : 2;
    }
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p, tid, nthr);
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
        // S[query 16qt+4g+r][key jp] and dP likewise
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
            const float pr = (kflag == 2 || qf == 2) ? 0.f : expf(v - lses[qi]) * invs[qi];
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
    if (real) {
        float* dKp = ka ? p.dKa + (size_t)(b * p.La + jloc) * p.lddka : p.dKb + (size_t)(b * p.Lb + jloc) * p.lddkb;
        float* dVp = ka ? p.dVa + (size_t)(b * p.La + jloc) * p.lddka : p.dVb + (size_t)(b * p.Lb + jloc) * p.lddkb;
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) {
            const int c = 16 * ct + 4 * g;
            if (c < DH) {
                *(f32x4*)(dKp + col0 + c) = dk[ct];
                *(f32x4*)(dVp + col0 + c) = dv[ct];
            }
        }
    }
}

}  // namespace segmm
