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
// A head here is tiny (40 queries x 140 keys x 48 dims, 69 KB of operands) and there are B*H = 8192 of them,
// so the kernels are organised for LATENCY, not for reuse: ONE WAVE owns one 16-row tile (16 queries of a
// (b, h) in the forward and the dQ kernel, 16 keys in the dK/dV kernel) and runs start to finish without
// LDS staging and without workgroup barriers (the only shared data is the key-mask byte vector).  Every
// MFMA operand is fetched from global memory directly in fragment form with vector BUFFER loads (one resource
// descriptor per tensor, a per-lane byte offset computed once, the key tile as the scalar offset: no per-tile
// address arithmetic; pad keys of a partial tile read the rows behind the block -- finite data or, past the end
// of the tensor, zeros -- and get probability 0), which works because two index orders are free to choose:
//   * the REDUCTION index of a product: lane (row, g) of a 16x16x4 operand holds, for step j = 4i + e, element
//     k = 16i + 4g + e -- so a lane reads one float4 per 16-float segment of its row (3 for dh = 48), the four
//     g-lanes of a row read one contiguous 64-byte line per instruction, and both operands use the same
//     permutation;
//   * the OUTPUT column order of O^T = V^T.P^T (and dQ, dK, dV): column tile ct, lane column c maps to head
//     column CT*c + ct, so a lane reads CT contiguous floats of a V row and ends up owning the dh/4
//     contiguous output columns [g*dh/4, (g+1)*dh/4) of its query row -> float4 stores.
// The score product is computed TRANSPOSED (S^T = K.Q^T): the MFMA result has the query on lane&15 and four
// consecutive keys in its four registers, so the softmax row reduce is register-local plus two wave
// shuffles and P^T feeds the second product straight from registers.
// The backward is two kernels, both recomputing S from Q, K and the saved softmax row statistics: a
// query-major one (dQ; D = rowsum(P*dP) = rowsum(dO*O) comes from the saved forward output, which makes it a
// single pass over the key tiles) and a key-major one (dK, dV) whose reductions over queries stay inside one
// wave => no atomics, bitwise reproducible.  K/V rows are re-read by the waves of the other
// query tiles of the same head (L1/L2 hits: the 4 waves of a workgroup work on the same head).
//
// Key blocks are padded separately to multiples of 16 (pad keys get probability 0); the dropout
// stream is indexed by (b, h, query, padded key) so 4 consecutive keys share one hash call.
#pragma once
#include "common.h"

namespace segmm {

// Input planes of the planes-in forward (attention_pl.h): the P32 fp16 planes of the Q / K / V column slices, as their producer
// GEMMs wrote them, with the site headers that say whether they are usable.  K and V of a key block are views of one buffer.
struct AttnInPlanes {
    const _Float16 *Qa, *Qb; int ldq2;      // same column slices as AttnArgs::Qa / Qb (null: no input planes)
    const _Float16 *baseA, *baseB;          // lowest plane address of key block a's / b's K and V views
    uint32_t offKa, offVa, offKb, offVb;    // byte offsets of the four views from their base
    uint32_t bytesA, bytesB, bytesQ;        // extents from the base / from the lower of Qa, Qb (buffer range check)
    int ldka2, ldkb2;
    const float *hdr_q, *hdr_ka, *hdr_kb;   // site headers: hdr[0] scale, hdr[1] overflow flag, partial maxima
};

struct AttnArgs {
    int B, H, Lq, La, Lb;
    const float *Qa, *Qb; int ldq;      // [B*Lq, ldq], head h at column h*DH
    const float *Ka, *Va; int ldka;     // [B*La, ldka]
    const float *Kb, *Vb; int ldkb;     // [B*Lb, ldkb]
    const uint8_t *mq, *mka, *mkb;      // [B,Lq] [B,La] [B,Lb]; nonzero = valid token
    float* O; int ldo;                  // [B*Lq, ldo]  (forward: output; backward: the saved forward output, read)
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
    int write_D;                        // dQ kernel: store Dvec (0 when a separate attn_D_kernel launch has produced it)
    int hpb;                            // adjacent heads per workgroup (divides H); waves per workgroup = hpb * row tiles
    uint32_t ka_bytes, kb_bytes, q_bytes, do_bytes;   // extents of the K/V (block a, b), Q and dO views (buffer range check)
    // optional partial maxima (AMAX_SLOTS each, common.h) of what the kernels write, for the fp16x3 GEMM engine:
    float* amax_o;                      // forward: |O|
    float *amax_q, *amax_ka, *amax_kb;  // backward: |dQa|,|dQb| ; |dKa|,|dVa| ; |dKb|,|dVb|
    // optional P32 plane outputs (common.h PlaneOut protocol): the forward's O, and -- fused backward only -- the three
    // gradient buffers (query side: dQa/dQb columns; key block a: dKa/dVa; key block b: dKb/dVb).  The plane pointers
    // address the same column slices as their fp32 twins (slice starts are multiples of 32 columns).
    PlaneOut po_o;
    _Float16 *dQap, *dQbp; int lddq2;
    _Float16 *dKap, *dVap; int lddka2;
    _Float16 *dKbp, *dVbp; int lddkb2;
    float *hdr_q, *hdr_ka, *hdr_kb;
    const float *sin_q, *sin_ka, *sin_kb;
    // fused backward: bit 0 = the gradients whose planes are written get NO fp32 copy (the copy only ever fed the consumers'
    // overflow fallback: 0.57 GB of stores per step at config 2); bit 1 = REPAIR pass of such a launch: same arguments, runs
    // after every producer of the sites has finished; a workgroup leaves at once unless a site's planes are unusable (overflow
    // flag up, or the maximum below the window), in which case it recomputes its tile and rewrites that site's planes with the
    // exact scale of the recorded maxima (headers untouched; segmm_site_fixup then records the scale)
    int pflags;
    AttnInPlanes in;
};
constexpr int ATT_PLANES_ONLY = 1, ATT_REPAIR = 2;

#ifndef ATT_PF
#define ATT_PF 1                        // key tiles prefetched ahead of the one being multiplied (forward); measured:
                                        // 1 -> 225 us, 2 -> 237 us, 3 -> 236 us (fewer resident waves cost more than
                                        // the extra loads in flight gain)
#endif
constexpr int ATT_MAX_THREADS = 320;    // forward: up to 5 waves = 5 row tiles of one (b, h) per workgroup
constexpr int ATT_BWD_THREADS = 768;    // backward kernels need ~160 registers: at most 12 waves per workgroup

__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }

template <int DH> struct AttnCfg {
    static constexpr int KS = DH / 4;                       // MFMA k-steps = floats per lane of a row fragment
    static constexpr int CT = (DH + 15) / 16;               // 16-wide column tiles of the head dim
    // float offset inside a row of lane-group g's first fragment element (see frag_load)
    __device__ static constexpr int row_off(int g) { return (KS % 4 == 0) ? 4 * g : KS * g; }
};

// Row fragment.  Lane (row, g) takes, from every 16-float segment i of its row, the 4 floats [16i + 4g, 16i + 4g + 4):
// its reduction index for step j = 4i + e is k = 16i + 4g + e.  (Not "KS contiguous floats per lane": then the four
// g-lanes of a row would touch four different cache lines in every load instruction; this way one instruction reads
// ONE 64-byte line per row.)  voff = byte offset of (row, column row_off(g)) in the tensor, soff = uniform byte offset.
template <int DH>
__device__ __forceinline__ void frag_load(float (&f)[DH / 4], __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    constexpr int KS = DH / 4;
    if (KS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = buf_load4(r, voff + 64 * i, soff);
            f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
        }
    } else if (KS == 2) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
        f[0] = __uint_as_float(v.x); f[1 % KS] = __uint_as_float(v.y);
    } else {
#pragma unroll
        for (int i = 0; i < KS; ++i) f[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff + 4 * i, (int)soff, 0));
    }
}
// the same fragment through a plain pointer (already at the lane's first element): once-per-wave query-side rows
template <int DH>
__device__ __forceinline__ void frag_load_ptr(float (&f)[DH / 4], const float* q) {
    constexpr int KS = DH / 4;
    if (KS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = *(const f32x4*)(q + 16 * i);
            f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < KS; ++i) f[i] = q[i];
    }
}
// Column fragment: lane (c, .) takes the CT contiguous floats [CT*c, CT*c + CT) of a row = its column of every
// column tile.  voff = byte offset of (row, column CT*c).
template <int DH>
__device__ __forceinline__ void col_load(float (&f)[(DH + 15) / 16], __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, int c) {
    constexpr int CT = (DH + 15) / 16;
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x3_t __attribute__((ext_vector_type(3)));
    if (CT * c < DH) {
        if (CT == 4) {
            const f32x4 v = buf_load4(r, voff, soff);
            f[0] = v.x; f[1 % CT] = v.y; f[2 % CT] = v.z; f[3 % CT] = v.w;
        } else if (CT == 3) {
            const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff, (int)soff, 0);
            f[0] = __uint_as_float(v.x); f[1 % CT] = __uint_as_float(v.y); f[2 % CT] = __uint_as_float(v.z);
        } else if (CT == 2) {
            const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
            f[0] = __uint_as_float(v.x); f[1 % CT] = __uint_as_float(v.y);
        } else {
            f[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
        }
    } else {
#pragma unroll
        for (int i = 0; i < CT; ++i) f[i] = 0.f;
    }
}
// result of a column-permuted product: lane (., g) register r of tile ct is head column CT*(4g + r) + ct, i.e. the
// lane owns the 4*CT contiguous columns starting at 4*CT*g
template <int DH>
__device__ __forceinline__ float col_store(float* rowp, const f32x4 (&o)[(DH + 15) / 16], int g, float am) {
    constexpr int CT = (DH + 15) / 16;
    if (4 * CT * g < DH) {
        float t[4 * CT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) t[CT * r + ct] = o[ct][r];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const f32x4 v = {t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
            *(f32x4*)(rowp + 4 * CT * g + 4 * i) = v;
            am = absmax4(am, v);
        }
    }
    return am;
}

// ---- plane stores of the attention kernels: 16 bytes per lane (8 hi terms or 8 lo terms of 8 consecutive columns), like
// plane_store4_pair (common.h), but the partner that holds the other 4 columns of an aligned group of 8 is either the SAME
// lane (a lane owns 4 CT consecutive columns) or the lane 16 further (g ^ 1): exchange through ds_swizzle (xor 16).
__device__ __forceinline__ uint32_t swz16(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
struct HL { uint32_t h0, h1, l0, l1; };
__device__ __forceinline__ HL split4(f32x4 v, float s) {
    HL r;
    splith_pair(v.x, v.y, s, r.h0, r.l0);
    splith_pair(v.z, v.w, s, r.h1, r.l1);
    return r;
}
__device__ __forceinline__ _Float16* p32_at(_Float16* p, int ld2, long long row, int c) { return p + row * ld2 + ((c >> 5) << 6) + (c & 31); }
// columns c .. c+7 (c % 8 == 0) held by ONE lane as two split groups a (c..c+3), b (c+4..c+7)
__device__ __forceinline__ void plane_store8(_Float16* p, int ld2, long long row, int c, const HL& a, const HL& b) {
    _Float16* o = p32_at(p, ld2, row, c);
    *(uint4*)o = make_uint4(a.h0, a.h1, b.h0, b.h1);
    *(uint4*)(o + 32) = make_uint4(a.l0, a.l1, b.l0, b.l1);
}
// columns c .. c+3 of this lane, the other half of the aligned 8 in lane ^ 16 (which calls this in the same instruction)
__device__ __forceinline__ void plane_store4_x16(_Float16* p, int ld2, long long row, int c, const HL& a) {
    const bool odd = (c & 4) != 0;
    const uint32_t r0 = swz16(odd ? a.h0 : a.l0), r1 = swz16(odd ? a.h1 : a.l1);
    _Float16* o = p32_at(p, ld2, row, c & ~7) + (odd ? 32 : 0);
    *(uint4*)o = odd ? make_uint4(r0, r1, a.l0, a.l1) : make_uint4(a.h0, a.h1, r0, r1);
}
// NG consecutive float4 groups of one row starting at column c0 (multiple of 4; c0 of the lane 16 further = c0 + 4 NG, same
// row; every lane of the 16-lane exchange pairs takes part).  Groups are paired into aligned 8-column chunks in-lane where
// possible, across lane ^ 16 at the seams.
template <int NG>
__device__ __forceinline__ void plane_store_groups(_Float16* p, int ld2, long long row, int c0, const f32x4 (&v)[NG], float s) {
    HL g[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = split4(v[i], s);
    if (NG % 2 == 0) {                        // c0 % 8 == 0 for every lane (4 NG % 8 == 0): all pairs in-lane
#pragma unroll
        for (int i = 0; i < NG; i += 2) plane_store8(p, ld2, row, c0 + 4 * i, g[i], g[i + 1]);
    } else {
        const bool odd = (c0 & 4) != 0;        // odd lanes: group 0 completes the previous lane's last group
        // seam: this lane's first (odd) or last (even) group with lane ^ 16
        const HL sm = odd ? g[0] : g[NG - 1];
        plane_store4_x16(p, ld2, row, odd ? c0 : c0 + 4 * (NG - 1), sm);
#pragma unroll
        for (int k = 0; k < NG / 2; ++k) {
            const HL a = odd ? g[2 * k + 1] : g[2 * k], b = odd ? g[2 * k + 2] : g[2 * k + 1];
            plane_store8(p, ld2, row, c0 + 4 * (2 * k) + (odd ? 4 : 0), a, b);
        }
    }
}

// the same store plus the P32 planes of the values (row = absolute row of the output tensor, col0 = first column of the head)
template <int DH>
__device__ __forceinline__ float col_store_p(float* rowp, _Float16* planes, int ld2, long long row, int col0, float ps,
                                             const f32x4 (&o)[(DH + 15) / 16], int g, float am) {
    constexpr int CT = (DH + 15) / 16;
    f32x4 vv[CT];
    if (4 * CT * g < DH) {
        float t[4 * CT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) t[CT * r + ct] = o[ct][r];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            vv[i] = f32x4{t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
            *(f32x4*)(rowp + 4 * CT * g + 4 * i) = vv[i];
            am = absmax4(am, vv[i]);
        }
        if (DH % 16 != 0 || (col0 & 7) != 0) {          // narrow heads (some g idle) or unaligned head start: 8-byte stores
#pragma unroll
            for (int i = 0; i < CT; ++i) plane_store4(planes, ld2, row, col0 + 4 * CT * g + 4 * i, vv[i], ps);
        }
    }
    if (DH % 16 == 0 && (col0 & 7) == 0) plane_store_groups<CT>(planes, ld2, row, col0 + 4 * CT * g, vv, ps);
    return am;
}

// kmask[jp] : 1 valid, 0 masked token (-10000 fill), 2 alignment pad (probability 0)
__device__ __forceinline__ void stage_kmask(uint8_t* km, const uint8_t* mka, const uint8_t* mkb, int b, int La, int Lb,
                                            int La_p, int Lb_p) {
    for (int j = threadIdx.x; j < La_p + Lb_p; j += blockDim.x) {
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

// scaled/masked/dropped logit.  The products must be ROUNDED fp32 values, identically in every kernel: with contraction on,
// the compiler folds the last multiply into the caller's "v - rowmax" as fma(t, scale, -rowmax) in some kernels and not in
// others; for a masked query row (every logit -10000*scale = -3535.5.., ulp 2.4e-4) the backward then recomputed
// exp(v - rowmax) as exp(+-1e-4) instead of exp(0) against the forward's 1/sum: a 1e-4 relative error on the probabilities
// of padded rows (measured on dV, tools/attn_err_probe.py).
__device__ __forceinline__ float logit_xform(float s, bool valid, float mult, float scale) {
#pragma clang fp contract(off)
    return (valid ? s : -10000.0f) * mult * scale;
}

// The two key blocks of one batch row, as seen by one lane: resources of the K and V tensors of each block, the
// lane's byte offsets at key 0 of the row (row-fragment form: key = lane&15; column-fragment form: key = 4*(lane>>4))
// and the row pitch.  tile() yields the operands of padded key tile t; which block it is in is wave-uniform.
template <int DH>
struct KeyBlocks {
    __amdgpu_buffer_rsrc_t ka, va, kb, vb;
    uint32_t row_a, row_b, col_a, col_b;     // lane byte offsets
    uint32_t pitch_a, pitch_b;               // bytes per key row
    int nta;
    __device__ __forceinline__ void init(const AttnArgs& p, int b, int col0, int l15, int g) {
        using C = AttnCfg<DH>;
        ka = make_rsrc(p.Ka, p.ka_bytes); va = make_rsrc(p.Va, p.ka_bytes);
        kb = make_rsrc(p.Kb, p.kb_bytes); vb = make_rsrc(p.Vb, p.kb_bytes);
        pitch_a = (uint32_t)p.ldka * 4u; pitch_b = (uint32_t)p.ldkb * 4u;
        row_a = ((uint32_t)(b * p.La + l15) * (uint32_t)p.ldka + col0 + C::row_off(g)) * 4u;
        row_b = ((uint32_t)(b * p.Lb + l15) * (uint32_t)p.ldkb + col0 + C::row_off(g)) * 4u;
        col_a = ((uint32_t)(b * p.La + 4 * g) * (uint32_t)p.ldka + col0 + C::CT * l15) * 4u;
        col_b = ((uint32_t)(b * p.Lb + 4 * g) * (uint32_t)p.ldkb + col0 + C::CT * l15) * 4u;
        nta = round16(p.La) >> 4;
    }
};

// ------------------------------------------------------------------------------------------ forward
template <int DH, int NT>
#ifndef ATT_FWD_WAVES
#define ATT_FWD_WAVES 0          // probe: minimum waves per SIMD the register allocator has to make room for (0 = compiler's choice)
#endif
#if ATT_FWD_WAVES
__global__ __launch_bounds__(ATT_MAX_THREADS, ATT_FWD_WAVES) void attn_fwd_kernel(const AttnArgs p) {
#else
__global__ __launch_bounds__(ATT_MAX_THREADS) void attn_fwd_kernel(const AttnArgs p) {
#endif
    using C = AttnCfg<DH>;
    const DropCfg drop_ = drop_live(p.drop);
    extern __shared__ uint8_t km[];
    // A workgroup = p.hpb ADJACENT heads of one batch row x wq row tiles: the heads read interleaved 4*DH-byte slices of
    // the same token rows, so their loads, issued together, touch each DRAM page / cache line once instead of once per
    // head at unrelated times.  XCD-aware block order keeps the remaining head groups of a batch row on one L2.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, wq = wpb / p.hpb;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = xcd_remap(blockIdx.x, gridDim.x) * p.hpb + wave / wq, b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    const int col0 = h * DH;
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p);
    __syncthreads();
    const int qt = blockIdx.y * wq + wave % wq;      // this wave's query tile
    if (16 * qt >= p.Lq) return;
    const int qi = 16 * qt + l15;                    // this lane's query
    const bool q_in = qi < p.Lq;
    const size_t qrow = (size_t)b * p.Lq + min(qi, p.Lq - 1);
    const bool q_ok = q_in && p.mq[qrow] != 0;
    KeyBlocks<DH> kbk;
    kbk.init(p, b, col0, l15, g);

    float qa[C::KS], qb[C::KS];
    frag_load_ptr<DH>(qa, p.Qa + qrow * p.ldq + col0 + C::row_off(g));
    frag_load_ptr<DH>(qb, p.Qb + qrow * p.ldq + col0 + C::row_off(g));
    f32x4 acc[NT];
    {   // S^T tiles: acc[t][r] = sum_c K[16t + 4g + r][c] Q[query][c]; the K fragments of the next ATT_PF tiles are in
        // flight under the MFMAs (the loads are L2/HBM-latency bound: 12 MFMAs = 384 cycles cover less than one L2 hit)
        float kf[ATT_PF + 1][C::KS];
        auto kfetch = [&](int t) {
            if (t < nta) frag_load<DH>(kf[t % (ATT_PF + 1)], kbk.ka, kbk.row_a, (uint32_t)(16 * t) * kbk.pitch_a);
            else frag_load<DH>(kf[t % (ATT_PF + 1)], kbk.kb, kbk.row_b, (uint32_t)(16 * (t - nta)) * kbk.pitch_b);
        };
#pragma unroll
        for (int t = 0; t < ATT_PF; ++t)
            if (t < NT && t < nt) kfetch(t);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < nt) {
                if (t + ATT_PF < NT && t + ATT_PF < nt) kfetch(t + ATT_PF);
                if (t < nta) {       // wave-uniform: no per-element select of the query projection
#pragma unroll
                    for (int c = 0; c < C::KS; ++c) acc[t] = MFMA16(kf[t % (ATT_PF + 1)][c], qa[c], acc[t]);
                } else {
#pragma unroll
                    for (int c = 0; c < C::KS; ++c) acc[t] = MFMA16(kf[t % (ATT_PF + 1)][c], qb[c], acc[t]);
                }
            }
        }
    }
    // first V rows on their way while the softmax runs
    float vf[ATT_PF + 1][4][C::CT];
    auto vfetch = [&](int t) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (t < nta) col_load<DH>(vf[t % (ATT_PF + 1)][s], kbk.va, kbk.col_a, (uint32_t)(16 * t + s) * kbk.pitch_a, l15);
            else col_load<DH>(vf[t % (ATT_PF + 1)][s], kbk.vb, kbk.col_b, (uint32_t)(16 * (t - nta) + s) * kbk.pitch_b, l15);
        }
    };
#pragma unroll
    for (int t = 0; t < ATT_PF; ++t)
        if (t < NT && t < nt) vfetch(t);

    // mask fill, dropout, scale; acc[t][r] is key jp = 16t + 4g + r of query qi
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            const uint32_t kb = *(const uint32_t*)(km + 16 * t + 4 * g);
            f32x4 mult = {1.f, 1.f, 1.f, 1.f};
            if (drop_.p > 0.f)
                mult = drop_apply4(drop_, (((uint64_t)bh * p.Lq + (q_in ? qi : 0)) * Tp + 16 * t + 4 * g) >> 2,
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
    const float inv = 1.0f / sum;
    if (g == 0 && q_in) {
        p.lse[(size_t)bh * p.Lq + qi] = mx;
        p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] = inv;
    }

    // O^T[c][query] = sum_key V[key][c] P^T[key][query]; step s of tile t contracts keys 16t + 4g + s
    f32x4 o[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            if (t + ATT_PF < NT && t + ATT_PF < nt) vfetch(t + ATT_PF);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float pb = acc[t][s] * inv;            // P^T[key 16t+4g+s][query]
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) o[ct] = MFMA16(vf[t % (ATT_PF + 1)][s][ct], pb, o[ct]);
            }
        }
    }
    float am = 0.f;
    const float ps = plane_scale(p.po_o);
    if (q_in) {
        if (ps > 0.f) am = col_store_p<DH>(p.O + qrow * p.ldo + col0, p.po_o.p, p.po_o.ld2, (long long)qrow, col0, ps, o, g, am);
        else am = col_store<DH>(p.O + qrow * p.ldo + col0, o, g, am);
    }
    plane_finish(p.po_o, p.amax_o, am, (blockIdx.x * gridDim.y + blockIdx.y) * wpb + wave, ps,
                 blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
}

// ------------------------------------------------------------------------------------------ forward, LDS-DMA staged (round 4)
// The direct-load forward above is bound by LATENCY, not by bytes or MFMAs: every wave walks a dependent chain of ~20 rounds of
// fragment-shaped global loads (one key tile of K, then of V, one tile prefetched) at 3 waves per SIMD, and the three query-tile
// waves of a head each fetch all of K and V.  Here ONE workgroup owns one (b, h): its waves first issue the head's whole K / V
// working set (both key blocks, 54 KB at config 2) as LDS-DMA (`buffer_load ... lds`, 16 B per lane, no registers) -- every load
// of the head in flight at once, a single round of memory latency -- then wait once (vmcnt(0) + one barrier) and run the same
// S^T = K Q^T -> softmax -> O^T = V^T P^T arithmetic as above with operands read from LDS in fragment form.  Two to three such
// workgroups share a CU, so one's staging flies under another's MFMAs.  Q rows (used by one wave only) go straight to registers.
// LDS image: rows of DH floats at a pitch of DH + 4 floats (an ODD number of 16-byte chunks: the 16 rows of a row-fragment
// read land in distinct bank groups); the pad chunk of a row and the slack behind an array are DMA'd from an out-of-range
// offset (zeros).  Arithmetic, masks and dropout stream are those of attn_fwd_kernel; with one key group per query tile the results
// are bit-identical, with several the softmax sums are merged in another order (differences at the 1e-7 level).
constexpr uint32_t ATT_BUF_OOB = 0x80000000u;
__device__ __forceinline__ void att_lds_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, SEGMM_ATT_AUX);
}
template <int DH> __host__ __device__ constexpr int att_lds_chunks(int rows) { return ((rows * (DH / 4 + 1)) + 63) & ~63; }      // 16-byte chunks of one staged array
template <int DH>
inline size_t attn_fwd_lds_bytes(int La, int Lb) {
    const int La_p = (La + 15) & ~15, Lb_p = (Lb + 15) & ~15;
    return (size_t)(2 * att_lds_chunks<DH>(La_p) + 2 * att_lds_chunks<DH>(Lb_p)) * 16 + (size_t)(La_p + Lb_p);
}

template <int DH, int NT>
__global__ __launch_bounds__(768) void attn_fwd_lds_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    constexpr int PC = DH / 4 + 1, PF = 4 * PC;            // row pitch in 16-byte chunks / in floats
    const DropCfg drop_ = drop_live(p.drop);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_att[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    const int col0 = h * DH;
    const int cha = att_lds_chunks<DH>(La_p), chb = att_lds_chunks<DH>(Lb_p);
    float* sKa = (float*)smem_att;
    float* sVa = sKa + 4 * cha;
    float* sKb = sVa + 4 * cha;
    float* sVb = sKb + 4 * chb;
    uint8_t* km = (uint8_t*)(sVb + 4 * chb);

    // ---- wave (qt, kg): 16-query tile qt, key-tile group kg of ksp -- the key tiles t = kg, kg + ksp, ... are this wave's, the
    // groups' partial softmax sums meet in LDS at the end.  More waves per head on the same staged K / V: with one wave per query
    // tile the compute phase (a chain of ds_read -> MFMA -> softmax -> MFMA per wave at ~1 wave per SIMD) was twice the staging.
    const int nqt = (p.Lq + 15) >> 4, ksp = nw / nqt;
    const int qt = wave % nqt, kg = wave / nqt;
    const int qi = 16 * qt + l15;
    const bool q_in = qi < p.Lq;
    const size_t qrow = (size_t)b * p.Lq + min(qi, p.Lq - 1);
    float qa[C::KS], qb[C::KS];
    frag_load_ptr<DH>(qa, p.Qa + qrow * p.ldq + col0 + C::row_off(g));
    frag_load_ptr<DH>(qb, p.Qb + qrow * p.ldq + col0 + C::row_off(g));
    const bool q_ok = q_in && p.mq[qrow] != 0;

    // ---- stage K and V of both key blocks: chunk pch of an array = (row pch / PC, 16-byte piece pch % PC); a wave issues whole
    // 64-chunk groups (1 KB of LDS per instruction, lane i at base + 16 i)
    auto stage = [&](const float* base, uint32_t bytes, int L, int Lp, int ld, float* dst, int nchunks) {
        if (Lp == 0) return;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, bytes);
        const uint32_t so = ((uint32_t)(b * L) * (uint32_t)ld + (uint32_t)col0) * 4u;
        const int total = Lp * PC;
        for (int c0 = 64 * wave; c0 < nchunks; c0 += 64 * nw) {
            const int pch = c0 + lane, row = pch / PC, c = pch - row * PC;
            uint32_t vo = (uint32_t)row * (uint32_t)ld * 4u + (uint32_t)c * 16u;
            if (c == PC - 1 || pch >= total) vo = ATT_BUF_OOB;          // pad chunk of a row / slack behind the array: zeros
            att_lds_dma16(rs, (char*)dst + (size_t)c0 * 16, vo, so);
        }
    };
#ifdef SEGMM_ATT_PROBE
    if (!(p.pflags & 512))            // (timing probes of -DSEGMM_ATT_PROBE builds, SEGMM_ATT_FWD_DBG: 256 = return after the staging, 512 = no staging; results wrong)
#endif
    {
        stage(p.Ka, p.ka_bytes, p.La, La_p, p.ldka, sKa, cha);
        stage(p.Kb, p.kb_bytes, p.Lb, Lb_p, p.ldkb, sKb, chb);
        stage(p.Va, p.ka_bytes, p.La, La_p, p.ldka, sVa, cha);
        stage(p.Vb, p.kb_bytes, p.Lb, Lb_p, p.ldkb, sVb, chb);
    }
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SEGMM_ATT_PROBE
    if (p.pflags & 256) return;
#endif

    // ---- S^T tiles from LDS: acc[u][r] = sum_c K[16t + 4g + r][c] Q[query][c], t = ksp u + kg
    f32x4 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t = ksp * u + kg;
        if (t < nt) {
            const float* kr = (t < nta ? sKa + (16 * t + l15) * PF : sKb + (16 * (t - nta) + l15) * PF) + C::row_off(g);
            float kf[C::KS];
            if (C::KS % 4 == 0) {
#pragma unroll
                for (int i = 0; i < C::KS / 4; ++i) {
                    const f32x4 v = *(const f32x4*)(kr + 16 * i);
                    kf[4 * i] = v.x; kf[4 * i + 1] = v.y; kf[4 * i + 2] = v.z; kf[4 * i + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < C::KS; ++i) kf[i] = kr[i];
            }
            if (t < nta) {
#pragma unroll
                for (int c = 0; c < C::KS; ++c) acc[u] = MFMA16(kf[c], qa[c], acc[u]);
            } else {
#pragma unroll
                for (int c = 0; c < C::KS; ++c) acc[u] = MFMA16(kf[c], qb[c], acc[u]);
            }
        }
    }
    // mask fill, dropout, scale; acc[u][r] is key jp = 16t + 4g + r of query qi
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int t = ksp * u + kg;
        if (t < nt) {
            const uint32_t kb = *(const uint32_t*)(km + 16 * t + 4 * g);
            f32x4 mult = {1.f, 1.f, 1.f, 1.f};
            if (drop_.p > 0.f)
                mult = drop_apply4(drop_, (((uint64_t)bh * p.Lq + (q_in ? qi : 0)) * Tp + 16 * t + 4 * g) >> 2,
                                   f32x4{1.f, 1.f, 1.f, 1.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t k = (kb >> (8 * r)) & 0xff;
                float v = logit_xform(acc[u][r], q_ok && k == 1, mult[r], p.scale);
                if (k == 2) v = -INFINITY;
                acc[u][r] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (ksp * u + kg < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fast_exp(acc[u][r] - mx);          // (a wave with no tile: mx = -inf, never evaluated)
                acc[u][r] = e;
                sum += e;
            }
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // one key group: normalised probabilities go into the second product, exactly like attn_fwd_kernel (bit-identical results);
    // several groups: unnormalised e = exp(logit - mx_group), normalised after the merge
    const float pnorm = ksp == 1 ? 1.0f / sum : 1.0f;
    // O^T[c][query] = sum_key V[key][c] P^T[key][query]; step s of tile t contracts keys 16t + 4g + s
    f32x4 o[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int t = ksp * u + kg;
        if (t < nt) {
            const float* vr = (t < nta ? sVa + (16 * t + 4 * g) * PF : sVb + (16 * (t - nta) + 4 * g) * PF) + C::CT * l15;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                float vf[C::CT];
                if (C::CT * l15 < DH) {
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) vf[ct] = vr[s_ * PF + ct];
                } else {
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) vf[ct] = 0.f;
                }
                const float pb = acc[u][s_] * pnorm;            // P^T (or e^T) [key 16t+4g+s][query]
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) o[ct] = MFMA16(vf[ct], pb, o[ct]);
            }
        }
    }
    // ---- merge the key groups of a query tile (ksp > 1): partials through LDS (the staged K / V are dead by now); group 0 combines
    // in group order (deterministic): m = max_k m_k, sum = sum_k s_k e^(m_k - m), O = sum_k O_k e^(m_k - m)
    if (ksp > 1) {
        __syncthreads();                                   // every wave is done reading K / V
        float* part = (float*)smem_att + (size_t)wave * (64 * 4 * C::CT + 128);          // [CT][64 lanes] float4 | mx[64] | sum[64]
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) *(f32x4*)(part + (ct * 64 + lane) * 4) = o[ct];
        part[64 * 4 * C::CT + lane] = mx;
        part[64 * 4 * C::CT + 64 + lane] = sum;
        __syncthreads();
        if (kg != 0) return;
        float m = mx;
        for (int k = 1; k < ksp; ++k) m = fmaxf(m, ((const float*)smem_att)[(size_t)(k * nqt + qt) * (64 * 4 * C::CT + 128) + 64 * 4 * C::CT + lane]);
        const float f0 = fast_exp(mx - m);
        sum *= f0;
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) o[ct] *= f0;
        for (int k = 1; k < ksp; ++k) {
            const float* pk = (const float*)smem_att + (size_t)(k * nqt + qt) * (64 * 4 * C::CT + 128);
            const float mk = pk[64 * 4 * C::CT + lane];
            const float fk = mk == -INFINITY ? 0.f : fast_exp(mk - m);          // a group without tiles contributes nothing
            sum += pk[64 * 4 * C::CT + 64 + lane] * fk;
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) o[ct] += *(const f32x4*)(pk + (ct * 64 + lane) * 4) * fk;
        }
        mx = m;
    }
    const float inv = ksp == 1 ? pnorm : 1.0f / sum;
    if (ksp > 1) {
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct) o[ct] *= inv;
    }
    if (g == 0 && q_in) {
        p.lse[(size_t)bh * p.Lq + qi] = mx;
        p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] = inv;
    }
    float am = 0.f;
    const float ps = plane_scale(p.po_o);
    if (q_in) {
        if (ps > 0.f) am = col_store_p<DH>(p.O + qrow * p.ldo + col0, p.po_o.p, p.po_o.ld2, (long long)qrow, col0, ps, o, g, am);
        else am = col_store<DH>(p.O + qrow * p.ldo + col0, o, g, am);
    }
    plane_finish(p.po_o, p.amax_o, am, blockIdx.x * nw + wave, ps, blockIdx.x == 0 && threadIdx.x == 0);
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ D)
// D[q] = sum_j P[q][j] dP[q][j] = sum_c dO[q][c] O[q][c] (O = P.V with the very same P), so with the saved forward
// output the kernel is ONE pass over the key tiles with nothing but the dQ accumulators carried along:
//   S^T_t = K_t.Q^T -> P^T_t ;  dP^T_t = V_t.dO^T ;  dS^T_t = P^T_t (dP^T_t - D) fac ;  dQ^T += K_t^T . dS^T_t
template <int DH, int NT>
__global__ __launch_bounds__(ATT_BWD_THREADS) void attn_bwd_dq_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    const DropCfg drop_ = drop_live(p.drop);
    extern __shared__ uint8_t km[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, wq = wpb / p.hpb;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = xcd_remap(blockIdx.x, gridDim.x) * p.hpb + wave / wq, b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    const int col0 = h * DH;
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p);
    __syncthreads();
    const int qt = blockIdx.y * wq + wave % wq;
    if (16 * qt >= p.Lq) return;
    const int qi = 16 * qt + l15;
    const bool q_in = qi < p.Lq;
    const size_t qrow = (size_t)b * p.Lq + min(qi, p.Lq - 1);
    const bool q_ok = q_in && p.mq[qrow] != 0;
    const float row_mx = q_in ? p.lse[(size_t)bh * p.Lq + qi] : 0.f;
    const float row_inv = q_in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + qi] : 0.f;
    KeyBlocks<DH> kbk;
    kbk.init(p, b, col0, l15, g);

    float qa[C::KS], qb[C::KS], dof[C::KS];
    float Dq = 0.f;
    {
        const size_t ro = col0 + C::row_off(g);
        float of[C::KS];
        frag_load_ptr<DH>(qa, p.Qa + qrow * p.ldq + ro);
        frag_load_ptr<DH>(qb, p.Qb + qrow * p.ldq + ro);
        frag_load_ptr<DH>(dof, p.dO + qrow * p.lddo + ro);
        frag_load_ptr<DH>(of, p.O + qrow * p.ldo + ro);
#pragma unroll
        for (int c = 0; c < C::KS; ++c) Dq += dof[c] * of[c];
        Dq += __shfl_xor(Dq, 16, 64);
        Dq += __shfl_xor(Dq, 32, 64);
        if (p.write_D && g == 0 && q_in) p.Dvec[(size_t)bh * p.Lq + qi] = Dq;
    }
    const float fac = drop_.scale * p.scale;        // d(logit)/d(raw) of a live, kept element

    // K and V row fragments of the NEXT tile are fetched under the current tile's MFMAs (two buffers); the K column
    // fragments of the current tile are fetched at its start and land under its 24 score MFMAs and the softmax
    float kf[2][C::KS], vr[2][C::KS], kc[4][C::CT];
    auto fetch_rows = [&](int buf, int t) {
        if (t < nta) {
            const uint32_t so = (uint32_t)(16 * t) * kbk.pitch_a;
            frag_load<DH>(kf[buf], kbk.ka, kbk.row_a, so);
            frag_load<DH>(vr[buf], kbk.va, kbk.row_a, so);
        } else {
            const uint32_t so = (uint32_t)(16 * (t - nta)) * kbk.pitch_b;
            frag_load<DH>(kf[buf], kbk.kb, kbk.row_b, so);
            frag_load<DH>(vr[buf], kbk.vb, kbk.row_b, so);
        }
    };
    auto fetch_cols = [&](int t) {
        if (t < nta) {
#pragma unroll
            for (int s = 0; s < 4; ++s) col_load<DH>(kc[s], kbk.ka, kbk.col_a, (uint32_t)(16 * t + s) * kbk.pitch_a, l15);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) col_load<DH>(kc[s], kbk.kb, kbk.col_b, (uint32_t)(16 * (t - nta) + s) * kbk.pitch_b, l15);
        }
    };
    f32x4 da[C::CT], db[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) { da[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; db[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    fetch_rows(0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            if (t + 1 < NT && t + 1 < nt) fetch_rows((t + 1) & 1, t + 1);
            fetch_cols(t);
            const bool isa = t < nta;
            f32x4 P = {0.f, 0.f, 0.f, 0.f}, dS = {0.f, 0.f, 0.f, 0.f};
            if (isa) {
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    P = MFMA16(kf[t & 1][c], qa[c], P);
                    dS = MFMA16(vr[t & 1][c], dof[c], dS);
                }
            } else {
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    P = MFMA16(kf[t & 1][c], qb[c], P);
                    dS = MFMA16(vr[t & 1][c], dof[c], dS);
                }
            }
            const uint32_t kb = *(const uint32_t*)(km + 16 * t + 4 * g);
            f32x4 mult = {1.f, 1.f, 1.f, 1.f};
            if (drop_.p > 0.f)
                mult = drop_apply4(drop_, (((uint64_t)bh * p.Lq + (q_in ? qi : 0)) * Tp + 16 * t + 4 * g) >> 2,
                                   f32x4{1.f, 1.f, 1.f, 1.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t k = (kb >> (8 * r)) & 0xff;
                const bool valid = q_ok && k == 1;
                const float v = logit_xform(P[r], valid, mult[r], p.scale);
                const float pr = (k == 2) ? 0.f : fast_exp(v - row_mx) * row_inv;
                dS[r] = (valid && mult[r] != 0.f) ? pr * (dS[r] - Dq) * fac : 0.f;
            }
            // dQ^T[c][query] += sum_key K[key][c] dS^T[key][query]; block a keys -> dQa, block b keys -> dQb
            if (isa) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) da[ct] = MFMA16(kc[s][ct], dS[s], da[ct]);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) db[ct] = MFMA16(kc[s][ct], dS[s], db[ct]);
            }
        }
    }
    float am = 0.f;
    if (q_in) {
        if (p.dQa) am = col_store<DH>(p.dQa + qrow * p.lddq + col0, da, g, am);      // null: empty key block (ablations)
        if (p.dQb) am = col_store<DH>(p.dQb + qrow * p.lddq + col0, db, g, am);
    }
    if (p.amax_q) amax_commit(p.amax_q, am, (blockIdx.x * gridDim.y + blockIdx.y) * wpb + wave);
}

// D[b, h, q] = sum_c dO[b, q, h*DH + c] * O[b, q, h*DH + c] on its own: lets the dQ and the dK/dV kernels of one backward
// run CONCURRENTLY on two streams (both are latency-bound; dK/dV needs D of every query before it starts).
// One thread per (b, q, h); adjacent threads read adjacent 4*DH-byte head slices of the same token row.
template <int DH>
__global__ __launch_bounds__(256) void attn_D_kernel(const AttnArgs p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * p.Lq * p.H) return;
    const int h = (int)(i % p.H);
    const long long row = i / p.H;              // b * Lq + q
    const int b = (int)(row / p.Lq), q = (int)(row % p.Lq);
    const float* dO = p.dO + row * p.lddo + h * DH;
    const float* O = p.O + row * p.ldo + h * DH;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const f32x4 a = *(const f32x4*)(dO + c), o = *(const f32x4*)(O + c);
        s += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
    }
    p.Dvec[((size_t)b * p.H + h) * p.Lq + q] = s;
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// One wave owns 16 keys (one padded key tile) and walks all query tiles.  The per-query statistics (row max,
// 1/sum, D, mask flag) are staged in LDS once per workgroup (reading them per wave straight from global memory
// was measured 12 % slower).
// NQT > 0: number of query tiles known at compile time (fully unrolled); NQT = 0: runtime loop.
template <int DH, int NQT>
__global__ __launch_bounds__(ATT_BWD_THREADS) void attn_bwd_dkv_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    const DropCfg drop_ = drop_live(p.drop);
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, wq = wpb / p.hpb;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = xcd_remap(blockIdx.x, gridDim.x) * p.hpb + wave / wq, b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, nt = Tp >> 4;
    const int Lq_p = round16(p.Lq), nqt = NQT > 0 ? NQT : (Lq_p >> 4);
    const int col0 = h * DH;
    const int hi = wave / wq;               // this wave's head inside the workgroup's head group
    float* s_mx = smem_f + hi * 3 * Lq_p;   // per head: [Lq_p] row max, [Lq_p] 1 / row sum, [Lq_p] rowsum(P dP)
    float* s_inv = s_mx + Lq_p;
    float* s_D = s_inv + Lq_p;
    uint8_t* qm = (uint8_t*)(smem_f + p.hpb * 3 * Lq_p);   // [Lq_p] 1 valid query, 0 masked, 2 pad (same for every head)
    uint8_t* km = qm + Lq_p;                // [Tp]
    {
        const int bh0 = bh - hi;
        for (int i = threadIdx.x; i < p.hpb * Lq_p; i += blockDim.x) {
            const int hh = i / Lq_p, q = i - hh * Lq_p;
            const bool in = q < p.Lq;
            float* base = smem_f + hh * 3 * Lq_p;
            base[q] = in ? p.lse[(size_t)(bh0 + hh) * p.Lq + q] : 0.f;
            base[Lq_p + q] = in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)(bh0 + hh) * p.Lq + q] : 0.f;
            base[2 * Lq_p + q] = in ? p.Dvec[(size_t)(bh0 + hh) * p.Lq + q] : 0.f;
        }
        for (int q = threadIdx.x; q < Lq_p; q += blockDim.x) qm[q] = q < p.Lq ? (p.mq[(size_t)b * p.Lq + q] ? 1 : 0) : 2;
    }
    stage_kmask(km, p.mka, p.mkb, b, p.La, p.Lb, La_p, Lb_p);
    __syncthreads();
    const int jt = blockIdx.y * wq + wave % wq;     // this wave's key tile
    if (jt >= nt) return;
    const bool isa = jt < nta;
    const int jp = 16 * jt + l15;                   // this lane's key (padded index)
    const uint8_t kflag = km[jp];
    KeyBlocks<DH> kbk;
    kbk.init(p, b, col0, l15, g);

    float kf[C::KS], vf[C::KS];
    if (isa) {
        frag_load<DH>(kf, kbk.ka, kbk.row_a, (uint32_t)(16 * jt) * kbk.pitch_a);
        frag_load<DH>(vf, kbk.va, kbk.row_a, (uint32_t)(16 * jt) * kbk.pitch_a);
    } else {
        frag_load<DH>(kf, kbk.kb, kbk.row_b, (uint32_t)(16 * (jt - nta)) * kbk.pitch_b);
        frag_load<DH>(vf, kbk.vb, kbk.row_b, (uint32_t)(16 * (jt - nta)) * kbk.pitch_b);
    }
    // query-side tensors: Q (the projection of this key block) and dO; lane offsets at query tile 0
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(isa ? p.Qa : p.Qb, p.q_bytes), rdo = make_rsrc(p.dO, p.do_bytes);
    const uint32_t pitch_q = (uint32_t)p.ldq * 4u, pitch_do = (uint32_t)p.lddo * 4u;
    const uint32_t qrow_off = ((uint32_t)(b * p.Lq + l15) * (uint32_t)p.ldq + col0 + C::row_off(g)) * 4u;
    const uint32_t dorow_off = ((uint32_t)(b * p.Lq + l15) * (uint32_t)p.lddo + col0 + C::row_off(g)) * 4u;
    const uint32_t qcol_off = ((uint32_t)(b * p.Lq + 4 * g) * (uint32_t)p.ldq + col0 + C::CT * l15) * 4u;
    const uint32_t docol_off = ((uint32_t)(b * p.Lq + 4 * g) * (uint32_t)p.lddo + col0 + C::CT * l15) * 4u;
    const float fscale = p.scale;

    f32x4 dk[C::CT], dv[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) { dk[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto tile = [&](int qt) {
        // row fragments of this query tile (lane&15 = query) for S and dP, column fragments (4 queries 4g+s) for dK, dV;
        // queries past Lq read the rows behind (finite, or zeros past the end) and are masked below
        float qf[C::KS], dof[C::KS];
        frag_load<DH>(qf, rq, qrow_off, (uint32_t)(16 * qt) * pitch_q);
        frag_load<DH>(dof, rdo, dorow_off, (uint32_t)(16 * qt) * pitch_do);
        float qc[4][C::CT], doc[4][C::CT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            col_load<DH>(qc[s], rq, qcol_off, (uint32_t)(16 * qt + s) * pitch_q, l15);
            col_load<DH>(doc[s], rdo, docol_off, (uint32_t)(16 * qt + s) * pitch_do, l15);
        }
        // statistics of this lane's 4 queries (16qt + 4g + r)
        const f32x4 mxq = *(const f32x4*)(s_mx + 16 * qt + 4 * g), invq = *(const f32x4*)(s_inv + 16 * qt + 4 * g);
        const f32x4 Dq = *(const f32x4*)(s_D + 16 * qt + 4 * g);
        const uint32_t qfl = *(const uint32_t*)(qm + 16 * qt + 4 * g);
        // S[query 16qt+4g+r][key jp] and dP likewise (two independent accumulators, interleaved)
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C::KS; ++c) {
            s = MFMA16(qf[c], kf[c], s);
            dp = MFMA16(dof[c], vf[c], dp);
        }
        f32x4 Pv, dSv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = 16 * qt + 4 * g + r;
            const uint32_t qf_ = (qfl >> (8 * r)) & 0xff;
            const bool valid = (qf_ == 1) && (kflag == 1);
            float mult = 1.f;
            if (drop_.p > 0.f && qf_ != 2) mult = drop_mult1(drop_, ((uint64_t)bh * p.Lq + qi) * Tp + jp);
            const float v = logit_xform(s[r], valid, mult, fscale);
            const float pr = (kflag == 2 || qf_ == 2) ? 0.f : fast_exp(v - mxq[r]) * invq[r];
            Pv[r] = pr;
            dSv[r] = valid ? pr * (dp[r] - Dq[r]) * mult * fscale : 0.f;
        }
        // dV^T[c][key] += dO[query][c] P[query][key];  dK^T[c][key] += Q[query][c] dS[query][key]
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) {
                dv[ct] = MFMA16(doc[s4][ct], Pv[s4], dv[ct]);
                dk[ct] = MFMA16(qc[s4][ct], dSv[s4], dk[ct]);
            }
        }
    };
    if (NQT > 0) {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) tile(qt);
    } else {
        for (int qt = 0; qt < nqt; ++qt) tile(qt);
    }
    // store: row = key jp, this lane's 4*CT contiguous columns
    const bool ka = jp < La_p;
    const int jloc = ka ? jp : jp - La_p;
    const bool real = ka ? (jloc < p.La) : (jloc < p.Lb);
    float am = 0.f;
    if (real) {
        float* dKp = ka ? p.dKa + (size_t)(b * p.La + jloc) * p.lddka : p.dKb + (size_t)(b * p.Lb + jloc) * p.lddkb;
        float* dVp = ka ? p.dVa + (size_t)(b * p.La + jloc) * p.lddka : p.dVb + (size_t)(b * p.Lb + jloc) * p.lddkb;
        am = col_store<DH>(dKp + col0, dk, g, am);
        am = col_store<DH>(dVp + col0, dv, g, am);
    }
    float* slot = isa ? p.amax_ka : p.amax_kb;          // wave-uniform: a key tile lies in one block
    if (slot) amax_commit(slot, am, (blockIdx.x * gridDim.y + blockIdx.y) * wpb + wave);
}

// ------------------------------------------------------------------------------------------ backward: fused dQ + dK + dV
// ONE workgroup per (b, h, key block) -- dQa depends on the block-a keys only and dQb on the block-b keys only, so the two
// blocks of a head are independent workgroups and several of them share a CU (one's staging loads run under another's MFMAs:
// with one workgroup per head and CU the prologue, bound by the CU's ~10 B/clk share of HBM, was 28 % of the time).
// The query-side tensors every key tile needs (Q of this block, dO: Lq x DH each) are staged ONCE in LDS
// in whole rows and read from there in both fragment forms, instead of being fetched from global memory in fragment shape
// (16 rows x 64 B per instruction) by every key-tile wave in row AND column form -- 20 fetches of Q and dO per head in the
// two-kernel version, which is bound by the vector-memory pipeline, not by HBM or the matrix cores.  S and dP are computed
// once per (query tile, key tile) pair instead of once in each of two kernels.
//   wave w owns key tile w of the block: its K / V row fragments and K column fragments come straight from global memory
//   (each is needed by this wave only); waves beyond the block's tile count only help staging;
//   per query tile: S = Q K^T, dP = dO V^T (operands from LDS) -> P, dS;  dV += P^T dO, dK += dS^T Q;
//   dS is transposed through a 16 x 16 per-wave LDS scratch, dQ^T = K^T dS^T of the (query tile, key tile) pair is formed in
//   registers and added into an LDS accumulator IN WAVE ORDER: a per-query-tile turn counter in LDS lets wave w add only
//   after wave w-1 has (plain read-modify-write, no float atomics: bitwise reproducible; the waves run the same work in
//   step, so the wait is short and no workgroup barrier is needed); the workgroup then stores dQ in whole rows.
// D = rowsum(dO * O) is formed during the staging (each staged dO chunk is multiplied with its O chunk, the DH/4 partial
// products of a row are summed in a fixed order after the barrier): no separate D launch in front of the backward.
// The query side is processed in chunks of 48 rows (K / V fragments and the dK / dV accumulators stay in registers across the
// chunks), so the LDS footprint does not depend on Lq: 3 x [48][DH + 4] floats + [48][DH/4] partials + statistics + scratch
// (41 KB at DH = 48, 7 waves).
#ifdef SEGMM_ATT_TRACE
__device__ unsigned long long g_att_trace[16 * 8];      // debug build (tools/attn_trace.py): phase stamps of every wave of one workgroup
#define ATT_MARK(ph) do { if (att_trace_on && lane == 0) g_att_trace[wave * 8 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT_MARK(ph) do { } while (0)
#endif
constexpr int ATT_FUSED_MAXW = 12;
constexpr int ATT_FUSED_QCHUNK = 48;              // queries staged at a time by attn_bwd_fused_kernel
// value of lane R of each aligned 4-lane group, in all four lanes of the group (DPP quad_perm broadcast)
template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xf, 0xf, true);
}
// NW = waves per workgroup the launch bound is made for (blockDim.x <= 64 NW); 3 waves per SIMD: up to 6 waves two workgroups share a CU
// (<= 168 registers), so that one head's staging / dQ reduction phases run under the other's MFMA phase.
// ONE: Lq <= 48, a single chunk known at compile time (the chunk loop disappears: 538 us instead of 597 at config 2).
// QCH: rows of the staged query chunk (16 / 32 for single-chunk launches with few queries -- config 3: Lq = 20 and 1 -- so that
// the staging loops, the zero fill and the LDS footprint follow the real row count; 48 otherwise)
template <int DH, int NW, bool ONE, int QCH = ATT_FUSED_QCHUNK>
__global__ __launch_bounds__(64 * NW, 4) void attn_bwd_fused_kernel(const AttnArgs p) {
    using C = AttnCfg<DH>;
    const DropCfg drop_ = drop_live(p.drop);
    constexpr int RS = DH + 4;                 // LDS row stride (floats): 16-byte aligned rows, conflict-free fragment reads
    constexpr int TS = 20;                     // row stride of the 16 x 16 transpose scratch
    constexpr int QC = QCH;                    // queries staged at a time (1 - 3 query tiles)
    constexpr int MAXQT = QC / 16;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // p.hpb: 0 / 1 = this launch handles key block a / b only (exact wave count per launch); 2 = both blocks in one launch,
    // workgroup 2 bh + blk, sized for the larger block -- the surplus waves of the smaller block's workgroups END here, before
    // any barrier (s_barrier waits for surviving waves only), so small and large workgroups mix on a CU
    // p.hpb == 3 ("merged", single-chunk launches of short heads -- config 3: 20 x (20 + 1) and 1 x (1 + 20)): ONE workgroup per
    // (b, h) for BOTH key blocks, waves 0 .. nta-1 on the tiles of block a, the next ntb on block b.  dO, O, the softmax
    // statistics and the flags are staged once instead of once per block (and the one-key block of a config-3 head no longer
    // pays a whole staging pass and launch of its own); each block keeps its own staged Q projection, its own dQ accumulator and
    // its own turn counters, so every sum is formed in the order of the per-block launches: bit-identical results.
    constexpr bool CAN_MERGE = ONE && QCH < ATT_FUSED_QCHUNK;          // (the host merges short single-chunk launches only: compile the
    const bool merged = CAN_MERGE && p.hpb == 3;                       //  48-row instances without the second accumulator / pointer set)
    const int wg = xcd_remap(blockIdx.x, gridDim.x), bh = p.hpb == 2 ? wg >> 1 : wg, b = bh / p.H, h = bh % p.H;
    const int La_p = round16(p.La), Lb_p = round16(p.Lb), Tp = La_p + Lb_p, nta = La_p >> 4, ntb = Lb_p >> 4;
    const bool isa = merged ? wave < nta : (p.hpb == 2 ? (wg & 1) == 0 : p.hpb == 0);
    const int ntk = merged ? nta + ntb : (isa ? nta : ntb);          // working waves of the workgroup
    if (wave >= ntk) return;                               // surplus wave, or empty block (CrossAtt / SelfAtt ablations)
    const int wib = (merged && !isa) ? wave - nta : wave;  // this wave's place among the waves of ITS key block (the dQ turn order)
    const int nthr = 64 * ntk;                             // surviving threads
    const int col0 = h * DH;
    const float* Qg = (isa || merged) ? p.Qa : p.Qb;      // (merged: Qa and Qb are both staged)
    float* dQg = isa ? p.dQa : p.dQb;
    _Float16* dQgp = isa ? p.dQap : p.dQbp;
    float s_q = ((merged ? (p.dQap || p.dQbp) : dQgp != nullptr) && p.sin_q) ? *p.sin_q : 0.f;
    const float* sin_k = isa ? p.sin_ka : p.sin_kb;
    float s_k = ((isa ? p.dKap : p.dKbp) && sin_k) ? *sin_k : 0.f;
    const bool repair = (p.pflags & ATT_REPAIR) != 0;
    const bool want_q = (merged ? (p.dQap || p.dQbp) : dQgp != nullptr) && p.sin_q, want_k = (isa ? p.dKap : p.dKbp) && sin_k;          // sites with plane outputs
    if (repair && merged) {          // (the decision to leave must be the same in every wave of the workgroup: both blocks' sites)
        const bool need_q = want_q && p.hdr_q[2] != 0.f;
        const bool need_ka = p.dKap && p.sin_ka && p.hdr_ka[2] != 0.f, need_kb = p.dKbp && p.sin_kb && p.hdr_kb[2] != 0.f;
        if (!need_q && !need_ka && !need_kb) return;
        s_q = need_q ? p.hdr_q[0] : 0.f;
        s_k = (isa ? need_ka : need_kb) ? (isa ? p.hdr_ka : p.hdr_kb)[0] : 0.f;
    } else if (repair) {
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
#ifdef SEGMM_ATT_TRACE
    const bool att_trace_on = blockIdx.x == gridDim.x / 2 + 5;
#endif
    ATT_MARK(0);
    const int nQ = merged ? 2 : 1;                         // staged Q projections / dQ accumulators
    float* sQ0 = smem_f;                                   // [nQ][QC][RS] query rows of the current chunk
    float* sdO = sQ0 + nQ * QC * RS;
    float* sdQ0 = sdO + QC * RS;
    float* sQ = sQ0 + ((merged && !isa) ? QC * RS : 0);   // this wave's block: its Q rows, its dQ accumulator
    float* sdQ = sdQ0 + ((merged && !isa) ? QC * RS : 0);
    float* s_mx = sdQ0 + nQ * QC * RS;
    float* s_inv = s_mx + QC;
    float* s_D = s_inv + QC;
    float* s_tr = s_D + QC + wave * (16 * TS);                              // this wave's transpose scratch
    int* s_turn0 = (int*)(s_D + QC + nw * (16 * TS));                       // [nQ][4] whose turn it is to add dQ of query tile qt
    int* s_turn = s_turn0 + ((merged && !isa) ? 4 : 0);
    float* s_Dp = (float*)(s_turn0 + 4 * nQ);                               // [QC][DH/4] partial products dO . O
    uint8_t* qm = (uint8_t*)(s_Dp + QC * (DH / 4));                         // [QC] 1 valid query, 0 masked, 2 pad
    uint8_t* km = qm + QC;                                                  // [Tp]
    // ---- this wave's key tile
    const int jt = (isa ? 0 : nta) + wib;                                   // padded key tile of this wave
    KeyBlocks<DH> kbk;
    kbk.init(p, b, col0, l15, g);
    float kf[C::KS], vf[C::KS], kc[4][C::CT];
    auto load_frags = [&]() {     // K / V row fragments and K column fragments of this wave's tile (global, fragment form)
        if (isa) {
            const uint32_t so = (uint32_t)(16 * jt) * kbk.pitch_a;
            frag_load<DH>(kf, kbk.ka, kbk.row_a, so);
            frag_load<DH>(vf, kbk.va, kbk.row_a, so);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) col_load<DH>(kc[s4], kbk.ka, kbk.col_a, (uint32_t)(16 * jt + s4) * kbk.pitch_a, l15);
        } else {
            const uint32_t so = (uint32_t)(16 * (jt - nta)) * kbk.pitch_b;
            frag_load<DH>(kf, kbk.kb, kbk.row_b, so);
            frag_load<DH>(vf, kbk.vb, kbk.row_b, so);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) col_load<DH>(kc[s4], kbk.kb, kbk.col_b, (uint32_t)(16 * (jt - nta) + s4) * kbk.pitch_b, l15);
        }
    };
    // single chunk (Lq <= 48): requested AFTER the staging loads (loads return in order: the first barrier then waits for the
    // staging data only; 542 -> 518 us); several chunks: requested first, their latency hides under the first chunk's staging
    // (the other order costs 18 % at Lq = 100).  (Round 4: two staging items per thread in flight + the fragments right behind
    // them, for the short chunks of config 3: no gain at Lq = 20, 86 -> 97 us at Lq = 1 -- removed.)
    if (!ONE) load_frags();
    for (int j = threadIdx.x; j < Tp; j += nthr) {         // key flags (stage_kmask with the surviving thread count)
        uint8_t v;
        if (j < La_p) v = (j < p.La) ? (p.mka[(size_t)b * p.La + j] ? 1 : 0) : 2;
        else { const int jb = j - La_p; v = (jb < p.Lb) ? (p.mkb[(size_t)b * p.Lb + jb] ? 1 : 0) : 2; }
        km[j] = v;
    }
    const float fscale = p.scale;
    const int jp = 16 * jt + l15;                          // this lane's key (padded index)
    f32x4 dk[C::CT], dv[C::CT];
#pragma unroll
    for (int ct = 0; ct < C::CT; ++ct) { dk[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float am_q = 0.f;

    // ---- the query side in chunks of QC rows: 41 KB of LDS whatever Lq is (user queries: Lq = 100 -> 3 chunks)
    for (int q0 = 0; ONE ? q0 < 1 : q0 < p.Lq; q0 += QC) {
        const int nq = min(QC, p.Lq - q0);                 // real queries of the chunk
        const int nqt = (nq + 15) >> 4;
        // stage whole rows (float4), rows >= nq zero; zero the dQ accumulator; partial products of D = rowsum(dO * O)
        for (int i = threadIdx.x; i < QC * (DH / 4); i += nthr) {
            const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vo = va, oo = va, vb = va;
            if (q < nq) {
                const size_t row = (size_t)b * p.Lq + q0 + q;
                va = *(const f32x4*)(Qg + row * p.ldq + col0 + c);
                if (merged) vb = *(const f32x4*)(p.Qb + row * p.ldq + col0 + c);
                vo = *(const f32x4*)(p.dO + row * p.lddo + col0 + c);
                oo = *(const f32x4*)(p.O + row * p.ldo + col0 + c);
            }
            *(f32x4*)((merged ? sQ0 : sQ) + q * RS + c) = va;
            *(f32x4*)(sdO + q * RS + c) = vo;
            *(f32x4*)((merged ? sdQ0 : sdQ) + q * RS + c) = f32x4{0.f, 0.f, 0.f, 0.f};
            if (merged) {
                *(f32x4*)(sQ0 + (QC + q) * RS + c) = vb;
                *(f32x4*)(sdQ0 + (QC + q) * RS + c) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            s_Dp[i] = (vo.x * oo.x + vo.y * oo.y) + (vo.z * oo.z + vo.w * oo.w);
        }
        for (int q = threadIdx.x; q < QC; q += nthr) {
            const bool in = q < nq;
            s_mx[q] = in ? p.lse[(size_t)bh * p.Lq + q0 + q] : 0.f;
            s_inv[q] = in ? p.lse[(size_t)p.B * p.H * p.Lq + (size_t)bh * p.Lq + q0 + q] : 0.f;
            qm[q] = in ? (p.mq[(size_t)b * p.Lq + q0 + q] ? 1 : 0) : 2;
        }
        if (threadIdx.x < 4 * nQ) s_turn0[threadIdx.x] = 0;
        if (ONE) load_frags();
        ATT_MARK(1);
        __syncthreads();
        for (int q = threadIdx.x; q < QC; q += nthr) {     // D[q]: the DH/4 partials of the row in index order (deterministic)
            float d_ = 0.f;
#pragma unroll
            for (int j = 0; j < DH / 4; ++j) d_ += s_Dp[q * (DH / 4) + j];
            s_D[q] = d_;
        }
        __syncthreads();
        ATT_MARK(2);
        const uint8_t kflag = km[jp];
#pragma unroll
        for (int qt = 0; qt < MAXQT; ++qt) {
            if (qt < nqt) {
                // row fragments (lane&15 = query): element k = 16 i + 4 g + e of the row, like frag_load
                float qf[C::KS], dof[C::KS];
                frag_load_ptr<DH>(qf, sQ + (16 * qt + l15) * RS + C::row_off(g));
                frag_load_ptr<DH>(dof, sdO + (16 * qt + l15) * RS + C::row_off(g));
                f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C::KS; ++c) {
                    sv = MFMA16(qf[c], kf[c], sv);
                    dp = MFMA16(dof[c], vf[c], dp);
                }
                const f32x4 mxq = *(const f32x4*)(s_mx + 16 * qt + 4 * g), invq = *(const f32x4*)(s_inv + 16 * qt + 4 * g);
                const f32x4 Dq = *(const f32x4*)(s_D + 16 * qt + 4 * g);
                const uint32_t qfl = *(const uint32_t*)(qm + 16 * qt + 4 * g);
                f32x4 Pv, dSv;
                // dropout multipliers of this lane's 4 (query, key) elements.  The random bits come in quads of 4 consecutive
                // keys of one query (drop_rand_quad): the 4 lanes of an aligned group hold the 4 keys of such quads, so each
                // lane hashes ONE query's quad (query 4g + (lane & 3)) and the group exchanges the words by DPP broadcasts --
                // one hash per lane instead of four, bit-identical to drop_mult1.
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
                    const float v = logit_xform(sv[r], valid, mult, fscale);
                    const float pr = (kflag == 2 || qf_ == 2) ? 0.f : fast_exp(v - mxq[r]) * invq[r];
                    Pv[r] = pr;
                    dSv[r] = valid ? pr * (dp[r] - Dq[r]) * mult * fscale : 0.f;
                }
                // column fragments from LDS: lane (c, g) takes rows 16 qt + 4 g + s4, head columns 16 ct + c
                // (=> result register r of tile ct is head column 16 ct + 4 g + r: one float4 per tile)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float* qr = sQ + (16 * qt + 4 * g + s4) * RS;
                    const float* dr = sdO + (16 * qt + 4 * g + s4) * RS;
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) {
                        const int cc = 16 * ct + l15;
                        const float qv = cc < DH ? qr[cc] : 0.f, dv_ = cc < DH ? dr[cc] : 0.f;
                        dv[ct] = MFMA16(dv_, Pv[s4], dv[ct]);
                        dk[ct] = MFMA16(qv, dSv[s4], dk[ct]);
                    }
                }
                // dS[query 4g+r][key l15] -> dS^T fragments (lane&15 = query, registers = keys 4g..4g+3) through the scratch
#pragma unroll
                for (int r = 0; r < 4; ++r) s_tr[(4 * g + r) * TS + l15] = dSv[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's own LDS writes have landed
                __builtin_amdgcn_wave_barrier();
                const f32x4 dST = *(const f32x4*)(s_tr + l15 * TS + 4 * g);
                __builtin_amdgcn_wave_barrier();
                f32x4 dqt[C::CT];
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct) dqt[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int ct = 0; ct < C::CT; ++ct) dqt[ct] = MFMA16(kc[s4][ct], dST[s4], dqt[ct]);
                // ordered accumulation: lane (query l15, g) holds head columns CT*(4g + r) + ct (col_load mapping of kc),
                // i.e. the 4*CT contiguous columns from 4*CT*g of row 16 qt + l15
                if (wib > 0)
                    while (__hip_atomic_load(s_turn + qt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != wib) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (4 * C::CT * g < DH) {
                    float* row = sdQ + (16 * qt + l15) * RS + 4 * C::CT * g;
                    float t[4 * C::CT];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ct = 0; ct < C::CT; ++ct) t[C::CT * r + ct] = dqt[ct][r];
#pragma unroll
                    for (int i = 0; i < C::CT; ++i) {
                        f32x4 a = *(f32x4*)(row + 4 * i);
                        a += f32x4{t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
                        *(f32x4*)(row + 4 * i) = a;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(s_turn + qt, wib + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        ATT_MARK(3);
        __syncthreads();                                   // every wave has added its dQ partials of this chunk
        ATT_MARK(4);
        for (int blk = 0; blk < nQ; ++blk) {              // (merged: dQa from the first accumulator, dQb from the second)
            const float* acc_ = merged ? sdQ0 + blk * QC * RS : sdQ;
            float* dq_ = merged ? (blk == 0 ? p.dQa : p.dQb) : dQg;
            _Float16* dqp_ = merged ? (blk == 0 ? p.dQap : p.dQbp) : dQgp;
            for (int i = threadIdx.x; i < nq * (DH / 4); i += nthr) {
                const int q = i / (DH / 4), c = (i - q * (DH / 4)) * 4;
                const size_t row = (size_t)b * p.Lq + q0 + q;
                const f32x4 v = *(const f32x4*)(acc_ + q * RS + c);
                if (f32_q) *(f32x4*)(dq_ + row * p.lddq + col0 + c) = v;
                if (s_q > 0.f && dqp_) {          // adjacent threads hold adjacent float4 groups of one row (DH / 4 even, col0 % 8 == 0)
                    if ((DH & 7) == 0 && (col0 & 7) == 0) plane_store4_pair(dqp_, p.lddq2, (long long)row, col0 + c, v, s_q);
                    else plane_store4(dqp_, p.lddq2, (long long)row, col0 + c, v, s_q);
                }
                am_q = absmax4(am_q, v);
            }
        }
        if (!ONE && q0 + QC < p.Lq) __syncthreads();       // the next chunk's staging overwrites what was just read
    }
    // dK / dV rows of this tile: lane (key l15, g), tile ct register r = head column 16 ct + 4 g + r
    {
        const bool ka = jp < La_p;
        const int jloc = ka ? jp : jp - La_p;
        const bool real = ka ? (jloc < p.La) : (jloc < p.Lb);
        float am = 0.f;
        if (real) {
            float* dKp = (ka ? p.dKa + (size_t)(b * p.La + jloc) * p.lddka : p.dKb + (size_t)(b * p.Lb + jloc) * p.lddkb) + col0;
            float* dVp = (ka ? p.dVa + (size_t)(b * p.La + jloc) * p.lddka : p.dVb + (size_t)(b * p.Lb + jloc) * p.lddkb) + col0;
            const long long krow = ka ? (long long)b * p.La + jloc : (long long)b * p.Lb + jloc;
            _Float16* dKpp = ka ? p.dKap : p.dKbp;
            _Float16* dVpp = ka ? p.dVap : p.dVbp;
            const int ldk2 = ka ? p.lddka2 : p.lddkb2;
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) {
                if (16 * ct + 4 * g < DH) {
                    if (f32_k) {
                        *(f32x4*)(dKp + 16 * ct + 4 * g) = dk[ct];
                        *(f32x4*)(dVp + 16 * ct + 4 * g) = dv[ct];
                    }
                    if (s_k > 0.f) {          // lane (key, g) and lane (key, g ^ 1) hold the two halves of an aligned 8
                        if (DH % 16 == 0 && (col0 & 7) == 0) {
                            plane_store4_x16(dKpp, ldk2, krow, col0 + 16 * ct + 4 * g, split4(dk[ct], s_k));
                            plane_store4_x16(dVpp, ldk2, krow, col0 + 16 * ct + 4 * g, split4(dv[ct], s_k));
                        } else {
                            plane_store4(dKpp, ldk2, krow, col0 + 16 * ct + 4 * g, dk[ct], s_k);
                            plane_store4(dVpp, ldk2, krow, col0 + 16 * ct + 4 * g, dv[ct], s_k);
                        }
                    }
                    am = absmax4(absmax4(am, dk[ct]), dv[ct]);
                }
            }
        }
        float* hk = isa ? p.hdr_ka : p.hdr_kb;
        float* slot = isa ? p.amax_ka : p.amax_kb;
        // the scale the planes were written with is recorded by ONE deterministic surviving wave per header: wave 0 of the
        // workgroup(s) of (b, h) = (0, 0) -- the 'a' workgroup for hdr_ka, the 'b' workgroup for hdr_kb, both for hdr_q.  (A
        // "key % 1024 == 0" rule left hdr_kb unwritten on small grids: consumers then saw s == 0 and took the fp32 path.)
        const bool hdr_writer = bh == 0 && wib == 0 && lane == 0;          // (merged: the first wave of each block for its key header)
        if (!repair) {          // (the repair pass leaves the headers as they are: every workgroup of it must read the same ones)
            if (s_k > 0.f) { site_commit(hk, am, blockIdx.x * nw + wave, s_k); if (hdr_writer) hk[0] = s_k; }
            else if (slot) amax_commit(slot, am, blockIdx.x * nw + wave);
            if (s_q > 0.f) { site_commit(p.hdr_q, am_q, blockIdx.x * nw + wave, s_q); if (hdr_writer) p.hdr_q[0] = s_q; }
            else if (p.amax_q) amax_commit(p.amax_q, am_q, blockIdx.x * nw + wave);
        }
    }
    ATT_MARK(5);
}

}  // namespace segmm
