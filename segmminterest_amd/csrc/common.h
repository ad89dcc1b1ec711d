// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the segment-interest path.
#pragma once
#ifndef SEGMM_NT_STORES
#define SEGMM_NT_STORES 0          // 1: streaming (nontemporal) stores for the big write-once outputs
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEGMM_WAVE 64

// ---------------------------------------------------------------- error plumbing (host)
extern thread_local char g_segmm_err[512];
int segmm_fail(int code, const char* fmt, ...);
#define SEGMM_CHECK_HIP(expr)                                                         \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return segmm_fail(-100 - (int)_e, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define SEGMM_REQUIRE(cond, ...)                         \
    do {                                                 \
        if (!(cond)) return segmm_fail(-1, __VA_ARGS__); \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- streaming row traffic.  The row kernels read and write every byte once; with the default cache policy those bytes pass through
// the XCD's L2 like anything else and push out the operand panels of the GEMMs that run beside them on the other stream.
// SEGMM_ROW_NT: 0 default policy, 1 nontemporal loads / stores for the LayerNorm kernels' fp32 rows, 2 also for the L1 normalisation,
// the column sums and the split-K combine (plane outputs stay cacheable: the next GEMM reads them at once)
#ifndef SEGMM_ROW_NT
#define SEGMM_ROW_NT 2
#endif
#ifndef SEGMM_ATT_AUX
#define SEGMM_ATT_AUX 0          // probe: cache policy of the planes-in attention kernels' operand loads (2 nt)
#endif
#ifndef SEGMM_E_AUX
#define SEGMM_E_AUX 0          // probe: cache policy of the GEMM epilogue's extra-operand loads (2 nt)
#endif
__device__ __forceinline__ f32x4 ld_row4(const float* p) {
#if SEGMM_ROW_NT >= 1
    return __builtin_nontemporal_load((const f32x4*)p);
#else
    return *(const f32x4*)p;
#endif
}
__device__ __forceinline__ void st_row4(float* p, f32x4 v) {
#if SEGMM_ROW_NT >= 1
    __builtin_nontemporal_store(v, (f32x4*)p);
#else
    *(f32x4*)p = v;
#endif
}
__device__ __forceinline__ f32x4 ld_row4b(const float* p) {
#if SEGMM_ROW_NT >= 2
    return __builtin_nontemporal_load((const f32x4*)p);
#else
    return *(const f32x4*)p;
#endif
}
__device__ __forceinline__ void st_row4b(float* p, f32x4 v) {
#if SEGMM_ROW_NT >= 2
    __builtin_nontemporal_store(v, (f32x4*)p);
#else
    *(f32x4*)p = v;
#endif
}

// ---------------------------------------------------------------- counter-based dropout stream
// Dropout masks are a pure function of (seed, site, element index), so the backward kernels
// regenerate them instead of storing them.  The generator is a stateless integer hash (two chained
// rounds of a 32-bit avalanche mix, ~14 VALU ops) that yields 4 x 16 random bits for the 4 consecutive
// elements of one "quad"; the first version used Philox4x32-10 (~120 ops per quad), which made the
// attention kernels VALU-bound (rocprof: 23 VALU instructions per MFMA).  16 bits per element
// quantise the drop probability to 1/65536 (0.1 -> 6554/65536 = 0.100006); the rescale uses the
// quantised value, so E[mask] = 1 exactly.
struct DropCfg {
    float p;              // drop probability (0 => disabled)
    float scale;          // 1/(1-p_quantised)
    uint32_t thresh;      // keep iff r16 >= thresh, thresh = round(p * 65536)
    uint32_t seed_lo, seed_hi;
    uint32_t site;        // distinct per dropout site in the model
    uint32_t live;        // != 0: the seed is XORed with the DEVICE-side step words (*st) when the kernel starts
    const struct StepState* st;          // the step state bound when the launch was made (segmm_step_bind); null unless live
};

// ---- device-side step state.  A training step captured in a hipGraph replays the SAME kernel arguments every step, so what
// must change from step to step lives in device memory and is advanced by a one-thread kernel at the head of the step
// (segmm_step_advance): two seed words for the dropout streams (kernels launched with a "live" seed -- bit 63 of the seed
// argument -- XOR them into their seed) and AdamW's bias corrections (segmm_adamw with step < 0 reads them).
// The state is a small struct in CALLER-OWNED device memory (segmm_step_state_bytes; segmm_step_bind names the one the following
// launches use -- two trainers in one process each bind their own before they step); with nothing bound the library falls back to
// one default state of its own.  The pointer travels inside the launch's arguments: no __device__ global.
struct StepState { uint32_t seed_lo, seed_hi; int step; float bc1, bc2_sqrt; uint32_t pad_[3]; };
extern StepState* g_segmm_step;          // host: the bound state (capi.hip)
StepState* segmm_step_current();         // host: the bound state, or the library's default one (allocated on first use)
__device__ __forceinline__ DropCfg drop_live(DropCfg d) {
    if (d.live) { d.seed_lo ^= d.st->seed_lo; d.seed_hi ^= d.st->seed_hi; }
    return d;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// 64 random bits for the 4 elements [4*q, 4*q+3] of a site (q = element_index / 4)
__device__ __forceinline__ uint2 drop_rand_quad(const DropCfg& d, uint64_t q) {
    const uint32_t k = d.seed_lo ^ (d.site * 0x9E3779B9u) ^ ((uint32_t)(q >> 32) * 0x85EBCA6Bu);
    const uint32_t a = mix32((uint32_t)q ^ k);
    const uint32_t b = mix32(a ^ d.seed_hi ^ 0x68E31DA4u);
    return make_uint2(a, b);
}
__device__ __forceinline__ f32x4 drop_apply4(const DropCfg& d, uint64_t q, f32x4 v) {
    const uint2 r = drop_rand_quad(d, q);
    v.x = ((r.x & 0xffffu) >= d.thresh) ? v.x * d.scale : 0.f;
    v.y = ((r.x >> 16) >= d.thresh) ? v.y * d.scale : 0.f;
    v.z = ((r.y & 0xffffu) >= d.thresh) ? v.z * d.scale : 0.f;
    v.w = ((r.y >> 16) >= d.thresh) ? v.w * d.scale : 0.f;
    return v;
}
// multiplier (0 or scale) of one element
__device__ __forceinline__ float drop_mult1(const DropCfg& d, uint64_t elem) {
    const uint2 r = drop_rand_quad(d, elem >> 2);
    const uint32_t lane = (uint32_t)(elem & 3);
    const uint32_t w = lane < 2 ? r.x : r.y;
    const uint32_t x = (lane & 1) ? (w >> 16) : (w & 0xffffu);
    return (x >= d.thresh) ? d.scale : 0.f;
}

static inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
    DropCfg d;
    d.p = p;
    double t = (double)p * 65536.0 + 0.5;
    d.thresh = p <= 0.f ? 0u : (t >= 65535.0 ? 65535u : (uint32_t)t);
    d.scale = p > 0.f ? (float)(65536.0 / (65536.0 - (double)d.thresh)) : 1.0f;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32) & 0x7fffffffu;
    d.site = site;
    d.live = (uint32_t)(seed >> 63);
    d.st = d.live ? segmm_step_current() : nullptr;
    return d;
}

// ---------------------------------------------------------------- math
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---------------------------------------------------------------- wave64 reductions (shuffle, no LDS)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// max over the wave, the same value in every lane.  Not the xor butterfly of wave_sum: six ds_bpermute round trips (the LDS
// crossbar, ~100 cycles each and serial) are what the attention kernels' per-tile scale derivation waited on.  Inside a row
// of 16 lanes the partners come through DPP (lane xor 1, xor 2, then the mirrored half-row and row: every lane of a row ends
// with the row's maximum), the four rows meet through v_readlane.  max is exact and order-independent, so nothing changes
// numerically.  Needs every lane of the wave active (like the shuffle form).
template <int CTRL>
__device__ __forceinline__ float dpp_row_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_row_move<0xB1>(v));           // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_row_move<0x4E>(v));           // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_row_move<0x141>(v));          // row_half_mirror
    v = fmaxf(v, dpp_row_move<0x140>(v));          // row_mirror
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// ---- running max|x| of a tensor, for the fp16x3 GEMM engine's per-tensor scale.  A producer kernel keeps a
// per-lane running maximum of what it stores and folds it, once per wave, into one of AMAX_SLOTS partial maxima
// with an INTEGER atomic max on the float bits (non-negative floats order like unsigned ints: exact, and
// independent of the order of arrival).  The slot array is zeroed by the host before the producer runs; the
// consuming GEMM reduces the slots itself.
constexpr int AMAX_SLOTS = 256;
__device__ __forceinline__ float absmax4(float m, f32x4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void amax_commit(float* slots, float m, unsigned key) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)slots + (key & (AMAX_SLOTS - 1)), __float_as_uint(m));
}
// ---- "site header" of a plane tensor (gemm_planes.h): SITE_HDR floats followed by the AMAX_SLOTS partial maxima.
// hdr[0] = power-of-two scale the fp16 (hi, lo) planes were written with (0: no planes written), hdr[1] (as an
// integer) != 0: some element left the fp16 range under that scale -> consumers take the fp32 copy instead.
constexpr int SITE_HDR = 8;
constexpr int SITE_FLOATS = SITE_HDR + AMAX_SLOTS;
__device__ __forceinline__ void site_commit(float* hdr, float m, unsigned key, float s_used) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) {
        atomicMax((unsigned int*)hdr + SITE_HDR + (key & (AMAX_SLOTS - 1)), __float_as_uint(m));
        if (s_used > 0.f && !(m * s_used < 65504.f)) ((volatile unsigned int*)hdr)[1] = 1u;      // NaN raises it too
    }
}
// power of two s with amax * s in [2^14, 2^15); exponent clamped to +-60 so that 1/(sa sb) stays finite
__device__ __forceinline__ float f16_scale_of(float amax) {
    const uint32_t u = __float_as_uint(amax);
    if (!(amax > 0.f) || (u >> 23) == 0xff) return 1.f;
    int se = 14 - ((int)(u >> 23) - 127);
    se = max(-60, min(60, se));
    return __uint_as_float((uint32_t)(se + 127) << 23);
}
// largest |x| the producers of a site recorded (AMAX_SLOTS partial maxima, four per lane): the same value in every wave
__device__ __forceinline__ float site_amax(const float* hdr, int lane) {
    static_assert(AMAX_SLOTS == 256, "one float4 per lane");
    const f32x4 v = *(const f32x4*)(hdr + SITE_HDR + lane * 4);
    return wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
}
// Planes written with the (delayed) scale s are usable iff the tensor's maximum sits inside the fp16 window: below 65504 -- the
// producers' overflow flag says the same -- AND not so far below it that the lo terms sink into the subnormals: a tensor that
// SHRANK by more than ~2^9 since its scale was derived (the gradients of a batch whose loss has collapsed) would silently keep
// 12 bits instead of 22.  max * s >= 2^-2 keeps every element's absolute error below 2^-22 max.  Checked by the CONSUMER, which
// sees the complete maxima (one 1 KB read per wave); block-uniform.
__device__ __forceinline__ bool site_planes_ok(const float* hdr, float s, int lane) {
    if (!(s > 0.f) || __float_as_uint(hdr[1]) != 0u) return false;
    const float m = site_amax(hdr, lane);
    // (s at its upper clamp 2^60 -- f16_scale_of / scales_update keep 1/(sa sb) finite: the fallback would use the same scale)
    return !(m > 0.f) || ((m * s >= 0.25f || s >= 0x1p60f) && m * s < 65504.f);
}
// (x0, x1) * s -> packed fp16 (hi0, hi1), (lo0, lo1); hi + lo = x s up to 2^-22 |x s|.
// Four VALU instructions per pair: v_fma_mix{lo,hi}_f16 multiply in fp32, round ONCE to fp16 and write one half
// of the destination, and take the fp16 hi term straight back as the addend of the lo term
// (lo = rn16(x s - hi), the fma is exact before that rounding).
__device__ __forceinline__ void splith_pair(float x0, float x1, float s, uint32_t& ph, uint32_t& pl) {
    uint32_t h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    ph = h;
    pl = l;
}

// ---- producer side of a plane tensor.  A kernel that writes a GEMM operand also writes its P32 fp16 planes (gemm_planes.h:
// per row and per 32 columns [32 hi | 32 lo]) with the DELAYED scale *scale_in (a power of two derived from the maxima this
// tensor site had on earlier steps; null or 0: no planes yet -> the host runs an exact split pass instead), records that
// scale in hdr[0] and folds the partial maxima / the overflow flag into the header.  Consumers fall back to the fp32 copy
// when the flag is up, so a scale that has become too large costs time, never correctness.
struct PlaneOut {
    _Float16* p; int ld2;         // plane view (null: no plane output)
    float* hdr;                   // site header of the tensor
    const float* scale_in;        // device scalar: the scale to write with
};
__device__ __forceinline__ float plane_scale(const PlaneOut& po) {
    return (po.p && po.scale_in) ? *po.scale_in : 0.f;
}
// 4 consecutive columns c .. c+3 (c % 4 == 0) of row `row`
__device__ __forceinline__ void plane_store4(_Float16* p, int ld2, long long row, int c, f32x4 v, float s) {
    uint32_t h0, l0, h1, l1;
    splith_pair(v.x, v.y, s, h0, l0);
    splith_pair(v.z, v.w, s, h1, l1);
    _Float16* o = p + row * ld2 + ((c >> 5) << 6) + (c & 31);
    *(uint2*)o = make_uint2(h0, h1);
    *(uint2*)(o + 32) = make_uint2(l0, l1);
}
// The same for kernels in which ADJACENT LANES hold ADJACENT float4 column groups of one row (lane l: columns c, lane l ^ 1:
// columns c ^ 4; both lanes of a pair active): the pair trades half of its terms through DPP (quad_perm 1,0,3,2), the even
// lane stores the 8 hi terms of columns [c & ~7, +8) and the odd lane the 8 lo terms -- ONE 16-byte store per lane, eight
// adjacent lanes write one whole 128-byte [32 hi | 32 lo] block.  (8-byte stores issued twice per lane ran the fused
// producers at a third of the HBM rate: 1.4 TB/s for the plane bytes.)
__device__ __forceinline__ uint32_t dpp_swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void plane_store4_pair(_Float16* p, int ld2, long long row, int c, f32x4 v, float s) {
    uint32_t h0, l0, h1, l1;
    splith_pair(v.x, v.y, s, h0, l0);
    splith_pair(v.z, v.w, s, h1, l1);
    const bool odd = (c & 4) != 0;
    const uint32_t r0 = dpp_swap1(odd ? h0 : l0), r1 = dpp_swap1(odd ? h1 : l1);      // odd lanes give away hi, even lanes lo
    const int cb = c & ~7;
    _Float16* o = p + row * ld2 + ((cb >> 5) << 6) + (cb & 31) + (odd ? 32 : 0);
    const uint4 w = odd ? make_uint4(r0, r1, l0, l1) : make_uint4(h0, h1, r0, r1);
#if SEGMM_NT_STORES
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u32x4{w.x, w.y, w.z, w.w}, (u32x4*)o);          // written once, read by a later kernel after GBs of other traffic
#else
    *(uint4*)o = w;
#endif
}
// end of a producer wave: partial maxima (+ flag) and the scale used.  The scale is stored by the waves whose key is a
// multiple of 1024 (key 0 always exists and always gets here; EVERY wave storing to that one word serialised the stores of
// a 51 200-row kernel at the memory side and more than doubled its run time).
__device__ __forceinline__ bool scale_writer(unsigned key) { return (threadIdx.x & 63) == 0 && (key & 1023u) == 0u; }
__device__ __forceinline__ void plane_finish(const PlaneOut& po, float* amax_slots, float am, unsigned key, float s, bool) {
    if (s > 0.f) {
        site_commit(po.hdr, am, key, s);
        if (scale_writer(key)) po.hdr[0] = s;
    } else if (amax_slots) {
        amax_commit(amax_slots, am, key);
    }
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---- raw buffer addressing: one resource descriptor (4 SGPRs) per tensor, a 32-bit per-lane byte offset and a
// scalar byte offset per access -- no 64-bit per-lane address arithmetic, and reads beyond [p, p + bytes) return 0
// (hardware range check) instead of faulting.  Tensors addressed this way must be smaller than 4 GiB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// XCD-aware bijective remap of a linear workgroup id (hardware deals consecutive ids round-robin over
// the 8 XCDs): gives every XCD a contiguous chunk of logical ids so tiles that share an operand panel
// hit the same L2.  Speed only; any mapping is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
