// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the segment-interest path.
#pragma once
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

// ---------------------------------------------------------------- Philox4x32-10 counter RNG
// Dropout masks are a pure function of (seed, site, element index), so the backward kernels
// regenerate them instead of storing them.  One call yields 4 x 32 bits = 4 consecutive elements.
struct DropCfg {
    float p;              // drop probability (0 => disabled)
    float scale;          // 1/(1-p)
    uint32_t thresh;      // keep iff r >= thresh, thresh = p * 2^32
    uint32_t seed_lo, seed_hi;
    uint32_t site;        // distinct per dropout site in the model
};

__device__ __forceinline__ uint4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// keep-mask bits for the 4 elements [4*q, 4*q+3] of a site (q = element_index / 4)
__device__ __forceinline__ uint4 drop_rand4(const DropCfg& d, uint64_t q) {
    return philox4x32_10(d.seed_lo, d.seed_hi, (uint32_t)q, (uint32_t)(q >> 32), d.site, 0x5e6d3u);
}
__device__ __forceinline__ f32x4 drop_apply4(const DropCfg& d, uint64_t q, f32x4 v) {
    const uint4 r = drop_rand4(d, q);
    v.x = (r.x >= d.thresh) ? v.x * d.scale : 0.f;
    v.y = (r.y >= d.thresh) ? v.y * d.scale : 0.f;
    v.z = (r.z >= d.thresh) ? v.z * d.scale : 0.f;
    v.w = (r.w >= d.thresh) ? v.w * d.scale : 0.f;
    return v;
}
// multiplier (0 or scale) of one element
__device__ __forceinline__ float drop_mult1(const DropCfg& d, uint64_t elem) {
    const uint4 r = drop_rand4(d, elem >> 2);
    const uint32_t lane = (uint32_t)(elem & 3);
    const uint32_t x = lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w;
    return (x >= d.thresh) ? d.scale : 0.f;
}

static inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
    DropCfg d;
    d.p = p;
    d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    double t = (double)p * 4294967296.0;
    d.thresh = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t);
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32);
    d.site = site;
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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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

// XCD-aware bijective remap of a linear workgroup id (hardware deals consecutive ids round-robin over
// the 8 XCDs): gives every XCD a contiguous chunk of logical ids so tiles that share an operand panel
// hit the same L2.  Speed only; any mapping is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
