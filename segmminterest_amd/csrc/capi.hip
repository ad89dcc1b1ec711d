// C ABI of libsegmm_hip.so (see include/segmm_hip.h).  Host-side launch logic only: shape checks,
// grid selection, error reporting.  Nothing here allocates, synchronises or keeps state.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/segmm_hip.h"
#include "attention16.h"
#include "attention_pl.h"
#include "common.h"
#include "evalops.h"
#include "gemm.h"
#include "gemm_split.h"
#include "gemm_planes.h"
#include "gemm_planes8.h"
#include "gemm_planes4.h"
#include "loss.h"
#include "rowops.h"

// compute units of the current device (cached; persistent kernels launch one workgroup per CU)
static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
#ifdef SEGMM_STAMPS
static unsigned long long* g_segmm_stamps = nullptr;
extern "C" int segmm_debug_set_stamps(void* p) { g_segmm_stamps = (unsigned long long*)p; return 0; }
#endif

thread_local char g_segmm_err[512] = {0};

int segmm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_segmm_err, sizeof(g_segmm_err), fmt, ap);
    va_end(ap);
    return code;
}

#define LAUNCH_CHECK() SEGMM_CHECK_HIP(hipGetLastError())

// ---- device-side step state (common.h StepState): caller-owned, named by segmm_step_bind; the library's own default state
// serves callers that never bind one (a single trainer per process)
StepState* g_segmm_step = nullptr;
static StepState* g_step_default = nullptr;
StepState* segmm_step_current() {
    if (g_segmm_step) return g_segmm_step;
    if (!g_step_default) {
        if (hipMalloc((void**)&g_step_default, sizeof(StepState)) != hipSuccess) return nullptr;
        (void)hipMemset(g_step_default, 0, sizeof(StepState));
    }
    return g_step_default;
}

using namespace segmm;

static int attn_fill(AttnArgs& a, int B, int H, int dh, int Lq, int La, int Lb, const float* Qa, const float* Qb, int ldq,
                     const float* Ka, const float* Va, int ldka, const float* Kb, const float* Vb, int ldkb,
                     const uint8_t* mq, const uint8_t* mka, const uint8_t* mkb, float drop_p, uint64_t seed,
                     uint32_t site, bool planes_in = false) {
    SEGMM_REQUIRE(B > 0 && H > 0 && Lq > 0 && La >= 0 && Lb >= 0 && La + Lb > 0, "attn: empty dimension");
    // One key block may be EMPTY (La == 0 or Lb == 0): the CrossAtt / SelfAtt ablations of the reference attend to one
    // block only (encoder.py:108-135).  Its pointers may then be null; reads are aliased to the other block's tensors
    // (never dereferenced for a key tile, the query fragment of the empty block is loaded but unused).
    if (La == 0) { Qa = Qb; Ka = Kb; Va = Vb; ldka = ldkb; mka = mkb; }
    if (Lb == 0) { Qb = Qa; Kb = Ka; Vb = Va; ldkb = ldka; mkb = mka; }
    // (with input planes the fp32 views of Q / K / V are optional: a caller whose projection GEMMs write planes only has none)
    SEGMM_REQUIRE(((Qa && Qb && Ka && Va && Kb && Vb) || (planes_in && !Qa && !Qb && !Ka && !Va && !Kb && !Vb)) && mq && mka && mkb, "attn: null pointer");
    SEGMM_REQUIRE(dh == 4 || dh == 8 || dh == 16 || dh == 32 || dh == 48 || dh == 64, "attn: head dim %d not built (4,8,16,32,48,64)", dh);
    SEGMM_REQUIRE(ldq % 4 == 0 && ldka % 4 == 0 && ldkb % 4 == 0, "attn: leading dims %% 4");
    SEGMM_REQUIRE(aligned16(Qa) && aligned16(Qb) && aligned16(Ka) && aligned16(Va) && aligned16(Kb) && aligned16(Vb), "attn: alignment");
    const int Tp = ((La + 15) & ~15) + ((Lb + 15) & ~15);
    SEGMM_REQUIRE(Tp <= 16 * 12, "attn: %d padded keys > 192 not built", Tp);
    a.B = B; a.H = H; a.Lq = Lq; a.La = La; a.Lb = Lb;
    a.Qa = Qa; a.Qb = Qb; a.ldq = ldq; a.Ka = Ka; a.Va = Va; a.ldka = ldka; a.Kb = Kb; a.Vb = Vb; a.ldkb = ldkb;
    a.mq = mq; a.mka = mka; a.mkb = mkb;
    a.scale = 1.0f / sqrtf((float)dh);
    a.drop = make_drop(drop_p, seed, site);
    {   // K/V/Q views are addressed through 32-bit buffer offsets (tile overhang of 15 rows included)
        const size_t ka = La ? ((size_t)B * La - 1) * ldka + (size_t)H * dh : 0, kb = Lb ? ((size_t)B * Lb - 1) * ldkb + (size_t)H * dh : 0;
        const size_t q = ((size_t)B * Lq - 1) * ldq + (size_t)H * dh;
        SEGMM_REQUIRE(((size_t)B * La + 16) * ldka * 4 < (1ull << 32) && ((size_t)B * Lb + 16) * ldkb * 4 < (1ull << 32) &&
                      ((size_t)B * Lq + 16) * ldq * 4 < (1ull << 32), "attn: a K/V/Q view exceeds the 4 GiB buffer-addressing window");
        a.ka_bytes = (uint32_t)(ka * 4); a.kb_bytes = (uint32_t)(kb * 4); a.q_bytes = (uint32_t)(q * 4);
    }
    return 0;
}

// ---------------------------------------------------------------- tuning / A-B knobs
// Every run-time choice of the library that is not an argument lives in ONE table: read from the environment once (first use of
// the library; SEGMM_<NAME>), listed by segmm_config_dump, changed afterwards only through segmm_config_set -- no getenv on a
// launch path.  Knobs whose non-default values give WRONG results (timing probes) do not exist in this build: they are compiled
// in by -DSEGMM_ATT_PROBE / -DSEGMM_GEMM_PROBE only.
enum {
    K_ATTN, K_ATT_FWD_PL, K_ATT_FWD_LDS, K_ATT_FWD_KSPLIT, K_ATT_FWD_LDS_PAD, K_ATT_FUSED_LAUNCH, K_ATT_MERGE, K_ATT_LDS_PAD, K_ATT_WAVES, K_ATT_WAVES_PL, K_ATT_REPAIR_WALK,
    K_ATT_HPB_FWD, K_ATT_HPB_DQ, K_ATT_HPB_DKV, K_L1NORM_REG, K_GEMM_BN, K_PL_VAR, K_PL_NJ, K_TN_VAR, K_LN_BWD_PARTS, K_COUNT
};
struct Knob { const char* name; int value; const char* doc; };
static Knob g_knobs[K_COUNT] = {
    {"ATTN", 1, "attention arithmetic: 0 (env: f32) exact-fp32 kernels only, 1 fp16x3 where measured faster, 2 (env: f16all) fp16x3 wherever built"},
    {"ATT_FWD_PL", 1, "planes-in forward when the caller hands input planes (0: never)"},
    {"ATT_FWD_LDS", 1, "LDS-DMA staged fp32 forward: 0 never, 1 heads with >= 4 query tiles, 2 wherever it fits"},
    {"ATT_FWD_KSPLIT", 1, "staged forward: key-tile groups per query tile"},
    {"ATT_FWD_LDS_PAD", 0, "probe: extra LDS bytes per forward workgroup (fewer workgroups per CU)"},
    {"ATT_FUSED_LAUNCH", 2, "fused backward: 2 one launch per key block, 1 one launch for both"},
    {"ATT_MERGE", 1, "short heads: one workgroup per head for both key blocks in the fused backward"},
    {"ATT_LDS_PAD", 0, "probe: extra LDS bytes per backward workgroup"},
    {"ATT_WAVES", 4, "fused fp16x3 backward: waves per workgroup (key tiles in passes)"},
    {"ATT_WAVES_PL", 3, "planes-in fused backward: waves per workgroup, 1..4 (3: 40.7 KB of LDS -> four workgroups per CU for a 100-key block too)"},
    {"ATT_REPAIR_WALK", 512, "planes-in backward, repair launch: workgroups that walk the heads (0: one workgroup per head)"},
    {"ATT_HPB_FWD", 0, "forward workgroup shape heads | tiles << 8 (env: \"heads[,tiles]\"; 0: built-in)"},
    {"ATT_HPB_DQ", 0, "dQ kernel workgroup shape, as above"},
    {"ATT_HPB_DKV", 0, "dK/dV kernel workgroup shape, as above"},
    {"L1NORM_REG", 1, "L1 normalisation with the row held in registers"},
    {"GEMM_BN", 0, "on-the-fly GEMM: tile width override (0: built-in choice)"},
    {"PL_VAR", 4, "plane NT GEMM: 4 gemm_pl_nt4 (round 6: 128 x 256 tiles, two workgroups per CU) for K >= 768 and N >= 768, gemm_pl_nt8 otherwise; 44 gemm_pl_nt4 always; 8 gemm_pl_nt8 (round 3); 1: the round-2 fallback kernel for every launch"},
    {"PL_NJ", 0, "plane NT GEMM: tile width in 64-column units (0: modelled choice)"},
    {"TN_VAR", 8, "plane TN GEMM: 8 gemm_pl_tn4 (round 6: 128 x 256 tiles, two workgroups per CU) for few-tile and 128-row matrices, gemm_pl_tn8 (round 3) otherwise; 4 gemm_pl_tn4 wherever it fits; 88 gemm_pl_tn8 wherever it fits; 0: the round-2 fallback kernel for every launch"},
    {"LN_BWD_PARTS", 0, "LayerNorm backward: most workgroups (= partial rows of its column sums) per launch; 0: as many four-wave workgroups as are resident at the row width (768 at d = 768: one full round, no under-occupied tail)"},
};
static bool g_knobs_ready = false;
static void knobs_init() {
    if (g_knobs_ready) return;
    g_knobs_ready = true;
    for (int k = 0; k < K_COUNT; ++k) {
        char name[64];
        snprintf(name, sizeof(name), "SEGMM_%s", g_knobs[k].name);
        const char* e = getenv(name);
        if (!e || !*e) continue;
        if (k == K_ATTN) g_knobs[k].value = !strcmp(e, "f32") ? 0 : !strcmp(e, "f16all") ? 2 : 1;
        else if (k == K_ATT_HPB_FWD || k == K_ATT_HPB_DQ || k == K_ATT_HPB_DKV) {
            const char* c = strchr(e, ',');
            g_knobs[k].value = (atoi(e) > 0 ? atoi(e) : 0) | ((c && atoi(c + 1) > 0 ? atoi(c + 1) : 0) << 8);
        } else g_knobs[k].value = atoi(e);
    }
}
static inline int knob(int k) { knobs_init(); return g_knobs[k].value; }
int segmm_config_set(const char* name, int value) {
    knobs_init();
    for (int k = 0; k < K_COUNT; ++k)
        if (!strcmp(name, g_knobs[k].name)) { const int prev = g_knobs[k].value; g_knobs[k].value = value; return prev; }
    return segmm_fail(-1, "segmm_config_set: no knob named %s (see segmm_config_dump)", name);
}
int segmm_config_dump(char* buf, int n) {          // "NAME=value  # doc\n" per knob; returns the length needed (without the terminator)
    knobs_init();
    int need = 0;
    for (int k = 0; k < K_COUNT; ++k) {
        char line[320];
        const int m = snprintf(line, sizeof(line), "SEGMM_%s=%d  # %s\n", g_knobs[k].name, g_knobs[k].value, g_knobs[k].doc);
        if (buf && need + m < n) memcpy(buf + need, line, (size_t)m + 1);
        need += m;
    }
    return need;
}

// SEGMM_ATTN=f32 keeps every attention kernel on the exact-fp32 matrix-core form (attention.h); default: the fp16x3 form
// (attention16.h) where it is built (head dims 16, 32, 48) and measured faster
static int attn_f16() { return knob(K_ATTN); }          // 0: exact-fp32 kernels only; 1 (default): fp16x3 where it is faster; 2: fp16x3 wherever it is built

// workgroup shape for n row tiles per head: wq tiles x hpb adjacent heads, at most max_waves waves, every wave busy
static void attn_shape(int n, int H, int max_waves, int want_default, int kn, int& wq, int& hpb) {
    wq = n;
    if (n > 5) {
        wq = 4;
        for (int w = 5; w >= 3; --w)
            if (n % w == 0) { wq = w; break; }
    }
    hpb = 1;
    const int e = knob(kn);          // A/B knob (SEGMM_ATT_HPB_FWD / _DQ / _DKV = "heads[,tiles]")
    int want = want_default;
    if (e & 0xff) {
        want = e & 0xff;
        if ((e >> 8) > 0 && (e >> 8) <= n) wq = e >> 8;
    }
    for (int c = want; c >= 1; --c)
        if (H % c == 0 && c * wq <= max_waves) { hpb = c; break; }
}

// round 5: the planes-in forward (attention_pl.h) when the caller hands the Q / K / V planes of the projection GEMMs and the
// shape qualifies.  SEGMM_ATT_FWD_PL=0 keeps the fp32-operand kernels (A/B, tests).
template <int DH>
static bool attn_fwd_pl_takes(const AttnArgs& a) {
    if constexpr (DH % 16 != 0) return false;
    else {
        if (knob(K_ATT_FWD_PL) == 0) return false;
        const int nqt = (a.Lq + 15) / 16, T = a.La + a.Lb;
        return a.in.Qa && a.La % 4 == 0 && a.Lb % 4 == 0 && nqt <= ATT_PL_MAXW && T <= 16 * 12 &&
               attn_fwd_pl_lds_bytes<DH>(a.La, a.Lb) <= 160 * 1024 && (((uintptr_t)a.mka | (uintptr_t)a.mkb) & 3u) == 0;
    }
}
template <int DH>
static int attn_launch_fwd_pl(AttnArgs& a, hipStream_t s) {
    if constexpr (DH % 16 == 0) {
        const int nqt = (a.Lq + 15) / 16, T = a.La + a.Lb;
        const size_t lds = attn_fwd_pl_lds_bytes<DH>(a.La, a.Lb);
        // instances: key-block-a length 40 with all 9 tiles present (configs 2 / 4 / 5: tile classes resolved at compile time, a
        // branch-free body), and the run-time forms for <= 4 / 9 / 12 key tiles
        const bool hot = DH == 48 && a.La == 40 && T > 128 && T <= 144;
        static bool optin = false;          // dynamic LDS above 64 KB needs the opt-in, once per kernel
        if (!optin) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_pl_kernel<DH, 4, -1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)attn_fwd_pl_kernel<DH, 9, -1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)attn_fwd_pl_kernel<DH, 12, -1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if constexpr (DH == 48) (void)hipFuncSetAttribute((const void*)attn_fwd_pl_kernel<48, 9, 40, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            optin = true;
        }
#ifdef SEGMM_ATT_PROBE
        if (getenv("SEGMM_ATT_FWD_DBG")) a.pflags |= atoi(getenv("SEGMM_ATT_FWD_DBG")) & (256 | 512);          // timing probes (results wrong; probe builds only)
#endif
        const dim3 grid(a.B * a.H), block(64 * nqt);
        if (hot) { if constexpr (DH == 48) hipLaunchKernelGGL((attn_fwd_pl_kernel<48, 9, 40, true>), grid, block, lds, s, a); }
        else if (T <= 64) hipLaunchKernelGGL((attn_fwd_pl_kernel<DH, 4, -1, false>), grid, block, lds, s, a);
        else if (T <= 144) hipLaunchKernelGGL((attn_fwd_pl_kernel<DH, 9, -1, false>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((attn_fwd_pl_kernel<DH, 12, -1, false>), grid, block, lds, s, a);
        LAUNCH_CHECK();
    }
    return 0;
}

template <int DH>
static int attn_launch_fwd(AttnArgs& a, hipStream_t s) {
    if (attn_fwd_pl_takes<DH>(a)) return attn_launch_fwd_pl<DH>(a, s);
    SEGMM_REQUIRE(a.Qa, "attn_fwd: no fp32 views were given and the planes-in forward does not take this call (it needs dh %% 16 == 0, "
                        "La %% 4 == 0, Lb %% 4 == 0, Lq <= 112, 4-byte aligned masks; knob ATT_FWD_PL)");
    const int Tp = ((a.La + 15) & ~15) + ((a.Lb + 15) & ~15);
    const int nqt = (a.Lq + 15) / 16;
    int wq, hpb;
    attn_shape(nqt, a.H, 5, 1, K_ATT_HPB_FWD, wq, hpb);          // measured: grouping heads does not pay in the forward
    // round 4: the LDS-DMA staged form (one workgroup per head: K / V of both key blocks staged once, every load of the head in
    // flight at once -- the staging alone runs at 5.7 TB/s) for heads with MORE than three query tiles: there one workgroup brings
    // enough waves (>= 4 per head, two heads per CU) to cover the per-wave instruction chains -- Lq = 100 (user queries of the full
    // layers of N >= 3 models): 523 -> 430 us.  At Lq = 40 (three waves per head, six per CU: the staged K / V of a head are 66 KB,
    // so two heads is what a CU holds) the compute phase alone takes 200 us -- issue-bound chains at 1.3 waves per SIMD, PMC in
    // profiles/r4/attention_fwd_lds_pmc.txt -- against 233 us for the whole direct-load kernel at 3.3 waves per SIMD, and splitting
    // the key tiles over two or three wave groups per query tile (SEGMM_ATT_FWD_KSPLIT) does not change that: the direct form stays.
    // SEGMM_ATT_FWD_LDS=0 / 2: never / wherever it fits (A/B, tests).
    const int fwd_lds = knob(K_ATT_FWD_LDS);
    if constexpr (DH >= 16) {
        const size_t lds2 = attn_fwd_lds_bytes<DH>(a.La, a.Lb);
        if ((fwd_lds == 2 || (fwd_lds == 1 && nqt >= 4)) && nqt <= 8 && lds2 <= 80 * 1024) {
            // key-tile groups per query tile (waves per head = nqt * ksp <= 12): more waves on the same staged K / V
            int ksp = knob(K_ATT_FWD_KSPLIT);
            const int ntile = Tp / 16;
            if (ksp > ntile) ksp = ntile;
            if (ksp < 1 || nqt * ksp > 12) ksp = 1;
            const dim3 grid2(a.B * a.H), block2(64 * nqt * ksp);
            const size_t merge = ksp > 1 ? (size_t)nqt * ksp * (64 * 4 * ((DH + 15) / 16) + 128) * 4 : 0;          // partials of the key groups
            const size_t lds3 = lds2 > merge ? lds2 : merge;
#ifdef SEGMM_ATT_PROBE
            if (getenv("SEGMM_ATT_FWD_DBG")) a.pflags |= atoi(getenv("SEGMM_ATT_FWD_DBG")) & (256 | 512);          // timing probes (results wrong; probe builds only)
#endif
            static bool lds_optin = false;          // dynamic LDS above 64 KB needs the opt-in, once per kernel
            if (!lds_optin) {
                (void)hipFuncSetAttribute((const void*)attn_fwd_lds_kernel<DH, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)attn_fwd_lds_kernel<DH, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)attn_fwd_lds_kernel<DH, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                lds_optin = true;
            }
            if (Tp <= 64) hipLaunchKernelGGL((attn_fwd_lds_kernel<DH, 4>), grid2, block2, lds3, s, a);
            else if (Tp <= 160) hipLaunchKernelGGL((attn_fwd_lds_kernel<DH, 10>), grid2, block2, lds3, s, a);
            else hipLaunchKernelGGL((attn_fwd_lds_kernel<DH, 12>), grid2, block2, lds3, s, a);
            LAUNCH_CHECK();
            return 0;
        }
    }
    a.hpb = hpb;
    dim3 grid(a.B * a.H / hpb, (nqt + wq - 1) / wq), block(64 * wq * hpb);       // one wave per 16-query tile of a head
    const int fpad = knob(K_ATT_FWD_LDS_PAD);      // probe: fewer workgroups per CU
    const size_t lds = (size_t)Tp + (size_t)fpad;
    if (Tp <= 64) hipLaunchKernelGGL((attn_fwd_kernel<DH, 4>), grid, block, lds, s, a);
    else if (Tp <= 160) hipLaunchKernelGGL((attn_fwd_kernel<DH, 10>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<DH, 12>), grid, block, lds, s, a);
    LAUNCH_CHECK();
    return 0;
}

template <int DH>
static int attn_launch_bwd(AttnArgs& a, int phase, hipStream_t s) {
    const int Tp = ((a.La + 15) & ~15) + ((a.Lb + 15) & ~15);
    if (phase == 1) {          // D only
        const long long n = (long long)a.B * a.Lq * a.H;
        hipLaunchKernelGGL((attn_D_kernel<DH>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        LAUNCH_CHECK();
        return 0;
    }
    if (phase >= 4) {          // fused dQ + dK + dV: one workgroup per (b, h, key block); forms D = rowsum(dO * O) itself
                               // (5 / 6: key block a / b only -- the two launches are independent and may run on two streams)
        const int nta = ((a.La + 15) & ~15) >> 4, ntb = ((a.Lb + 15) & ~15) >> 4;
        // LDS is sized for one chunk of the query side: 48 rows, or 16 / 32 when all queries fit (fp32 kernel only)
        const int Lq_small = a.Lq <= 16 ? 16 : a.Lq <= 32 ? 32 : ATT_FUSED_QCHUNK;
        int Lq_p = ATT_FUSED_QCHUNK;
        SEGMM_REQUIRE(nta <= ATT_FUSED_MAXW && ntb <= ATT_FUSED_MAXW, "attn_bwd phase 4: built for <= 12 key tiles per block (%d + %d tiles)", nta, ntb);
        const int fmode_env = knob(K_ATT_FUSED_LAUNCH);
        // the repair pass is (almost always) a launch of workgroups that leave at once: one launch for both key blocks
        const int fmode = (a.pflags & ATT_REPAIR) ? 1 : fmode_env;
        const int nmax = nta > ntb ? nta : ntb;
        // short heads (config 3: 20 x (20 + 1) and 1 x (1 + 20), three key tiles): ONE workgroup per head for both key blocks
        // (p.hpb == 3, attention.h) -- the query side is staged once, one launch instead of two; bit-identical to the per-block
        // launches.  SEGMM_ATT_MERGE=0 restores them (A/B, tests)
        {
            const bool f16_takes_it = (DH % 16 == 0 && DH <= 48) && attn_f16() >= 2;
            if (!a.in.Qa && knob(K_ATT_MERGE) != 0 && phase == 4 && a.Lq <= 32 && nta > 0 && ntb > 0 && nta + ntb <= 4 && !f16_takes_it) {
                const int nw = nta + ntb;
                a.hpb = 3;
                const dim3 grid(a.B * a.H), block(64 * nw);
                Lq_p = Lq_small;
                const size_t lds = ((size_t)5 * Lq_p * (DH + 4) + 3 * Lq_p + (size_t)nw * 16 * 20 + 8 + (size_t)Lq_p * (DH / 4)) * 4 + Lq_p + Tp;
                if (Lq_p == 16) hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, 4, true, 16>), grid, block, lds, s, a);
                else hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, 4, true, 32>), grid, block, lds, s, a);
                LAUNCH_CHECK();
                return 0;
            }
        }
        for (int blk = 0; blk < 2; ++blk) {
            // fmode 2 (default): one launch per key block with its exact wave count -- 539-549 us at config 2;
            // fmode 1: ONE launch for both key blocks (workgroups of 64 * max(nta, ntb) threads, surplus waves end at once) -- 680 us
            if (fmode == 1 && blk == 1) break;
            if (fmode != 1 && phase >= 5 && blk != phase - 5) continue;
            const int nw = fmode == 1 ? nmax : (blk == 0 ? nta : ntb);
            if (nw == 0) continue;
            a.hpb = fmode == 1 ? 2 : blk;
            const dim3 grid((fmode == 1 ? 2 : 1) * a.B * a.H), block(64 * nw);
            const bool one = a.Lq <= ATT_FUSED_QCHUNK;
            Lq_p = ATT_FUSED_QCHUNK;
            size_t lds = ((size_t)3 * Lq_p * (DH + 4) + 3 * Lq_p + (size_t)nw * 16 * 20 + 4 + 36 + (size_t)Lq_p * (DH / 4)) * 4 + Lq_p + Tp;
            const int lds_pad = knob(K_ATT_LDS_PAD);      // probe: fewer workgroups per CU
            lds += (size_t)lds_pad;
            if constexpr (DH % 16 == 0 && DH <= 48) {
                if (a.in.Qa) {          // round 5: Q / K / V from the projection GEMMs' planes (attention_pl.h); always in passes of <= 4 waves
                    // (single chunk: the key tiles of a block in passes over <= ATT_WAVES_PL waves.  Three: the 7 tiles of a 100-key block as 3 + 3 + 1,
                    // 40.7 instead of 44.0 KB of LDS -> FOUR workgroups per CU like the 40-key block; same time stand-alone, 494 -> 480 us beside
                    // the weight-gradient GEMMs of the step, +0.5 % of the step: profiles/r6/att_waves_ab.txt.  Results do not depend on it.)
                    const int wcap_pl = knob(K_ATT_WAVES_PL) >= 1 && knob(K_ATT_WAVES_PL) <= 4 ? knob(K_ATT_WAVES_PL) : 3;
                    const int nwp = one && nw > wcap_pl ? wcap_pl : nw;
                    const size_t ldsp = attn_bwd_pl_lds_bytes<DH>(Lq_p, nwp, Tp);
                    const dim3 blockp(64 * nwp);
                    // the repair launch: 512 workgroups that walk the heads (and normally leave at once) instead of one per head
                    const bool walk = (a.pflags & ATT_REPAIR) != 0;
                    const unsigned nwalk = (unsigned)knob(K_ATT_REPAIR_WALK) & ~1u;          // (even: a workgroup keeps its key block)
                    const dim3 gridp(walk && nwalk && grid.x > nwalk ? nwalk : grid.x);
#define FUSEDPL(NWV) do { if (walk) { if (one) hipLaunchKernelGGL((attn_bwd_pl_kernel<DH, NWV, true, true>), gridp, blockp, ldsp, s, a); \
                                      else hipLaunchKernelGGL((attn_bwd_pl_kernel<DH, NWV, false, true>), gridp, blockp, ldsp, s, a); } \
                          else if (one) hipLaunchKernelGGL((attn_bwd_pl_kernel<DH, NWV, true>), grid, blockp, ldsp, s, a); \
                          else hipLaunchKernelGGL((attn_bwd_pl_kernel<DH, NWV, false>), grid, blockp, ldsp, s, a); } while (0)
                    if (nwp <= 4) FUSEDPL(4);
                    else if (nwp <= 8) FUSEDPL(8);
                    else FUSEDPL(12);
#undef FUSEDPL
                    continue;
                }
                // fp16x3 matrix-core form (attention16.h).  By default only single-chunk launches with more than 32 queries: with
                // several query chunks the kernel needs more than the 128 registers that keep two 7-wave workgroups on a CU and
                // loses to the fp32 form (Lq = 100: 1 459 vs 1 199 us), and with a handful of queries (config 3: Lq = 20 and 1)
                // the in-place conversion pass costs more than the products save (1.81 vs 1.65 ms of attention per step).
                // segmm_attn_mode(2) / SEGMM_ATTN=f16all forces it everywhere it is built (parity tests)
                if (attn_f16() >= ((one && a.Lq > 32) ? 1 : 2)) {
                    // single chunk: at most `wcap` waves per workgroup, a wave walks its key tiles in passes (attention16.h) --
                    // the 7 tiles of a 100-key block as 4 waves x 2 passes, four workgroups per CU instead of two
                    const int wcap = knob(K_ATT_WAVES);
#ifdef SEGMM_ATT_PROBE
                    if (getenv("SEGMM_ATT_BWD_DBG")) a.pflags |= atoi(getenv("SEGMM_ATT_BWD_DBG")) & (1024 | 2048 | 4096);          // timing probes (results wrong)
#endif
                    const int nw16 = one && wcap >= 1 && nw > wcap ? wcap : nw;
                    const dim3 block16(64 * nw16);
                    const size_t lds16 = lds - (size_t)(nw - nw16) * 16 * 20 * 4;
#define FUSED16(NWV) do { if (one) hipLaunchKernelGGL((attn_bwd_fused16_kernel<DH, NWV, true>), grid, block16, lds16, s, a); \
                          else hipLaunchKernelGGL((attn_bwd_fused16_kernel<DH, NWV, false>), grid, block16, lds16, s, a); } while (0)
                    if (nw16 <= 4) FUSED16(4);
                    else if (nw16 <= 8) FUSED16(8);
                    else FUSED16(12);
#undef FUSED16
                    continue;
                }
            }
            SEGMM_REQUIRE(a.Qa, "attn_bwd: no fp32 views were given and the planes-in backward does not take this call (head dims 16, 32, 48)");
            Lq_p = Lq_small;
            lds = ((size_t)3 * Lq_p * (DH + 4) + 3 * Lq_p + (size_t)nw * 16 * 20 + 4 + (size_t)Lq_p * (DH / 4)) * 4 + Lq_p + Tp;
#define FUSED(NWV) do { if (Lq_p == 16) hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, NWV, true, 16>), grid, block, lds, s, a); \
                        else if (Lq_p == 32) hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, NWV, true, 32>), grid, block, lds, s, a); \
                        else if (one) hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, NWV, true>), grid, block, lds, s, a); \
                        else hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, NWV, false>), grid, block, lds, s, a); } while (0)
            if (nw <= 4) FUSED(4);
            else if (nw <= 8) FUSED(8);
            else FUSED(12);
#undef FUSED
        }
        LAUNCH_CHECK();
        return 0;
    }
    a.write_D = phase == 0;
    if (phase == 0 || phase == 2) {
        const int nqt = (a.Lq + 15) / 16;
        int wq, hpb;
        attn_shape(nqt, a.H, 12, 1, K_ATT_HPB_DQ, wq, hpb);
        a.hpb = hpb;
        dim3 grid(a.B * a.H / hpb, (nqt + wq - 1) / wq), block(64 * wq * hpb);
        if (Tp <= 64) hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, 4>), grid, block, Tp, s, a);
        else if (Tp <= 160) hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, 10>), grid, block, Tp, s, a);
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, 12>), grid, block, Tp, s, a);
        LAUNCH_CHECK();
    }
    if (phase == 0 || phase == 3) {
        const int nt = Tp / 16;
        int wq, hpb;
        attn_shape(nt, a.H, 12, 1, K_ATT_HPB_DKV, wq, hpb);
        // small workgroups: at 3 waves/SIMD a CU holds 12 waves = six 2-wave groups, but only two 5-wave ones
        // (measured for 10 key tiles: 2 waves 533 us, 1 wave 544, 4 waves 611, 5 waves 722)
        if (!knob(K_ATT_HPB_DKV)) wq = nt >= 2 ? 2 : 1;
        a.hpb = hpb;
        dim3 grid(a.B * a.H / hpb, (nt + wq - 1) / wq), block(64 * wq * hpb);   // one wave per 16-key tile of a head
        const int Lq_p = (a.Lq + 15) & ~15;
        const size_t lds = (size_t)hpb * Lq_p * 12 + Lq_p + Tp;          // per head 3 float vectors; query flags; key flags
        if (Lq_p == 48) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, 3>), grid, block, lds, s, a);
        else if (Lq_p == 16) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, 1>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, 0>), grid, block, lds, s, a);
        LAUNCH_CHECK();
    }
    return 0;
}

#define ATTN_DISPATCH(FN, dh, ...)                  \
    switch (dh) {                                   \
        case 4: return FN<4>(__VA_ARGS__);          \
        case 8: return FN<8>(__VA_ARGS__);          \
        case 16: return FN<16>(__VA_ARGS__);        \
        case 32: return FN<32>(__VA_ARGS__);        \
        case 48: return FN<48>(__VA_ARGS__);        \
        case 64: return FN<64>(__VA_ARGS__);        \
        default: return segmm_fail(-1, "attn: head dim %d", dh); \
    }

__global__ void dropout_mult_kernel(float* out, long long n, DropCfg d) {
    d = drop_live(d);
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < ((n + 3) >> 2); q += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = drop_apply4(d, (uint64_t)q, f32x4{1.f, 1.f, 1.f, 1.f});
        for (int k = 0; k < 4; ++k)
            if (4 * q + k < n) out[4 * q + k] = v[k];
    }
}


// Diagnostic (bench.py roofline context): the sustained issue rate of the plane GEMMs' MFMA instruction with their accumulator
// pattern, their occupancy (8 waves per CU) and RANDOM operand bits, registers only.  With constant operands the MI355X holds
// ~2.47 PFLOP/s; with random bits the power management clocks it down (32x32x16: 1.6-1.78 PFLOP/s, 16x16x32: ~1.85 PFLOP/s,
// box to box) -- the ceiling a real-data GEMM can reach on this part, below the 2.5 PFLOP/s datasheet figure bench.py prices against.
namespace segmm {
__global__ __launch_bounds__(512, 2) void mfma_rate_kernel(float* out, int iters, uint32_t seed) {
    // the plane GEMMs' instruction (round 3: v_mfma_f32_16x16x32_f16 -- the chip holds a higher clock on it than on 32x32x16),
    // tile grid (8 x 4 accumulators of 16 x 16 per wave) and accumulator order (three products per accumulator, back to back)
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 a[8], b[4];
    uint32_t h = (threadIdx.x + blockIdx.x * 977u) * 2654435761u + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) { h = h * 1664525u + 1013904223u; a[i][r] = (_Float16)(((int)(h >> 16) - 32768) * (1.f / 32768.f)); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) { h = h * 1664525u + 1013904223u; b[i][r] = (_Float16)(((int)(h >> 16) - 32768) * (1.f / 32768.f)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    if (t == 12345.678f) out[0] = t;          // never true: keeps the accumulators live
}
}  // namespace segmm

namespace segmm {
__global__ void step_set_kernel(StepState* st, uint32_t lo, uint32_t hi, int step, float b1, float b2) {
    st->seed_lo = lo; st->seed_hi = hi; st->step = step;
    st->bc1 = step > 0 ? (float)(1.0 - pow((double)b1, (double)step)) : 1.f;
    st->bc2_sqrt = step > 0 ? (float)sqrt(1.0 - pow((double)b2, (double)step)) : 1.f;
}
__global__ void step_advance_kernel(StepState* st, float b1, float b2) {
    const int t = st->step + 1;
    st->step = t;
    const uint32_t lo = mix32(st->seed_lo + 0x9E3779B9u * (uint32_t)t);
    st->seed_hi = mix32(st->seed_hi ^ lo ^ 0x85EBCA6Bu) & 0x7fffffffu;
    st->seed_lo = lo;
    st->bc1 = (float)(1.0 - pow((double)b1, (double)t));
    st->bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
}  // namespace segmm

extern "C" {

const char* segmm_last_error(void) { return g_segmm_err; }
int segmm_abi_version(void) { return 29; }
int segmm_attn_mode(int mode) { const int prev = attn_f16(); if (mode >= 0 && mode <= 2) g_knobs[K_ATTN].value = mode; return prev; }

static PlaneOut plane_out(uint16_t* planes, int ld2, float* hdr, const float* scale_in) {
    PlaneOut po;
    po.p = (_Float16*)planes; po.ld2 = ld2; po.hdr = hdr; po.scale_in = scale_in;
    return po;
}
#define PLANE_OUT_CHECK(what, cols)                                                                                          \
    SEGMM_REQUIRE(!planes || (hdr && (cols) % 32 == 0 && ld2 % 64 == 0 && ld2 >= 2 * (cols) && aligned16(planes)),             \
                  what ": plane output needs a header, cols %% 32, ld2 %% 64, 16-byte alignment")

int segmm_l1norm(const float* x, float* y, float* inv_scale, int64_t rows, int D, float* amax, uint16_t* planes, int ld2, float* hdr,
                 const float* scale_in, segmm_stream_t stream) {
    PLANE_OUT_CHECK("l1norm", D);
    SEGMM_REQUIRE(x && (y || inv_scale || (planes && scale_in)), "l1norm: null pointer");
    SEGMM_REQUIRE(D > 0 && D % 4 == 0 && aligned16(x) && (!y || aligned16(y)), "l1norm: D %% 4 / alignment (D=%d)", D);
    if (rows <= 0) return 0;
    const int wpb = 4;
    const int reg_form = knob(K_L1NORM_REG);
    const dim3 grid((unsigned)((rows + wpb - 1) / wpb)), block(64 * wpb);
#define L1R(V) hipLaunchKernelGGL((l1norm_reg_kernel<V>), grid, block, 0, (hipStream_t)stream, x, y, inv_scale, (long long)rows, D, amax, plane_out(planes, ld2, hdr, scale_in))
    if (reg_form && D <= 256) L1R(1);
    else if (reg_form && D <= 512) L1R(2);
    else if (reg_form && D <= 768) L1R(3);
    else if (reg_form && D <= 1024) L1R(4);
    else if (reg_form && D <= 1536) L1R(6);
    else if (reg_form && D <= 2048) L1R(8);
    else hipLaunchKernelGGL(l1norm_kernel, grid, block, 0, (hipStream_t)stream, x, y, inv_scale, (long long)rows, D, amax, plane_out(planes, ld2, hdr, scale_in));
#undef L1R
    LAUNCH_CHECK();
    return 0;
}

static int gemm_impl(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
                     int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
                     float* workspace, int accumulate, int engine, const uint16_t* a_planes, long long a_pstride,
                     const uint16_t* b_planes, long long b_pstride, int nplanes, const float* a_amax, int a_namax,
                     const float* b_amax, int b_namax, float* c_amax, segmm_stream_t stream) {
    SEGMM_REQUIRE(layout >= 0 && layout <= 2, "gemm: bad layout %d", layout);
    if (engine == 2) nplanes = 2;
    SEGMM_REQUIRE(nplanes == 3 || nplanes == 2, "gemm: nplanes %d (3 = six products, 2 = three products)", nplanes);
    if (engine == 2)
        SEGMM_REQUIRE(a_amax && b_amax && a_namax > 0 && b_namax > 0 && a_namax <= 1024 && b_namax <= 1024,
                      "gemm: the fp16x3 engine needs the partial maxima of both operands (1..1024 each)");
    if (a_planes || b_planes) {
        SEGMM_REQUIRE((engine == 1 || engine == 2) && layout == 0, "gemm: pre-split operands need engine 1/2 and the NT layout");
        SEGMM_REQUIRE(K % 8 == 0 && (!a_planes || (lda % 8 == 0 && aligned16(a_planes) && a_pstride % 8 == 0)) &&
                      (!b_planes || (ldb % 8 == 0 && aligned16(b_planes) && b_pstride % 8 == 0)), "gemm: plane operands need K, ld, stride %% 8 == 0 and 16-byte alignment");
    }
    if (a_planes && !A) A = (const float*)a_planes;      // only the planes are read
    if (b_planes && !B) B = (const float*)b_planes;
    SEGMM_REQUIRE(engine >= 0 && engine <= 2, "gemm: engine %d (0 = f32 MFMA, 1 = bf16x6 split MFMA, 2 = fp16x3 split MFMA)", engine);
    SEGMM_REQUIRE(A && B && C, "gemm: null operand");
    if (M <= 0 || N <= 0) return 0;
    SEGMM_REQUIRE(K > 0, "gemm: K=%d", K);
    SEGMM_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "gemm: N/ld not multiples of 4 (N=%d lda=%d ldb=%d ldc=%d)", N, lda, ldb, ldc);
    SEGMM_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), "gemm: operands must be 16-byte aligned");
    if (layout == 0 || layout == 1) SEGMM_REQUIRE(K % 4 == 0, "gemm: K %% 4 != 0 for a k-contiguous operand (K=%d)", K);
    if (layout == 2) SEGMM_REQUIRE(M % 4 == 0, "gemm: M %% 4 != 0 for the transposed A operand (M=%d)", M);
    SEGMM_REQUIRE(!bias || aligned16(bias), "gemm: bias alignment");
    SEGMM_REQUIRE(!residual || (aligned16(residual) && ldr % 4 == 0 && res_period > 0), "gemm: residual alignment/period");
    SEGMM_REQUIRE(activation >= 0 && activation <= 4, "gemm: activation %d", activation);
    SEGMM_REQUIRE(activation == 0 || activation == EPI_RELU || (aux && aligned16(aux) && ldaux % 4 == 0), "gemm: activation needs aux");
    SEGMM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm: dropout p=%f", drop_p);
    if (splits < 1) splits = 1;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
    g.bias = bias; g.row_scale = row_scale;
    g.residual = residual; g.ldr = ldr; g.res_period = res_period > 0 ? res_period : 1;
    g.aux = aux; g.ldaux = ldaux; g.epi = activation;
    g.drop = make_drop(drop_p, seed, site);
    g.amax_out = c_amax;
    // workgroup tile width (fp16x3 engine): 128 x 256 for the weight-gradient form (TN: both operands fp32, k-strided,
    // transposed in registers) once it has >= 36 wide tiles -- measured on one box, same run: TN 1536x768x51200 634 -> 562 us,
    // TN 3072x768x20480 491 -> 470 us, but TN 768x768xK 139 -> 154 us and every NT shape (weights pre-split) 2-7 % slower
    // at the two workgroups per CU the wide tile allows.  SEGMM_GEMM_BN=128 / 256 forces one of them (A/B knob).
    const int bn_env = knob(K_GEMM_BN);
    const bool wide = engine == 2 && !a_planes && N > 128 && bn_env != 128 &&
                      (bn_env == 256 || (layout == 2 && (long long)((M + 127) / 128) * ((N + 255) / 256) >= 36));
    const int BN = wide ? 256 : GBN;
    {   // extents of the operand views (rows x ld, last row only as wide as it is read); the split engines address
        // them through 32-bit buffer offsets, tile overhang included
        const size_t a_rows = layout == 2 ? (size_t)K : (size_t)M, a_cols = layout == 2 ? (size_t)M : (size_t)K;
        const size_t b_rows = layout == 0 ? (size_t)N : (size_t)K, b_cols = layout == 0 ? (size_t)K : (size_t)N;
        const size_t a_ext = ((a_rows - 1) * lda + a_cols) * 4, b_ext = ((b_rows - 1) * ldb + b_cols) * 4;
        const size_t a_reach = (layout == 2 ? (size_t)K + GBK : (size_t)((M + GBM - 1) / GBM) * GBM) * lda * 4 + 4096;
        const size_t b_reach = (layout == 0 ? (size_t)((N + BN - 1) / BN) * BN : (size_t)K + GBK) * ldb * 4 + 4096;
        if (engine != 0)
            SEGMM_REQUIRE(a_reach < (1ull << 32) && b_reach < (1ull << 32), "gemm: operand view above the 4 GiB buffer-addressing window of the split engines (%zu / %zu bytes)", a_reach, b_reach);
        g.a_bytes = (uint32_t)(a_ext < (1ull << 32) ? a_ext : 0xffffffffull);
        g.b_bytes = (uint32_t)(b_ext < (1ull << 32) ? b_ext : 0xffffffffull);
    }
    g.nbm = (M + GBM - 1) / GBM; g.nbn = (N + BN - 1) / BN;
    int ktiles = (K + GBK - 1) / GBK;
    if (splits > ktiles) splits = ktiles;
    if (splits > 1) {
        SEGMM_REQUIRE(workspace && aligned16(workspace), "gemm: split-K needs a workspace");
        SEGMM_REQUIRE(!bias && !row_scale && !residual && activation == 0 && drop_p == 0.f && !c_amax, "gemm: split-K supports no epilogue");
        const int tps = (ktiles + splits - 1) / splits;
        splits = (ktiles + tps - 1) / tps;
        g.k_per_split = tps * GBK;
        g.C = workspace; g.ldc = N; g.slab_stride = (long long)M * N;
    } else {
        g.k_per_split = ktiles * GBK;
        g.C = C; g.ldc = ldc; g.slab_stride = 0;
        if (accumulate) {
            SEGMM_REQUIRE(!residual, "gemm: accumulate and residual are exclusive");
            g.residual = C; g.ldr = ldc; g.res_period = M;
        }
    }
    dim3 grid(g.nbm * g.nbn, 1, splits), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (engine == 0) {
        if (layout == 0) hipLaunchKernelGGL((gemm_f32_mfma<true, true>), grid, block, 0, s, g);
        else if (layout == 1) hipLaunchKernelGGL((gemm_f32_mfma<true, false>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_f32_mfma<false, false>), grid, block, 0, s, g);
    } else {
        GemmPlanes q;
        q.Ap = (const __bf16*)a_planes; q.a_pstride = a_pstride;
        q.Bp = (const __bf16*)b_planes; q.b_pstride = b_pstride;
        q.a_amax = a_amax; q.a_namax = a_namax; q.b_amax = b_amax; q.b_namax = b_namax;
#define X6(AK, BK, AP, BP, NP) hipLaunchKernelGGL((gemm_split_mfma<AK, BK, AP, BP, NP>), grid, block, 0, s, g, q)
#define H3(AK, BK, AP, BP) hipLaunchKernelGGL((gemm_split_mfma<AK, BK, AP, BP, 2, true>), grid, block, 0, s, g, q)
#define H3W(AK, BK, BP) hipLaunchKernelGGL((gemm_split_mfma<AK, BK, false, BP, 2, true, 4>), grid, block, 0, s, g, q)
        if (engine == 2 && wide) {
            if (layout == 0) {
                if (b_planes) H3W(true, true, true);
                else H3W(true, true, false);
            } else if (layout == 1) H3W(true, false, false);
            else H3W(false, false, false);
        } else if (engine == 2) {
            if (layout == 0) {
                if (a_planes && b_planes) H3(true, true, true, true);
                else if (b_planes) H3(true, true, false, true);
                else if (a_planes) H3(true, true, true, false);
                else H3(true, true, false, false);
            } else if (layout == 1) H3(true, false, false, false);
            else H3(false, false, false, false);
        } else if (nplanes == 3) {
            if (layout == 0) {
                if (a_planes && b_planes) X6(true, true, true, true, 3);
                else if (b_planes) X6(true, true, false, true, 3);
                else if (a_planes) X6(true, true, true, false, 3);
                else X6(true, true, false, false, 3);
            } else if (layout == 1) X6(true, false, false, false, 3);
            else X6(false, false, false, false, 3);
        } else {
            SEGMM_REQUIRE(!a_planes && !b_planes, "gemm: nplanes 2 is implemented for fp32 operands only");
            if (layout == 0) X6(true, true, false, false, 2);
            else if (layout == 1) X6(true, false, false, false, 2);
            else X6(false, false, false, false, 2);
        }
#undef X6
#undef H3
#undef H3W
    }
    LAUNCH_CHECK();
    if (splits > 1) {
        const long long n4 = (long long)M * (N / 4);
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, s, (const float*)workspace, splits,
                           (long long)M * N, C, ldc, M, N, accumulate, (const float*)nullptr, (float*)nullptr);
        LAUNCH_CHECK();
    }
    return 0;
}

int segmm_gemm(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
               const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
               int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
               float* workspace, int accumulate, int engine, segmm_stream_t stream) {
    return gemm_impl(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, row_scale, residual, ldr, res_period, activation, aux, ldaux,
                     drop_p, seed, site, splits, workspace, accumulate, engine, nullptr, 0, nullptr, 0, 3, nullptr, 0, nullptr, 0, nullptr, stream);
}

int segmm_gemm_x(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
                 int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
                 float* workspace, int accumulate, const uint16_t* a_planes, int64_t a_pstride,
                 const uint16_t* b_planes, int64_t b_pstride, int nplanes, segmm_stream_t stream) {
    return gemm_impl(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, row_scale, residual, ldr, res_period, activation, aux, ldaux,
                     drop_p, seed, site, splits, workspace, accumulate, 1, a_planes, a_pstride, b_planes, b_pstride, nplanes,
                     nullptr, 0, nullptr, 0, nullptr, stream);
}

int segmm_gemm_h(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
                 int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
                 float* workspace, int accumulate, const uint16_t* a_planes, int64_t a_pstride,
                 const uint16_t* b_planes, int64_t b_pstride, const float* a_amax, int a_namax, const float* b_amax,
                 int b_namax, float* c_amax, segmm_stream_t stream) {
    return gemm_impl(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, row_scale, residual, ldr, res_period, activation, aux, ldaux,
                     drop_p, seed, site, splits, workspace, accumulate, 2, a_planes, a_pstride, b_planes, b_pstride, 2,
                     a_amax, a_namax, b_amax, b_namax, c_amax, stream);
}

/* ---- plane-operand GEMM (gemm_planes.h) */
int segmm_gemm_p(int layout, int M, int N, int K, const uint16_t* a_planes, int lda2, const float* a_hdr, const float* a_f32, int ldaf,
                 const uint16_t* b_planes, int ldb2, const float* b_hdr, const float* b_f32, int ldbf, float* C, int ldc,
                 uint16_t* c_planes, int ldc2, float* c_hdr, const float* c_scale_in, int write_c, const float* bias, const float* row_scale,
                 const float* residual, int ldr, int res_period, int activation, float* aux, int ldaux, float drop_p,
                 uint64_t seed, uint32_t site, int splits, float* workspace, int accumulate, float* colsum_out, segmm_stream_t stream) {
    SEGMM_REQUIRE(layout == 0 || layout == 2, "gemm_p: layout %d (0 = NT, 2 = TN)", layout);
    SEGMM_REQUIRE(!colsum_out || layout == 2, "gemm_p: colsum_out is an output of the TN form");
    SEGMM_REQUIRE(a_planes && b_planes && a_hdr && b_hdr, "gemm_p: null plane operand / header");
    SEGMM_REQUIRE(C || (c_planes && !(write_c & 1)), "gemm_p: no output");
    if (M <= 0 || N <= 0) return 0;
    SEGMM_REQUIRE(K > 0 && (layout == 2 || K % 32 == 0), "gemm_p: K %% 32 != 0 (K=%d)", K);      // TN: token tails are zero-filled by the buffer range check
    SEGMM_REQUIRE(N % 4 == 0 && (!C || (ldc % 4 == 0 && aligned16(C))), "gemm_p: N/ldc %% 4, alignment");
    SEGMM_REQUIRE(lda2 % 64 == 0 && ldb2 % 64 == 0 && aligned16(a_planes) && aligned16(b_planes), "gemm_p: plane strides %% 64 halves / alignment");
    SEGMM_REQUIRE(!c_planes || (c_hdr && ldc2 % 64 == 0 && N % 32 == 0 && aligned16(c_planes)), "gemm_p: plane output needs a header, N %% 32, ldc2 %% 64");
    SEGMM_REQUIRE(!a_f32 || (aligned16(a_f32) && ldaf % 4 == 0), "gemm_p: fp32 copy of A alignment");
    SEGMM_REQUIRE(!b_f32 || (aligned16(b_f32) && ldbf % 4 == 0), "gemm_p: fp32 copy of B alignment");
    SEGMM_REQUIRE(!bias || aligned16(bias), "gemm_p: bias alignment");
    SEGMM_REQUIRE(!residual || (aligned16(residual) && ldr % 4 == 0 && res_period > 0), "gemm_p: residual alignment/period");
    SEGMM_REQUIRE(activation >= 0 && activation <= 4, "gemm_p: activation %d", activation);
    SEGMM_REQUIRE(activation == 0 || activation == EPI_RELU || (aux && aligned16(aux) && ldaux % 4 == 0), "gemm_p: activation needs aux");
    SEGMM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_p: dropout p=%f", drop_p);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.row_scale = row_scale;
    g.residual = residual; g.ldr = ldr; g.res_period = res_period > 0 ? res_period : 1;
    g.aux = aux; g.ldaux = ldaux; g.epi = activation;
    g.drop = make_drop(drop_p, seed, site);
    g.amax_out = nullptr;
    g.C = C; g.ldc = ldc;
    PGemmX q;
    memset(&q, 0, sizeof(q));
    q.A.p = (const _Float16*)a_planes; q.A.ld2 = lda2; q.A.hdr = a_hdr; q.A.f32 = a_f32; q.A.ldf = ldaf;
    q.B.p = (const _Float16*)b_planes; q.B.ld2 = ldb2; q.B.hdr = b_hdr; q.B.f32 = b_f32; q.B.ldf = ldbf;
    q.Cp = (_Float16*)c_planes; q.ldc2 = ldc2; q.c_hdr = c_hdr; q.c_scale_in = c_scale_in; q.write_c = C ? ((write_c & 1) != 0) : 0;
    q.repair = (write_c & 2) != 0;
    SEGMM_REQUIRE(!q.repair || (layout == 0 && c_planes && c_hdr && !q.write_c && !aux && !residual && !row_scale && !accumulate),
                  "gemm_p: a repair launch (write_c bit 1) rewrites the planes of a planes-only NT output: c_planes + c_hdr, no fp32 C, no aux / residual");
#ifdef SEGMM_GEMM_PROBE
    static const int pl_flags = getenv("SEGMM_PL_FLAGS") ? atoi(getenv("SEGMM_PL_FLAGS")) : 0;          // timing ablations (results wrong; probe builds only)
#else
    constexpr int pl_flags = 0;
#endif
    q.dbg = pl_flags;
    hipStream_t s = (hipStream_t)stream;
    if (splits < 1) splits = 1;
    if (layout == 0) {
        SEGMM_REQUIRE(splits == 1, "gemm_p: the NT form has no split-K");
        const size_t a_ext = ((size_t)(M - 1) * lda2 + 2 * (size_t)K) * 2, b_ext = ((size_t)(N - 1) * ldb2 + 2 * (size_t)K) * 2;
        SEGMM_REQUIRE(a_ext < (1ull << 32) && b_ext < (1ull << 32), "gemm_p: operand view above the 4 GiB buffer window");
        q.A.bytes = (uint32_t)a_ext; q.B.bytes = (uint32_t)b_ext;
        if (accumulate) {
            SEGMM_REQUIRE(!residual && C, "gemm_p: accumulate and residual are exclusive");
            g.residual = C; g.ldr = ldc; g.res_period = M;
        }
        const int pl_var = knob(K_PL_VAR);          // 8: gemm_pl_nt8 (round 3); anything else: the round-2 fallback kernel for every launch
#ifdef SEGMM_STAMPS
        q.stamps = g_segmm_stamps;
#endif
        if (pl_var == 8 || pl_var == 4 || pl_var == 44) {          // round-3 / round-6 forms: 16x16x32 MFMA, register epilogue (gemm_planes8.h, gemm_planes4.h)
            // every epilogue access is a buffer operation with a 32-bit offset whose top bit masks out-of-range columns
            const long long lim = 1ll << 31;
            const bool fits = !row_scale && (long long)M * ldc * 4 < lim && (!aux || (long long)M * ldaux * 4 < lim) &&
                              (!residual || (long long)(res_period < M ? res_period : M) * ldr * 4 < lim) &&
                              (!c_planes || (long long)M * ldc2 * 2 < lim) &&
                              !(residual && (activation == EPI_DGELU || activation == EPI_DRELU));      // one extra operand per element
            // round 6: 128 x 256 tiles, four waves, two workgroups resident per CU (results bitwise those of gemm_pl_nt8).  Taken for
            // K >= 768 and N >= 768 (config 3's input-gradient GEMMs have K = 2048 .. 3072 but N = 512: they lose on it too): in the step it wins at configs 2 / 4 / 5 (K = 768 .. 3072: +0.7 % of the step, the NT launches 191 -> 175 us) and
            // loses at config 3 (K = 512, N = 512 .. 2048: 195.7 -> 192.3 k interactions/s -- a short k-loop leaves the second resident
            // workgroup less to hide, and gemm_pl_nt8's 192-wide tiles fit N = 512 better) -- profiles/r6/ab_kernel_generations.txt.
            // SEGMM_PL_VAR=44 forces it for every launch (tests, A/B)
            if (fits && (pl_var == 44 || (pl_var == 4 && K >= 768 && N >= 768))) {
                g.nbm = (M + P4_BM - 1) / P4_BM; g.nbn = (N + P4_BN - 1) / P4_BN;
                hipLaunchKernelGGL(gemm_pl_nt4, dim3(g.nbm * g.nbn), dim3(256), 0, s, g, q);
                LAUNCH_CHECK();
                return 0;
            }
            if (fits) {
                // tile width 64 NJ: the one that needs the fewest CU-rounds.  Measured tile times at K = 768 (stand-alone, round 3):
                // 60.7 / 49 / 38 us for NJ = 4 / 3 / 2 = 15 + 11.4 NJ us -- a narrower tile carries the same A traffic, prologue and
                // barrier count for less work -- so it only wins when it saves a whole round or more; near-ties go to the widest tile
                const int nj_env = knob(K_PL_NJ);
                int best = 4;
                if (nj_env >= 2 && nj_env <= 4) best = nj_env;
                else {
                    double best_cost = 0.0;
                    const int ncu = num_cus(), bm = (M + PBM - 1) / PBM;
                    for (int nj = 4; nj >= 2; --nj) {
                        const long long tiles = (long long)bm * ((N + 64 * nj - 1) / (64 * nj));
                        const double cost = (double)((tiles + ncu - 1) / ncu) * (15.0 + 11.4 * nj * (K / 768.0));
                        if (nj == 4 || cost < 0.93 * best_cost) { best = nj; best_cost = cost; }
                    }
                }
                g.nbm = (M + PBM - 1) / PBM; g.nbn = (N + 64 * best - 1) / (64 * best);
                const dim3 grid(g.nbm * g.nbn);
                // (the tile width of a repair launch need not match the first launch's: both write whole elements)
                if (best == 4) hipLaunchKernelGGL(gemm_pl_nt8<4>, grid, dim3(512), 0, s, g, q);
                else if (best == 3) hipLaunchKernelGGL(gemm_pl_nt8<3>, grid, dim3(512), 0, s, g, q);
                else hipLaunchKernelGGL(gemm_pl_nt8<2>, grid, dim3(512), 0, s, g, q);
                LAUNCH_CHECK();
                return 0;
            }
        }
        SEGMM_REQUIRE(!q.repair, "gemm_p: this launch does not run on gemm_pl_nt8, which alone has the repair mode");
        // the round-2 NT kernels below judge and fall back for operand A only: B must be an exact-split operand (weight planes)
        SEGMM_REQUIRE(!b_f32, "gemm_p NT: this launch (row-scaled output, residual with a d-activation, or an extent >= 2^31) runs on the "
                              "round-2 kernel, which has no fp32 fallback for operand B -- pass B without an fp32 copy (exact-split planes)");
        g.nbm = (M + PBM - 1) / PBM; g.nbn = (N + PBN - 1) / PBN;
        hipLaunchKernelGGL(gemm_pl_nt, dim3(g.nbm * g.nbn), dim3(512), 0, s, g, q);
        LAUNCH_CHECK();
        return 0;
    }
    // TN: A planes [K][2M], B planes [K][2N]; split-K over blockIdx.z
    SEGMM_REQUIRE(M % 32 == 0 && N % 32 == 0, "gemm_p TN: M, N %% 32 (M=%d N=%d)", M, N);
    SEGMM_REQUIRE(!c_planes && !bias && !row_scale && activation == 0 && drop_p == 0.f, "gemm_p TN: no epilogue besides residual/accumulate");
    {
        const size_t a_ext = ((size_t)(K - 1) * lda2 + 2 * (size_t)M) * 2, b_ext = ((size_t)(K - 1) * ldb2 + 2 * (size_t)N) * 2;
        SEGMM_REQUIRE(a_ext + 4096 < (1ull << 32) && b_ext + 4096 < (1ull << 32), "gemm_p: operand view above the 4 GiB buffer window");
        q.A.bytes = (uint32_t)a_ext; q.B.bytes = (uint32_t)b_ext;
    }
    g.nbm = (M + PBM - 1) / PBM; g.nbn = (N + PBN - 1) / PBN;
    const int ktiles = (K + 31) / 32;
    if (splits > ktiles) splits = ktiles;
    if (splits > 1) {
        SEGMM_REQUIRE(workspace && aligned16(workspace), "gemm_p: split-K needs a workspace");
        SEGMM_REQUIRE(!residual, "gemm_p: split-K supports no residual");
        const int tps = (ktiles + splits - 1) / splits;
        splits = (ktiles + tps - 1) / tps;
        g.k_per_split = tps * 32;
        g.C = workspace; g.ldc = N; g.slab_stride = (long long)M * N;
        q.colsum_ws = workspace + (size_t)splits * M * N;          // the caller sizes the workspace splits * (M * N + M) with colsum_out
    } else {
        g.k_per_split = ktiles * 32;
        g.slab_stride = 0;
        if (accumulate) {
            SEGMM_REQUIRE(!residual, "gemm_p: accumulate and residual are exclusive");
            g.residual = C; g.ldr = ldc; g.res_period = M;
        }
    }
    q.colsum_out = colsum_out;
    // round-6 form (gemm_planes4.h): whole 128 x 256 tiles; round-3 form (gemm_planes8.h): whole 256 x 256 tiles; both: plain or
    // split-K stores (no accumulate into C), 32-bit output offsets.  Everything else: the round-2 kernel
    const int tn_var = knob(K_TN_VAR);
    const bool small_out = !g.residual && (long long)M * (splits > 1 ? N : ldc) * 4 < (1ll << 31);
    // Stand-alone gemm_pl_tn4 is the faster kernel (+4 .. 37 %, profiles/r6/gemm4_standalone.txt).  In the step (config 2, same-box A/Bs,
    // profiles/r6/ab_kernel_generations.txt) it LOSES 2.1-2.5 % when every weight gradient takes it -- its 504 workgroups hold one slot of
    // EVERY CU at once, and the main stream's input-gradient GEMM beside it then runs one workgroup per CU next to a weight-gradient
    // workgroup instead of two of its own -- and WINS +0.5 .. 1.5 % when only the few-tile matrices take it (768 x 768: 9 tiles of
    // 256 x 256, 28 splits: short k-loops whose prologue / epilogue the second resident workgroup hides; the row kernels of the step's
    // tail fit beside them).  Default (TN_VAR 8): gemm_pl_tn4 for at most 9 tiles of 256 x 256 of width >= 768 and for matrices that are whole 128-
    // but not 256-row tiles, gemm_pl_tn8 otherwise; 4: gemm_pl_tn4 wherever it fits; 88: gemm_pl_tn8 wherever it fits
    const bool fits8 = M % PBM == 0 && N % PBN == 0 && small_out, fits4 = M % P4_BM == 0 && N % P4_BN == 0 && small_out;
    // (few-tile AND at least 768 wide: config 3's d = 512 matrices -- 4 .. 12 tiles -- read 187.9 k/s with gemm_pl_tn4 against 191.7 k/s)
    const bool few = (long long)((M + PBM - 1) / PBM) * ((N + PBN - 1) / PBN) <= 9 && M >= 768 && N >= 768;
    const bool tn4 = fits4 && (tn_var == 4 || (tn_var == 8 && (few || !fits8)));
    const bool tn8 = !tn4 && fits8 && (tn_var == 8 || tn_var == 88 || tn_var == 4);
    if (tn4) {
        g.nbm = M / P4_BM; g.nbn = N / P4_BN;
        hipLaunchKernelGGL(gemm_pl_tn4, dim3(g.nbm * g.nbn, 1, splits), dim3(256), 0, s, g, q);
    } else if (tn8) hipLaunchKernelGGL(gemm_pl_tn8, dim3(g.nbm * g.nbn, 1, splits), dim3(512), 0, s, g, q);
    else hipLaunchKernelGGL(gemm_pl_tn, dim3(g.nbm * g.nbn, 1, splits), dim3(512), 0, s, g, q);
    LAUNCH_CHECK();
    if (splits > 1) {          // one combine launch for the slabs AND the folded column sums
        const long long n4 = (long long)M * (N / 4);
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, s, (const float*)workspace, splits, (long long)M * N, C, ldc, M, N, accumulate,
                           (const float*)q.colsum_ws, colsum_out);
        LAUNCH_CHECK();
    }
    return 0;
}

int segmm_split_p32(const float* x, int64_t rows, int cols, int ld, uint16_t* planes, int ld2, float* hdr, int mode, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && planes && hdr && rows >= 0 && cols > 0 && cols % 32 == 0 && ld % 4 == 0 && ld >= cols && ld2 % 64 == 0 && ld2 >= 2 * cols
                  && aligned16(x) && aligned16(planes), "split_p32: cols %% 32, ld %% 4, ld2 %% 64, alignment");
    SEGMM_REQUIRE(mode == 0 || mode == 1, "split_p32: mode %d", mode);
    if (rows == 0) return 0;
    const long long n4 = rows * (cols / 4);
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_p32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, cols, ld, (_Float16*)planes, ld2, hdr, mode);
    LAUNCH_CHECK();
    return 0;
}

int segmm_wsplit_p32(const float* flat, const void* desc, int n_mats, int n_tiles, float* hdr, uint16_t* wpl, uint16_t* wTpl,
                     segmm_stream_t stream) {
    SEGMM_REQUIRE(flat && desc && hdr && wpl && n_mats > 0 && n_tiles > 0 && aligned16(flat) && aligned16(wpl), "wsplit_p32: arguments");
    static_assert(AMAX_SLOTS == 64 * 4, "wabsmax: one wave per slot");
    hipLaunchKernelGGL(wabsmax_kernel, dim3(64, n_mats), dim3(256), 0, (hipStream_t)stream, flat, (const WMat*)desc, hdr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(wsplit_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, flat, (const WMat*)desc, n_mats, hdr,
                       (_Float16*)wpl, (_Float16*)wTpl);
    LAUNCH_CHECK();
    return 0;
}

int segmm_split_p32_transpose(const float* x, int R, int Cc, int ld, uint16_t* planes, int ld2, const float* hdr, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && planes && hdr && R > 0 && Cc > 0 && R % 32 == 0 && ld >= Cc && ld2 % 64 == 0 && ld2 >= 2 * R && aligned16(planes),
                  "split_p32_transpose: R %% 32, ld2 %% 64, alignment");
    hipLaunchKernelGGL(split_p32_transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, R, Cc, ld,
                       (_Float16*)planes, ld2, hdr);
    LAUNCH_CHECK();
    return 0;
}

int segmm_absmax(const float* x, int64_t rows, int cols, int ld, float* out, int nparts, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && out && rows >= 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols && aligned16(x), "absmax: cols/ld %% 4, alignment");
    SEGMM_REQUIRE(nparts >= 1 && nparts <= 1024, "absmax: 1..1024 partials");
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, cols, ld, out);
    LAUNCH_CHECK();
    return 0;
}

int segmm_split2h(const float* x, uint16_t* planes, int64_t n, int64_t pstride, const float* amax, int namax, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && planes && amax && namax >= 1 && namax <= 1024 && n % 4 == 0 && pstride % 4 == 0 && aligned16(x) && (((uintptr_t)planes) & 7) == 0, "split2h: n/stride %% 4, alignment, partial maxima");
    if (n <= 0) return 0;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splith_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)planes, (long long)n, (long long)pstride, amax, namax);
    LAUNCH_CHECK();
    return 0;
}

int segmm_split2h_transpose(const float* x, int R, int Cc, int ld, uint16_t* planes, int64_t pstride, const float* amax, int namax, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && planes && amax && namax >= 1 && namax <= 1024 && R > 0 && Cc > 0 && ld >= Cc, "split2h_transpose: bad args");
    hipLaunchKernelGGL((split3_transpose_kernel<true>), dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, R, Cc, ld,
                       (__bf16*)planes, (long long)pstride, amax, namax);
    LAUNCH_CHECK();
    return 0;
}

int segmm_split3(const float* x, uint16_t* planes, int64_t n, int64_t pstride, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && planes && n % 4 == 0 && pstride % 4 == 0 && aligned16(x) && (((uintptr_t)planes) & 7) == 0, "split3: n/stride %% 4 and alignment");
    if (n <= 0) return 0;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split3_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)planes, (long long)n, (long long)pstride);
    LAUNCH_CHECK();
    return 0;
}

int segmm_split3_transpose(const float* x, int R, int Cc, int ld, uint16_t* planes, int64_t pstride, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && planes && R > 0 && Cc > 0 && ld >= Cc, "split3_transpose: bad args");
    hipLaunchKernelGGL((split3_transpose_kernel<false>), dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, R, Cc, ld,
                       (__bf16*)planes, (long long)pstride, (const float*)nullptr, 0);
    LAUNCH_CHECK();
    return 0;
}

static int ln_fwd_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                         int64_t rows, int d, float eps, float drop_p, uint64_t seed, uint32_t site, float* amax,
                         uint16_t* planes, int ld2, float* hdr, const float* scale_in, const float* dot_w, const float* dot_b, float* dot_out,
                         segmm_stream_t stream) {
    SEGMM_REQUIRE(x && gamma && beta && (y || (planes && hdr)) && mean && rstd, "layernorm_fwd: null pointer");
    SEGMM_REQUIRE(!dot_w || (dot_out && aligned16(dot_w)), "layernorm_fwd_dot: head weight / output");
    PLANE_OUT_CHECK("layernorm_fwd", d);
    SEGMM_REQUIRE(d > 0 && d % 4 == 0 && d <= 256 * ROW_MAXV, "layernorm_fwd: d=%d unsupported", d);
    SEGMM_REQUIRE(aligned16(x) && (!y || aligned16(y)) && aligned16(gamma) && aligned16(beta), "layernorm_fwd: alignment");
    if (rows <= 0) return 0;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const DropCfg dc = make_drop(drop_p, seed, site);
#define LNF(V) hipLaunchKernelGGL((layernorm_fwd_kernel<V>), grid, block, 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, (long long)rows, d, eps, dc, amax, plane_out(planes, ld2, hdr, scale_in), dot_w, dot_b, dot_out)
    if (d <= 256) LNF(1); else if (d <= 512) LNF(2); else if (d <= 768) LNF(3); else if (d <= 1024) LNF(4); else LNF(8);
#undef LNF
    LAUNCH_CHECK();
    return 0;
}

int segmm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int64_t rows, int d, float eps, float drop_p, uint64_t seed, uint32_t site, float* amax,
                        uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream) {
    return ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, d, eps, drop_p, seed, site, amax, planes, ld2, hdr, scale_in, nullptr, nullptr, nullptr, stream);
}

int segmm_layernorm_fwd_dot(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int64_t rows, int d, float eps, float drop_p, uint64_t seed, uint32_t site, float* amax,
                            uint16_t* planes, int ld2, float* hdr, const float* scale_in, const float* dot_w, const float* dot_b,
                            float* dot_out, segmm_stream_t stream) {
    SEGMM_REQUIRE(dot_w && dot_out, "layernorm_fwd_dot: null pointer");
    return ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, d, eps, drop_p, seed, site, amax, planes, ld2, hdr, scale_in, dot_w, dot_b, dot_out, stream);
}

// Workgroups of a LayerNorm-backward launch = partial rows of its column sums.  A workgroup is four waves that walk rows; what is resident
// per CU follows the kernel's registers at the row width (layernorm_bwd_kernel<V>: 82 / 116 / 152 / 174 / 372 registers for V = 1 / 2 / 3 /
// 4 / 8).  The launch asks for exactly ONE round of resident workgroups: with 1024 workgroups at d = 768 (768 resident) the last 256 ran
// alone, one per CU, at 40 % of the bandwidth -- 133 -> 110 us at 51 200 rows (4.7 -> 5.7 TB/s), 59 -> 47 us at 20 480 rows, +0.8 % of the
// step (profiles/r6/ln_parts_ab.txt).  Knob LN_BWD_PARTS overrides (A/B).
static int ln_bwd_cap(int d) {
    const int k = knob(K_LN_BWD_PARTS);
    if (k >= 64) return k;
    return d <= 512 ? 1024 : d <= 768 ? 768 : d <= 1024 ? 512 : 256;
}
int segmm_layernorm_bwd_parts(int64_t rows, int d) {
    int64_t b = (rows + 3) / 4;
    const int cap = ln_bwd_cap(d);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

static int ln_bwd_launch(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                         int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                         float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, float* part_pos, int parts, segmm_stream_t stream,
                         const float* dy_col = nullptr) {
    SEGMM_REQUIRE(dy && x && mean && rstd && gamma && dx && part_dgamma && part_dbeta, "layernorm_bwd: null pointer");
    PLANE_OUT_CHECK("layernorm_bwd", d);
    SEGMM_REQUIRE(d > 0 && d % 4 == 0 && d <= 256 * ROW_MAXV, "layernorm_bwd: d=%d unsupported", d);
    SEGMM_REQUIRE((dy_col ? aligned16(dy_col) : aligned16(dy)) && aligned16(x) && aligned16(dx) && aligned16(gamma) && (!dx_drop || aligned16(dx_drop)), "layernorm_bwd: alignment");
    const DropCfg dy_ = make_drop(drop_y_p, seed, drop_y_site), db_ = make_drop(drop_b_p, seed, drop_b_site);
#define LNB(V) hipLaunchKernelGGL((layernorm_bwd_kernel<V>), dim3(parts), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, part_dsum, (long long)rows, d, dy_, db_, amax, plane_out(planes, ld2, hdr, scale_in), part_pos, dy_col)
    if (d <= 256) LNB(1); else if (d <= 512) LNB(2); else if (d <= 768) LNB(3); else if (d <= 1024) LNB(4); else LNB(8);
#undef LNB
    LAUNCH_CHECK();
    return 0;
}

int segmm_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                        int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                        float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream) {
    return ln_bwd_launch(dy, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, part_dsum, rows, d, drop_y_p, drop_y_site, drop_b_p,
                         drop_b_site, seed, amax, planes, ld2, hdr, scale_in, nullptr, segmm_layernorm_bwd_parts(rows, d), stream);
}

int segmm_layernorm_bwd_outer(const float* dy_row, const float* dy_col, const float* x, const float* mean, const float* rstd, const float* gamma,
                              float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                              int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                              float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream) {
    SEGMM_REQUIRE(dy_row && dy_col, "layernorm_bwd_outer: null pointer");
    return ln_bwd_launch(dy_row, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, part_dsum, rows, d, drop_y_p, drop_y_site, drop_b_p,
                         drop_b_site, seed, amax, planes, ld2, hdr, scale_in, nullptr, segmm_layernorm_bwd_parts(rows, d), stream, dy_col);
}

// workgroups (4 waves each) of the per-position form: the wave stride 4 * parts must be a multiple of the sequence length
int segmm_layernorm_bwd_pos_parts(int64_t rows, int period, int d) {
    if (period <= 0 || rows <= 0 || rows % period) return 0;
    int g = period % 4 == 0 ? 4 : (period % 2 == 0 ? 2 : 1);
    const int unit = period / g;                     // parts must be a multiple of this
    int64_t want = (rows + 3) / 4;
    const int cap = ln_bwd_cap(d);
    if (want > cap) want = cap;
    const int64_t k = want / unit;
    return k < 1 ? (unit <= 1024 ? unit : 0) : (int)(k * unit);
}

int segmm_layernorm_bwd_pos(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                            float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                            int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                            float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, float* part_pos, int period,
                            segmm_stream_t stream) {
    const int parts = segmm_layernorm_bwd_pos_parts(rows, period, d);
    SEGMM_REQUIRE(part_pos && parts > 0 && (4 * parts) % period == 0, "layernorm_bwd_pos: %lld rows with period %d have no per-position grid", (long long)rows, period);
    return ln_bwd_launch(dy, x, mean, rstd, gamma, dx, dx_drop, part_dgamma, part_dbeta, part_dsum, rows, d, drop_y_p, drop_y_site, drop_b_p,
                         drop_b_site, seed, amax, planes, ld2, hdr, scale_in, part_pos, parts, stream);
}

int segmm_colsum_pos(const float* part, int n_rows, int period, int d, float* out, segmm_stream_t stream) {
    SEGMM_REQUIRE(part && out && n_rows > 0 && period > 0 && d > 0 && d % 4 == 0 && aligned16(part) && aligned16(out), "colsum_pos: arguments");
    hipLaunchKernelGGL(colsum_pos_kernel, dim3((unsigned)((d / 4 + 15) / 16), (unsigned)period), dim3(256), 0, (hipStream_t)stream, part, n_rows, period, d, out);
    LAUNCH_CHECK();
    return 0;
}

int segmm_colsum_chunks(int64_t M) {
    int64_t c = (M + 15) / 16;          // >= 16 rows per chunk; up to 256 chunks so that a 768-column sum still fills the chip
    if (c > 256) c = 256;
    if (c < 1) c = 1;
    return (int)c;
}

int segmm_colsum3(const float* X0, const float* X1, const float* X2, int ld, int64_t M, int N, float* out0, float* out1,
                  float* out2, float* workspace, segmm_stream_t stream) {
    SEGMM_REQUIRE(X0 && out0 && workspace && (!X1 == !out1) && (!X2 == !out2) && (X1 || !X2), "colsum3: pointers");
    SEGMM_REQUIRE(N % 4 == 0 && ld % 4 == 0 && aligned16(X0) && aligned16(out0) && (!X1 || (aligned16(X1) && aligned16(out1))) &&
                  (!X2 || (aligned16(X2) && aligned16(out2))) && aligned16(workspace), "colsum3: N/ld %% 4 and alignment");
    if (M <= 0) return 0;
    const int nmat = X2 ? 3 : (X1 ? 2 : 1);
    const int chunks = segmm_colsum_chunks(M);
    const int rpc = (int)((M + chunks - 1) / chunks);
    Colsum3 a;
    a.X[0] = X0; a.X[1] = X1; a.X[2] = X2; a.out[0] = out0; a.out[1] = out1; a.out[2] = out2;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial3_kernel, dim3((N + 255) / 256, chunks, nmat), dim3(256), 0, s, a, ld, (long long)M, N, workspace,
                       rpc > 0 ? rpc : 1);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final3_kernel, dim3((N + 255) / 256, 1, nmat), dim3(256), 0, s, (const float*)workspace, chunks, N, a);
    LAUNCH_CHECK();
    return 0;
}

int segmm_colsum(const float* X, int ld, const float* w, int64_t M, int N, float* out, int accumulate,
                 float* workspace, segmm_stream_t stream) {
    SEGMM_REQUIRE(X && out && workspace, "colsum: null pointer");
    SEGMM_REQUIRE(N > 0 && N % 4 == 0 && ld % 4 == 0 && aligned16(X) && aligned16(workspace) && aligned16(out), "colsum: N/ld %% 4 / alignment (N=%d ld=%d)", N, ld);
    const int chunks = segmm_colsum_chunks(M);
    const int rpc = (int)((M + chunks - 1) / chunks);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, s, X, ld, w, (long long)M, N,
                       workspace, rpc > 0 ? rpc : 1);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (const float*)workspace, chunks, N, out,
                       accumulate);
    LAUNCH_CHECK();
    return 0;
}

// input planes of the attention kernels (round 5): the views of an empty key block are aliased to the other block's, like attn_fill
static int attn_fill_in(AttnArgs& a, const segmm_attn_planes_t* pl, int B, int H, int dh, int Lq, int La, int Lb, const char* who) {
    if (!(pl && pl->qa_in)) return 0;
    const uint16_t *ka = pl->ka_in, *va = pl->va_in, *kb = pl->kb_in, *vb = pl->vb_in;
    const float *ha = pl->hdr_ka_in, *hb = pl->hdr_kb_in;
    int lda2 = pl->ldka2_in, ldb2 = pl->ldkb2_in;
    if (La == 0) { ka = kb; va = vb; ha = hb; lda2 = ldb2; }
    if (Lb == 0) { kb = ka; vb = va; hb = ha; ldb2 = lda2; }
    const uint16_t *qa = pl->qa_in, *qb = pl->qb_in ? pl->qb_in : pl->qa_in;
    SEGMM_REQUIRE(ka && va && kb && vb && pl->hdr_q_in && ha && hb, "%s: input planes need every view and header", who);
    SEGMM_REQUIRE(pl->ldq2_in % 64 == 0 && lda2 % 64 == 0 && ldb2 % 64 == 0, "%s: input plane strides %% 64", who);
    SEGMM_REQUIRE(aligned16(qa) && aligned16(qb) && aligned16(ka) && aligned16(va) && aligned16(kb) && aligned16(vb), "%s: input plane alignment", who);
    AttnInPlanes& in = a.in;
    in.Qa = (const _Float16*)qa; in.Qb = (const _Float16*)qb; in.ldq2 = pl->ldq2_in;
    in.hdr_q = pl->hdr_q_in; in.hdr_ka = ha; in.hdr_kb = hb;
    in.ldka2 = lda2; in.ldkb2 = ldb2;
    const uintptr_t bA = (uintptr_t)ka < (uintptr_t)va ? (uintptr_t)ka : (uintptr_t)va, bB = (uintptr_t)kb < (uintptr_t)vb ? (uintptr_t)kb : (uintptr_t)vb;
    const size_t offKa = (uintptr_t)ka - bA, offVa = (uintptr_t)va - bA, offKb = (uintptr_t)kb - bB, offVb = (uintptr_t)vb - bB;
    // extent of a view: last row's start + the head columns' planes (2 bytes x 2 terms per column, rounded up to a whole block)
    const size_t headb = (size_t)((H * dh + 31) / 32) * 128;
    const size_t extA = ((size_t)B * (La ? La : Lb) - 1) * (size_t)lda2 * 2 + headb;
    const size_t extB = ((size_t)B * (Lb ? Lb : La) - 1) * (size_t)ldb2 * 2 + headb;
    const size_t bytesA = (offKa > offVa ? offKa : offVa) + extA, bytesB = (offKb > offVb ? offKb : offVb) + extB;
    const size_t bytesQ = ((size_t)B * Lq - 1) * (size_t)pl->ldq2_in * 2 + headb;
    SEGMM_REQUIRE(bytesA < (1ull << 31) && bytesB < (1ull << 31) && bytesQ < (1ull << 31), "%s: an input plane view exceeds the 2 GiB buffer-addressing window", who);
    in.baseA = (const _Float16*)bA; in.baseB = (const _Float16*)bB;
    in.offKa = (uint32_t)offKa; in.offVa = (uint32_t)offVa; in.offKb = (uint32_t)offKb; in.offVb = (uint32_t)offVb;
    in.bytesA = (uint32_t)bytesA; in.bytesB = (uint32_t)bytesB; in.bytesQ = (uint32_t)bytesQ;
    return 0;
}

int segmm_attn_fwd(int B, int H, int dh, int Lq, int La, int Lb, const float* Qa, const float* Qb, int ldq,
                   const float* Ka, const float* Va, int ldka, const float* Kb, const float* Vb, int ldkb,
                   const uint8_t* mq, const uint8_t* mka, const uint8_t* mkb, float* O, int ldo, float* lse,
                   float drop_p, uint64_t seed, uint32_t site, float* amax_o, const segmm_attn_planes_t* pl, segmm_stream_t stream) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    int rc = attn_fill(a, B, H, dh, Lq, La, Lb, Qa, Qb, ldq, Ka, Va, ldka, Kb, Vb, ldkb, mq, mka, mkb, drop_p, seed, site, pl && pl->qa_in);
    if (rc) return rc;
    SEGMM_REQUIRE(O && lse && aligned16(O) && ldo % 4 == 0, "attn_fwd: output pointer/alignment");
    a.O = O; a.ldo = ldo; a.lse = lse; a.amax_o = amax_o;
    if (pl && pl->o) {
        SEGMM_REQUIRE(pl->hdr_o && pl->ldo2 % 64 == 0 && aligned16(pl->o) && (H * dh) % 32 == 0, "attn_fwd: plane output needs a header, ld2 %% 64, width %% 32");
        a.po_o = plane_out(pl->o, pl->ldo2, pl->hdr_o, pl->sin_o);
    }
    rc = attn_fill_in(a, pl, B, H, dh, Lq, La, Lb, "attn_fwd");
    if (rc) return rc;
    ATTN_DISPATCH(attn_launch_fwd, dh, a, (hipStream_t)stream);
}

int segmm_attn_bwd(int B, int H, int dh, int Lq, int La, int Lb, const float* Qa, const float* Qb, int ldq,
                   const float* Ka, const float* Va, int ldka, const float* Kb, const float* Vb, int ldkb,
                   const uint8_t* mq, const uint8_t* mka, const uint8_t* mkb, const float* lse, const float* O, int ldo,
                   const float* dO, int lddo, float* Dvec, float* dQa, float* dQb, int lddq, float* dKa, float* dVa, int lddka,
                   float* dKb, float* dVb, int lddkb, float drop_p, uint64_t seed, uint32_t site,
                   float* amax_q, float* amax_ka, float* amax_kb, int phase, const segmm_attn_planes_t* pl, segmm_stream_t stream) {
    SEGMM_REQUIRE(phase >= 0 && phase <= 6, "attn_bwd: phase %d (0 all, 1 D, 2 dQ, 3 dK/dV, 4 fused dQ+dK+dV, 5 / 6 fused, key block a / b only)", phase);
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    const bool planes_in = pl && pl->qa_in;
    SEGMM_REQUIRE(!planes_in || phase >= 4, "attn_bwd: input planes need the fused backward (phase 4 / 5 / 6)");
    int rc = attn_fill(a, B, H, dh, Lq, La, Lb, Qa, Qb, ldq, Ka, Va, ldka, Kb, Vb, ldkb, mq, mka, mkb, drop_p, seed, site, planes_in);
    if (rc) return rc;
    rc = attn_fill_in(a, pl, B, H, dh, Lq, La, Lb, "attn_bwd");
    if (rc) return rc;
    SEGMM_REQUIRE(!planes_in || (dh % 16 == 0 && dh <= 48), "attn_bwd: input planes are built for head dims 16, 32, 48 (got %d)", dh);
    SEGMM_REQUIRE(lse && O && dO && Dvec, "attn_bwd: null pointer");
    SEGMM_REQUIRE((La == 0 || (dQa && dKa && dVa)) && (Lb == 0 || (dQb && dKb && dVb)), "attn_bwd: null gradient pointer of a non-empty key block");
    if (La == 0) { dQa = nullptr; dKa = dKb; dVa = dVb; lddka = lddkb; }      // dQ of an empty block is not written
    if (Lb == 0) { dQb = nullptr; dKb = dKa; dVb = dVa; lddkb = lddka; }
    SEGMM_REQUIRE(lddo % 4 == 0 && ldo % 4 == 0 && lddq % 4 == 0 && lddka % 4 == 0 && lddkb % 4 == 0, "attn_bwd: leading dims %% 4");
    SEGMM_REQUIRE(aligned16(O) && aligned16(dO) && (!dQa || aligned16(dQa)) && (!dQb || aligned16(dQb)) && aligned16(dKa) && aligned16(dVa) && aligned16(dKb) && aligned16(dVb), "attn_bwd: alignment");
    a.lse = (float*)lse; a.O = (float*)O; a.ldo = ldo; a.dO = dO; a.lddo = lddo; a.Dvec = Dvec;
    SEGMM_REQUIRE(((size_t)B * Lq + 16) * lddo * 4 < (1ull << 32), "attn_bwd: dO exceeds the 4 GiB buffer-addressing window");
    a.do_bytes = (uint32_t)((((size_t)B * Lq - 1) * lddo + (size_t)H * dh) * 4);
    a.dQa = dQa; a.dQb = dQb; a.lddq = lddq; a.dKa = dKa; a.dVa = dVa; a.lddka = lddka; a.dKb = dKb; a.dVb = dVb; a.lddkb = lddkb;
    a.amax_q = amax_q; a.amax_ka = amax_ka; a.amax_kb = amax_kb;
    if (pl && phase >= 4 && (pl->dqa || pl->dqb || pl->dka || pl->dkb)) {          // plane outputs: fused backward only
        SEGMM_REQUIRE(pl->lddq2 % 64 == 0 && pl->lddka2 % 64 == 0 && pl->lddkb2 % 64 == 0, "attn_bwd: plane strides %% 64");
        SEGMM_REQUIRE((!(pl->dqa || pl->dqb) || pl->hdr_q) && (!pl->dka || (pl->dva && pl->hdr_ka)) && (!pl->dkb || (pl->dvb && pl->hdr_kb)),
                      "attn_bwd: plane outputs need their headers (and dK and dV planes come in pairs)");
        a.dQap = (_Float16*)pl->dqa; a.dQbp = (_Float16*)pl->dqb; a.lddq2 = pl->lddq2;
        a.dKap = (_Float16*)pl->dka; a.dVap = (_Float16*)pl->dva; a.lddka2 = pl->lddka2;
        a.dKbp = (_Float16*)pl->dkb; a.dVbp = (_Float16*)pl->dvb; a.lddkb2 = pl->lddkb2;
        a.hdr_q = pl->hdr_q; a.hdr_ka = pl->hdr_ka; a.hdr_kb = pl->hdr_kb;
        a.sin_q = pl->sin_q; a.sin_ka = pl->sin_ka; a.sin_kb = pl->sin_kb;
        if (La == 0) { a.dQap = nullptr; a.dKap = a.dKbp; a.dVap = a.dVbp; a.lddka2 = a.lddkb2; a.hdr_ka = a.hdr_kb; a.sin_ka = a.sin_kb; }
        if (Lb == 0) { a.dQbp = nullptr; a.dKbp = a.dKap; a.dVbp = a.dVap; a.lddkb2 = a.lddka2; a.hdr_kb = a.hdr_ka; a.sin_kb = a.sin_ka; }
        SEGMM_REQUIRE((pl->flags & ~3) == 0, "attn_bwd: plane flags %d", pl->flags);
        a.pflags = pl->flags;
    } else {
        SEGMM_REQUIRE(!pl || pl->flags == 0, "attn_bwd: plane flags need the fused backward (phase 4) with plane outputs");
    }
    ATTN_DISPATCH(attn_launch_bwd, dh, a, phase, (hipStream_t)stream);
}

namespace segmm {
__global__ __launch_bounds__(256) void site_fixup_kernel(float* h0, float* h1, float* h2, float* h3, float* stats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* h = wave == 0 ? h0 : wave == 1 ? h1 : wave == 2 ? h2 : h3;
    if (!h) return;
    if (!site_planes_ok(h, h[0], lane)) {          // (also a site written with no scale yet: h[0] == 0)
        const float sx = f16_scale_of(site_amax(h, lane));
        if (lane == 0) { h[0] = sx; ((volatile unsigned int*)h)[1] = 0u; h[2] = 1.f; if (stats) atomicAdd(stats, 1.0f); }
    }
}
}  // namespace segmm
int segmm_site_fixup(float* hdr0, float* hdr1, float* hdr2, float* hdr3, float* stats, segmm_stream_t stream) {
    if (!hdr0 && !hdr1 && !hdr2 && !hdr3) return 0;
    hipLaunchKernelGGL(site_fixup_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, hdr0, hdr1, hdr2, hdr3, stats);
    LAUNCH_CHECK();
    return 0;
}

int segmm_rowdot(const float* x, int ld, const float* w, const float* bias, float* out, int64_t rows, int d,
                 int accumulate, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && w && out && d % 4 == 0 && ld % 4 == 0 && aligned16(x) && aligned16(w), "rowdot: pointer/alignment");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ld, w, bias, out,
                       (long long)rows, d, accumulate);
    LAUNCH_CHECK();
    return 0;
}

int segmm_rowscale_bcast(const float* g, const float* w, float* dx, int ld, int64_t rows, int d, int accumulate,
                         segmm_stream_t stream) {
    SEGMM_REQUIRE(g && w && dx && d % 4 == 0 && ld % 4 == 0 && aligned16(dx) && aligned16(w), "rowscale_bcast: pointer/alignment");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowscale_bcast_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, w, dx, ld,
                       (long long)rows, d, accumulate);
    LAUNCH_CHECK();
    return 0;
}

int segmm_rowdot_pair(const float* a, int lda, const float* b, int ldb, float* out, int64_t rows, int d,
                      int accumulate, segmm_stream_t stream) {
    SEGMM_REQUIRE(a && b && out && d % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned16(a) && aligned16(b), "rowdot_pair: pointer/alignment");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowdot_pair_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out,
                       (long long)rows, d, accumulate);
    LAUNCH_CHECK();
    return 0;
}

int segmm_rowscale_mat(const float* g, const float* X, int ldx, float* out, int ldo, int64_t rows, int d,
                       int accumulate, segmm_stream_t stream) {
    SEGMM_REQUIRE(g && X && out && d % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && aligned16(X) && aligned16(out), "rowscale_mat: pointer/alignment");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowscale_mat_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, X, ldx, out, ldo,
                       (long long)rows, d, accumulate);
    LAUNCH_CHECK();
    return 0;
}

int segmm_vecsum(const float* v, int64_t n, float* out, int accumulate, segmm_stream_t stream) {
    SEGMM_REQUIRE(v && out, "vecsum: null pointer");
    hipLaunchKernelGGL(vecsum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, (long long)n, out, accumulate);
    LAUNCH_CHECK();
    return 0;
}

int segmm_embed_id_vid(const int64_t* item_id, const float* table, int dhalf, const float* frame_w,
                       const float* frame_b, const float* pe, const float* frame_pos, float* out, int B, int S, int64_t n_rows,
                       segmm_stream_t stream) {
    SEGMM_REQUIRE(item_id && table && frame_w && frame_b && out && n_rows > 0, "embed_id_vid: null pointer / empty table");
    SEGMM_REQUIRE(dhalf % 4 == 0 && aligned16(table) && aligned16(frame_w) && aligned16(frame_b) && aligned16(pe) && aligned16(out), "embed_id_vid: d/2 %% 4 / alignment");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(embed_id_vid_kernel, dim3(B * S), dim3(64), 0, (hipStream_t)stream, (const long long*)item_id, table, dhalf,
                       frame_w, frame_b, pe, frame_pos, out, B, S, (long long)n_rows);
    LAUNCH_CHECK();
    return 0;
}

int segmm_embed_id_usr(const int64_t* user_id, const float* table, int d, const float* pe, float* out, int B,
                       int64_t n_rows, segmm_stream_t stream) {
    SEGMM_REQUIRE(user_id && table && out && n_rows > 0 && d % 4 == 0 && aligned16(table) && aligned16(pe) && aligned16(out), "embed_id_usr: pointer/alignment");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(embed_id_usr_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const long long*)user_id, table, d, pe, out, B, (long long)n_rows);
    LAUNCH_CHECK();
    return 0;
}

int segmm_embed_id_bwd(const float* dpre, int tokens_per_row, int ld, int col0, int width, const int32_t* order,
                       const int64_t* ids, float* dtable, int B, int64_t n_rows, segmm_stream_t stream) {
    SEGMM_REQUIRE(dpre && order && ids && dtable && n_rows > 0, "embed_id_bwd: null pointer / empty table");
    SEGMM_REQUIRE(width % 4 == 0 && ld % 4 == 0 && col0 % 4 == 0 && aligned16(dpre) && aligned16(dtable), "embed_id_bwd: alignment");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(embed_id_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dpre, tokens_per_row, ld, col0, width,
                       (const int*)order, (const long long*)ids, dtable, B, (long long)n_rows);
    LAUNCH_CHECK();
    return 0;
}

int segmm_argsort_ids(const int64_t* ids, int n, int32_t* order, segmm_stream_t stream) {
    SEGMM_REQUIRE(ids && order && n >= 0 && n <= ARGSORT_MAX, "argsort_ids: 0 <= n <= %d (n=%d)", ARGSORT_MAX, n);
    if (n == 0) return 0;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    hipLaunchKernelGGL(argsort_ids_kernel, dim3(1), dim3(np2 >= 2048 ? 1024 : (np2 >= 128 ? np2 / 2 : 64)), (size_t)np2 * 8, (hipStream_t)stream,
                       (const long long*)ids, n, np2, (int*)order);
    LAUNCH_CHECK();
    return 0;
}

int segmm_argsort_ids_ws(const int64_t* ids, int n, int32_t* order, uint64_t* keys_ws, segmm_stream_t stream) {
    SEGMM_REQUIRE(ids && order && n >= 0 && n <= (1 << 24), "argsort_ids_ws: 0 <= n <= 2^24 (n=%d)", n);
    if (n <= ARGSORT_MAX) return segmm_argsort_ids(ids, n, order, stream);
    SEGMM_REQUIRE(keys_ws && (((uintptr_t)keys_ws) & 7u) == 0, "argsort_ids_ws: %d ids need a workspace of the next power of two x 8 bytes", n);
    int np2 = 2 * ARGSORT_MAX;
    while (np2 < n) np2 <<= 1;
    hipStream_t s = (hipStream_t)stream;
    static bool optin = false;          // 64 KB of dynamic LDS: at the default limit, the opt-in is harmless
    if (!optin) { (void)hipFuncSetAttribute((const void*)argsort_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ARGSORT_MAX * 8); optin = true; }
    unsigned long long* keys = (unsigned long long*)keys_ws;
    hipLaunchKernelGGL(argsort_keys_init_kernel, dim3(np2 / 256), dim3(256), 0, s, (const long long*)ids, n, np2, keys);
    hipLaunchKernelGGL(argsort_chunk_kernel, dim3(np2 / ARGSORT_MAX), dim3(1024), (size_t)ARGSORT_MAX * 8, s, keys, 2);
    for (int k = 2 * ARGSORT_MAX; k <= np2; k <<= 1) {
        for (int j = k >> 1; j >= ARGSORT_MAX; j >>= 1)
            hipLaunchKernelGGL(argsort_global_step_kernel, dim3(np2 / 512), dim3(256), 0, s, keys, np2, k, j);
        hipLaunchKernelGGL(argsort_chunk_kernel, dim3(np2 / ARGSORT_MAX), dim3(1024), (size_t)ARGSORT_MAX * 8, s, keys, k);
    }
    hipLaunchKernelGGL(argsort_keys_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned long long*)keys, n, (int*)order);
    LAUNCH_CHECK();
    return 0;
}

int segmm_label_stats_unpack(const float* gathered, int G, int B, float* v_all, float* v2_all, float* norms, segmm_stream_t stream) {
    SEGMM_REQUIRE(gathered && v_all && v2_all && norms && G > 0 && B > 0, "label_stats_unpack: arguments");
    int blocks = (G * B + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(label_stats_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gathered, G, B, v_all, v2_all, norms);
    LAUNCH_CHECK();
    return 0;
}

int segmm_zero_rows(float* table, int width, const int64_t* ids, int n, int64_t n_rows, segmm_stream_t stream) {
    SEGMM_REQUIRE(table && ids && width > 0 && width % 4 == 0 && aligned16(table) && n_rows > 0, "zero_rows: arguments");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, table, width, (const long long*)ids, (long long)n_rows);
    LAUNCH_CHECK();
    return 0;
}

int segmm_pe_grad(const float* dpre, int ld, int B, int S, int d, float* dpe, int accumulate, segmm_stream_t stream) {
    SEGMM_REQUIRE(dpre && dpe && d % 4 == 0 && ld % 4 == 0 && aligned16(dpre) && aligned16(dpe), "pe_grad: pointer/alignment");
    if (S <= 0) return 0;
    hipLaunchKernelGGL(pe_grad_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, dpre, ld, B, S, d, dpe, accumulate);
    LAUNCH_CHECK();
    return 0;
}

int segmm_label_stats(const int64_t* gt, int B, int S, int rewritten, float* v, float* v2, float* norms,
                      segmm_stream_t stream) {
    SEGMM_REQUIRE(gt && v && v2 && norms && B > 0 && S > 0, "label_stats: bad args");
    hipLaunchKernelGGL(label_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long long*)gt, B, S, rewritten, v, v2, norms);
    LAUNCH_CHECK();
    return 0;
}

int segmm_loss_finish(const float* parts, int B, const float* coef, float* losses, float* total, const float* dlogits, int64_t n_dl,
                      float* site_scale, const float* gain, int n_sites, float* gmax, int target, segmm_stream_t stream) {
    SEGMM_REQUIRE(parts && coef && losses && total && B > 0, "loss_finish: arguments");
    SEGMM_REQUIRE(!dlogits || (site_scale && gain && gmax && n_sites >= 0 && n_dl >= 0 && target >= 2 && target <= 15), "loss_finish: scale arguments");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, parts, B, coef, losses, total, dlogits, (long long)n_dl,
                       site_scale, gain, n_sites, gmax, target);
    LAUNCH_CHECK();
    return 0;
}

int segmm_loss_fwd_bwd(int B, int S, const float* logits, const int64_t* gt, const float* bias_w,
                       const float* bias_b, const float* exposure, const float* coef, const int* enabled,
                       int rewritten_ce, int rewritten_kl, int use_mask, const float* norms, const float* v_all,
                       const float* v2_all, int Bg, float* logits_out, float* dlogits, float* parts,
                       segmm_stream_t stream) {
    SEGMM_REQUIRE(logits && gt && exposure && coef && enabled && norms && v_all && v2_all && logits_out && parts, "loss: null pointer");
    SEGMM_REQUIRE(S >= 1 && S <= 64, "loss: S=%d must be in [1,64]", S);
    SEGMM_REQUIRE((bias_w == nullptr) == (bias_b == nullptr), "loss: bias_w/bias_b must come together");
    if (B <= 0) return 0;
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.S = S; a.logits = logits; a.gt = (const long long*)gt; a.bias_w = bias_w; a.bias_b = bias_b;
    a.exposure = exposure;
    for (int k = 0; k < L_NPART; ++k) { a.coef[k] = coef[k]; a.enabled[k] = enabled[k]; }
    a.gt_rewritten_for_ce = rewritten_ce; a.gt_rewritten_for_kl = rewritten_kl;
    a.use_mask = use_mask; a.norms = norms;
    a.v_all = v_all; a.v2_all = v2_all; a.Bg = Bg;
    a.logits_out = logits_out; a.dlogits = dlogits; a.parts = parts;
    hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}

int segmm_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int step, segmm_stream_t stream) {
    SEGMM_REQUIRE(p && g && m && v && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw: pointer/alignment");
    SEGMM_REQUIRE(step >= 1 || step == -1, "adamw: step=%d (>= 1, or -1: the device-side step state)", step);
    if (n <= 0) return 0;
    const double bc1 = step > 0 ? 1.0 - pow((double)beta1, step) : 1.0, bc2 = step > 0 ? 1.0 - pow((double)beta2, step) : 1.0;
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), step < 0 ? (const StepState*)segmm_step_current() : (const StepState*)nullptr);
    LAUNCH_CHECK();
    return 0;
}

int segmm_adamw_table(float* p, const float* g, float* m, float* v, int64_t n_rows, int width, const int64_t* ids, int n_ids,
                      uint32_t* flags, float lr, float beta1, float beta2, float eps, float weight_decay, int step, int phase,
                      segmm_stream_t stream) {
    SEGMM_REQUIRE(p && m && v && flags && (ids || n_ids == 0) && aligned16(p) && aligned16(m) && aligned16(v), "adamw_table: pointer/alignment");
    SEGMM_REQUIRE(phase == 0 || (phase == 1 && g && aligned16(g)), "adamw_table: phase %d (0: rows without a gradient, 1: the listed rows; needs g)", phase);
    SEGMM_REQUIRE(width > 0 && width % 4 == 0 && n_rows >= 0 && n_ids >= 0, "adamw_table: width %% 4, sizes");
    SEGMM_REQUIRE(step >= 1 || step == -1, "adamw_table: step=%d (>= 1, or -1: the device-side step state)", step);
    if (n_rows == 0) return 0;
    const double bc1 = step > 0 ? 1.0 - pow((double)beta1, step) : 1.0, bc2 = step > 0 ? 1.0 - pow((double)beta2, step) : 1.0;
    hipStream_t s = (hipStream_t)stream;
    const int w4 = width / 4;
    if (phase == 0) {
        if (n_ids > 0) {
            hipLaunchKernelGGL(table_mark_kernel, dim3((n_ids + 255) / 256), dim3(256), 0, s, (const long long*)ids, n_ids, (long long)n_rows, flags);
            LAUNCH_CHECK();
        }
        long long blocks = (n_rows * w4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(adamw_table_rest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, m, v, (long long)n_rows, w4, (const unsigned int*)flags,
                           lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), step < 0 ? (const StepState*)segmm_step_current() : (const StepState*)nullptr);
        LAUNCH_CHECK();
        return 0;
    }
    if (n_ids == 0) return 0;
    hipLaunchKernelGGL(adamw_table_rows_kernel, dim3((n_ids + 3) / 4), dim3(256), 0, s, p, g, m, v, (long long)n_rows, w4, (const long long*)ids, n_ids,
                       flags, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), step < 0 ? (const StepState*)segmm_step_current() : (const StepState*)nullptr);
    LAUNCH_CHECK();
    return 0;
}

int segmm_step_state_bytes(void) { return (int)sizeof(StepState); }
int segmm_step_bind(void* state) {
    SEGMM_REQUIRE(!state || aligned16(state), "step_bind: the state must be 16-byte aligned device memory");
    g_segmm_step = (StepState*)state;
    return 0;
}
int segmm_step_set(uint64_t seed, int step, float beta1, float beta2, segmm_stream_t stream) {
    SEGMM_REQUIRE(step >= 0, "step_set: step=%d", step);
    StepState* st = segmm_step_current();
    SEGMM_REQUIRE(st, "step_set: no step state");
    hipLaunchKernelGGL(step_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, st, (uint32_t)seed, (uint32_t)(seed >> 32) & 0x7fffffffu, step, beta1, beta2);
    LAUNCH_CHECK();
    return 0;
}
int segmm_step_advance(float beta1, float beta2, segmm_stream_t stream) {
    StepState* st = segmm_step_current();
    SEGMM_REQUIRE(st, "step_advance: no step state");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, st, beta1, beta2);
    LAUNCH_CHECK();
    return 0;
}
int segmm_step_get(uint64_t* seed, int* step, float* bias_corrections, segmm_stream_t stream) {
    const StepState* st = segmm_step_current();
    SEGMM_REQUIRE(st, "step_get: no step state");
    StepState h;
    hipError_t e = hipMemcpyAsync(&h, st, sizeof(StepState), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    SEGMM_CHECK_HIP(e);
    if (seed) *seed = ((uint64_t)h.seed_hi << 32) | h.seed_lo;
    if (step) *step = h.step;
    if (bias_corrections) { bias_corrections[0] = h.bc1; bias_corrections[1] = h.bc2_sqrt; }
    return 0;
}

int segmm_dropout_mult(float* out, int64_t n, float p, uint64_t seed, uint32_t site, segmm_stream_t stream) {
    SEGMM_REQUIRE(out && p >= 0.f && p < 1.f, "dropout_mult: bad args");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_mult_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, (long long)n, make_drop(p, seed, site));
    LAUNCH_CHECK();
    return 0;
}


int segmm_rank_leave(const float* x, int ldx, const int64_t* gt, const int32_t* perm, int B, int S, int masked, int seq_valid,
                     int32_t* ranks, int32_t* hist, segmm_stream_t stream) {
    SEGMM_REQUIRE(x && gt && ranks && hist && S > 0 && ldx >= S, "rank_leave: null pointer / S / ldx");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(rank_leave_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, (const long long*)gt,
                       (const int*)perm, B, S, masked, seq_valid, (int*)ranks, (int*)hist);
    LAUNCH_CHECK();
    return 0;
}

int segmm_auc_counts(const float* score, const int8_t* label, const int64_t* seg_off, int n_seg, int64_t* out,
                     segmm_stream_t stream) {
    SEGMM_REQUIRE(score && label && seg_off && out, "auc_counts: null pointer");
    if (n_seg <= 0) return 0;
    hipLaunchKernelGGL(auc_counts_kernel, dim3(n_seg), dim3(256), 0, (hipStream_t)stream, score, (const signed char*)label,
                       (const long long*)seg_off, (long long*)out);
    LAUNCH_CHECK();
    return 0;
}

int segmm_survival(const float* interest, int ld, const int64_t* gt, float* surv, int8_t* label, int B, int S,
                   segmm_stream_t stream) {
    SEGMM_REQUIRE(interest && gt && surv && label && S > 0 && ld >= S, "survival: null pointer / S / ld");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(survival_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, interest, ld, (const long long*)gt,
                       surv, (signed char*)label, B, S);
    LAUNCH_CHECK();
    return 0;
}

int segmm_scales_update(const float* arena, const int32_t* site_idx, int n_rows, float* site_scale, float* stats, int target,
                        float* gain, const float* gmax, segmm_stream_t stream) {
    SEGMM_REQUIRE(arena && site_idx && site_scale && stats && n_rows >= 0 && target >= 2 && target <= 15 && (!gain == !gmax), "scales_update: arguments");
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(scales_update_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, arena, (const int*)site_idx, n_rows,
                       site_scale, stats, target, gain, gmax);
    LAUNCH_CHECK();
    return 0;
}

int segmm_gather_l1(const float* table, int64_t n_lines, int D, const int64_t* idx, int64_t rows, int normalize, float* out,
                    uint8_t* mask, float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream) {
    PLANE_OUT_CHECK("gather_l1", D);
    SEGMM_REQUIRE(table && idx && out && n_lines > 0 && D > 0 && D % 4 == 0 && aligned16(table) && aligned16(out), "gather_l1: pointer / D %% 4 / alignment");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(gather_l1_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, table, (long long)n_lines, D,
                       (const long long*)idx, (long long)rows, normalize, out, mask, amax, plane_out(planes, ld2, hdr, scale_in));
    LAUNCH_CHECK();
    return 0;
}

int segmm_segment_weighted_sum(const float* pred, const float* weight, const int64_t* duration, int64_t rows, int S, float* out,
                               segmm_stream_t stream) {
    SEGMM_REQUIRE(pred && out && S > 0, "segment_weighted_sum: null pointer / S");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(segment_weighted_sum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pred, weight,
                       (const long long*)duration, (long long)rows, S, out);
    LAUNCH_CHECK();
    return 0;
}

int segmm_pool_tokens(const float* U, int Lu, const float* V, int Lv, float* out, int B, int d, int bins, segmm_stream_t stream) {
    SEGMM_REQUIRE(U && V && out && Lu > 0 && Lv > 0 && bins > 0 && d % 4 == 0 && aligned16(U) && aligned16(V) && aligned16(out),
                  "pool_tokens: pointer/alignment");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pool_tokens_kernel, dim3((unsigned)(B * bins)), dim3(256), 0, (hipStream_t)stream, U, Lu, V, Lv, out, d, bins);
    LAUNCH_CHECK();
    return 0;
}

int segmm_pool_tokens_bwd(const float* dOut, float* dU, int Lu, float* dV, int Lv, int B, int d, int bins, segmm_stream_t stream) {
    SEGMM_REQUIRE(dOut && dU && dV && Lu > 0 && Lv > 0 && bins > 0 && d % 4 == 0 && aligned16(dOut) && aligned16(dU) && aligned16(dV),
                  "pool_tokens_bwd: pointer/alignment");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pool_tokens_bwd_kernel, dim3((unsigned)(B * (Lu + Lv))), dim3(256), 0, (hipStream_t)stream, dOut, dU, Lu, dV, Lv,
                       d, bins);
    LAUNCH_CHECK();
    return 0;
}

#ifdef SEGMM_GEMM_TRACE
int segmm_debug_gemm_trace(unsigned long long* host, int n) {      // debug build only (-DSEGMM_GEMM_TRACE): phase timestamps
    SEGMM_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(segmm::g_gemm_trace), sizeof(unsigned long long) * (size_t)n));
    return 0;
}
#endif

#ifdef SEGMM_ATT_TRACE
int segmm_debug_attn_trace(unsigned long long* host, int n) {      // debug build only (-DSEGMM_ATT_TRACE)
    SEGMM_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(segmm::g_att_trace), sizeof(unsigned long long) * (size_t)n));
    return 0;
}
#endif

int segmm_probe_mfma_rate(int workgroups, int iters, float* scratch, double* flops_out, segmm_stream_t stream) {
    SEGMM_REQUIRE(workgroups > 0 && iters > 0 && scratch, "probe_mfma_rate: workgroups/iters > 0 and a scratch float");
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, scratch, iters, 12345u);
    LAUNCH_CHECK();
    if (flops_out) *flops_out = (double)workgroups * 8.0 * iters * 96.0 * 16384.0;          // 96 MFMAs of 16 x 16 x 32 per wave and iteration
    return 0;
}

/* ---- recorded launch sequences (include/segmm_hip.h: "Recorded launch sequences") */
int segmm_bias_grad(const float* dl, int B, int S, float* g_bias_weight, float* g_bias_bias, segmm_stream_t stream) {
    SEGMM_REQUIRE(dl && g_bias_weight && g_bias_bias && B > 0 && S > 0, "bias_grad: null pointer / empty");
    hipLaunchKernelGGL(bias_grad_kernel, dim3((S + 63) / 64), dim3(64), 0, (hipStream_t)stream, dl, B, S, g_bias_weight, g_bias_bias);
    LAUNCH_CHECK();
    return 0;
}
int segmm_focal_relabel(int64_t* gt, int64_t n, segmm_stream_t stream) {
    SEGMM_REQUIRE(gt && n >= 0, "focal_relabel: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(focal_relabel_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long*)gt, (long long)n);
    LAUNCH_CHECK();
    return 0;
}
int segmm_rand_uniform(float* out, int64_t n, uint64_t seed, uint32_t site, segmm_stream_t stream) {
    SEGMM_REQUIRE(out && n >= 0, "rand_uniform: null pointer");
    if (n == 0) return 0;
    const long long q = (n + 1) / 2;
    hipLaunchKernelGGL(rand_uniform_kernel, dim3((unsigned)((q + 255) / 256 > 4096 ? 4096 : (q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (long long)n, make_drop(0.5f, seed, site));
    LAUNCH_CHECK();
    return 0;
}
int segmm_rand_ids(int64_t* out, int64_t n, int64_t lo, int64_t hi, uint64_t seed, uint32_t site, segmm_stream_t stream) {
    SEGMM_REQUIRE(out && n >= 0 && hi > lo && hi - lo < (1ll << 31), "rand_ids: null pointer / empty or too wide a range");
    if (n == 0) return 0;
    const long long q = (n + 1) / 2;
    hipLaunchKernelGGL(rand_ids_kernel, dim3((unsigned)((q + 255) / 256 > 1024 ? 1024 : (q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long*)out, (long long)n, (long long)lo,
                       (long long)hi, make_drop(0.5f, seed, site));
    LAUNCH_CHECK();
    return 0;
}
int segmm_rand_perm_rows(float* out, int rows, int S, uint64_t seed, uint32_t site, segmm_stream_t stream) {
    SEGMM_REQUIRE(out && rows >= 0 && S >= 1 && S <= 64, "rand_perm_rows: null pointer / S = %d (1 .. 64)", S);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rand_perm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, out, rows, S, make_drop(0.5f, seed, site));
    LAUNCH_CHECK();
    return 0;
}

int segmm_fill_zero(void* p, int64_t bytes, segmm_stream_t stream) {
    SEGMM_REQUIRE(p && bytes >= 0, "fill_zero: null pointer / negative size");
    if (bytes == 0) return 0;
    const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    SEGMM_REQUIRE(e == hipSuccess, "fill_zero: hipMemsetAsync: %s", hipGetErrorString(e));
    return 0;
}

int segmm_copy_bytes(void* dst, const void* src, int64_t bytes, segmm_stream_t stream) {
    SEGMM_REQUIRE(dst && src && bytes >= 0, "copy_bytes: null pointer / negative size");
    if (bytes == 0) return 0;
    const hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    SEGMM_REQUIRE(e == hipSuccess, "copy_bytes: hipMemcpyAsync: %s", hipGetErrorString(e));
    return 0;
}

#include "cmd_dispatch.inc"

int segmm_cmd_op_count(void) { return SEGMM_N_CMD_OPS; }
const char* segmm_cmd_op_name(int op) { return (op >= 0 && op < SEGMM_N_CMD_OPS) ? segmm_cmd_names[op] : nullptr; }

int segmm_run_phase(const segmm_phase_t* ph, const segmm_stream_t* streams, int n_streams, void* const* events) {
    SEGMM_REQUIRE(ph && ph->n_cmds >= 0 && (ph->n_cmds == 0 || ph->cmds), "run_phase: null descriptor");
    SEGMM_REQUIRE(ph->kind >= 0 && ph->kind < SEGMM_PHASE_KINDS, "run_phase: phase kind %d", ph->kind);
    SEGMM_REQUIRE(n_streams >= 0 && n_streams <= SEGMM_MAX_STREAMS && (n_streams == 0 || streams), "run_phase: %d streams (<= %d)", n_streams, SEGMM_MAX_STREAMS);
    for (int i = 0; i < ph->n_cmds; ++i) {
        const segmm_cmd_t& c = ph->cmds[i];
        SEGMM_REQUIRE(c.stream >= 0 && c.stream < SEGMM_MAX_STREAMS, "run_phase: command %d names stream slot %d (0 main, 1 side, 2 auxiliary)", i, c.stream);
        if (c.op == SEGMM_OP_FORK || c.op == SEGMM_OP_JOIN) {
            const bool fork = c.op == SEGMM_OP_FORK;
            SEGMM_REQUIRE(c.stream >= 1 && c.stream < n_streams && events && events[2 * (c.stream - 1)] && events[2 * (c.stream - 1) + 1],
                          "run_phase: command %d, a %s of stream slot %d, needs that stream and its two events", i, fork ? "fork" : "join", c.stream);
            hipEvent_t ev = (hipEvent_t)events[2 * (c.stream - 1) + (fork ? 0 : 1)];
            hipStream_t from = (hipStream_t)(fork ? streams[0] : streams[c.stream]), to = (hipStream_t)(fork ? streams[c.stream] : streams[0]);
            hipError_t e = hipEventRecord(ev, from);
            if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
            SEGMM_REQUIRE(e == hipSuccess, "run_phase: command %d (%s): %s", i, fork ? "fork" : "join", hipGetErrorString(e));
            continue;
        }
        SEGMM_REQUIRE(c.op >= 0 && c.op < SEGMM_N_CMD_OPS, "run_phase: command %d has op %d (0 .. %d)", i, c.op, SEGMM_N_CMD_OPS - 1);
        SEGMM_REQUIRE(c.stream < n_streams || c.stream == 0, "run_phase: command %d is for stream slot %d, %d streams given", i, c.stream, n_streams);
        const int rc = segmm_cmd_dispatch(c.op, c.a, n_streams > 0 ? streams[c.stream] : nullptr);
        if (rc != 0) return rc;          // (segmm_last_error holds the failing entry point's message)
    }
    return 0;
}
#define SEGMM_PHASE_ENTRY(fn, KIND)                                                                                                   \
    int fn(const segmm_phase_t* ph, const segmm_stream_t* streams, int n_streams, void* const* events) {                              \
        SEGMM_REQUIRE(ph && ph->kind == KIND, #fn ": the descriptor is not a " #KIND " phase");                                       \
        return segmm_run_phase(ph, streams, n_streams, events);                                                                       \
    }
SEGMM_PHASE_ENTRY(segmm_step_begin, SEGMM_PHASE_STEP_BEGIN)
SEGMM_PHASE_ENTRY(segmm_embed_fwd, SEGMM_PHASE_EMBED_FWD)
SEGMM_PHASE_ENTRY(segmm_layer_fwd, SEGMM_PHASE_LAYER_FWD)
SEGMM_PHASE_ENTRY(segmm_head_loss_fwd, SEGMM_PHASE_HEAD_LOSS_FWD)
SEGMM_PHASE_ENTRY(segmm_head_loss_bwd, SEGMM_PHASE_HEAD_LOSS_BWD)
SEGMM_PHASE_ENTRY(segmm_layer_bwd, SEGMM_PHASE_LAYER_BWD)
SEGMM_PHASE_ENTRY(segmm_embed_bwd, SEGMM_PHASE_EMBED_BWD)
SEGMM_PHASE_ENTRY(segmm_step_tail, SEGMM_PHASE_STEP_TAIL)
#undef SEGMM_PHASE_ENTRY

}  // extern "C"
