// Leave/skip label losses, forward + backward in one pass, one wave per interaction row.
// Follows MultiScaleTemporalDetrLeaveFocal.compute_loss (MMinterest/models/decoder_leave_focal.py:490-572)
// and the loss functions it calls (:35-97, :99-161, :163-221, :273-286), with the literal 40 generalised to S.
// No host syncs: the reference's .item()/boolean-index/python-row-loop (:175-178, :554-555) become
// per-row predicates; every cross-row normaliser (valid-row count, batch size, mask count) is an
// argument so that a data-parallel shard scales its rows by the GLOBAL count.
#pragma once
#include "common.h"

namespace segmm {

enum { L_BPR = 0, L_FOCAL, L_SCE, L_ICE, L_IKL, L_HUBER, L_HAZARD, L_MSE, L_MSE2, L_NPART };
constexpr int L_PSTRIDE = 12;   // row stride of `parts` (padded to a multiple of 4 for the float4 column sum)

struct LossArgs {
    int B, S;                       // local rows, segments (S <= 64)
    const float* logits;            // [B,S] head output (before the learnable position bias)
    const long long* gt;            // [B,S] in {1,0,-1,-2}
    const float* bias_w;            // [S] or null   learnable_bias (decoder_leave_focal.py:442-444,497-504)
    const float* bias_b;
    const float* exposure;          // [S]
    float coef[L_NPART];            // loss weight if the loss is selected, else 0 (mse/mse2: logged only)
    int enabled[L_NPART];
    int gt_rewritten_for_ce;        // 'focal' precedes interestCE in loss_type_list (in-place gt rewrite :534-535)
    int gt_rewritten_for_kl;
    int gt_rewritten_for_mse2;
    int use_mask;                   // model_cfg.mask_loss
    // global normalisers, DEVICE array [3] = {rows with view_len < S, batch rows, sum of (gt != -2)}
    // (produced by label_stats_kernel, summed over data-parallel ranks by the trainer; no host sync)
    const float* norms;
    const float* v_all;             // [Bg] view lengths of every row of the global batch (huber / mse broadcast)
    const float* v2_all;            // [Bg] (gt >= 0).sum per row (mse2)
    int Bg;
    // outputs
    float* logits_out;              // [B,S] logits incl. bias
    float* dlogits;                 // [B,S] d(total loss)/d(logits incl. bias); may be null
    float* parts;                   // [B, L_PSTRIDE] per-row, already normalised, contributions (cols >= 9 zero)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bce_logits(float x, float t) {
    return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void loss_fwd_bwd_kernel(const LossArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const int S = a.S;
    const bool in = lane < S;
    const float n_valid_bpr = fmaxf(a.norms[0], 1.f);
    const float Bg = a.norms[1];
    const float mask_sum_global = fmaxf(a.norms[2], 1.f);
    float z = in ? a.logits[(size_t)row * S + lane] : 0.f;
    if (in && a.bias_w) z += (float)(lane + 1) * a.bias_w[lane] + a.bias_b[lane];
    const int gt = in ? (int)a.gt[(size_t)row * S + lane] : -2;
    const bool m = in && gt != -2;
    const float mf = m ? 1.f : 0.f;
    const int v = (int)wave_sum((in && gt == 1) ? 1.f : 0.f);       // view length = index of the leave segment
    const int dur = (int)wave_sum(mf);
    const float p = sigmoidf_(z);
    const float logp = in ? logf(p) : 0.f;
    const float h = wave_scan_incl(logp, lane);
    const float surv = in ? expf(h) : 0.f;
    float dz = 0.f;          // d total / d z (this lane's position)
    float qs = 0.f;          // d total / d surv_j  (survival-based losses share one suffix scan)
    float part[L_NPART];
#pragma unroll
    for (int k = 0; k < L_NPART; ++k) part[k] = 0.f;

    // ---- interestBPR (compute_interest_BPR_all, :163-221)
    if (a.enabled[L_BPR] && v < S) {
        const float pos = __shfl(z, v, 64);
        const bool neg = in && lane != v;
        const float mx = wave_max(neg ? z : -INFINITY);
        const float e = neg ? expf(z - mx) : 0.f;
        const float w = e / wave_sum(e);
        const float sg = neg ? sigmoidf_(z - pos) : 0.f;
        const float A = wave_sum(sg * w);
        const float Ac = fminf(fmaxf(A, 1e-8f), 1.0f - 1e-8f);
        part[L_BPR] = -logf(Ac) / n_valid_bpr;
        const float dA = (A >= 1e-8f && A <= 1.0f - 1e-8f) ? -1.0f / (A * n_valid_bpr) : 0.f;
        const float dpos = -wave_sum(w * sg * (1.f - sg));
        float gz = neg ? w * (sg * (1.f - sg) + sg - A) : 0.f;
        if (lane == v) gz = dpos;
        dz += a.coef[L_BPR] * dA * gz;
    }
    // ---- focal (my_sigmoid_focal_loss :35-59, alpha .5, gamma 2, exposure-corrected p; sum/bsz :536-538)
    if (a.enabled[L_FOCAL]) {
        float t = (gt > 0) ? 1.f : 0.f;                  // after the in-place rewrite: >0 -> 1, -1 -> 0
        const float ex = in ? a.exposure[lane] : 1.f;
        const float ce = bce_logits(z, t);
        const float pe = p * ex;
        const float pt = pe * t + (1.f - pe) * (1.f - t);
        const float om = 1.f - pt;
        const float fl = m ? 0.5f * ce * om * om : 0.f;
        part[L_FOCAL] = wave_sum(fl) / Bg;
        if (m) {
            const float dpt = (2.f * t - 1.f) * ex * p * (1.f - p);
            dz += a.coef[L_FOCAL] * 0.5f * ((p - t) * om * om - 2.f * ce * om * dpt) / Bg;
        }
    }
    // ---- surviveCE (compute_leave_prob_CE :68-97): BCE-with-logits on exp(h_t), masked mean
    if (a.enabled[L_SCE]) {
        const float y = (gt == 1) ? 1.f : 0.f;
        const float ce = m ? bce_logits(surv, y) : 0.f;
        part[L_SCE] = wave_sum(ce) / mask_sum_global;
        if (m) qs += a.coef[L_SCE] * (sigmoidf_(surv) - y) / mask_sum_global;
    }
    // ---- interestCE / interestKL (compute_interest_leave_CE :99-161)
    if (a.enabled[L_ICE] || a.enabled[L_IKL]) {
        const float mz = wave_max(in ? z : -INFINITY);
        const float ez = in ? expf(z - mz) : 0.f;
        const float sz = wave_sum(ez);
        const float ni = ez / sz;
        const float logni = z - mz - logf(sz);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int L = which == 0 ? L_ICE : L_IKL;
            if (!a.enabled[L]) continue;
            const int rew = which == 0 ? a.gt_rewritten_for_ce : a.gt_rewritten_for_kl;
            // gt_nonleave = (gt != 0) on the (possibly rewritten) labels
            const bool nz = in && (rew ? (gt == 1 || gt == -2) : (gt != 0));
            const float n1 = wave_sum(nz ? 1.f : 0.f);
            // softmax of a 0/1 vector the way torch does it (subtract the max)
            const float gmax = n1 > 0.f ? 1.f : 0.f;
            const float eg = in ? expf((nz ? 1.f : 0.f) - gmax) : 0.f;
            const float ng = eg / wave_sum(eg);
            float c, val;
            if (a.use_mask) {
                c = m ? ng / (float)dur : 0.f;
                val = (which == 0) ? -c * logni : (m ? c * (logf(ng) - logni) : 0.f);
            } else {
                c = in ? ng : 0.f;
                val = (which == 0) ? (in ? -ng * logni : 0.f) : (in ? ng * (logf(ng) - logni) : 0.f);
            }
            part[L] = wave_sum(val) / Bg;
            const float csum = wave_sum(c);
            if (in) dz += a.coef[L] * (ni * csum - c) / Bg;
        }
    }
    // ---- huber (huber_loss :61-66 on [B] vs [B,1] => [B,B] broadcast, :540) and mse / mse2 (:552-558)
    const float hz = m ? 1.f - surv : 0.f;
    const float ssum_h = wave_sum(hz);                   // sum of masked hazard
    const float ssum = wave_sum(m ? surv : 0.f);         // sum of masked survival
    {
        const int dlast = dur > 0 ? dur - 1 : S - 1;
        const float ssum2 = wave_sum(in ? (lane == dlast ? 1.f : (m ? surv : 0.f)) : 0.f);
        float hub = 0.f, dhub = 0.f, e1 = 0.f, e2 = 0.f;
        for (int i = lane; i < a.Bg; i += 64) {
            const float vi = a.v_all[i];
            if (a.enabled[L_HUBER]) {
                const float err = ssum_h - vi, ae = fabsf(err);
                hub += ae < 1.f ? 0.5f * err * err : ae - 0.5f;
                dhub += ae < 1.f ? err : (err > 0.f ? 1.f : -1.f);
            }
            const float d1 = ssum - vi, d2 = ssum2 - a.v2_all[i];
            e1 += d1 * d1;
            e2 += d2 * d2;
        }
        const float inv = 1.0f / (Bg * Bg);
        part[L_MSE] = wave_sum(e1) * inv;
        part[L_MSE2] = wave_sum(e2) * inv;
        if (a.enabled[L_HUBER]) {
            part[L_HUBER] = wave_sum(hub) * inv;
            const float dLds = wave_sum(dhub) * inv;     // d/d(sum of masked hazard)
            if (m) qs += a.coef[L_HUBER] * (-dLds);
        }
    }
    // ---- hazard (compute_partial_likelihood_loss :273-286)
    if (a.enabled[L_HAZARD] && v < S) {
        const float ht = __shfl(hz, v, 64) + 1e-6f;
        const float R = wave_sum((in && lane >= v) ? hz : 0.f) + 1e-6f;
        part[L_HAZARD] = -(logf(ht) - logf(R)) / Bg;
        if (m) {
            float dh = 0.f;                               // d L / d hz_j
            if (lane == v) dh -= 1.f / ht;
            if (lane >= v) dh += 1.f / R;
            qs += a.coef[L_HAZARD] * (dh / Bg) * (-1.f);  // hz = 1 - surv
        }
    }
    // ---- survival chain: d surv_j / d z_k = surv_j (1 - p_k) for k <= j  => suffix sum over j >= k
    {
        const float u = in ? qs * surv : 0.f;
        const float c = wave_scan_incl(u, lane);
        const float tot = __shfl(c, 63, 64);
        if (in) dz += (1.f - p) * (tot - c + u);
    }
    if (in) {
        a.logits_out[(size_t)row * S + lane] = z;
        if (a.dlogits) a.dlogits[(size_t)row * S + lane] = dz;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < L_PSTRIDE; ++k) a.parts[(size_t)row * L_PSTRIDE + k] = k < L_NPART ? part[k] : 0.f;
    }
}

// Per-row label statistics (view length v = #(gt==1), v2 = #(gt>=0) on the possibly focal-rewritten
// labels) and the three cross-row normalisers; one workgroup, deterministic.
__global__ __launch_bounds__(1024) void label_stats_kernel(const long long* __restrict__ gt, int B, int S, int rewritten,
                                                           float* __restrict__ v, float* __restrict__ v2,
                                                           float* __restrict__ norms) {
    __shared__ float red[2][16];
    float nvalid = 0.f, msum = 0.f;
    for (int r = threadIdx.x; r < B; r += blockDim.x) {
        int c1 = 0, c2 = 0, cm = 0;
        for (int j = 0; j < S; ++j) {
            const int g = (int)gt[(size_t)r * S + j];
            c1 += g == 1;
            cm += g != -2;
            c2 += rewritten ? (g != -2) : (g >= 0);
        }
        v[r] = (float)c1; v2[r] = (float)c2;
        nvalid += c1 < S ? 1.f : 0.f;
        msum += (float)cm;
    }
    nvalid = wave_sum(nvalid); msum = wave_sum(msum);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nvalid; red[1][threadIdx.x >> 6] = msum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a0 += red[0][w]; a1 += red[1][w]; }
        norms[0] = a0; norms[1] = (float)B; norms[2] = a1;
    }
}

// losses[c] = sum_b parts[b][c] (c < 12) and total[0] = sum_c coef[c] * losses[c], one workgroup, fixed summation order:
// replaces a column-sum launch pair + a dot-product launch between the loss kernel and the backward.
// dlogits != null (plane engine, training): also gmax[0] = max |d loss / d logits| of this step, and every site with a recorded
// gain (scales_update_kernel) gets the delayed scale that puts gain * gmax at 2^target -- the backward tensors are linear in
// d loss / d logits, so a batch whose loss has collapsed (or spiked) no longer throws them out of the fp16 window.
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ parts, int B, const float* __restrict__ coef,
                                                          float* __restrict__ losses, float* __restrict__ total,
                                                          const float* __restrict__ dlogits, long long n_dl, float* site_scale,
                                                          const float* __restrict__ gain, int n_sites, float* gmax, int target) {
    __shared__ float red[16][16];
    if (dlogits) {
        __shared__ float gred[4];
        float g = 0.f;
        // one workgroup walks B*S values: 16-byte loads, four of them in flight per thread (the scalar loop was 80 dependent
        // round trips at config 2 -- most of this launch's 32 us on the step's critical path between forward and backward)
        const long long n4 = ((reinterpret_cast<uintptr_t>(dlogits) & 15) == 0) ? (n_dl >> 2) : 0;
        const f32x4* d4 = (const f32x4*)dlogits;
        long long i4 = threadIdx.x;
        for (; i4 + 3 * (long long)blockDim.x < n4; i4 += 4 * (long long)blockDim.x) {
            const f32x4 a = d4[i4], b = d4[i4 + blockDim.x], c2 = d4[i4 + 2 * blockDim.x], e = d4[i4 + 3 * blockDim.x];
            g = absmax4(absmax4(absmax4(absmax4(g, a), b), c2), e);
        }
        for (; i4 < n4; i4 += blockDim.x) g = absmax4(g, d4[i4]);
        for (long long i = 4 * n4 + threadIdx.x; i < n_dl; i += blockDim.x) g = fmaxf(g, fabsf(dlogits[i]));
        g = wave_max(g);
        if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = g;
        __syncthreads();
        g = fmaxf(fmaxf(gred[0], gred[1]), fmaxf(gred[2], gred[3]));
        if (threadIdx.x == 0) gmax[0] = g;
        const uint32_t gu = __float_as_uint(g);
        if (g > 0.f && (gu >> 23) != 0xff) {
            for (int i = threadIdx.x; i < n_sites; i += blockDim.x) {
                const float pred = gain[i] * g;
                const uint32_t u = __float_as_uint(pred);
                if (gain[i] > 0.f && pred > 0.f && (u >> 23) != 0xff && (u >> 23) != 0) {
                    int se = (target - 1) - ((int)(u >> 23) - 127);
                    se = max(-60, min(60, se));
                    site_scale[i] = __uint_as_float((uint32_t)(se + 127) << 23);
                }
            }
        }
        __syncthreads();
    }
    const int c = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    float a = 0.f;
    if (c < 12) {          // (fixed summation order per column: rows r0, r0 + 16, ... in four interleaved chains, then combined)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = r0;
        for (; r + 48 < B; r += 64) {
            a0 += parts[(size_t)r * 12 + c]; a1 += parts[(size_t)(r + 16) * 12 + c];
            a2 += parts[(size_t)(r + 32) * 12 + c]; a3 += parts[(size_t)(r + 48) * 12 + c];
        }
        for (; r < B; r += 16) a0 += parts[(size_t)r * 12 + c];
        a = (a0 + a1) + (a2 + a3);
    }
    red[r0][c] = a;
    __syncthreads();
    if (threadIdx.x < 16) {
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
        red[0][threadIdx.x] = threadIdx.x < 12 ? s : 0.f;
        if (threadIdx.x < 12) losses[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 12; ++i) t += coef[i] * red[0][i];
        total[0] = t;
    }
}


// ---------------------------------------------------------------- small kernels that keep the step free of torch ops (round 5)
// learnable_bias (decoder_leave_focal.py:497-504,649-658: logits += (s + 1) bias_weight[s] + bias_bias[s]): its two gradients,
// d bias_bias[s] = sum_b dl[b, s] (rows in index order: deterministic) and d bias_weight[s] = (s + 1) d bias_bias[s].
__global__ void bias_grad_kernel(const float* __restrict__ dl, int B, int S, float* __restrict__ gbw, float* __restrict__ gbb) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += dl[(size_t)b * S + s];
    gbb[s] = t;
    gbw[s] = (float)(s + 1) * t;
}
// focal loss first in the list: the reference rewrites the labels IN PLACE after the loss (decoder_leave_focal.py:534-535:
// gt[gt > 0] = 1; gt[gt == -1] = 0)
__global__ void focal_relabel_kernel(long long* __restrict__ gt, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long v = gt[i];
        if (v > 0) gt[i] = 1;
        else if (v == -1) gt[i] = 0;
    }
}
// noUser / noUser_SelfAtt (main_for_seq_leave_earlystop_SegMM.py:275-280: torch.rand_like user features, random user ids): draws from
// the counter hash of the dropout streams (24 random bits per float, ids by multiply-shift) -- the distribution of the reference's
// draws, not its bit stream
__global__ void rand_uniform_kernel(float* __restrict__ out, long long n, DropCfg d0) {
    const DropCfg d = drop_live(d0);
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; 2 * q < n; q += (long long)gridDim.x * blockDim.x) {
        const uint2 r = drop_rand_quad(d, (uint64_t)q);
        out[2 * q] = (float)(r.x >> 8) * (1.0f / 16777216.0f);
        if (2 * q + 1 < n) out[2 * q + 1] = (float)(r.y >> 8) * (1.0f / 16777216.0f);
    }
}
__global__ void rand_ids_kernel(long long* __restrict__ out, long long n, long long lo, long long hi, DropCfg d0) {
    const DropCfg d = drop_live(d0);
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; 2 * q < n; q += (long long)gridDim.x * blockDim.x) {
        const uint2 r = drop_rand_quad(d, (uint64_t)q);
        const unsigned long long span = (unsigned long long)(hi - lo);
        out[2 * q] = lo + (long long)(((unsigned long long)r.x * span) >> 32);
        if (2 * q + 1 < n) out[2 * q + 1] = lo + (long long)(((unsigned long long)r.y * span) >> 32);
    }
}
// noPos (encoder.py:428-429: a fresh torch.randperm(S) per row): one wave per row, lane i < S draws a 32-bit key; its position in
// the permutation is the number of lanes with a smaller (key, index) -- a uniformly random permutation
__global__ void rand_perm_rows_kernel(float* __restrict__ out, int rows, int S, DropCfg d0) {
    const DropCfg d = drop_live(d0);
    const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint2 r = drop_rand_quad(d, (uint64_t)row * 64u + (uint64_t)lane);
    const uint32_t key = lane < S ? r.x : 0xffffffffu;
    int rank = 0;
    for (int j = 0; j < S; ++j) {
        const uint32_t kj = (uint32_t)__shfl((int)key, j, 64);
        rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
    }
    if (lane < S) out[(size_t)row * S + rank] = (float)lane;          // out[row, :] is a permutation of 0 .. S-1 (as floats: frame positions)
}

}  // namespace segmm
