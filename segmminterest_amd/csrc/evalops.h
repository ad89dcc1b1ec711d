// the callers on either side of the training step, as integer/byte kernels.
//
//  * on-device evaluation (my_evaluation.py:73-231, SegRec/main.py:101-117): the rank of the leave segment is an
//    INTEGER count ("how many positions sort before the target, ties by the lower index" -- what np.argsort gives),
//    AUC is the Mann-Whitney statistic as INTEGER pair counts (2*less + equal), so both are bit-exact against the
//    numpy/sklearn reference for the same float inputs; nothing here rounds.  The reference moves the whole dev set
//    to the host every 30 steps to do this with numpy.
//  * feature gather (dataloader_SegMM.py:271-362): per row up to S + Lt feature vectors picked from a resident
//    [n_lines, D] table by index, padded, L1-normalised (main...SegMM.py:272-273) and masked -- HBM-bound,
//    (S + Lt) * D * 4 bytes per row in, the same out, instead of a 573 KB/row host->device copy.
#pragma once
#include "common.h"

namespace segmm {

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- leave-segment rank (TOP_K_leave / TOP_K_leave_mask)
// One wave per row.  gt in {1, 0, -1, -2} (watched, leave, unwatched, padding).  view_len = #(gt == 1).
//   plain  (my_evaluation.py:180-231): row valid iff view_len < seq_valid; candidates = all S positions as they are
//   masked (my_evaluation.py:137-178): row valid iff view_len != #(gt != -2); padded positions score 1.1
// perm (optional, [B, S] int32): pred[j] = x[perm[j]], target = position of view_len in perm (the reference shuffles
// the candidates with np.random; the host supplies the very same permutations, so ranks stay bit-exact).
// rank = 1 + #{j : pred[j] < pred[t]  or  (pred[j] == pred[t] and j < t)};  ranks[b] = 0 for invalid rows.
// hist[r] += 1 (integer atomics: order-independent) for r in 1..S; hist[0] counts invalid rows.
__global__ __launch_bounds__(256) void rank_leave_kernel(const float* __restrict__ x, int ldx, const long long* __restrict__ gt,
                                                         const int* __restrict__ perm, int B, int S, int masked,
                                                         int seq_valid, int* __restrict__ ranks, int* __restrict__ hist) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const long long* g = gt + (size_t)b * S;
    int vl = 0, dur = 0;
    for (int j = lane; j < S; j += 64) {
        const long long v = g[j];
        vl += v == 1;
        dur += v != -2;
    }
    vl = wave_sum_i(vl);
    dur = wave_sum_i(dur);
    const bool valid = masked ? (vl != dur) : (vl < seq_valid);
    int rank = 0;
    if (valid) {
        const float* xr = x + (size_t)b * ldx;
        const int* pr = perm ? perm + (size_t)b * S : nullptr;
        // target position t in candidate order
        int t = vl;
        if (pr) {
            int found = S;
            for (int j = lane; j < S; j += 64)
                if (pr[j] == vl) found = min(found, j);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) found = min(found, __shfl_xor(found, o, 64));
            t = min(found, S - 1);          // (a permutation always contains view_len; a malformed one must not index out of range)
        }
        const int ts = pr ? min(max(pr[t], 0), S - 1) : t;      // source index of the target
        float tv = xr[ts];
        if (masked && g[ts] == -2) tv = 1.1f;
        int cnt = 0;
        for (int j = lane; j < S; j += 64) {
            const int js = pr ? min(max(pr[j], 0), S - 1) : j;
            float v = xr[js];
            if (masked && g[js] == -2) v = 1.1f;
            cnt += (v < tv) || (v == tv && j < t);
        }
        rank = wave_sum_i(cnt) + 1;
    }
    if (lane == 0) {
        ranks[b] = rank;
        atomicAdd(hist + rank, 1);
    }
}

// ---------------------------------------------------------------- AUC as integer pair counts, per segment
// seg_off[s] .. seg_off[s+1]: elements of segment s (a user for wuAUC; one segment = the batch for ProbAUC).
// label: 1 positive, 0 negative, anything else ignored (padding cells).  out[s] = {U2, npos, nneg} with
// U2 = sum over positives of (2 * #negatives below + #negatives equal)  =>  AUC = U2 / (2 npos nneg), exactly the
// midrank Mann-Whitney value sklearn.roc_auc_score computes.  One workgroup per segment; every thread owns a strided
// set of positives and streams the segment's negatives through LDS.
__global__ __launch_bounds__(256) void auc_counts_kernel(const float* __restrict__ score, const signed char* __restrict__ label,
                                                         const long long* __restrict__ seg_off, long long* __restrict__ out) {
    __shared__ float tile[1024];
    __shared__ unsigned long long red[3][4];
    const long long a = seg_off[blockIdx.x], e = seg_off[blockIdx.x + 1];
    unsigned long long u2 = 0, npos = 0, nneg = 0;
    for (long long base = a; base < e; base += 1024) {             // negatives of this chunk -> LDS (+inf marks "not a negative")
        const int n = (int)min((long long)1024, e - base);
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += 256) {
            float v = INFINITY;
            if (i < n && label[base + i] == 0) { v = score[base + i]; ++nneg; }
            tile[i] = v;
        }
        __syncthreads();
        for (long long p = a + threadIdx.x; p < e; p += 256) {     // every positive of the segment against the chunk
            if (label[p] != 1) continue;
            const float sp = score[p];
            unsigned int less = 0, eq = 0;
            for (int i = 0; i < n; ++i) {
                const float v = tile[i];
                less += v < sp;
                eq += v == sp;
            }
            u2 += 2ull * less + eq;
        }
    }
    for (long long p = a + threadIdx.x; p < e; p += 256) npos += label[p] == 1;
    // integer reductions: exact
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long v[3] = {u2, npos, nneg};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if (lane == 0) red[k][wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) out[(size_t)blockIdx.x * 3 + threadIdx.x] = (long long)(red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// survival[b, s] = exp(sum_{j <= s} log interest[b, j]) (main_eval_batch, my_evaluation.py:270: the running sum is the
// sequential fp32 cumsum) and the AUC label of the cell: 1 watched, 0 leave/unwatched, -1 (ignored) padding.
__global__ __launch_bounds__(256) void survival_kernel(const float* __restrict__ interest, int ld, const long long* __restrict__ gt,
                                                       float* __restrict__ surv, signed char* __restrict__ label, int B, int S) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float h = 0.f;
    for (int s = 0; s < S; ++s) {
        h += logf(interest[(size_t)b * ld + s]);
        surv[(size_t)b * S + s] = expf(h);
        const long long g = gt[(size_t)b * S + s];
        label[(size_t)b * S + s] = g == -2 ? (signed char)-1 : (g == 1 ? (signed char)1 : (signed char)0);
    }
}

// ---------------------------------------------------------------- resident-table feature gather (+ pad, mask, L1 norm)
// out[r, :] = table[idx[r], :] / (sum|table[idx[r], :]| + 1e-6) for idx[r] in [0, n_lines), zeros and mask 0 otherwise
// (negative index = padding slot of the collator, dataloader_SegMM.py:345-350).  One wave per output row.
__global__ __launch_bounds__(256) void gather_l1_kernel(const float* __restrict__ table, long long n_lines, int D,
                                                        const long long* __restrict__ idx, long long rows, int normalize,
                                                        float* __restrict__ out, unsigned char* __restrict__ mask, float* amax,
                                                        PlaneOut po) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long i = idx[r];
    const bool ok = i >= 0 && i < n_lines;
    if (mask && lane == 0) mask[r] = ok ? 1 : 0;
    float* o = out + r * D;
    const float ps = plane_scale(po);
    if (!ok) {
        for (int c = lane * 4; c < D; c += 256) {
            *(f32x4*)(o + c) = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ps > 0.f) plane_store4_pair(po.p, po.ld2, r, c, f32x4{0.f, 0.f, 0.f, 0.f}, ps);
        }
        plane_finish(po, amax, 0.f, (unsigned)r, ps, r == 0 && lane == 0);
        return;
    }
    const float* t = table + i * D;
    float s = 0.f;
    if (normalize) {
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 v = *(const f32x4*)(t + c);
            s += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
        }
        s = wave_sum(s) + 1e-6f;
    }
    float am = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = *(const f32x4*)(t + c);
        if (normalize) { v.x /= s; v.y /= s; v.z /= s; v.w /= s; }
        *(f32x4*)(o + c) = v;
        if (ps > 0.f) plane_store4_pair(po.p, po.ld2, r, c, v, ps);
        am = absmax4(am, v);
    }
    plane_finish(po, amax, am, (unsigned)r, ps, r == 0 && lane == 0);
}

// ---------------------------------------------------------------- delayed scaling: end-of-pass update of the site scales
// arena: n_rows site headers of the pass that just ended; site_idx[r] = index of row r's tensor site in site_scale (< 0:
// none).  For every row that was produced (max > 0): site_scale[idx] = the power of two s with max * s in
// [2^(target-1), 2^target) -- the scale the NEXT pass writes this site's planes with (fp16 tops out at 2^16: target 12
// leaves head-room for step-to-step growth AND shrinkage; a tensor that leaves the window is refused by the consumers, which
// take their fp32 path).  stats[0] += rows whose planes were outside the window (diagnostics / tests).
// gain != null (the BACKWARD pass): also gain[idx] = max|x| / gmax[0], the site's size relative to the largest d loss / d logits
// of the step it was measured in -- every backward tensor is linear in d loss / d logits, so the next step predicts its maximum
// as gain * (that step's gmax) (loss_finish_kernel) instead of assuming it repeats.
__global__ __launch_bounds__(256) void scales_update_kernel(const float* __restrict__ arena, const int* __restrict__ site_idx, int n_rows,
                                                            float* site_scale, float* stats, int target, float* gain, const float* gmax) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int idx = site_idx[r];
    if (idx < 0) return;
    const float* hdr = arena + (size_t)r * SITE_FLOATS;
    float m = 0.f;
    for (int i = lane; i < AMAX_SLOTS; i += 64) m = fmaxf(m, hdr[SITE_HDR + i]);
    m = wave_max(m);
    if (lane == 0) {
        const uint32_t u = __float_as_uint(m);
        if (m > 0.f && (u >> 23) != 0xff) {
            int se = (target - 1) - ((int)(u >> 23) - 127);
            se = max(-60, min(60, se));
            site_scale[idx] = __uint_as_float((uint32_t)(se + 127) << 23);
            if (gain) gain[idx] = gmax[0] > 0.f ? m / gmax[0] : 0.f;
        }
        // tensors whose planes the consumers had to refuse: overflow flag up, or the maximum below the window (site_planes_ok)
        const float s_used = hdr[0];
        if (__float_as_uint(hdr[1]) != 0u || (s_used > 0.f && s_used < 0x1p60f && m > 0.f && m * s_used < 0.25f)) atomicAdd(stats, 1.0f);
    }
}

// ---------------------------------------------------------------- SegRec weighted head (ClipRec.forward, ClipRec.py:134-198)
// out[b, i] = sum_seg pred[b, i, seg] * weight[b, i, seg] * (seg < duration[b, i]): the per-segment interest logits
// written by the inference script re-enter the recommender as segment weights ('c_interest_weight',
// SegRec/models/BaseModel.py:262-408; weight == null = all ones, duration == null = no duration mask).
// One wave per (b, i).
__global__ __launch_bounds__(256) void segment_weighted_sum_kernel(const float* __restrict__ pred, const float* __restrict__ weight,
                                                                   const long long* __restrict__ duration, long long rows, int S,
                                                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long dur = duration ? duration[r] : (long long)S;
    float s = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float m = j < dur ? 1.f : 0.f;
        s += pred[r * S + j] * (weight ? weight[r * S + j] : 1.f) * m;
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

}  // namespace segmm
