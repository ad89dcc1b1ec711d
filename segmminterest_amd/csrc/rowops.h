// HBM-bound row kernels of the path: L1 normalisation, LayerNorm forward/backward, column sums,
// the Linear(d,1) interest head, id-embedding gather, fused multi-tensor AdamW.  All are one-wave-
// per-row (wave64 shuffle reductions, float4 coalesced traffic, no LDS for the row reduce).
#pragma once
#include "common.h"

namespace segmm {
constexpr int ROW_MAXV = 8;      // float4 per lane kept in registers => d <= 2048

// ---------------------------------------------------------------- a1: x / (sum|x| + 1e-6)
// trainer-side normalisation, main_for_seq_leave_earlystop_SegMM.py:272-273.  If y == null only the
// reciprocal scale is written (consumed by the GEMM row_scale epilogue: the fused a1+a2 path).
__global__ __launch_bounds__(256) void l1norm_kernel(const float* __restrict__ x, float* y, float* inv_scale, long long rows, int D,
                                                     float* amax, PlaneOut po) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
        s += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
    }
    s = wave_sum(s);
    const float inv = 1.0f / (s + 1e-6f);
    if (inv_scale && lane == 0) inv_scale[row] = inv;
    const float ps = plane_scale(po);
    if (y || ps > 0.f) {          // y == null with a plane output: the normalised rows are written as planes ONLY
        const float den = s + 1e-6f;
        float am = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            f32x4 v = *(const f32x4*)(xr + c);
            v.x /= den; v.y /= den; v.z /= den; v.w /= den;
            if (y) *(f32x4*)(y + row * D + c) = v;
            if (ps > 0.f) plane_store4_pair(po.p, po.ld2, row, c, v, ps);
            am = absmax4(am, v);
        }
        plane_finish(po, amax, am, (unsigned)row, ps, row == 0 && lane == 0);      // partial maxima of |y| for the GEMM that reads y
    }
}

// The same with the row held in registers between the two passes (V float4 per lane, D <= 1024 V): one read of x instead of two --
// the second pass of the loop form came back from HBM for ~45 % of its bytes (PMC: 160 MB read per launch for 110 MB of rows).
// Element for element the arithmetic of l1norm_kernel (same summation order per lane, same wave reduction, true division).
template <int V>
__global__ __launch_bounds__(256) void l1norm_reg_kernel(const float* __restrict__ x, float* y, float* inv_scale, long long rows, int D,
                                                         float* amax, PlaneOut po) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    f32x4 v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + i * 256;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) { v[i] = ld_row4b(xr + c); s += fabsf(v[i].x) + fabsf(v[i].y) + fabsf(v[i].z) + fabsf(v[i].w); }
    }
    s = wave_sum(s);
    const float inv = 1.0f / (s + 1e-6f);
    if (inv_scale && lane == 0) inv_scale[row] = inv;
    const float ps = plane_scale(po);
    if (y || ps > 0.f) {
        const float den = s + 1e-6f;
        float am = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < D) {
                f32x4 o = v[i];
                o.x /= den; o.y /= den; o.z /= den; o.w /= den;
                if (y) st_row4b(y + row * D + c, o);
                if (ps > 0.f) plane_store4_pair(po.p, po.ld2, row, c, o, ps);
                am = absmax4(am, o);
            }
        }
        plane_finish(po, amax, am, (unsigned)row, ps, row == 0 && lane == 0);
    }
}

// ---------------------------------------------------------------- LayerNorm forward (eps 1e-12, encoder.py:39-40)
// y = LN(x) * gamma + beta, optional dropout on y (embedding, encoder.py:461,471); saves mean / rstd.
// V = float4 per lane (d <= 256 V): the row lives in 4V registers, so the register count -- and with it the
// number of rows in flight per CU, which is what hides the HBM latency of this streaming kernel -- follows d.
template <int V>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ y, float* mean_out,
                                     float* rstd_out, long long rows, int d, float eps, DropCfg drop, float* amax, PlaneOut po,
                                     const float* __restrict__ dot_w = nullptr, const float* __restrict__ dot_b = nullptr,
                                     float* __restrict__ dot_out = nullptr) {
    // dot_w != null: also dot_out[row] = y[row, :] . dot_w (+ dot_b[0]) -- the Linear(d, 1) interest head on the backbone's last
    // LayerNorm (decoder_leave_focal.py:451,596), taken from the registers that hold y instead of by a pass over y (rowdot_kernel's
    // per-lane order and wave reduction)
    drop = drop_live(drop);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    float ps = plane_scale(po);
    f32x4 v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + i * 256;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < d) { v[i] = ld_row4(xr + c); s += v[i].x + v[i].y + v[i].z + v[i].w; }
    }
    if (y == nullptr) {
        // PLANES ONLY (no fp32 copy a consumer could fall back on): the planes are written with the scale of the output's BOUND,
        // |y| <= (max|gamma| sqrt(d - 1) + max|beta|) / (1 - p) for any input row -- it cannot overflow whatever the batch, needs no
        // history and no repair pass, and sits ~sqrt(d) / 4 above the typical maximum (3 of the 9 binades of the consumers' window)
        float gm = 0.f, bm = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < d) { gm = absmax4(gm, *(const f32x4*)(gamma + c)); bm = absmax4(bm, *(const f32x4*)(beta + c)); }
        }
        ps = f16_scale_of((wave_max(gm) * sqrtf((float)d) + wave_max(bm)) * drop.scale);
    }
    const float mean = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < d) {
            const f32x4 t = v[i] - mean;
            q += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / d + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    float am = 0.f, dsum = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < d) {
            f32x4 o = (v[i] - mean) * rstd * *(const f32x4*)(gamma + c) + *(const f32x4*)(beta + c);
            if (drop.p > 0.f) o = drop_apply4(drop, ((uint64_t)row * d + c) >> 2, o);
            if (y) st_row4(y + row * d + c, o);
            if (ps > 0.f) plane_store4_pair(po.p, po.ld2, row, c, o, ps);
            am = absmax4(am, o);
            if (dot_w) {
                const f32x4 b = *(const f32x4*)(dot_w + c);
                dsum += o.x * b.x + o.y * b.y + o.z * b.z + o.w * b.w;
            }
        }
    }
    if (dot_w) {
        dsum = wave_sum(dsum);
        if (lane == 0) dot_out[row] = dot_b ? dsum + dot_b[0] : dsum;
    }
    plane_finish(po, amax, am, (unsigned)row, ps, row == 0 && lane == 0);
}

// ---------------------------------------------------------------- LayerNorm backward
// dy_eff = dy (.) dropmask (if the forward dropped y);  dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)).
// Outputs: dx; optionally dx_drop = dx (.) dropmask2 (the gradient that continues through the
// residual branch "x = res + dropout(z)": dz = dx_drop); per-workgroup partial dgamma / dbeta and, optionally,
// the partial COLUMN SUMS of the forwarded gradient (dx_drop, or dx without it) -- the bias gradient of the
// Linear that produced z, which would otherwise cost a second pass over that tensor.
template <int V>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ gamma, float* __restrict__ dx,
                                     float* __restrict__ dx_drop, float* __restrict__ part_dgamma,
                                     float* __restrict__ part_dbeta, float* __restrict__ part_dsum, long long rows, int d,
                                     DropCfg drop_y, DropCfg drop_branch, float* amax, PlaneOut po, float* __restrict__ part_pos = nullptr,
                                     const float* __restrict__ dy_col = nullptr) {
    // dy_col != null: the incoming gradient is an OUTER PRODUCT dy[row, c] = dy[row] * dy_col[c] (``dy`` then holds one value per
    // row) -- the gradient the Linear(d, 1) interest head sends into the last LayerNorm (d logits[row] * w[c],
    // decoder_leave_focal.py:451,596): formed here instead of being written out by one kernel and read back by this one.
    drop_y = drop_live(drop_y); drop_branch = drop_live(drop_branch);
    __shared__ f32x4 red[4][64];
    const float ps = plane_scale(po);
    float am = 0.f;          // max |dx_drop| (or |dx| when there is no dropped copy): the tensor the GEMMs consume
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // part_pos (embedding LayerNorms): this WAVE's sum of dx over the rows it walks, one partial row per wave.  The host picks the
    // grid so that the wave stride gridDim * nw is a multiple of the sequence length L: every row of a wave then sits at the
    // same position s = (blockIdx * nw + wave) % L, and d pe[s, :] = the sum of the partial rows p = s (mod L) (segmm_colsum_pos)
    // -- 12 MB of partials instead of a second pass over the 157 MB gradient (the positional-embedding gradient, encoder.py:450-471)
    f32x4 ag[V], ab[V], as[V], gm[V], ap[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i]; as[i] = ag[i]; ap[i] = ag[i];
        const int c = lane * 4 + i * 256;
        gm[i] = (c < d) ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long long row = (long long)blockIdx.x * nw + wave; row < rows; row += (long long)gridDim.x * nw) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 g[V], xh[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = lane * 4 + i * 256;
            g[i] = f32x4{0.f, 0.f, 0.f, 0.f}; xh[i] = g[i];
            if (c < d) {
                f32x4 t = dy_col ? *(const f32x4*)(dy_col + c) * dy[row] : ld_row4(dy + row * d + c);
                if (drop_y.p > 0.f) t = drop_apply4(drop_y, ((uint64_t)row * d + c) >> 2, t);
                xh[i] = (ld_row4(x + row * d + c) - mu) * rs;
                ab[i] += t;
                ag[i] += t * xh[i];
                g[i] = t * gm[i];
                s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                const f32x4 u = g[i] * xh[i];
                s2 += u.x + u.y + u.z + u.w;
            }
        }
        s1 = wave_sum(s1) / d;
        s2 = wave_sum(s2) / d;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < d) {
                const f32x4 o = (g[i] - s1 - xh[i] * s2) * rs;
                st_row4(dx + row * d + c, o);
                f32x4 od = o;
                if (dx_drop) {
                    if (drop_branch.p > 0.f) od = drop_apply4(drop_branch, ((uint64_t)row * d + c) >> 2, o);
                    st_row4(dx_drop + row * d + c, od);
                }
                if (ps > 0.f) plane_store4_pair(po.p, po.ld2, row, c, od, ps);
                as[i] += od;
                ap[i] += o;
                am = absmax4(am, od);
            }
        }
    }
    plane_finish(po, amax, am, blockIdx.x * nw + wave, ps, blockIdx.x == 0 && threadIdx.x == 0);
    if (part_pos) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < d) *(f32x4*)(part_pos + ((size_t)blockIdx.x * nw + wave) * d + c) = ap[i];
        }
    }
    // cross-wave reduce of the partials, one partial row per workgroup.  One 256-column chunk at a time through a 4 KB
    // buffer: the kernel usually runs NEXT TO a GEMM that holds 120 of the CU's 160 KB of LDS, and a 12 KB buffer
    // would cap it at three workgroups per CU there.
    for (int pass = 0; pass < (part_dsum ? 3 : 2); ++pass) {
        float* out = (pass == 0 ? part_dgamma : (pass == 1 ? part_dbeta : part_dsum)) + (size_t)blockIdx.x * d;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = lane * 4 + i * 256;
            __syncthreads();
            red[wave][lane] = pass == 0 ? ag[i] : (pass == 1 ? ab[i] : as[i]);
            __syncthreads();
            if (wave == 0 && c < d) {
                f32x4 s4 = red[0][lane];
                for (int w = 1; w < nw; ++w) s4 += red[w][lane];
                *(f32x4*)(out + c) = s4;
            }
        }
    }
}

// ---------------------------------------------------------------- column sums (bias grads, head weight grad)
// partial[chunk][n] = sum over the chunk's rows of w[m] * X[m][n]   (w optional)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int ld, const float* __restrict__ w, long long M,
                                      int N, float* __restrict__ partial, int rows_per_chunk) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + tx) * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    const long long r1 = min(M, r0 + rows_per_chunk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        for (long long r = r0 + ty; r < r1; r += 4) {
            f32x4 v = ld_row4b(X + r * ld + n);
            if (w) v *= w[r];
            acc += v;
        }
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        const f32x4 s = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
        *(f32x4*)(partial + (size_t)blockIdx.y * N + n) = s;
    }
}
// out[n] (+)= sum_p partial[p][n]
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int P, int N, float* out, int accumulate) {
    // 64 float4 columns x 4 partial-row lanes per workgroup: the P partials of a column are summed by 4
    // threads in a fixed order (deterministic), not by one thread walking P dependent loads.
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + tx) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N)
        for (int p = ty; p < P; p += 4) acc += *(const f32x4*)(partial + (size_t)p * N + n);
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        f32x4 s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        if (accumulate) s += *(const f32x4*)(out + n);
        *(f32x4*)(out + n) = s;
    }
}

// up to three column sums of same-shaped matrices in ONE launch pair (blockIdx.z selects the matrix): the per-workgroup
// partials that a LayerNorm backward leaves behind (d gamma, d beta, column sums of the forwarded gradient) were six tiny
// dependent launches per LayerNorm; this makes them two
struct Colsum3 { const float* X[3]; float* out[3]; };
__global__ __launch_bounds__(256) void colsum_partial3_kernel(Colsum3 a, int ld, long long M, int N, float* __restrict__ partial,
                                                              int rows_per_chunk) {
    __shared__ f32x4 red[4][64];
    const float* X = a.X[blockIdx.z];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + tx) * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    const long long r1 = min(M, r0 + rows_per_chunk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N)
        for (long long r = r0 + ty; r < r1; r += 4) acc += ld_row4b(X + r * ld + n);
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N)
        *(f32x4*)(partial + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * N + n) = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}
__global__ __launch_bounds__(256) void colsum_final3_kernel(const float* __restrict__ partial, int P, int N, Colsum3 a) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + tx) * 4;
    const float* part = partial + (size_t)blockIdx.z * P * N;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N)
        for (int p = ty; p < P; p += 4) acc += *(const f32x4*)(part + (size_t)p * N + n);
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N) *(f32x4*)(a.out[blockIdx.z] + n) = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// out[s][n] = sum over the partial rows p = s, s + period, s + 2 period, ... of part[p][n] (a fixed order: deterministic):
// the per-position sums of a LayerNorm backward's per-wave partials (layernorm_bwd_kernel part_pos).  A workgroup takes 64
// columns of one position; 16 row lanes share the position's partial rows (a short sequence has few positions and many rows
// each: one thread per column walked 1 024 dependent loads at period 1), then one lane adds the 16 partials in lane order.
__global__ __launch_bounds__(256) void colsum_pos_kernel(const float* __restrict__ part, int P, int period, int N, float* __restrict__ out) {
    __shared__ f32x4 red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n = (blockIdx.x * 16 + tx) * 4, s_ = blockIdx.y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N)
        for (long long p_ = s_ + (long long)period * ty; p_ < P; p_ += 16ll * period) acc += *(const f32x4*)(part + (size_t)p_ * N + n);
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        f32x4 t = red[0][tx];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][tx];
        *(f32x4*)(out + (size_t)s_ * N + n) = t;
    }
}

// ---------------------------------------------------------------- interest head: Linear(d,1)  (decoder_leave_focal.py:451,596)
// out[m] (+)= x[m,:].w (+ bias)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w,
                              const float* __restrict__ bias, float* out, long long rows, int d, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 a = *(const f32x4*)(x + row * ld + c), b = *(const f32x4*)(w + c);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (bias) s += bias[0];
        out[row] = accumulate ? out[row] + s : s;
    }
}
// dx[m,:] (+)= g[m] * w
__global__ __launch_bounds__(256) void rowscale_bcast_kernel(const float* __restrict__ g, const float* __restrict__ w, float* dx, int ld,
                                      long long rows, int d, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float gv = g[row];
    for (int c = lane * 4; c < d; c += 256) {
        f32x4 v = *(const f32x4*)(w + c) * gv;
        float* o = dx + row * ld + c;
        if (accumulate) v += *(const f32x4*)o;
        *(f32x4*)o = v;
    }
}
// out[m] (+)= sum_n a[m,n] * b[m,n]      (bilinear fusion head, decoder_leave_focal.py:417-421)
__global__ __launch_bounds__(256) void rowdot_pair_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                          int ldb, float* out, long long rows, int d, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 x = *(const f32x4*)(a + row * lda + c), y = *(const f32x4*)(b + row * ldb + c);
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = accumulate ? out[row] + s : s;
}
// out[m,:] (+)= g[m] * X[m,:]
__global__ __launch_bounds__(256) void rowscale_mat_kernel(const float* __restrict__ g, const float* __restrict__ X, int ldx,
                                                           float* out, int ldo, long long rows, int d, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float gv = g[row];
    for (int c = lane * 4; c < d; c += 256) {
        f32x4 v = *(const f32x4*)(X + row * ldx + c) * gv;
        float* o = out + row * ldo + c;
        if (accumulate) v += *(const f32x4*)o;
        *(f32x4*)o = v;
    }
}
// deterministic single-workgroup sum of a vector: out[0] (+)= sum v
__global__ __launch_bounds__(1024) void vecsum_kernel(const float* __restrict__ v, long long n, float* out, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        out[0] = accumulate ? out[0] + t : t;
    }
}

// ---------------------------------------------------------------- id-mode embedding (encoder.py:426-435,445,484-486)
// vid[b,s,:] = cat(E_item[item_id[b]], frame_w * pos[b,s] + frame_b) + pe[s]     (pre-LayerNorm; pos[b,s] = s unless given)
__global__ __launch_bounds__(256) void embed_id_vid_kernel(const long long* __restrict__ item_id, const float* __restrict__ table, int dhalf,
                                    const float* __restrict__ frame_w, const float* __restrict__ frame_b,
                                    const float* __restrict__ pe, const float* __restrict__ frame_pos,
                                    float* __restrict__ out, int B, int S, long long n_rows) {
    const int d = 2 * dhalf;
    const long long row = blockIdx.x;       // b*S + s
    const int b = (int)(row / S), s = (int)(row % S);
    const float fpos = frame_pos ? frame_pos[row] : (float)s;     // 'noPos' ablation: shuffled positions (encoder.py:428-429)
    const long long id = item_id[b];
    const bool ok = id >= 0 && id < n_rows;  // torch.nn.Embedding raises on such an id; here the row is poisoned with NaN
    const float nan = __uint_as_float(0x7fc00000u);
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        f32x4 v;
        if (c < dhalf) v = ok ? *(const f32x4*)(table + id * dhalf + c) : f32x4{nan, nan, nan, nan};
        else v = *(const f32x4*)(frame_w + (c - dhalf)) * fpos + *(const f32x4*)(frame_b + (c - dhalf));
        if (pe) v += *(const f32x4*)(pe + (size_t)s * d + c);          // pe == null: --use_pe 0 (encoder.py:450-458)
        *(f32x4*)(out + row * d + c) = v;
    }
}
// usr[b,0,:] = E_user[user_id[b]] + pe[0]
__global__ __launch_bounds__(256) void embed_id_usr_kernel(const long long* __restrict__ user_id, const float* __restrict__ table, int d,
                                    const float* __restrict__ pe, float* __restrict__ out, int B, long long n_rows) {
    const int b = blockIdx.x;
    const long long id = user_id[b];
    const bool ok = id >= 0 && id < n_rows;
    const float nan = __uint_as_float(0x7fc00000u);
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        f32x4 v = ok ? *(const f32x4*)(table + id * d + c) : f32x4{nan, nan, nan, nan};
        if (pe) v += *(const f32x4*)(pe + c);
        *(f32x4*)(out + (size_t)b * d + c) = v;
    }
}
// order[k] = index of the k-th smallest id, ties in index order (a STABLE argsort, like torch.argsort(ids, stable=True)): one
// workgroup, bitonic network over the 64-bit keys (id << 32 | index) in LDS, n <= 8192 (a data-parallel node of 8 ranks x 1024
// rows).  Replaces the five ATen launches of torch.argsort on the id-mode step path.
constexpr int ARGSORT_MAX = 8192;
__global__ __launch_bounds__(1024) void argsort_ids_kernel(const long long* __restrict__ ids, int n, int npow2, int* __restrict__ order) {
    extern __shared__ unsigned long long keys[];
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        // ids are row numbers of an embedding table (< 2^31); out-of-range ones sort by their low 31 bits + the sign bit, any
        // fixed order is fine for them (the scatter kernel skips them).  Padding keys are larger than every real key.
        const unsigned long long id = i < n ? ((unsigned long long)ids[i] & 0xFFFFFFFFull) ^ 0x80000000ull : 0xFFFFFFFFull;
        keys[i] = i < n ? ((id << 32) | (unsigned)i) : 0xFFFFFFFFFFFFFFFFull;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npow2 >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));          // lower index of the pair (bit j clear)
                const int l = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long a = keys[i], b = keys[l];
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = (int)(keys[i] & 0xFFFFFFFFull);
}

// ---- the same argsort for MORE than ARGSORT_MAX ids (a data-parallel node whose gathered id list outgrows one workgroup: 8 ranks x
// 2048 rows and up): the bitonic network over a key array in global memory (caller's workspace, npow2 x u64), chunks of
// ARGSORT_MAX keys sorted / merged in LDS by one workgroup each, the strides >= ARGSORT_MAX as one global compare-exchange
// launch per stride.  Directions come from the GLOBAL index, so the chunk kernels are the single-workgroup network restricted
// to a chunk.  Deterministic, no host sync: 5 launches for 16 384 ids, 8 for 32 768.
__global__ __launch_bounds__(256) void argsort_keys_init_kernel(const long long* __restrict__ ids, int n, int npow2, unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npow2) return;
    const unsigned long long id = i < n ? ((unsigned long long)ids[i] & 0xFFFFFFFFull) ^ 0x80000000ull : 0xFFFFFFFFull;
    keys[i] = i < n ? ((id << 32) | (unsigned)i) : 0xFFFFFFFFFFFFFFFFull;
}
// one chunk of ARGSORT_MAX keys per workgroup: kfirst == 2: the whole network up to k = ARGSORT_MAX (a sorted chunk, ascending or
// descending by the chunk's position); kfirst > ARGSORT_MAX: only the strides j < ARGSORT_MAX of stage k = kfirst (the tail of a merge)
__global__ __launch_bounds__(1024) void argsort_chunk_kernel(unsigned long long* __restrict__ gkeys, int kfirst) {
    extern __shared__ unsigned long long keys[];
    const int base = blockIdx.x * ARGSORT_MAX;
    for (int i = threadIdx.x; i < ARGSORT_MAX; i += blockDim.x) keys[i] = gkeys[base + i];
    __syncthreads();
    const int klast = kfirst == 2 ? ARGSORT_MAX : kfirst;
    for (int k = kfirst; k <= klast; k <<= 1) {
        for (int j = (k > ARGSORT_MAX ? ARGSORT_MAX : k) >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (ARGSORT_MAX >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((base + i) & k) == 0;
                const unsigned long long a = keys[i], b = keys[l];
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
        if (k >= ARGSORT_MAX) break;
    }
    for (int i = threadIdx.x; i < ARGSORT_MAX; i += blockDim.x) gkeys[base + i] = keys[i];
}
__global__ __launch_bounds__(256) void argsort_global_step_kernel(unsigned long long* __restrict__ keys, int npow2, int k, int j) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (npow2 >> 1)) return;
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const int l = i | j;
    const bool up = (i & k) == 0;
    const unsigned long long a = keys[i], b = keys[l];
    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
}
__global__ __launch_bounds__(256) void argsort_keys_finish_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ order) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[i] = (int)(keys[i] & 0xFFFFFFFFull);
}

// Data-parallel label statistics: `gathered` = G records [v (B) | v2 (B) | norms (3)] in rank order (one all-gather).  Splits them
// into the contiguous v_all / v2_all [G * B] the loss kernel takes and sums the three normalisers over the ranks, in rank order
// (counts: exact in fp32).  One launch instead of three strided ATen kernels inside the step.
__global__ __launch_bounds__(256) void label_stats_unpack_kernel(const float* __restrict__ gathered, int G, int B, float* __restrict__ v_all,
                                                                  float* __restrict__ v2_all, float* __restrict__ norms) {
    const int n = 2 * B + 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < G * B; i += gridDim.x * 256) {
        const int g = i / B, b = i - g * B;
        v_all[i] = gathered[(size_t)g * n + b];
        v2_all[i] = gathered[(size_t)g * n + B + b];
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += gathered[(size_t)g * n + 2 * B + threadIdx.x];
        norms[threadIdx.x] = s;
    }
}

// table[ids[k]][:] = 0 for every k (ids outside the table are skipped): clears the rows the previous step scattered into a
// dense embedding-table gradient instead of re-filling the whole table
__global__ __launch_bounds__(64) void zero_rows_kernel(float* __restrict__ table, int width, const long long* __restrict__ ids, long long n_rows) {
    const long long id = ids[blockIdx.x];
    if (id < 0 || id >= n_rows) return;
    for (int c = threadIdx.x * 4; c < width; c += 256) *(f32x4*)(table + id * width + c) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// backward of the gathers: dense table gradients (torch.nn.Embedding semantics), deterministic and
// sync-free: `order` = batch rows sorted by id (host-side torch.sort, no size-dependent output).
// Workgroup k owns sorted position k; it is a segment head iff its id differs from position k-1,
// and then sums every following row with the same id, in sorted order.
__global__ __launch_bounds__(256) void embed_id_bwd_kernel(const float* __restrict__ dpre, int tok_per_row, int ld,
                                    int col0, int width, const int* __restrict__ order,
                                    const long long* __restrict__ ids, float* __restrict__ dtable, int B, long long n_rows) {
    const int k0 = blockIdx.x;
    const long long id = ids[order[k0]];
    if (id < 0 || id >= n_rows) return;      // out-of-range id: its forward row was poisoned, nothing to scatter
    if (k0 > 0 && ids[order[k0 - 1]] == id) return;
    for (int c = threadIdx.x * 4; c < width; c += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < B && ids[order[k]] == id; ++k) {
            const long long tok0 = (long long)order[k] * tok_per_row;
            for (int t = 0; t < tok_per_row; ++t) acc += *(const f32x4*)(dpre + (tok0 + t) * ld + col0 + c);
        }
        float* o = dtable + id * width + c;
        *(f32x4*)o = *(const f32x4*)o + acc;
    }
}
// frame-position Linear(1, d/2) grads and positional-embedding grads need column sums over b with s fixed:
// dpe[s, :] = sum_b dpre[b*S+s, :]   (one workgroup per s; deterministic)
__global__ __launch_bounds__(256) void pe_grad_kernel(const float* __restrict__ dpre, int ld, int B, int S, int d, float* __restrict__ dpe,
                               int accumulate) {
    const int s = blockIdx.x;
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) acc += *(const f32x4*)(dpre + ((size_t)b * S + s) * ld + c);
        float* o = dpe + (size_t)s * d + c;
        if (accumulate) acc += *(const f32x4*)o;
        *(f32x4*)o = acc;
    }
}

// One AdamW update of four elements.  Floating-point contraction is OFF: adamw_kernel and the two table kernels below must give
// bit-identical results for the same element whatever the surrounding code lets the compiler fuse.
__device__ __forceinline__ void adamw_elem4(f32x4& pp, const f32x4 gg, f32x4& mm, f32x4& vv, float lr, float b1, float b2, float eps, float wd,
                                            float step, float bc2_sqrt) {
#pragma clang fp contract(off)
    pp *= (1.0f - lr * wd);
    mm = mm * b1 + gg * (1.0f - b1);
    vv = vv * b2 + gg * gg * (1.0f - b2);
    f32x4 den;
    den.x = sqrtf(vv.x) / bc2_sqrt + eps; den.y = sqrtf(vv.y) / bc2_sqrt + eps;
    den.z = sqrtf(vv.z) / bc2_sqrt + eps; den.w = sqrtf(vv.w) / bc2_sqrt + eps;
    pp.x -= step * (mm.x / den.x); pp.y -= step * (mm.y / den.y);
    pp.z -= step * (mm.z / den.z); pp.w -= step * (mm.w / den.w);
}
// ---------------------------------------------------------------- fused AdamW over a flat range (a13 / K9)
// torch.optim.AdamW single-tensor update order (decoupled decay first), bias corrections passed in.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, const StepState* live) {
    if (live) { bc1 = live->bc1; bc2_sqrt = live->bc2_sqrt; }          // bias corrections of the device-side step count
    const long long n4 = n >> 2;
    const float step = lr / bc1;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = ((f32x4*)p)[i];
        const f32x4 gg = ((const f32x4*)g)[i];
        f32x4 mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        adamw_elem4(pp, gg, mm, vv, lr, b1, b2, eps, wd, step, bc2_sqrt);
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
    // tail (n % 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        f32x4 pp = {p[i], 0.f, 0.f, 0.f}, mm = {m[i], 0.f, 0.f, 0.f}, vv = {v[i], 0.f, 0.f, 0.f};
        adamw_elem4(pp, f32x4{g[i], 0.f, 0.f, 0.f}, mm, vv, lr, b1, b2, eps, wd, step, bc2_sqrt);
        p[i] = pp.x; m[i] = mm.x; v[i] = vv.x;
    }
}

// AdamW of an id-embedding table in two passes (segmm_adamw_table): the arithmetic of adamw_kernel, element for element
__global__ __launch_bounds__(256) void table_mark_kernel(const long long* __restrict__ ids, int n, long long n_rows, unsigned int* __restrict__ flags) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        const long long id = ids[k];
        if (id >= 0 && id < n_rows) flags[id] = 1u;
    }
}
// every row WITHOUT a mark, g = 0
__global__ __launch_bounds__(256) void adamw_table_rest_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, long long n_rows,
                                                               int w4, const unsigned int* __restrict__ flags, float lr, float b1, float b2,
                                                               float eps, float wd, float bc1, float bc2_sqrt, const StepState* live) {
    if (live) { bc1 = live->bc1; bc2_sqrt = live->bc2_sqrt; }
    const float step = lr / bc1;
    const long long n4 = n_rows * w4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        if (flags[i / w4] != 0u) continue;
        f32x4 pp = ((f32x4*)p)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        adamw_elem4(pp, zero, mm, vv, lr, b1, b2, eps, wd, step, bc2_sqrt);
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
}
// the marked rows, each once: one wave per list entry; lane 0 claims (and clears) the row's mark
__global__ __launch_bounds__(256) void adamw_table_rows_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, long long n_rows, int w4, const long long* __restrict__ ids,
                                                               int n_ids, unsigned int* __restrict__ flags, float lr, float b1, float b2, float eps,
                                                               float wd, float bc1, float bc2_sqrt, const StepState* live) {
    if (live) { bc1 = live->bc1; bc2_sqrt = live->bc2_sqrt; }
    const float step = lr / bc1;
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n_ids) return;
    const long long id = ids[k];
    if (id < 0 || id >= n_rows) return;
    unsigned int mine = 0u;
    if (lane == 0) mine = atomicExch(flags + id, 0u);
    mine = (unsigned int)__builtin_amdgcn_readfirstlane((int)mine);
    if (mine == 0u) return;          // another entry of the list holds the same id and has taken the row
    for (int c = lane; c < w4; c += 64) {
        const long long i = id * w4 + c;
        f32x4 pp = ((f32x4*)p)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        adamw_elem4(pp, ((const f32x4*)g)[i], mm, vv, lr, b1, b2, eps, wd, step, bc2_sqrt);
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
}

// ---------------------------------------------------------------- CrossMLP ablation: AdaptiveAvgPool1d over tokens
// out[b, i, :] = mean over t in [floor(i T / bins), ceil((i+1) T / bins)) of cat(U[b], V[b])[t, :], T = Lu + Lv
// (encoder.py:396,503-506: nn.AdaptiveAvgPool1d(40) over the token axis of the concatenated user and video tokens;
// neighbouring bins overlap when T is not a multiple of bins).
__device__ __forceinline__ void pool_bin(int i, int T, int bins, int& t0, int& t1) {
    t0 = (int)(((long long)i * T) / bins);
    t1 = (int)((((long long)(i + 1)) * T + bins - 1) / bins);
}
__global__ __launch_bounds__(256) void pool_tokens_kernel(const float* __restrict__ U, int Lu, const float* __restrict__ V, int Lv,
                                                          float* __restrict__ out, int d, int bins) {
    const int b = blockIdx.x / bins, i = blockIdx.x % bins, T = Lu + Lv;
    int t0, t1;
    pool_bin(i, T, bins, t0, t1);
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = t0; t < t1; ++t)
            acc += t < Lu ? *(const f32x4*)(U + ((size_t)b * Lu + t) * d + c) : *(const f32x4*)(V + ((size_t)b * Lv + (t - Lu)) * d + c);
        const float n = (float)(t1 - t0);
        *(f32x4*)(out + (size_t)blockIdx.x * d + c) = f32x4{acc.x / n, acc.y / n, acc.z / n, acc.w / n};
    }
}
// dU / dV[b, t, :] = sum over the bins i that contain t of dOut[b, i, :] / |bin i|   (fixed order: deterministic)
__global__ __launch_bounds__(256) void pool_tokens_bwd_kernel(const float* __restrict__ dOut, float* __restrict__ dU, int Lu,
                                                              float* __restrict__ dV, int Lv, int d, int bins) {
    const int T = Lu + Lv, b = blockIdx.x / T, t = blockIdx.x % T;
    float* dst = t < Lu ? dU + ((size_t)b * Lu + t) * d : dV + ((size_t)b * Lv + (t - Lu)) * d;
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < bins; ++i) {
            int t0, t1;
            pool_bin(i, T, bins, t0, t1);
            if (t >= t0 && t < t1) {
                const f32x4 g = *(const f32x4*)(dOut + ((size_t)b * bins + i) * d + c);
                const float n = (float)(t1 - t0);
                acc += f32x4{g.x / n, g.y / n, g.z / n, g.w / n};
            }
        }
        *(f32x4*)(dst + c) = acc;
    }
}

}  // namespace segmm
