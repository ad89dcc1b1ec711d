/* segmm_hip.h -- C ABI of libsegmm_hip.so, the MI355X (gfx950) kernels of the segment-interest training path.
 *
 * The reference (hezy18/SegMMInterest) is pure Python/PyTorch and has NO native boundary; its boundary is the
 * Python plugin surface `MMinterest/models/__init__.py:1-5`.  This header defines the native layer UNDERNEATH that
 * surface: each entry point replaces the chain of stock ATen ops the reference executes at the
 * cited file:line.  Host bindings: `segmminterest_amd/hipabi.py` (ctypes); INTEGRATION.md shows the stub a
 * maintainer of the reference adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library allocates nothing persistent,
 *    keeps no mutable global state and never synchronises: it only enqueues on `stream` (a hipStream_t);
 *  - fp32 row-major everywhere; feature/leading dimensions must be multiples of 4 floats and 16-byte aligned;
 *  - return value 0 = ok, negative = error (message via segmm_last_error(), thread local); nothing throws;
 *  - re-entrant: forward runs on the Python thread, backward on an autograd-engine thread;
 *  - dropout is a counter-based hash stream addressed by (seed, site, element index): with p = 0 results are
 *    bitwise reproducible run to run; with p > 0 they are reproducible for a fixed (seed, site).
 */
#ifndef SEGMM_HIP_H
#define SEGMM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* segmm_stream_t; /* hipStream_t */

const char* segmm_last_error(void);
int segmm_abi_version(void);

/* a1 -- trainer L1 normalisation  x / (sum|x| + 1e-6)  (main_for_seq_leave_earlystop_SegMM.py:272-273).
 * y may be NULL: then only inv_scale[row] = 1/(sum|x|+1e-6) is produced, for the fused a1+a2 GEMM (row_scale) -- or, with a
 * plane output, the normalised rows are written as planes only (|y| <= 1: with *scale_in = 2^14 the planes can neither overflow
 * nor fall below the fp16 window, so their consumers need no fp32 copy to fall back on).
 * amax: optional zeroed [SEGMM_AMAX_SLOTS] array receiving the partial maxima of |y| (see segmm_gemm_h). */
int segmm_l1norm(const float* x, float* y, float* inv_scale, int64_t rows, int D, float* amax, uint16_t* planes, int ld2, float* hdr,
                 const float* scale_in, segmm_stream_t stream);
/* PLANE OUTPUTS of producers (the four trailing arguments `planes, ld2, hdr, scale_in` of segmm_l1norm, segmm_gather_l1,
 * segmm_layernorm_fwd / _bwd; segmm_attn_planes_t; c_planes / c_hdr / c_scale_in of segmm_gemm_p): a kernel that writes a
 * tensor some GEMM reads can also write that tensor's P32 fp16 planes (format: segmm_gemm_p) with the DELAYED scale
 * *scale_in -- a device scalar, the power of two that segmm_scales_update derived from the maxima the tensor site had on
 * earlier passes.  The kernel stores the scale it used in hdr[0], folds the partial maxima of what it wrote into the header's
 * slots and raises hdr[1] if an element left the fp16 range (consumers then read the fp32 copy).  planes == NULL, scale_in ==
 * NULL or *scale_in == 0: no planes are written (the caller runs segmm_split_p32 with the exact scale instead). */

/* K2/K3/K5/K6 -- every nn.Linear of the path and its gradients (encoder.py:95-104,163-167,183-184,438,445;
 * kn_util/nn_utils/layers/mlp.py:17-23), on the f32 MFMA.
 *   layout 0 (NT): C[M,N] = A[M,K] . B[N,K]^T    forward, B = Linear.weight
 *   layout 1 (NN): C[M,N] = A[M,K] . B[K,N]      dgrad
 *   layout 2 (TN): C[M,N] = A[K,M]^T . B[K,N]    wgrad (use splits > 1: K is the token dimension)
 * epilogue, in this order: * row_scale[m]; + bias[n]; activation (0 none, 1 erf-GELU saving the pre-activation to
 * aux, 2 multiply by GELU'(aux), 3 ReLU, 4 zero where aux <= 0 -- ReLU backward, aux = the forward output); dropout(p, seed, site) on element m*N+n; + residual[(m % res_period), n].
 * residual may alias C (accumulate).  splits > 1: partial slabs in `workspace` (splits*M*N floats), then a
 * deterministic combine; only `accumulate` (C += result) applies in that mode.
 * engine 0: v_mfma_f32_32x32x2_f32 (exact fp32 products).  engine 1: "bf16x6" -- operands are split exactly into
 * three bf16 terms on the fly and six v_mfma_f32_32x32x16_bf16 partial products are accumulated in fp32 (error below
 * that of a plain fp32 GEMM, 2.67x less matrix-pipe time); same layouts, epilogue and split-K. */
int segmm_gemm(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
               const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
               int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
               float* workspace, int accumulate, int engine, segmm_stream_t stream);

/* Same GEMM on the bf16x6 engine with optional PRE-SPLIT operands: a_planes / b_planes point at bf16 planes
 * [nplanes][rows][ld] (plane p at base + p*pstride elements, rows k-contiguous, NT layout only) produced by
 * segmm_split3 / segmm_split3_transpose -- the main loop then does no conversion work for that operand.  Weights are
 * split once per optimizer step; W^T planes turn the dgrad (dX = dY.W) into the NT form as well.
 * nplanes = 2 keeps only hi and mid (three partial products, ~2e-5 relative error): offered for weight gradients,
 * which are leaves of the backward graph. */
int segmm_gemm_x(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
                 int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
                 float* workspace, int accumulate, const uint16_t* a_planes, int64_t a_pstride,
                 const uint16_t* b_planes, int64_t b_pstride, int nplanes, segmm_stream_t stream);

/* fp16x3 engine: x*s = hi + lo in two fp16 terms (22 mantissa bits), s a per-tensor power of two derived inside the
 * kernel from PARTIAL MAXIMA of |x| (a_amax[0..a_namax), b_amax[0..b_namax), device arrays written by
 * segmm_absmax or by the operand's producer; no host sync); three products hh + hl + lh on
 * v_mfma_f32_32x32x16_f16, per-product error <= 3 * 2^-22.  Half the matrix-pipe work of segmm_gemm_x.
 * Optional pre-split operands are fp16 planes [2][rows][ld] from segmm_split2h / segmm_split2h_transpose made with
 * the SAME partial maxima that are passed here.  All layouts, epilogues and split-K as segmm_gemm. */
int segmm_gemm_h(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, const float* row_scale, const float* residual, int ldr, int res_period,
                 int activation, float* aux, int ldaux, float drop_p, uint64_t seed, uint32_t site, int splits,
                 float* workspace, int accumulate, const uint16_t* a_planes, int64_t a_pstride,
                 const uint16_t* b_planes, int64_t b_pstride, const float* a_amax, int a_namax, const float* b_amax,
                 int b_namax, float* c_amax, segmm_stream_t stream);

/* Plane-operand GEMM: the fp16x3 arithmetic of segmm_gemm_h with operands that arrive PRE-SPLIT by their producers.
 * P32 plane format of a matrix X[R][C], C % 32 == 0: one fp16 array with row stride ld2 (halves); per row and per block
 * of 32 columns [32 hi | 32 lo] (128 bytes) -- element (r, c): hi at r*ld2 + (c/32)*64 + c%32, lo 32 further.
 * Site header `hdr` of a plane tensor: SEGMM_SITE_HDR floats followed by SEGMM_AMAX_SLOTS partial maxima:
 *   hdr[0] scale s the planes were written with (power of two; 0 = no planes written),
 *   hdr[1] != 0 (as integer): an element left the fp16 range under s (delayed scaling) -> consumers read the fp32 copy.
 * a_f32 / b_f32 (optional): fp32 copy of the operand; taken (exact split on the fly, slow) whenever the planes are not
 * usable -- results are never silently wrong.  Replaces the same reference ops as segmm_gemm (encoder.py:95-104,
 * 163-167,183-184,438,445; kn_util/nn_utils/layers/mlp.py:17-23).
 *   layout 0 (NT): C[M,N] = A[M,K] . B[N,K]^T  A planes [M][2K], B planes [N][2K]          (forward; dgrad on W^T planes)
 *   layout 2 (TN): C[M,N] = A[K,M]^T . B[K,N]  A planes [K][2M], B planes [K][2N], split-K  (weight gradients)
 * Output: fp32 C (unless bit 0 of write_c is clear) and/or P32 planes c_planes written with the scale *c_scale_in; the partial
 * maxima of |C|, the overflow flag and the scale used are folded into c_hdr (caller zeroes the header).  Epilogue as
 * segmm_gemm.  NT: K % 32 == 0; TN: M, N % 32 == 0 (token tails are zero-filled).
 * PLANES-ONLY outputs (round 5; NT, write_c = 0: the projection outputs the planes-in attention kernels read): with no fp32 copy
 * a consumer cannot fall back when the delayed scale turns out wrong, so the caller enqueues the SAME call once more with bit 1
 * of write_c set -- the REPAIR launch: every workgroup judges the output site from c_hdr (scale, flag, complete maxima) and
 * leaves at once when the planes are usable; otherwise it recomputes its tile and rewrites the planes with the exact scale of
 * the recorded maxima, leaving c_hdr untouched.  Consumers judge the same header and derive the same scale. */
#define SEGMM_SITE_HDR 8
int segmm_gemm_p(int layout, int M, int N, int K, const uint16_t* a_planes, int lda2, const float* a_hdr, const float* a_f32, int ldaf,
                 const uint16_t* b_planes, int ldb2, const float* b_hdr, const float* b_f32, int ldbf, float* C, int ldc,
                 uint16_t* c_planes, int ldc2, float* c_hdr, const float* c_scale_in, int write_c, const float* bias, const float* row_scale,
                 const float* residual, int ldr, int res_period, int activation, float* aux, int ldaux, float drop_p,
                 uint64_t seed, uint32_t site, int splits, float* workspace, int accumulate, float* colsum_out, segmm_stream_t stream);
/* colsum_out (TN only, optional): [M] floats receiving sum_k A[k, m] -- the bias gradient of the Linear whose weight gradient
 * the call computes (dW = dY^T . X, db = column sums of dY), formed inside the same kernel (+)= with `accumulate`; with
 * splits > 1 the workspace must hold splits * (M * N + M) floats. */
/* fp32 [rows, cols] (row stride ld) -> P32 planes.  mode 0: exact scale from the header's partial maxima (complete when
 * this runs), written to hdr[0], flag cleared.  mode 1: scale = hdr[0] as given; maxima and flag folded into hdr. */
int segmm_split_p32(const float* x, int64_t rows, int cols, int ld, uint16_t* planes, int ld2, float* hdr, int mode,
                    segmm_stream_t stream);
/* All GEMM weight matrices of a model in two launches per optimizer step (absmax; exact split + transposed split).
 * desc: device array of n_mats records {int64 flat offset (floats); int32 R, C, needs_transpose, first_tile, tile_cols}
 * (32 x 32 tiles, n_tiles in total, matrices in ascending first_tile order); hdr: [n_mats][SEGMM_SITE_HDR +
 * SEGMM_AMAX_SLOTS] zeroed site headers (receive maxima and scale); W planes at wpl + 2 * offset (ld2 = 2 C), W^T planes
 * at wTpl + 2 * offset (ld2 = 2 R).  C % 32 == 0 (and R % 32 == 0 for transposed matrices). */
int segmm_wsplit_p32(const float* flat, const void* desc, int n_mats, int n_tiles, float* hdr, uint16_t* wpl, uint16_t* wTpl,
                     segmm_stream_t stream);
/* P32 planes of the transpose of x[R, C]: plane row c holds x[:, c] (R % 32 == 0), scale hdr[0]. */
int segmm_split_p32_transpose(const float* x, int R, int C, int ld, uint16_t* planes, int ld2, const float* hdr,
                              segmm_stream_t stream);
/* Producer side of those partial maxima.  segmm_gemm_h (c_amax), segmm_layernorm_fwd/bwd (amax), segmm_attn_fwd
 * (amax_o) and segmm_attn_bwd (amax_q / amax_ka / amax_kb) take OPTIONAL arrays of SEGMM_AMAX_SLOTS floats that
 * the CALLER HAS ZEROED; each wave folds the maximum of what it stored into one slot with an integer atomic max on
 * the float bits (exact, order-independent, so results stay bitwise reproducible).  Several producers may share
 * one array (the attention backward of both sides writes column blocks of the same dY buffer). */
#define SEGMM_AMAX_SLOTS 256
/* out[0..nparts) = partial maxima of |x| over the [rows, cols] view with row stride ld (1 <= nparts <= 1024). */
int segmm_absmax(const float* x, int64_t rows, int cols, int ld, float* out, int nparts, segmm_stream_t stream);
int segmm_split2h(const float* x, uint16_t* planes, int64_t n, int64_t pstride, const float* amax, int namax,
                  segmm_stream_t stream);
int segmm_split2h_transpose(const float* x, int R, int Cc, int ld, uint16_t* planes, int64_t pstride,
                            const float* amax, int namax, segmm_stream_t stream);
/* exact split x = hi + mid + lo into three bf16 planes: planes[p*pstride + i] (flat), or the transposed copy
 * planes[p*pstride + c*R + r] of an [R, C] row-major matrix with leading dimension ld. */
int segmm_split3(const float* x, uint16_t* planes, int64_t n, int64_t pstride, segmm_stream_t stream);
int segmm_split3_transpose(const float* x, int R, int C, int ld, uint16_t* planes, int64_t pstride, segmm_stream_t stream);

/* LayerNorm(d, eps) forward/backward (encoder.py:39-40,170-171,185-186,203,206,383-385,455,465).
 * forward: optional dropout on the output (embedding dropout, encoder.py:461,471).  y == NULL with a plane output (round 5):
 * PLANES ONLY -- the planes are written with the scale of the output's bound (max|gamma| sqrt(d) + max|beta|) / (1 - p), derived
 * in the kernel (scale_in is ignored) and recorded in hdr[0]: no input can overflow it, so the consumers need no fp32 fallback.
 * backward: dy is first multiplied by the forward's output-dropout mask (drop_y_*); dx_drop (may be NULL) receives
 * dx times the mask of the residual-branch dropout "x = res + dropout(branch)" (drop_b_*), i.e. d(branch).
 * part_dgamma/part_dbeta: [nparts, d] per-workgroup partials, nparts = segmm_layernorm_bwd_parts(rows, d)
 * (one round of the workgroups that are resident at that row width: 768 at d = 768).
 * part_dsum (optional, same shape): partial column sums of the forwarded gradient (dx_drop if given, else dx) =
 * the bias gradient of the Linear whose output entered the LayerNorm through the residual branch. */
int segmm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int64_t rows, int d, float eps, float drop_p, uint64_t seed, uint32_t site, float* amax,
                        uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream);
/* segmm_layernorm_fwd that also returns dot_out[row] = y[row, :] . dot_w (+ dot_b[0]): the Linear(d, 1) interest head
 * (decoder_leave_focal.py:451,596) applied to the backbone's last LayerNorm output without a second pass over it. */
int segmm_layernorm_fwd_dot(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int64_t rows, int d, float eps, float drop_p, uint64_t seed, uint32_t site, float* amax,
                            uint16_t* planes, int ld2, float* hdr, const float* scale_in, const float* dot_w, const float* dot_b,
                            float* dot_out, segmm_stream_t stream);
int segmm_layernorm_bwd_parts(int64_t rows, int d);
int segmm_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                        int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                        float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream);
/* segmm_layernorm_bwd whose incoming gradient is the outer product dy[row, c] = dy_row[row] * dy_col[c]: what the Linear(d, 1)
 * interest head (decoder_leave_focal.py:451,596) sends into the last LayerNorm of the backbone -- formed inside the launch
 * instead of by segmm_rowscale_bcast (same products, bit-identical results). */
int segmm_layernorm_bwd_outer(const float* dy_row, const float* dy_col, const float* x, const float* mean, const float* rstd, const float* gamma,
                              float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                              int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                              float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream);
/* The embedding LayerNorms' backward (encoder.py:450-471: y = LN(proj(x) + pe[s])): the same launch on a grid of
 * segmm_layernorm_bwd_pos_parts(rows, period, d) workgroups (0: no such grid; part_dgamma / part_dbeta / part_dsum then have that
 * many rows) whose waves each walk rows of ONE position s = row mod period, and leave their sum of dx in part_pos[4 * parts, d]
 * (partial row p holds position p mod period).  segmm_colsum_pos: out[s, :] = sum of the partial rows p = s (mod period), in
 * index order -- the positional-embedding gradient without a second pass over dx. */
int segmm_layernorm_bwd_pos_parts(int64_t rows, int period, int d);
int segmm_layernorm_bwd_pos(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                            float* dx, float* dx_drop, float* part_dgamma, float* part_dbeta, float* part_dsum, int64_t rows,
                            int d, float drop_y_p, uint32_t drop_y_site, float drop_b_p, uint32_t drop_b_site, uint64_t seed,
                            float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, float* part_pos, int period,
                            segmm_stream_t stream);
int segmm_colsum_pos(const float* part, int n_rows, int period, int d, float* out, segmm_stream_t stream);

/* out[n] (+)= sum_m w[m] * X[m,n]  (bias gradients, LayerNorm partial combine, head weight gradient).
 * workspace: segmm_colsum_chunks(M) * N floats. */
int segmm_colsum_chunks(int64_t M);
int segmm_colsum(const float* X, int ld, const float* w, int64_t M, int N, float* out, int accumulate,
                 float* workspace, segmm_stream_t stream);

/* K4 -- joint self+cross attention of one side (encoder.py:44-73,138-161).  Queries Qa/Qb (two projections of the
 * same Lq tokens) against key blocks a (La tokens) and b (Lb tokens); all tensors are [B*L, ld] with head h at
 * columns [h*dh, (h+1)*dh).  Masks are uint8 (torch.bool).  lse: [2, B, H, Lq] floats
 * (plane 0 = softmax row max, plane 1 = 1/row sum, written by the forward); Dvec: [B, H, Lq] floats.
 * One key block may be empty (La == 0 or Lb == 0, its pointers null): the CrossAtt / SelfAtt ablations.
 * segmm_attn_bwd phase: 0 = whole backward on `stream` (dQ kernel, which also writes Dvec, then dK/dV kernel);
 * 1 = Dvec = rowsum(dO * O) only; 2 = dQa/dQb only (Dvec not written); 3 = dKa/dVa/dKb/dVb only (reads Dvec) -- phases 2
 * and 3 are independent once phase 1 is complete and may run concurrently on two streams; 4 = dQ, dK and dV in ONE kernel
 * (one workgroup per (b, h, key block), query side staged in LDS in chunks of 48 rows, D formed inside -- Dvec is not
 * used; <= 12 key tiles per block); 5 / 6 = the same kernel for key block a / b only (the two launches of phase 4 are
 * independent: disjoint outputs, shared maxima slots are integer atomic maxima -- they may run on two streams). */
/* optional plane outputs of the attention kernels (see "PLANE OUTPUTS of producers"): the forward's O; in the fused backward
 * (phase 4) the query-side gradients dQa / dQb (one site: they are columns of the same buffer) and the key-side gradients
 * of block a (dKa, dVa) and block b (dKb, dVb).  Each plane pointer addresses the same column slice as its fp32 twin. */
typedef struct {
    uint16_t* o; int ldo2; float* hdr_o; const float* sin_o;
    uint16_t *dqa, *dqb; int lddq2;
    uint16_t *dka, *dva; int lddka2;
    uint16_t *dkb, *dvb; int lddkb2;
    float *hdr_q, *hdr_ka, *hdr_kb;
    const float *sin_q, *sin_ka, *sin_kb;
    /* fused backward only.  bit 0 (SEGMM_ATTN_PLANES_ONLY): gradients whose planes are written get no fp32 copy -- the caller
     * then passes the consumers no fp32 fallback for that site and, after EVERY producer of the site has been enqueued, calls
     * segmm_site_fixup on the site headers and then segmm_attn_bwd again with the same arguments and bit 1
     * (SEGMM_ATTN_REPAIR) set: the repair launch ends at once unless segmm_site_fixup found the planes of a site unusable
     * (overflow flag, maximum below the fp16 window, or no scale yet), in which case it rewrites them with the exact scale of
     * the recorded maxima. */
    int flags;
    /* forward and fused backward (round 5): INPUT planes -- the P32 planes of the Q / K / V column slices as their producers
     * (the fused projection GEMMs) wrote them, each addressing the same columns as its fp32 twin, with the headers of the sites
     * they belong to (queries; key block a: K and V of one buffer; key block b).  When qa_in / qb_in are given (all of the *_in
     * fields must be, for the non-empty key blocks) and the shape qualifies (dh in {16, 32, 48}, La % 4 == 0, Lb % 4 == 0, at most
     * 112 tokens per side), segmm_attn_fwd stages K / V by LDS-DMA and segmm_attn_bwd (fused phases 4-6) loads the fragments
     * straight from the planes; every product runs on the fp16 matrix cores (three partial products, the GEMM engine's
     * arithmetic).  The fp32 views (Qa ... Vb) may then be NULL -- a caller whose projection GEMMs write planes ONLY: a site
     * whose planes are unusable under its header's scale must have been rewritten by its producer's repair launch
     * (segmm_gemm_p, write_c bit 1), and the kernels take the exact scale of the recorded maxima like that launch did.  With
     * fp32 views given, the forward stages such a site from them inside the same launch.  Shapes that do not qualify run the
     * fp32-view kernels and refuse a call without views. */
    const uint16_t *qa_in, *qb_in; int ldq2_in; const float* hdr_q_in;
    const uint16_t *ka_in, *va_in; int ldka2_in; const float* hdr_ka_in;
    const uint16_t *kb_in, *vb_in; int ldkb2_in; const float* hdr_kb_in;
} segmm_attn_planes_t;
#define SEGMM_ATTN_PLANES_ONLY 1
#define SEGMM_ATTN_REPAIR 2
/* Between the producers of planes-only sites and their repair launches: for each non-NULL site header whose planes are
 * unusable under hdr[0], put the exact scale of the recorded maxima in hdr[0], clear the overflow flag, set hdr[2] (the repair
 * launches act on it) and count the site in stats[0] (the counter segmm_scales_update keeps for refused planes; may be NULL). */
int segmm_site_fixup(float* hdr0, float* hdr1, float* hdr2, float* hdr3, float* stats, segmm_stream_t stream);
int segmm_attn_fwd(int B, int H, int dh, int Lq, int La, int Lb, const float* Qa, const float* Qb, int ldq,
                   const float* Ka, const float* Va, int ldka, const float* Kb, const float* Vb, int ldkb,
                   const uint8_t* mq, const uint8_t* mka, const uint8_t* mkb, float* O, int ldo, float* lse,
                   float drop_p, uint64_t seed, uint32_t site, float* amax_o, const segmm_attn_planes_t* planes,
                   segmm_stream_t stream);
int segmm_attn_bwd(int B, int H, int dh, int Lq, int La, int Lb, const float* Qa, const float* Qb, int ldq,
                   const float* Ka, const float* Va, int ldka, const float* Kb, const float* Vb, int ldkb,
                   const uint8_t* mq, const uint8_t* mka, const uint8_t* mkb, const float* lse, const float* O, int ldo,
                   const float* dO, int lddo, float* Dvec, float* dQa, float* dQb, int lddq, float* dKa, float* dVa, int lddka,
                   float* dKb, float* dVb, int lddkb, float drop_p, uint64_t seed, uint32_t site,
                   float* amax_q, float* amax_ka, float* amax_kb, int phase, const segmm_attn_planes_t* planes,
                   segmm_stream_t stream);

/* K7 -- interest head Linear(d,1) (decoder_leave_focal.py:451,596): out[m] (+)= x[m,:].w (+ bias[0]) and
 * dx[m,:] (+)= g[m]*w.  segmm_vecsum: deterministic out[0] (+)= sum v. */
int segmm_rowdot(const float* x, int ld, const float* w, const float* bias, float* out, int64_t rows, int d,
                 int accumulate, segmm_stream_t stream);
int segmm_rowscale_bcast(const float* g, const float* w, float* dx, int ld, int64_t rows, int d, int accumulate,
                         segmm_stream_t stream);
int segmm_vecsum(const float* v, int64_t n, float* out, int accumulate, segmm_stream_t stream);
/* bilinear fusion head InteractionAggregation (decoder_leave_focal.py:392-423): out[m] (+)= sum_n a[m,n]*b[m,n] and
 * out[m,:] (+)= g[m]*X[m,:]; the per-head x_h W_h products run on segmm_gemm. */
int segmm_rowdot_pair(const float* a, int lda, const float* b, int ldb, float* out, int64_t rows, int d,
                      int accumulate, segmm_stream_t stream);
int segmm_rowscale_mat(const float* g, const float* X, int ldx, float* out, int ldo, int64_t rows, int d,
                       int accumulate, segmm_stream_t stream);

/* K2' -- id-mode embeddings (encoder.py:426-435,445,484-486), pre-LayerNorm:
 *   vid[b,s,:] = cat(item_table[item_id[b]], frame_w*pos[b,s] + frame_b) + vid_pe[s];   usr[b,0,:] = user_table[uid[b]] + usr_pe[0]
 *   (frame_pos = null: pos[b,s] = s; the 'noPos' ablation passes per-row shuffled positions [B*S], encoder.py:428-429;
 *   pe = null: no positional embedding is added -- the trainers' --use_pe 0, encoder.py:450-471 else branches)
 * backward: dense table gradients accumulated deterministically over batch rows grouped by id
 * (order = batch rows sorted by id, from a host-side torch.sort; no data-dependent sizes, no sync),
 * and dpe[s,:] = sum_b dpre[b,s,:].  n_rows = rows of the table: an id outside [0, n_rows) (torch.nn.Embedding raises
 * on it) never touches memory -- its forward row is filled with NaN so the loss shows it, its backward is skipped. */
int segmm_embed_id_vid(const int64_t* item_id, const float* table, int dhalf, const float* frame_w,
                       const float* frame_b, const float* pe, const float* frame_pos, float* out, int B, int S,
                       int64_t n_rows, segmm_stream_t stream);
int segmm_embed_id_usr(const int64_t* user_id, const float* table, int d, const float* pe, float* out, int B,
                       int64_t n_rows, segmm_stream_t stream);
int segmm_embed_id_bwd(const float* dpre, int tokens_per_row, int ld, int col0, int width, const int32_t* order,
                       const int64_t* ids, float* dtable, int B, int64_t n_rows, segmm_stream_t stream);
/* order[k] = index of the k-th smallest id, equal ids in index order (torch.argsort(ids, stable=True), which
 * segmm_embed_id_bwd needs as `order`); n <= 8192, one workgroup, no host sync. */
int segmm_argsort_ids(const int64_t* ids, int n, int32_t* order, segmm_stream_t stream);
/* The same for any n <= 2^24: beyond 8192 ids the bitonic network runs over `keys_ws` (caller-owned, next power of two >= n
 * 64-bit words), chunks of 8192 keys per workgroup in LDS, larger strides as global compare-exchange launches (round 6: the
 * gathered id list of a data-parallel node outgrows one workgroup from 8 ranks x 2048 rows on). */
int segmm_argsort_ids_ws(const int64_t* ids, int n, int32_t* order, uint64_t* keys_ws, segmm_stream_t stream);
/* table[ids[k], :] = 0 for k < n (ids outside [0, n_rows) are skipped): the rows a previous segmm_embed_id_bwd scattered into a
 * dense [n_rows, width] table gradient, cleared without re-filling the table. */
int segmm_zero_rows(float* table, int width, const int64_t* ids, int n, int64_t n_rows, segmm_stream_t stream);
/* Data parallel: the all-gathered label statistics of G ranks -- G records [v (B) | v2 (B) | norms (3)] in rank order -- split into
 * v_all / v2_all [G * B] (what segmm_loss_fwd_bwd takes as the global view lengths, decoder_leave_focal.py:163-221 on the GLOBAL
 * batch) and the three normalisers summed over the ranks in rank order.  One launch; no torch kernel inside the DP step. */
int segmm_label_stats_unpack(const float* gathered, int G, int B, float* v_all, float* v2_all, float* norms, segmm_stream_t stream);
int segmm_pe_grad(const float* dpre, int ld, int B, int S, int d, float* dpe, int accumulate, segmm_stream_t stream);

/* The loss scalars of compute_loss (decoder_leave_focal.py:490-572) from the per-row terms segmm_loss_fwd_bwd wrote:
 * losses[c] = sum_b parts[b][c] for the 12 loss slots and total[0] = sum_c coef[c] * losses[c] (the weighted sum of
 * :560-571), one launch, fixed summation order.  dlogits != NULL (the n_dl values of d loss / d logits): also gmax[0] = their
 * largest magnitude, and site_scale[i] for every i < n_sites with gain[i] > 0 becomes the power of two that puts gain[i] * gmax at
 * 2^target -- the delayed scales of the BACKWARD tensors, which are linear in d loss / d logits (see segmm_scales_update). */
int segmm_loss_finish(const float* parts, int B, const float* coef, float* losses, float* total, const float* dlogits, int64_t n_dl,
                      float* site_scale, const float* gain, int n_sites, float* gmax, int target, segmm_stream_t stream);

/* K8 -- compute_loss forward + backward in one launch (decoder_leave_focal.py:490-572).
 * part order: 0 interestBPR 1 focal 2 surviveCE 3 interestCE 4 interestKL 5 huber 6 hazard 7 mse 8 mse2.
 * coef/enabled are host arrays of 9; parts: [B, 12] (row stride 12, columns 9..11 zero) per-row contributions already divided by the GLOBAL
 * normalisers, so a column sum (segmm_colsum) over rows -- and over data-parallel ranks -- gives each loss.
 * segmm_label_stats fills v[B] = #(gt==1), v2[B] = #(gt>=0) (after the optional focal rewrite) and the device
 * array norms[3] = {#rows with v < S, B, #(gt != -2)}; a data-parallel trainer all-gathers v/v2 and sums norms
 * over ranks before the loss call, so every rank normalises by the GLOBAL counts without a host sync. */
int segmm_label_stats(const int64_t* gt, int B, int S, int rewritten, float* v, float* v2, float* norms,
                      segmm_stream_t stream);
int segmm_loss_fwd_bwd(int B, int S, const float* logits, const int64_t* gt, const float* bias_w,
                       const float* bias_b, const float* exposure, const float* coef, const int* enabled,
                       int rewritten_ce, int rewritten_kl, int use_mask, const float* norms, const float* v_all,
                       const float* v2_all, int Bg, float* logits_out, float* dlogits, float* parts,
                       segmm_stream_t stream);

/* K9 -- fused AdamW over one flat fp32 range (torch.optim.AdamW semantics; main_for_seq_leave_earlystop_SegMM.py:226,299) */
int segmm_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int step, segmm_stream_t stream);
/* AdamW over an id-embedding table [n_rows, width] (encoder.py:352-362: nn.Embedding inputs) whose gradient is zero outside the
 * rows listed in ids -- the batch's rows (duplicates allowed; ids outside [0, n_rows) are ignored) -- in TWO passes that together
 * are the dense torch.optim.AdamW step (moments and weights of rows without a gradient still decay), element for element with the
 * arithmetic of segmm_adamw:
 *   phase 0: marks the listed rows in `flags`, then updates every UNMARKED row with g = 0.  It reads no gradient, so it can be
 *            enqueued at the START of the step on a stream of its own (the forward and backward only read the listed rows of
 *            the table): 28 B x 90 M parameters of HBM traffic leave the end of the step (config 3's 352 494 x 256 item table);
 *   phase 1: the listed rows, each once (a mark is claimed and cleared by whoever updates the row), with their gradient rows
 *            from g (the dense [n_rows, width] gradient).  Enqueue it after phase 0 AND the backward have completed.
 * flags: n_rows 32-bit words owned by the caller, zero before the first call; zero again after phase 1.  step as in segmm_adamw. */
int segmm_adamw_table(float* p, const float* g, float* m, float* v, int64_t n_rows, int width, const int64_t* ids, int n_ids,
                      uint32_t* flags, float lr, float beta1, float beta2, float eps, float weight_decay, int step, int phase,
                      segmm_stream_t stream);
/* Device-side step state, so that a whole training step (main_for_seq_leave_earlystop_SegMM.py:265-300) can be replayed from its
 * recorded launch sequences with unchanged kernel arguments: two dropout seed words and the optimizer's step count with its bias
 * corrections live in device memory -- a struct of segmm_step_state_bytes() bytes in CALLER-OWNED, 16-byte aligned device memory
 * that segmm_step_bind names for the calls that follow (host-side: each launch carries the pointer in its arguments, the library
 * keeps no __device__ state; several trainers in one process bind their own state before they step).  With nothing bound (or
 * segmm_step_bind(NULL)) the calls use one default state the library allocates on first use.  segmm_step_set initialises the
 * bound state, segmm_step_advance (one thread; the first launch of a
 * step) increments the count, derives new seed words and the corrections 1 - beta^t, segmm_step_get reads them back (it
 * synchronises the stream).  A dropout seed argument with bit 63 set is "live": the kernel XORs the device words into it;
 * segmm_adamw with step == -1 takes the corrections from the device state. */
int segmm_step_state_bytes(void);
int segmm_step_bind(void* state);
int segmm_step_set(uint64_t seed, int step, float beta1, float beta2, segmm_stream_t stream);
int segmm_step_advance(float beta1, float beta2, segmm_stream_t stream);
int segmm_step_get(uint64_t* seed, int* step, float* bias_corrections, segmm_stream_t stream);


/* ---- the callers either side of the training step -------------------------------------------
 * (f)-2 on-device evaluation (my_evaluation.py:73-231, SegRec/main.py:101-117).  Integer results only: for the same
 * float inputs they are bit-exact against numpy/sklearn.
 * segmm_rank_leave: TOP_K_leave (masked = 0: row valid iff view_len < seq_valid) / TOP_K_leave_mask (masked = 1: row
 *   valid iff view_len != #unpadded, padded positions score 1.1).  x [B, ldx] interests, gt [B, S] in {1,0,-1,-2},
 *   perm [B, S] int32 candidate permutations or null.  ranks[B]: 1-based rank of the leave segment in ascending
 *   order with ties by the lower candidate index (np.argsort), 0 for invalid rows; hist[S + 1] (caller-zeroed) is
 *   incremented at [rank].
 * segmm_auc_counts: per segment s (elements seg_off[s] .. seg_off[s+1]) out[s] = {U2, npos, nneg}, U2 = sum over
 *   positives of 2 * #(negatives below) + #(negatives equal); AUC = U2 / (2 npos nneg) = sklearn.roc_auc_score.
 *   label: 1 positive, 0 negative, other values ignored.  One segment = ProbAUC of a batch; one per user = wuAUC.
 * segmm_survival: surv = exp(cumsum(log interest)) (sequential fp32 sum) and the AUC label of each cell. */
int segmm_rank_leave(const float* x, int ldx, const int64_t* gt, const int32_t* perm, int B, int S, int masked, int seq_valid,
                     int32_t* ranks, int32_t* hist, segmm_stream_t stream);
int segmm_auc_counts(const float* score, const int8_t* label, const int64_t* seg_off, int n_seg, int64_t* out,
                     segmm_stream_t stream);
int segmm_survival(const float* interest, int ld, const int64_t* gt, float* surv, int8_t* label, int B, int S,
                   segmm_stream_t stream);
/* (f)-1 resident-table feature gather (dataloader_SegMM.py:271-362 + the trainer's L1 normalisation): out[r, :] =
 * table[idx[r], :] (/ (sum|.| + 1e-6) if normalize), mask[r] = 1, for idx[r] in [0, n_lines); zeros / 0 otherwise. */
int segmm_gather_l1(const float* table, int64_t n_lines, int D, const int64_t* idx, int64_t rows, int normalize, float* out,
                    uint8_t* mask, float* amax, uint16_t* planes, int ld2, float* hdr, const float* scale_in, segmm_stream_t stream);
/* Delayed scaling, end of a pass: arena = the pass's n_rows site headers, site_idx[r] = index of row r's tensor site in
 * site_scale (< 0: none).  site_scale[idx] = the power of two s with max|x| * s in [2^(target-1), 2^target) for every row
 * that was produced; stats[0] += number of rows whose planes were outside the fp16 window.  gain / gmax non-NULL (backward pass):
 * also gain[idx] = max|x| / gmax[0] (segmm_loss_finish turns it into the next step's scale). */
int segmm_scales_update(const float* arena, const int32_t* site_idx, int n_rows, float* site_scale, float* stats, int target,
                        float* gain, const float* gmax, segmm_stream_t stream);
/* DIAGNOSTIC, not part of the reference path: launches `workgroups` x 512 threads that issue `iters` x 48 v_mfma_f32_16x16x32_f16
 * per wave on random operand bits (registers only, the plane GEMM's accumulator order and occupancy); *flops_out = the fp16 MFMA
 * FLOPs of the launch.  bench.py times it to report the SUSTAINED matrix-core rate of the part beside the datasheet peak
 * (power management clocks a random-data MFMA stream down; constant operands do not show it). */
int segmm_probe_mfma_rate(int workgroups, int iters, float* scratch, double* flops_out, segmm_stream_t stream);

/* Arithmetic of the attention kernels (models/encoder.py:44-73,138-161): 0 = exact-fp32 matrix-core products everywhere,
 * 1 = fp16x3 products (22-bit operands, the GEMM engine's arithmetic) where that form is built and measured faster (default;
 * env SEGMM_ATTN=f32|f16|f16all sets the initial value), 2 = fp16x3 wherever it is built.  Returns the previous mode; mode < 0
 * only queries.  Results of the two forms agree to ~1e-6 relative; both are deterministic. */
int segmm_attn_mode(int mode);

/* Tuning / A-B knobs of the library (tile shapes, kernel forms, probes that change occupancy -- none changes results beyond the
 * documented equivalences).  They are read from the environment ONCE, at the first use of the library (SEGMM_<NAME>), and live
 * in one table: segmm_config_dump writes "SEGMM_<NAME>=<value>  # what it selects" lines into buf (at most n bytes incl. the
 * terminator) and returns the length the full listing needs; segmm_config_set changes a knob by its name (without the SEGMM_
 * prefix) and returns its previous value, or a negative error code for an unknown name.  No launch path reads the environment.
 * Timing probes whose results are wrong exist in -DSEGMM_ATT_PROBE / -DSEGMM_GEMM_PROBE builds only. */
int segmm_config_set(const char* name, int value);
int segmm_config_dump(char* buf, int n);

/* (f)-3 SegRec weighted head (ClipRec.forward, SegRec/models/context/ClipRec.py:163-181): out[r] = sum_seg pred[r, seg] *
 * weight[r, seg] * (seg < duration[r]); weight == null: ones, duration == null: no duration mask. */
int segmm_segment_weighted_sum(const float* pred, const float* weight, const int64_t* duration, int64_t rows, int S, float* out,
                               segmm_stream_t stream);

/* up to three column sums out_i[n] = sum_m X_i[m, n] of same-shaped matrices in one launch pair (X1/X2 may be null);
 * workspace: 3 * segmm_colsum_chunks(M) * N floats.  Used for the partial buffers of segmm_layernorm_bwd. */
int segmm_colsum3(const float* X0, const float* X1, const float* X2, int ld, int64_t M, int N, float* out0, float* out1,
                  float* out2, float* workspace, segmm_stream_t stream);

/* CrossMLP ablation (encoder.py:392-396,503-506): nn.AdaptiveAvgPool1d(bins) over the token axis of cat(U[B,Lu,d], V[B,Lv,d]):
 * out[b,i,:] = mean of tokens [floor(i T/bins), ceil((i+1) T/bins)), T = Lu+Lv; and its backward (dU, dV overwritten). */
int segmm_pool_tokens(const float* U, int Lu, const float* V, int Lv, float* out, int B, int d, int bins, segmm_stream_t stream);
int segmm_pool_tokens_bwd(const float* dOut, float* dU, int Lu, float* dV, int Lv, int B, int d, int bins, segmm_stream_t stream);

/* ---- Recorded launch sequences: the training step (main_for_seq_leave_earlystop_SegMM.py:265-300) enqueued from C -----------
 * SURVEY.md 8(b) asks for coarse entry points (embedding, encoder layer, head + loss, optimizer tail) so that a host language
 * does not pay a foreign-function call per kernel.  A PHASE is the resolved launch list of one such part of the step: an array
 * of commands, each naming one stream-taking entry point of this header (op = index into segmm_cmd_op_name) with its arguments
 * in declaration order as 8-byte slots (pointers and integers in .i / .p, float parameters as double in .f) and the stream slot
 * it is enqueued on (0 = main stream, 1 = side stream, 2 = auxiliary stream).  Two pseudo-ops order a stream s >= 1 against the
 * main stream: SEGMM_OP_FORK (stream s waits for everything enqueued on the main stream so far) and SEGMM_OP_JOIN (the main
 * stream waits for stream s); s travels in the command's `stream` field.
 * The host builds a phase ONCE per workload shape -- segmminterest_amd/hipabi.py records the per-op calls of one eager step --
 * and replays it every step: kernel arguments do not change from step to step when the per-step state lives on the device
 * (segmm_step_advance: dropout seed words, optimizer step count) and buffers are persistent; the few that do (the batch's
 * tensors) are patched in place by the host before the call.  Ownership: the command array, structs and host arrays its
 * pointer arguments name (segmm_attn_planes_t, the loss coefficient arrays) belong to the caller and must outlive the call.
 * segmm_run_phase returns the first non-zero return code of a command (0 = everything enqueued); it never synchronises.
 * The named entry points are segmm_run_phase restricted to one kind of phase (they refuse a descriptor of another kind):
 * they are what a maintainer binds per part of the reference's step -- _get_embedding (encoder.py:425-473), one
 * SegFormerXEncoderLayer (encoder.py:189-208) forward / backward, head + compute_loss (decoder_leave_focal.py:490-658),
 * optimizer.step (main...SegMM.py:299). */
#define SEGMM_CMD_MAX_ARGS 48
typedef union { int64_t i; double f; void* p; } segmm_arg_t;
typedef struct { int32_t op; int32_t stream; segmm_arg_t a[SEGMM_CMD_MAX_ARGS]; } segmm_cmd_t;
#define SEGMM_OP_FORK (-1)
#define SEGMM_OP_JOIN (-2)
enum { SEGMM_PHASE_STEP_BEGIN = 0, SEGMM_PHASE_EMBED_FWD = 1, SEGMM_PHASE_LAYER_FWD = 2, SEGMM_PHASE_HEAD_LOSS_FWD = 3,
       SEGMM_PHASE_HEAD_LOSS_BWD = 4, SEGMM_PHASE_LAYER_BWD = 5, SEGMM_PHASE_EMBED_BWD = 6, SEGMM_PHASE_STEP_TAIL = 7, SEGMM_PHASE_KINDS = 8 };
typedef struct {
    int32_t kind;          /* SEGMM_PHASE_* */
    int32_t backbone;      /* 0 / 1: which backbone of a two-tower ('both') model; 0 otherwise */
    int32_t layer;         /* encoder layer of a LAYER_* phase, else 0 */
    int32_t n_cmds;
    const segmm_cmd_t* cmds;
} segmm_phase_t;
/* number of dispatchable entry points / the name of op `op` (NULL outside [0, n)): hosts map names to op ids at load time */
int segmm_cmd_op_count(void);
const char* segmm_cmd_op_name(int op);
/* streams[0 .. n_streams): hipStream_t of the slots (streams[0] = main; n_streams <= SEGMM_MAX_STREAMS); events[2 (s - 1)] and
 * events[2 (s - 1) + 1]: two hipEvent_t of the caller (disable-timing events) per stream s >= 1, used by its fork / join pseudo-ops */
#define SEGMM_MAX_STREAMS 3
int segmm_run_phase(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_step_begin(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_embed_fwd(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_layer_fwd(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_head_loss_fwd(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_head_loss_bwd(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_layer_bwd(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_embed_bwd(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
int segmm_step_tail(const segmm_phase_t* phase, const segmm_stream_t* streams, int n_streams, void* const* events);
/* bytes of p[0 .. bytes) = 0 on `stream` (the clears of the step path as a recordable command) */
/* Small launches that keep the reference's remaining step variants free of host-side tensor ops, so that they can be recorded
 * and replayed like the default step (round 5):
 * segmm_bias_grad: learnable_bias (decoder_leave_focal.py:497-504,649-658) -- g_bias_bias[s] = sum_b dl[b, s] (rows in index
 *   order), g_bias_weight[s] = (s + 1) g_bias_bias[s];
 * segmm_focal_relabel: focal loss first in the loss list rewrites the labels in place (:534-535): gt > 0 -> 1, gt == -1 -> 0;
 * segmm_rand_uniform / segmm_rand_ids: the noUser ablations' random user features in [0, 1) and random user ids in [lo, hi)
 *   (main_for_seq_leave_earlystop_SegMM.py:275-280), segmm_rand_perm_rows: the noPos ablation's fresh permutation of 0 .. S-1 per
 *   row (encoder.py:428-429; S <= 64, written as floats) -- drawn from the counter hash of the dropout streams (seed with bit
 *   63 set: the device-side step words are XORed in): the reference's distributions, not torch's bit streams. */
int segmm_bias_grad(const float* dl, int B, int S, float* g_bias_weight, float* g_bias_bias, segmm_stream_t stream);
int segmm_focal_relabel(int64_t* gt, int64_t n, segmm_stream_t stream);
int segmm_rand_uniform(float* out, int64_t n, uint64_t seed, uint32_t site, segmm_stream_t stream);
int segmm_rand_ids(int64_t* out, int64_t n, int64_t lo, int64_t hi, uint64_t seed, uint32_t site, segmm_stream_t stream);
int segmm_rand_perm_rows(float* out, int rows, int S, uint64_t seed, uint32_t site, segmm_stream_t stream);
int segmm_fill_zero(void* p, int64_t bytes, segmm_stream_t stream);
/* dst[0 .. bytes) = src[0 .. bytes), device to device, on `stream` (non-overlapping ranges) */
int segmm_copy_bytes(void* dst, const void* src, int64_t bytes, segmm_stream_t stream);

/* test hook: multiplier (0 or 1/(1-p)) of elements [0,n) of a dropout site */
int segmm_dropout_mult(float* out, int64_t n, float p, uint64_t seed, uint32_t site, segmm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
