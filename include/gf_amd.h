/* gf_amd.h — C ABI of libgf_amd.so, the MI355X (gfx950) matcher hot-path library.
 *
 * The reference (cvg/glue-factory) is pure Python on PyTorch and has no FFI of its own; the
 * entry points below are what its matcher modules would bind for the hot path, one per fused
 * op, each citing the reference lines it replaces (paths relative to /root/reference).
 * INTEGRATION.md shows the ctypes stubs a glue-factory maintainer would add.
 *
 * Conventions
 *  - Plain pointers to DEVICE memory owned by the caller (PyTorch's caching allocator in the
 *    shipped host code); the library allocates nothing and keeps no global state (ABI 14: the
 *    Sinkhorn schedule, the last process-wide setting, became an argument of the calls).
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues kernels on it (no host
 *    synchronisation, hipGraph-capturable).
 *  - `dtype`: GF_DTYPE_F32 (exact fp32 MFMA, parity mode) or GF_DTYPE_BF16 (bf16 operands, fp32
 *    accumulation and statistics, perf mode).  Statistics / scores / indices are always
 *    float / float / int64.
 *  - Strides are in ELEMENTS; the innermost (channel) dimension is contiguous.  Base pointers
 *    and strides must keep 16-byte alignment of every row (GF_ERR_ALIGN otherwise).
 *  - Return value: 0 on success, a negative GF_ERR_* for rejected arguments, or a positive
 *    hipError_t from the launch.
 */
#ifndef GF_AMD_H
#define GF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_DTYPE_F32 0
#define GF_DTYPE_BF16 1

#define GF_ERR_UNSUPPORTED (-1) /* e.g. a head_dim outside {32, 64, 128} */
#define GF_ERR_SHAPE (-2)
#define GF_ERR_ALIGN (-3)
#define GF_ERR_DTYPE (-4)

/* ABI version; bumped on any signature or workspace-size change (2: gf_attn_bwd's delta workspace doubled; 3: line head + gf_bgemm; 4: smallops; 5: gf_attn_bwd_acc; 9: cast entries with leading dimensions, gf_fold_linear_*, double betas in gf_multi_adam; 10: gf_attn_fwd_ex / GF_ATTN_SPLIT, gf_topk_candidates; 14: gf_sinkhorn_* take `schedule`, gf_sinkhorn_mode removed, gf_probe_hold_cus, gf_linear_dw2, gf_gemm_res2, gf_rowdot2_*; gf_rowdot_fwd / gf_rotary_qk_bwd take a device bias / a base sum; 16: the test diagnostic gf_probe_hold_cus left the product ABI for tests/csrc/gf_test_probe.hip; 17: gf_conv3x3_c64_ld). */
#define GF_AMD_ABI_VERSION 17
int gf_abi_version(void);

/* ---- multi-head attention over keypoints --------------------------------------------------
 * softmax(scale * q k^T) v, flash style (no N x N tensor in HBM).
 * Replaces gluefactory/models/matchers/lightglue.py:97-128 (Attention), :161 (SelfBlock),
 * :203-216 (CrossBlock: call twice, (qk0,qk1,v1) and (qk1,qk0,v0)),
 * gluefactory_nonfree/superglue.py:112-116 and gluefactory/models/matchers/gluestick.py:524-529.
 * q [B,Nq,H,D], k/v [B,Nk,H,D], o [B,Nq,H,D] with strides {batch, token, head}; D == 64 (the tuned LDS-DMA kernels) or
 * D in {32, 128} (generic register-staged kernels; fp32 at 128: forward only, the backward's staging exceeds the LDS).
 * lse [B,H,Nq] = log sum_j exp(scale * q_i.k_j)  (saved for the backward). */
int gf_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                int B, int H, int Nq, int Nk, int D,
                const int64_t* q_strides, const int64_t* k_strides,
                const int64_t* v_strides, const int64_t* o_strides,
                float scale, int dtype, void* stream);
/* gf_attn_fwd_ex: the same with `flags`.  GF_ATTN_SPLIT (bf16 only; ignored for fp32): the softmax weights enter the second
 * product as a hi + lo pair of bf16 values (16 mantissa bits), i.e. scores, softmax and weighted sum are fp32-equivalent on the
 * bf16 operands -- the arithmetic gluestick.py:18-22, 524-529 (@AMP_CUSTOM_FWD_F32) prescribes for GlueStick's attention under
 * mixed precision.  o32 (GF_ATTN_SPLIT only, may be NULL): fp32 copy of the output, [B, Nq, H, 64] contiguous -- the reference's
 * fp32 attention hands its output on in fp32.  gf_attn_bwd_acc takes the same flag (P and dS split in front of dV / dK / dQ);
 * its `o` / `o_strides` then describe that fp32 copy (delta = sum o dO without the bf16 rounding of o). */
#define GF_ATTN_SPLIT 4
int gf_attn_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                   int B, int H, int Nq, int Nk, int D,
                   const int64_t* q_strides, const int64_t* k_strides,
                   const int64_t* v_strides, const int64_t* o_strides,
                   float scale, int dtype, int flags, float* o32, void* stream);

/* gf_attn_cross_bwd (csrc/attention_xbwd.hip): backward of LightGlue's BIDIRECTIONAL cross attention with its shared
 * similarity (lightglue.py:203-216: sim = qk0 qk1^T, m0 = softmax(sim) v1, m1 = softmax(sim^T) v0) for B2 stacked images
 * where image b is paired with image (b + pair) mod B2 (pair = B2 / 2 for [image 0 batch | image 1 batch]).  qk, v
 * [B2,N,H,64] views (strides {batch, token, head}); o = the forward's messages, dout their gradient; lse [B2,H,N] of the
 * direction in which the image's tokens are the QUERIES; stat: workspace of 2 * B2 * H * N floats.  Writes dqk and dv (every
 * row once: nothing to accumulate between launches).  One kernel per image side takes all of its gradients from ONE score
 * tile: 10 MFMA products per tile pair instead of the 14 of two gf_attn_bwd_acc calls.  bf16, D == 64, H <= 4, N % 64 == 0
 * (else GF_ERR_UNSUPPORTED: call gf_attn_bwd_acc twice). */
int gf_attn_cross_bwd(const void* qk, const void* v, const void* o, const void* dout, const float* lse, float* stat,
                      void* dqk, void* dv, int B2, int pair, int H, int N, int D,
                      const int64_t* qk_strides, const int64_t* v_strides, const int64_t* o_strides,
                      const int64_t* do_strides, const int64_t* dqk_strides, const int64_t* dv_strides,
                      float scale, int dtype, void* stream);

/* Backward of gf_attn_fwd (what autograd derives from the lines above).  delta is WORKSPACE of 2*B*H*Nq floats,
 * written by the call: the two per-row vectors the dQ kernel hands to the dK/dV kernel (fp32: delta = rowsum(dout * o)
 * in the first plane; bf16: -lse in the exponent's units and -delta, the initial values of its accumulators). */
int gf_attn_bwd(const void* q, const void* k, const void* v, const void* o,
                const void* dout, const float* lse, float* delta,
                void* dq, void* dk, void* dv,
                int B, int H, int Nq, int Nk, int D,
                const int64_t* q_strides, const int64_t* k_strides,
                const int64_t* v_strides, const int64_t* o_strides,
                const int64_t* do_strides, const int64_t* dq_strides,
                const int64_t* dk_strides, const int64_t* dv_strides,
                float scale, int dtype, void* stream);
/* The same with accumulation: flags bit 0 (1): dq += instead of dq =; bit 1 (2): dk += (fp32 sum of the stored value and
 * the new gradient, one rounding).  Bidirectional cross attention (lightglue.py:203-216) uses one tensor as query in one
 * direction and as key in the other: the second direction's backward adds straight into the first one's results. */
int gf_attn_bwd_acc(const void* q, const void* k, const void* v, const void* o,
                    const void* dout, const float* lse, float* delta,
                    void* dq, void* dk, void* dv,
                    int B, int H, int Nq, int Nk, int D,
                    const int64_t* q_strides, const int64_t* k_strides,
                    const int64_t* v_strides, const int64_t* o_strides,
                    const int64_t* do_strides, const int64_t* dq_strides,
                    const int64_t* dk_strides, const int64_t* dv_strides,
                    float scale, int dtype, int flags, void* stream);

/* ---- assignment head: double softmax with dustbins ------------------------------------------
 * S = a b^T with a [B,M,D], b [B,N,D] (the final_proj outputs, already scaled by D^-1/4),
 * D % 16 == 0, D <= 256, rows contiguous (row stride D, batch strides M*D / N*D).
 *
 * gf_rows_lse: lse[b,i] = log sum_j exp(S_ij + colbias[b,j])   (colbias may be NULL)
 *   Called twice — (a,b) and (b,a) — it gives the row and column normalisers of
 *   lightglue.py:262-263 (log_softmax over dim 2 of sim and of sim^T); with colbias it also
 *   serves the bin-augmented softmax of gluestick.py:772-783 by the caller adding the bin. */
int gf_rows_lse(const void* a, const void* b, const float* colbias, float* lse,
                int B, int M, int N, int D, int dtype, void* stream);

/* gf_rows_argmax: for every row i of a: max_j / argmax_j over j < N of
 *   (alpha * S_ij + colbias[b,j])        -> rowmax[b,i] (float), rowarg[b,i] (int64)
 * With alpha = 2, colbias_j = logsigmoid(z1_j) - c_j this is the row arg-max of the core of the
 * log assignment (lightglue.py:295 `scores[:, :-1, :-1].max(2)`, and the per-layer arg-max of
 * TokenConfidence.loss lightglue.py:81-94) without materialising it; called with (b,a) it
 * gives the column arg-max. */
int gf_rows_argmax(const void* a, const void* b, const float* colbias, float alpha,
                   float* rowmax, int64_t* rowarg,
                   int B, int M, int N, int D, int dtype, void* stream);

/* gf_rows_lse_argmax: ONE pass over b for both per-row statistics of a LightGlue head
 * (lightglue.py:262-263 log_softmax normaliser + lightglue.py:295 / 81-94 row arg-max):
 *   lse[b,i]              = log sum_j exp(S_ij)                      (skipped when lse == NULL)
 *   rowmax / rowarg [b,i] = max / argmax_j alpha * S_ij + logsigmoid(bias_z[b,j]) - bias_n[b,j]
 * Three passes give everything a layer's loss needs: c = gf_rows_lse(b, a);
 * (r, max0, arg0) = gf_rows_lse_argmax(a, b, z1, c); (max1, arg1) = gf_rows_lse_argmax(b, a, z0, r, lse=NULL). */
int gf_rows_lse_argmax(const void* a, const void* b, const float* bias_z, const float* bias_n,
                       float alpha, float* lse, float* rowmax, int64_t* rowarg,
                       int B, int M, int N, int D, int dtype, void* stream);

/* gf_assign_write: materialise the log assignment (lightglue.py:256-268)
 *   out[b,i,j] = alpha*S_ij + rowbias[b,i] + colbias[b,j]      i<M, j<N
 *   out[b,i,N] = bin_col[b,i];  out[b,M,j] = bin_row[b,j];  out[b,M,N] = corner
 * out is [B, M+1, N+1] fp32 contiguous.  LightGlue: alpha=2, rowbias = logsig(z0) - r,
 * colbias = logsig(z1) - c, bin_col = logsig(-z0), bin_row = logsig(-z1), corner = 0.
 * GlueStick (gluestick.py:772-783): alpha=1, rowbias=-r/2, colbias=-c/2, bins from the caller.
 * expsum (NULL or [B] fp32): expsum[b] = sum_{i<M, j<=N} exp(out[b,i,j]), accumulated while the entries are in
 * registers -- the reference's `row_norm` statistic (lightglue.py:602: scores.exp()[:, :-1].sum(2).mean(1) =
 * expsum / M) without re-reading the matrix; fp32 atomics, so its last bits depend on the scheduling. */
int gf_assign_write(const void* a, const void* b, const float* rowbias, const float* colbias,
                    const float* bin_col, const float* bin_row, float alpha, float corner,
                    float* out, float* expsum, int B, int M, int N, int D, int dtype, void* stream);

/* gf_dual_softmax_bwd: the N x N part of the head's backward.  With r_i = LSE_j S_ij and
 * c_j = LSE_i S_ij (gf_rows_lse), gr = dL/dr and gc = dL/dc:
 *   dS[b,i,j] = galpha * G[b,i,j] + exp(S_ij - r_i) * gr[b,i] + exp(S_ij - c_j) * gc[b,j]
 * written in `dtype` ([B,M,N] contiguous) for the two GEMMs dA = dS b, dB = dS^T a.
 * G (fp32, rows of stride ldg inside a [B, M+1, ldg] buffer, i.e. the upstream gradient of the
 * materialised log assignment) may be NULL: in the sparse-loss training path the positives'
 * direct term reaches a/b through autograd of a gather in the host code. */
int gf_dual_softmax_bwd(const void* a, const void* b, const float* r, const float* c,
                        const float* gr, const float* gc, const float* G, int64_t ldg,
                        float galpha, void* dS, int B, int M, int N, int D, int dtype,
                        void* stream);

/* gf_head_bwd: the same backward WITHOUT the dS tensor and without the two GEMMs (bf16, D == 256; anything else
 * returns GF_ERR_UNSUPPORTED and the caller uses gf_dual_softmax_bwd + its own products):
 *   da[b,i,:] = sum_j dS_ij b[b,j,:],   db[b,j,:] = sum_i dS_ij a[b,i,:],
 *   dS_ij = exp(S_ij - r_i) gr[b,i] + exp(S_ij - c_j) gc[b,j]            (S = a b^T, lightglue.py:256-290 autograd)
 * Two passes of one kernel (owner rows of b, then of a): S tiles recomputed on the matrix cores, dS formed in
 * registers and fed straight into the second product.  da [B,M,D], db [B,N,D] in `dtype`, contiguous rows. */
int gf_head_bwd(const void* a, const void* b, const float* r, const float* c, const float* gr, const float* gc,
                void* da, void* db, int B, int M, int N, int D, int dtype, void* stream);

/* ---- mutual nearest neighbour filter (lightglue.py:293-309, superglue.py:301-311,
 * gluestick.py:321-334) from the row/column arg-max vectors:
 * max0 [B,M] (log-score of the row maximum), arg0 [B,M], arg1 [B,N] ->
 * m0 [B,M], m1 [B,N] (int64, -1 = unmatched), s0 [B,M], s1 [B,N]. */
int gf_filter_matches(const float* max0, const int64_t* arg0, const int64_t* arg1, float th,
                      int64_t* m0, int64_t* m1, float* s0, float* s1,
                      int B, int M, int N, void* stream);

/* ---- log-domain Sinkhorn optimal transport (gluefactory_nonfree/superglue.py:186-214) --------
 * Z [B, M+1, N+1] fp32 couplings (scores augmented with the bin score), iterated `iters`
 * times: u = log_mu - LSE_j(Z + v), v = log_nu - LSE_i(Z + u) with the marginals of
 * superglue.py:206-209.  Stores every iterate in u_hist [iters, B, M+1], v_hist [iters, B, N+1]
 * (all the backward needs) and writes out = Z + u + v - norm ([B, M+1, N+1], norm = -log(M+N)).
 * ws: device workspace of gf_sinkhorn_ws_bytes(B, M, N, iters) bytes (same buffer size for both calls).
 * Two schedules of the same recurrence: streaming kernels (one sweep of Z per iteration, two launches each) and, for
 * N % 256 == 0, N <= 2048, chip-resident sweeps (a chunk of <= 8-16 pairs is loaded once and stays in registers + LDS for
 * all iterations, one persistent launch per chunk with per-pair workgroup barriers; csrc/sinkhorn_resident.h).
 * `schedule` (per call; the library keeps no setting): bits 0-1 = 0 streaming only, 1 resident from 5 pairs per launch (the host
 * code's default), 2 resident whenever the problem fits; bit 2 = placement-independent hand-offs only (the resident kernel
 * otherwise publishes through the shared L2 when it FINDS all workgroups of a pair on one XCD at run time -- a speed path,
 * same numbers); bits 8-31 = bound of every inter-workgroup wait of the resident kernel in milliseconds (0 = 10 000).  The resident kernel's workgroups need not be co-resident (a CU held by another
 * stream's kernel only delays them); when a wait outlasts the bound, the pair's out / gZ is NaN in every row -- a loud,
 * skippable failure (train.py:477-480), never a silently wrong number or a hung device.  Same `schedule` for fwd and bwd. */
/* Host-only query (no device needed): the chip-resident distribution for this problem on a device of `ncu` compute units:
 * out[8] = {pairs per launch, workgroups per pair, waves per pair, rows per wave, waves holding one more row, float4 columns
 * per workgroup in the column phase, N / 256, dynamic LDS bytes}.  1 = resident path (out filled), 0 = streaming kernels. */
int gf_sinkhorn_plan(int B, int M, int N, int ncu, int backward, int schedule, int64_t* out);
int64_t gf_sinkhorn_ws_bytes(int B, int M, int N, int iters);
int gf_sinkhorn_fwd(const float* Z, float* out, float* u_hist, float* v_hist, void* ws,
                    int B, int M, int N, int iters, int schedule, void* stream);
/* Backward: gout [B,M+1,N+1] and its row / column sums gsum_row [B,M+1], gsum_col [B,N+1]
 * -> gZ [B,M+1,N+1] (the caller reduces the bin row/column/corner of gZ to d bin_score). */
int gf_sinkhorn_bwd(const float* Z, const float* gout, const float* gsum_row, const float* gsum_col,
                    const float* u_hist, const float* v_hist, float* gZ, void* ws,
                    int B, int M, int N, int iters, int schedule, void* stream);

/* ---- GlueStick line message passing (gluestick.py:589-691): endpoints e = 0..E-1 (E = 2 Nl, partner e ^ 1) sit on
 * junctions idx[b, e] in [0, N).
 * gf_line_csr:    order [B, E] int32 (endpoints grouped by junction, stable) and seg [B, N+1] int32 (segment starts).
 * gf_line_gather: msg [B, E, 3D] = [x[idx[e]] | x[idx[e^1]] | enc[e]]  (get_endpoint_update's MLP input, :609-621).
 * gf_line_segsum: out [B, N, D] = base (or 0) + scale_j * sum_{e on j} (s0[b, e, :D] + s1[b, e^1, :D]); s0 / s1 are
 *                 [B, E, ld] views (row strides ld0 / ld1 in elements, s1 may be NULL); mode 0: scale = 1 (weighted
 *                 sum of line attention, :660-680), mode 1: scale = 1 / count (scatter_reduce "mean", :683-697).
 * gf_line_expand: d [B, E, D] = scale_{idx[e]} * g[b, idx[e]]  (backward of the aggregation). */
int gf_line_csr(const int64_t* idx, int* order, int* seg, int B, int E, int N, void* stream);
int gf_line_gather(const void* x, const int64_t* idx, const void* enc, void* msg, int B, int E, int N, int D,
                   int dtype, void* stream);
int gf_line_segsum(const void* s0, int64_t ld0, const void* s1, int64_t ld1, const int* order, const int* seg,
                   const void* base, void* out, int B, int E, int N, int D, int mode, int dtype, void* stream);
int gf_line_expand(const void* g, const int64_t* idx, const int* seg, void* d, int B, int E, int N, int D,
                   int mode, int dtype, void* stream);

/* ---- GlueStick line-matching head on dense score matrices (gluestick.py:336-376, :772-783; csrc/line_head.hip) ----
 * gf_rows_gather:      out [B,E,D] = x[b, idx[b,e], :] (x [B,N,D]; endpoint descriptors; backward = gf_line_segsum).
 * gf_line_pair_scores: backward == 0: out = raw [B,M,N], raw[a,c] = 1/2 max(S[2a,2c] + S[2a+1,2c+1], S[2a,2c+1] +
 *                      S[2a+1,2c]) for the endpoint scores S [B,2M,2N] (:349-354); backward != 0: out = dS [B,2M,2N]
 *                      for draw [B,M,N] (the winning pairing, recomputed from S, receives 1/2 draw).
 * gf_dense_rowcol:     rows [B,M] / cols [B,N] (either may be NULL) of the [B,M,N] fp32 view z (batch stride, row stride
 *                      ld in elements): mode 0 log-sum-exp (the normalisers of :772-783 before the bin), mode 1 sum.
 * gf_dense_assign:     out [B,M+1,N+1] = [[raw + row_bias_i + col_bias_j, bin_row_i], [bin_col_j, corner]].
 * gf_dense_assign_bwd: draw = G[:, :M, :N] - exp(raw - r_i) A_i - exp(raw - c_j) Bv_j  (G [B,M+1,N+1]). */
int gf_rows_gather(const void* x, const int64_t* idx, void* out, int B, int E, int N, int D, int dtype, void* stream);
int gf_line_pair_scores(const float* S, const float* draw, float* out, int B, int M, int N, int backward, void* stream);
int gf_dense_rowcol(const float* z, int64_t batch_stride, int64_t ld, float* rows, float* cols, int B, int M, int N,
                    int mode, void* stream);
int gf_dense_assign(const float* raw, const float* row_bias, const float* col_bias, const float* bin_row,
                    const float* bin_col, float corner, float* out, int B, int M, int N, void* stream);
int gf_dense_assign_bwd(const float* raw, const float* r, const float* c, const float* A, const float* Bv,
                        const float* G, float* draw, int B, int M, int N, void* stream);

/* ---- per-step utility kernels (csrc/smallops.hip) ---------------------------------------------------------------
 * gf_multi_cast_transpose: ONE launch converts every fp32 master parameter of a model into the compute dtype and writes
 *   the transposed copy of every matrix (the weight of the input-gradient GEMM dx = dy W).  `table` is a DEVICE array of
 *   n_entries records {const float* src; void* dst; void* dst_t; int rows, cols, tile0, tiles_x; const int* perm;
 *   const float* rscale; float scale; int ldt; int flags; int ldd; const int* cperm; int lds; int pad;} (gf_cast_entry_bytes() = 88 each;
 *   lds / ldd = leading dimensions of src / dst, 0 = cols: column blocks of a wider tensor; dst or dst_t may be
 *   NULL), tile0 = running count of the 32 x 32 tiles of the preceding entries, tiles_x = ceil(cols / 32); total_tiles =
 *   the grid.  dst[r][c] = src[perm ? perm[r] : r][cperm ? cperm[c] : c] * (rscale ? rscale[r] : 1) * scale, dst_t[c * ldt + r] the same
 *   value; flags bit 0: dst is fp32 (biases).  Plain parameters: perm = rscale = NULL, scale = 1, ldt = rows.  DERIVED
 *   weights -- the matcher's prepared projections, e.g. LightGlue's Wqkv with its rows gathered into ({q,k,v}, head,
 *   channel) order and the softmax scale folded into the q rows (lightglue.py:97-128), or the stacked (to_qk | to_v) of the
 *   cross block (:196-221) -- are one entry per row block of the same output.
 * gf_weight_grad_map: the gradient of such a row block back to its parameter: out[perm[r]][cperm[c]] = g[r][c] rscale[r] scale.
 * gf_colsum_f32: out[g, c] = sum_r x[g, r, c] for fp32 x [G, R, C], deterministic (ws: gf_colsum_ws_floats(G, C) floats).
 * gf_small_dw:   dw[o, k] = sum_m dy[m, o] x[m, k], fp32, dy [M, O], x [M, K], K <= 8, O * K <= 256 (the gradient of
 *   lightglue.py:52-65 posenc.Wr, of the keypoint / line-endpoint encoders' first layer superglue.py:82-91, gluestick.py:489-521); ws: gf_small_dw_ws_floats(O, K) floats. */
int gf_multi_cast_transpose(const void* table, int n_entries, int total_tiles, int dtype, void* stream);
/* Folding a linear into its consumer (csrc/fold.hip): h = ffn.0(cat[x, out_proj(ctx)]) = [W0a | W0b Wo] cat[x, ctx] + (b0 + W0b bo)
 * (lightglue.py:131-163, :166-221 to_out; superglue.py:137-160 merge -> mlp.0) -- weights only, fp32 FMA arithmetic.
 * gf_fold_linear_fwd: DEVICE table of n_entries records {const float* W0 [R, ldw0]; const float* Wo [K, N]; const float* b0;
 *   const float* bo; const int* cperm; float* Wc [R, N]; float* bc [R]; int R, K, N, ldw0, c0, tile0, tiles_x, pad;}
 *   (gf_fold_entry_bytes() = 88): Wc = W0[:, c0:c0+K] Wo[:, cperm], bc = b0 + W0[:, c0:c0+K] bo (b0 / bo / cperm / bc may be
 *   NULL); an entry owns ceil(R / 64) * (tiles_x + 1) blocks (the extra tile column computes bc), tile0 = running count of
 *   the preceding entries' blocks, tiles_x = ceil(N / 64), total_tiles = the grid.  One launch for all
 *   blocks of a model, in front of gf_multi_cast_transpose, whose entries stack [W0a | Wc] into the compute-dtype weight.
 * gf_fold_linear_bwd: g [R, c0 + N] (gradient of the stacked weight), gb [R] or NULL (gradient of bc) ->
 *   dW0 [R, ldw0] = [g[:, :c0] | g[:, c0:] Wo[:, cperm]^T + gb bo^T], dWo[:, cperm] = W0[:, c0:]^T g[:, c0:] ([K, N]),
 *   dbo [K] = W0[:, c0:]^T gb (written when bo != NULL); the gradient of b0 is gb itself.  Needs ldw0 >= c0 + K. */
int gf_fold_entry_bytes(void);
int gf_fold_linear_fwd(const void* table, int n_entries, int total_tiles, void* stream);
int gf_fold_linear_bwd(const float* g, const float* gb, const float* W0, const float* Wo, const float* bo, const int* cperm,
                       float* dW0, float* dWo, float* dbo, int R, int K, int N, int ldw0, int c0, void* stream);
int gf_cast_entry_bytes(void);
int gf_weight_grad_map(const float* g, float* out, const int* perm, const int* cperm, const float* rscale, float scale,
                       int rows, int cols, void* stream);
int gf_colsum_f32(const float* x, float* ws, float* out, int G, int R, int C, void* stream);
int gf_colsum_ws_floats(int G, int C);
int gf_small_dw(const float* dy, const float* x, float* ws, float* dw, int M, int O, int K, void* stream);
int gf_small_dw_ws_floats(int O, int K);
/* the forward of the same linears: y[m, o] = sum_k x[m, k] w[o, k], fp32, K <= 8 */
int gf_small_fwd(const float* x, const float* w, float* y, int M, int O, int K, void* stream);
/* gf_multi_adam: the Adam update (torch.optim.Adam semantics, amsgrad = maximize = False; train.py:513) of every parameter
 * tensor in ceil(n_entries / 80) launches.  `table`: HOST array of n_entries records {float* p; const float* g; float* m;
 * float* v; long long n; int pad[2];} (gf_adam_entry_bytes() = 48) of DEVICE pointers; it travels by value in the kernel
 * arguments.  lr, step: device fp32 scalars (step = number of updates done so far; it is incremented here); found_inf (device
 * fp32, may be NULL) > 0 skips update and count; grad_scale (device fp32 or NULL) divides the gradients.  beta1 / beta2 are
 * doubles: the bias corrections 1 - beta^t are evaluated in double (as torch's python-float arithmetic does). */
int gf_adam_entry_bytes(void);
int gf_multi_adam(const void* table, int n_entries, const float* lr, float* step, const float* found_inf,
                  const float* grad_scale, double beta1, double beta2, float eps, float weight_decay, void* stream);

/* ---- small batched GEMM with arbitrary element strides (csrc/bgemm.hip): C[b,i,j] = alpha sum_k A[b,i,k] B[b,k,j],
 * strides {batch, row, column} of A [M,K], B [K,N], C [M,N]; fp32 operands use the exact-fp32 MFMA.  The products of
 * two ACTIVATION tensors outside the fused kernels: the line head's endpoint scores and their gradients
 * (gluestick.py:345-347), the fp32 parity mode of the assignment-head backward (lightglue.py:256-290 autograd). */
int gf_bgemm(const void* a, const void* b, void* c, int batch, int M, int N, int K,
             const int64_t* a_strides, const int64_t* b_strides, const int64_t* c_strides,
             float alpha, int dtype, void* stream);

/* ---- weight-streaming GEMM of the tall-and-skinny linear layers (default path of every nn.Linear / Conv1d(k=1)
 * forward and, with the transposed weight, of every input-gradient GEMM: lightglue.py:131-221,271-290,
 * superglue.py:70-160, gluestick.py:465-586):
 *   y[m, n] = sum_k [x0 | x1][m, k] w[n, k] + bias[n] (+ res[m, n]), then optionally the rotary rotation
 *   (lightglue.py:42-49) of channel pairs (2i, 2i+1) of the output channels [0, rot_n) by cs[m, (n mod 64)] =
 *   interleaved (cos, sin) of token m (cs [M, 64] fp32; head dim 64).
 * x0 [M, K0] (row stride ld0), x1 [M, K1] or NULL (K1 = 0 or K1 == K0: the FFN input cat[x, message] without the
 * concatenation), w [N, K0+K1] (row stride ldw), res / y [M, N] (y may alias res), bias fp32 or NULL.
 * dtype GF_F32 (exact fp32 MFMA) or GF_BF16 (fp32 accumulation).  Supported: N % 32 == 0 and
 * K0 + K1 in {32, 64, 128, 256, 512} (fp32: <= 256); otherwise GF_ERR_UNSUPPORTED (the host splits K or uses the
 * library).  Rows of x / w 16-byte aligned, rows of y / res 8-byte (bf16) or 16-byte (fp32) aligned. */
int gf_gemm(const void* x0, const void* x1, const void* w, const float* bias, const void* res, void* y,
            const float* cs, int rot_n, int M, int N, int K0, int K1,
            int64_t ld0, int64_t ld1, int64_t ldw, int64_t ldr, int64_t ldy, int dtype, void* stream);
/* gf_gemm_res2: the same with TWO residual inputs, y = [x0 | x1] W^T + bias + res + res_b -- the input-gradient GEMM at the
 * point where a per-layer loss head's gradient and the block's residual gradient meet (autograd of lightglue.py:131-221 +
 * :271-290): both ride in the epilogue instead of being added by a separate elementwise kernel first.  bf16, the streamed-
 * activation kernel's shapes only (K0 + K1 in {256, 512}, N % 256 == 0, M % 64 == 0); GF_ERR_UNSUPPORTED otherwise. */
int gf_gemm_res2(const void* x0, const void* x1, const void* w, const float* bias, const void* res, const void* res_b,
                 void* y, int M, int N, int K0, int K1, int64_t ld0, int64_t ld1, int64_t ldw, int64_t ldr,
                 int64_t ldr_b, int64_t ldy, int dtype, void* stream);

/* ---- weight / bias gradient of a linear layer (autograd of every nn.Linear on the path,
 * lightglue.py:131-221, 271-290): dW[n][k] = sum_m dY[m][n] X[m][k], db[n] = sum_m dY[m][n]
 * with dY [M,Nout], X [M,K] row-major in `dtype` and fp32 outputs dW [Nout,K], db [Nout]
 * (db may be NULL).  Split-M MFMA kernel + deterministic slice reduction; ws must hold
 * gf_linear_dw_ws_bytes(M, Nout, K) bytes.  Nout and K must be multiples of 8 (bf16) / 4 (f32). */
int64_t gf_linear_dw_ws_bytes(int M, int Nout, int K);
int gf_linear_dw(const void* dy, const void* x, float* dw, float* db, void* ws,
                 int M, int Nout, int K, int dtype, void* stream);
/* gf_linear_dw2: the same for a TWO-SOURCE input x = [x1 | x2] (x1 [M, K1], x2 [M, K - K1], each contiguous) without building
 * the concatenation -- the weight gradient of `ffn.0(cat[x, message])`, lightglue.py:140-148,196-221 -- in ONE launch: dY is
 * streamed once instead of twice, dw [Nout, K] comes out whole (no cat of two halves).  bf16, Nout, K1 and K - K1 multiples
 * of 128 (GF_ERR_UNSUPPORTED otherwise: call gf_linear_dw per source); workspace of gf_linear_dw_ws_bytes(M, Nout, K). */
int gf_linear_dw2(const void* dy, const void* x1, const void* x2, int K1, float* dw, float* db, void* ws,
                  int M, int Nout, int K, int dtype, void* stream);

/* ---- BatchNorm1d (+ReLU) over channels-last activations [M, C] (training: batch statistics) ------
 * Replaces the Conv1d -> BatchNorm1d -> ReLU tails of superglue.py:70-79 / gluestick.py:465-474.
 * gf_bn_stats:     part [gf_bn_nblk(M)][2][C] = per-block (sum x, sum x^2); the caller reduces the
 *                  blocks (and all-reduces across ranks for SyncBatchNorm) and derives mean / rstd.
 * gf_bn_act_fwd:   y = act((x - mean) * rstd * gamma + beta), act = ReLU if relu else identity.
 * gf_bn_bwd_stats: part = per-block (sum dz, sum dz * xhat), dz = dy * (z > 0 or 1).
 * gf_bn_bwd_dx:    dx = gamma * rstd * (dz - m1 - xhat * m2) with m1 = sum dz / n, m2 = sum dz*xhat / n
 *                  (eval mode: pass zeros for m1/m2).  C must be a multiple of 8 (bf16) / 4 (f32). */
int gf_bn_nblk(int M);
int gf_bn_stats(const void* x, float* part, int M, int C, int dtype, void* stream);
int gf_bn_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, void* y, int M, int C, int relu, int dtype, void* stream);
int gf_bn_bwd_stats(const void* x, const void* dy, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, float* part, int M, int C, int relu,
                    int dtype, void* stream);
int gf_bn_bwd_dx(const void* x, const void* dy, const float* mean, const float* rstd,
                 const float* gamma, const float* beta, const float* m1, const float* m2,
                 void* dx, int M, int C, int relu, int dtype, void* stream);
/* gf_bn_finalize_fwd / _bwd: the glue between the partial sums and the apply pass in ONE launch each (single
 * process; under SyncBatchNorm the caller all-reduces the summed partials itself and finalises on the host side):
 *   fwd: mean, var (biased), rstd = 1/sqrt(var + eps) from part = gf_bn_stats' output and the row count n; when
 *        run_mean / run_var are given they are updated in place as nn.BatchNorm1d does (momentum, unbiased variance);
 *   bwd: dbeta = sum dz, dgamma = sum dz xhat and m1 = dbeta / n, m2 = dgamma / n from gf_bn_bwd_stats' output. */
int gf_bn_finalize_fwd(const float* part, int nblk, int C, float n, float eps, float momentum,
                       float* mean, float* var, float* rstd, float* run_mean, float* run_var, void* stream);
int gf_bn_finalize_bwd(const float* part, int nblk, int C, float n, float* dbeta, float* dgamma,
                       float* m1, float* m2, void* stream);
/* SyncBatchNorm over several statistics sets with ONE exchange per direction (ABI 16; train.py:338 converts every BatchNorm1d of
 * superglue.py:70-79 / gluestick.py:465-474, which the reference calls once per image): gf_bn_pack_sums reduces the block
 * sums of `sets` gf_bn_stats / gf_bn_bwd_stats outputs (part = [sets][nblk][2][C]) into packed = [sets][2][C] sums followed by
 * the `sets` row counts (n_local each) -- the ONE buffer the host all-reduces across ranks -- and, when local_copy is given, keeps
 * the un-reduced sums ([sets][2][C]: the local dbeta / dgamma that DDP averages).  gf_bn_finalize_sets_fwd: mean / biased var /
 * rstd of every set (mvr = [sets][3][C]) from the reduced buffer, the running statistics updated set after set with the
 * reduced counts.  gf_bn_finalize_sets_bwd: m12 = [sets][2][C] = reduced sums / counts (counts: the forward's reduced counts).
 * gf_bn_replay_running_n: gf_bn_replay_running with per-set device-side counts. */
int gf_bn_pack_sums(const float* part, int sets, int nblk, int C, float n_local, float* packed, float* local_copy,
                    void* stream);
int gf_bn_finalize_sets_fwd(const float* packed, int sets, int C, float eps, float momentum, float* mvr,
                            float* run_mean, float* run_var, void* stream);
int gf_bn_finalize_sets_bwd(const float* packed, const float* counts, int sets, int C, float* m12, void* stream);
int gf_bn_replay_running_n(const float* mvr, const float* counts, int sets, int C, float momentum, float* run_mean,
                           float* run_var, const float* skip, void* stream);
/* gf_bn_replay_running (ABI 15): the running statistics take the batch statistics of `sets` forward calls a SECOND time,
 * set after set (mvr = [sets][3][C] rows mean / biased var / rstd as gf_bn_finalize_fwd wrote them, n rows per set).  The
 * reference wraps its GNN layers in torch.utils.checkpoint while training (gluefactory_nonfree/superglue.py:160-169 always;
 * gluefactory/models/matchers/gluestick.py:724-757 with `checkpointed: true`): the backward re-runs their forward in
 * training mode, so every BatchNorm1d inside updates running_mean / running_var / num_batches_tracked twice per step.
 * Launched from the backward of the fused BatchNorm op to leave the same buffers behind.  `skip` (nullable): a device-side
 * flag; when it is non-zero (or NaN) the launch leaves the buffers alone -- the reference `continue`s BEFORE its backward on
 * a non-finite / non-differentiable loss (train.py:477-488), so such a step updates the statistics once, not twice. */
int gf_bn_replay_running(const float* mvr, int sets, int C, float n, float momentum, float* run_mean, float* run_var,
                         const float* skip, void* stream);

/* ---- ground-truth nearest neighbours under a homography (gluefactory/geometry/gt_generation.py:120-150)
 * For every point i of the "own" set [B,No,2] (own = its coordinates, own_warped = the same points
 * warped into the other image) against the "other" set [B,Ns,2] (+ its warped copy):
 *   d_own(i,j) = |own_warped_i - oth_j|^2,  d_oth(i,j) = |own_i - oth_warped_j|^2,  d = max(d_own, d_oth)
 *   arg[b,i] = argmin_j d (lowest index on ties), dmin[b,i] = min_j d, own_min[b,i] = min_j d_own.
 * Called as (kp0, H kp0, kp1, H^-1 kp1) and (kp1, H^-1 kp1, kp0, H kp0) it yields everything the
 * labelling needs (mutual arg-mins, positive / negative thresholds) without any [B,M,N] tensor. */
int gf_gt_nn(const float* own, const float* own_warped, const float* oth, const float* oth_warped,
             int64_t* arg, float* dmin, float* own_min, int B, int No, int Ns, void* stream);

/* ---- frozen SuperPoint extractor tails (gluefactory/models/extractors/superpoint_open.py; the
 * convolutions stay on the stock library).
 * gf_bias_act_bn_nhwc: one pass for the VGGBlock tail Conv2d(no bias) -> +bias -> ReLU -> BatchNorm2d(eval)
 *   (superpoint_open.py:37-75) on a channels-last activation x [B,H,W,C] (C fastest, C % (16/sizeof) == 0):
 *   y = act(x + bias[c]) * scale[c] + shift[c], scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
 *   pool != 0 additionally applies MaxPool2d(2,2) (:98-105) and writes y [B,H/2,W/2,C]; otherwise y may alias x.
 * gf_nms_scores: simple_nms (:19-34) with radius 1..4 on scores [B,H,W] fp32 -> out (0 where suppressed),
 *   plus the border removal of :160-164 (out = -1 inside `border` pixels of the frame; 0 disables). */
int gf_bias_act_bn_nhwc(const void* x, void* y, const float* bias, const float* scale, const float* shift,
                        int B, int H, int W, int C, int relu, int pool, int dtype, void* stream);
/* gf_conv1_bias_act_bn: the whole first VGG block on a 1-channel image (superpoint_open.py:98-100:
 *   Conv2d(1, 64, 3, padding=1) -> +bias -> ReLU -> BatchNorm2d(eval)) in one pass: img [B,H,W], w [64,1,3,3] (both
 *   `dtype`), out [B,H,W,64] channels-last -- the largest activation of the extractor is written exactly once. */
int gf_conv1_bias_act_bn(const void* img, const void* w, const float* bias, const float* scale, const float* shift,
                         void* out, int B, int H, int W, int C, int relu, int dtype, void* stream);
/* gf_conv3x3_c64: one 64 -> 64 channel VGG block (superpoint_open.py:37-75 VGGBlock, :98-105 backbone.0.1 / 1.0 / 1.1:
 *   Conv2d(64, 64, 3, padding=1) -> +bias -> ReLU -> BatchNorm2d(eval) [-> MaxPool2d(2, 2)]) as one implicit-GEMM
 *   kernel.  x [B,H,W,64] channels-last bf16, w [9 taps (ky*3+kx)][64 c_out][64 c_in] bf16, y [B,H,W,64] or, with
 *   pool != 0, [B,H/2,W/2,64].  dtype must be GF_BF16 (GF_ERR_DTYPE), H % 8 == 0 and W % 32 == 0
 *   (GF_ERR_UNSUPPORTED: the caller uses the library convolution + gf_bias_act_bn_nhwc). */
int gf_conv3x3_c64(const void* x, const void* w, const float* bias, const float* scale, const float* shift, void* y,
                   int B, int H, int W, int relu, int pool, int dtype, void* stream);
/* gf_conv3x3_c64_ld (ABI 17): the same with the output's pixel stride ldy (elements, >= 64, % 8 == 0) as an argument: the kernel
 *   writes the 64 channels of each pixel into a slice of a WIDER channels-last tensor -- a 64 -> 128 block (backbone.2.0,
 *   superpoint_open.py:101-103) is two calls, one per half of the output channels (w / bias / scale / shift of that half,
 *   y + 64 * half), tail fused, instead of the library convolution + a tail pass. */
int gf_conv3x3_c64_ld(const void* x, const void* w, const float* bias, const float* scale, const float* shift, void* y,
                      int64_t ldy, int B, int H, int W, int relu, int pool, int dtype, void* stream);
int gf_nms_scores(const float* scores, float* out, int B, int H, int W, int radius, int border, void* stream);
/* gf_nms_candidates: the same NMS, but instead of the dense map the surviving maxima with a POSITIVE score outside the
 * border go to per-image lists cand_scores / cand_idx [B, cap] (flat pixel index y W + x), cap = gf_nms_candidates_cap(H, W,
 * radius): a fixed segment per kernel tile, filled deterministically, unused slots keep the caller's fill (use a negative
 * score); a segment holds every set of points more than `radius` apart that fits its tile, so entries are dropped only
 * on plateaus of exactly tied scores.  A top-k over the list = the top-k over the dense map (superpoint_open.py:165-170)
 * as long as it does not run into the zeros. */
int gf_nms_candidates_cap(int H, int W, int radius);
/* gf_topk_candidates (csrc/topk.hip): the K largest entries of each of B lists scores [B, n] (fp32; a NEGATIVE score marks
 * an unfilled slot and ranks below every real entry, in list order), sorted by descending score, ties by ascending list
 * position: out_scores [B, K] fp32 and out_payload [B, K] int64 = payload [B, n] (int32) of the selected entries -- the
 * `torch.topk(scores, k, sorted=True)` + index gather of superpoint_open.py:165-170 on the candidate lists, without the
 * hipMemsetAsync nodes of the library implementation (a captured graph holding those faults on its second replay on
 * ROCm 7.2).  K <= min(n, 4096), else GF_ERR_UNSUPPORTED. */
int gf_topk_candidates(const float* scores, const int* payload, float* out_scores, int64_t* out_payload, int B, int n, int K,
                       void* stream);
int gf_nms_candidates(const float* scores, float* cand_scores, int* cand_idx, int B, int H, int W, int radius, int border,
                      void* stream);
/* gf_detector_scores: tail of the detector head (superpoint_open.py:105-108 detector.1 = Conv2d(256,65,1) [+ReLU] + BatchNorm(eval),
 * :141-147 softmax over the 65 channels, dustbin dropped, 8 x 8 cells unfolded): y [B,h,w,65] = the bias-free convolution
 * output, channels-last, 16-byte aligned; scores [B, 8h, 8w] fp32. */
int gf_detector_scores(const void* y, const float* bias, const float* scale, const float* shift, float* scores,
                       int B, int h, int w, int relu, int dtype, void* stream);
/* gf_sample_descriptors: sample_descriptors (:10-16) fused with the dense map's L2 normalisation (:149):
 *   out[b,n,:] = normalize(sum over the 4 bilinear corners of w_k * normalize(map[b,y_k,x_k,:])), zero padding,
 *   x_pix = (kp_x + 0.5) / stride - 0.5.  map [B,h,w,C] channels-last in `dtype` (C % 64 == 0), kpts [B,N,2] fp32
 *   pixel coordinates WITHOUT the +0.5 output offset, out [B,N,C] fp32. */
int gf_sample_descriptors(const void* map, const float* kpts, float* out, int B, int N, int h, int w, int C,
                          int stride, int dtype, void* stream);

/* ---- single-output linear heads z[m] = x[m,:] . w + b (matchability / token-confidence logits,
 * lightglue.py:71,275-276,285-286).  x [M,C] in `dtype`, w [C] fp32, z/dz [M] fp32.
 * gf_rowdot_bwd writes dx = dz * w (+ base, when base != NULL: the running sum of a gradient chain; dx may alias base)
 * when dx != NULL (pass NULL for a detached input) and per-block
 * partials part [gf_rowdot_nblk(M)][C+1] whose column sums are (dw[0..C), db). */
int gf_rowdot_nblk(int M);
int gf_rowdot_fwd(const void* x, const float* w, float bias, const float* bias_dev, float* z, int M, int C, int dtype,
                  void* stream);      /* z = x w + bias + (bias_dev ? *bias_dev : 0): bias_dev = a DEVICE scalar (the nn.Linear bias) */
int gf_rowdot_bwd(const void* x, const float* dz, const float* w, void* dx, const void* base, float* part,
                  int M, int C, int dtype, void* stream);
/* gf_rowdot2_*: TWO single-output heads on the same rows -- a LightGlue layer's matchability and token-confidence logits
 * (lightglue.py:275-276 / :285-286, :71) -- with one read of x: z0 = x w0 + *b0, z1 = x w1 + *b1 (b0 / b1: device scalars or
 * NULL).  Backward: dx = base + dz0 * w0 (only head 0 sees the un-detached descriptors, lightglue.py:81-94; dx may be NULL)
 * and per-block partials part [gf_rowdot_nblk(M)][2][C+1] whose column sums are (dw0, db0) and (dw1, db1). */
int gf_rowdot2_fwd(const void* x, const float* w0, const float* w1, const float* b0, const float* b1, float* z0, float* z1,
                   int M, int C, int dtype, void* stream);
int gf_rowdot2_bwd(const void* x, const float* dz0, const float* dz1, const float* w0, void* dx, const void* base,
                   float* part, int M, int C, int dtype, void* stream);

/* ---- fused deep-supervision loss of one LightGlue layer (lightglue.py:598-657 `loss`,
 * utils/losses.py:6-73 NLL with dustbins on the non-zero weights, lightglue.py:81-94 token confidence).
 * Inputs are the head statistics, never the [B,M+1,N+1] matrix: md0 [B,M,D] / md1 [B,N,D] (final_proj
 * outputs, scaled by D^-1/4), matchability logits z0 [B,M] / z1 [B,N], r [B,M] / c [B,N] (gf_rows_lse),
 * positives as COO (pos_b, pos_i, pos_j)[P] (any order, duplicates allowed; entries with pos_j < 0 are
 * skipped, so a fixed-length "one slot per row" list needs no compaction), dustbin weights neg0 / neg1.
 *   A_ij = 2 md0_i.md1_j - r_i - c_j + logsig(z0_i) + logsig(z1_j),  A_i,N = logsig(-z0_i),  A_M,j = logsig(-z1_j)
 * gf_lg_loss_fwd: acc [B,4] = { sum_pos A_ij, sum_i neg0 A_i,N + sum_j neg1 A_M,j, sum_i bce(t0_i, tgt0_i),
 *   sum_j bce(t1_j, tgt1_j) }.  With token logits t0 / t1 (NULL for the last layer): v0/a0, v1/a1 are the
 *   gf_rows_lse_argmax results, fin0 / fin1 the final layer's arg-max incl. dustbin, and
 *   tgt0 / tgt1 receive the 0/1 targets (layer arg-max incl. dustbin == final) for the backward.
 * gf_lg_loss_bwd_tokens: from gacc = dL/dacc [B,4]: dz0, dz1, dt0, dt1 (dense) and gr = dL/dr, gc = dL/dc
 *   restricted to the direct terms (the caller feeds them to gf_dual_softmax_bwd).
 * gf_lg_loss_bwd_rows: dmd0[b,i,:] += 2 gacc[b,0] md1[b,j,:], dmd1[b,j,:] += 2 gacc[b,0] md0[b,i,:]
 *   (atomic accumulation INTO dmd, after the dense part has been written). */
int gf_lg_loss_fwd(const void* md0, const void* md1, const float* z0, const float* z1,
                   const float* r, const float* c,
                   const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int64_t P,
                   const float* neg0, const float* neg1,
                   const float* t0, const float* t1,
                   const float* v0, const int64_t* a0, const float* v1, const int64_t* a1,
                   const int64_t* fin0, const int64_t* fin1,
                   float* tgt0, float* tgt1, float* acc,
                   int B, int M, int N, int D, int dtype, void* stream);
int gf_lg_loss_bwd_tokens(const float* z0, const float* z1, const float* neg0, const float* neg1,
                          const float* t0, const float* t1, const float* tgt0, const float* tgt1,
                          const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int64_t P,
                          const float* gacc, float* dz0, float* dz1, float* dt0, float* dt1,
                          float* gr, float* gc, int B, int M, int N, void* stream);
int gf_lg_loss_bwd_rows(const void* md0, const void* md1,
                        const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int64_t P,
                        const float* gacc, void* dmd0, void* dmd1,
                        int B, int M, int N, int D, int dtype, void* stream);

/* ---- fused elementwise ops of the transformer block ---------------------------------------
 * Rotary embedding applied in place to the q and k thirds of a fused [B,N,3,H,D] projection
 * (lightglue.py:42-49,159-160): cs [B,N,D] holds cos in the even and sin in the odd slot of
 * each pair (one angle per pair).  `inverse` applies the transpose rotation (backward). */
int gf_rotary_qk(void* qkv, const float* cs, int B, int N, int H, int D, int inverse,
                 int dtype, void* stream);
/* Backward of the rotation: dqkv holds the gradients w.r.t. the ROTATED q,k (first two thirds,
 * rotated back in place) and qkv_rot the rotated values saved by the forward;
 * dtheta [B,N,D/2] (fp32) receives the gradient w.r.t. the pair angles
 * sum_{q,k,heads} (g_odd * y_even - g_even * y_odd), which autograd carries to posenc.Wr;
 * dtheta_base (may be NULL, may alias dtheta): added to it -- the angles are shared by all L layers, so the running sum of
 * the layers whose backward already ran rides in this launch instead of L - 1 separate adds. */
int gf_rotary_qk_bwd(void* dqkv, const void* qkv_rot, const float* cs, float* dtheta, const float* dtheta_base,
                     int B, int N, int H, int D, int dtype, void* stream);
/* LayerNorm(affine) + GELU(erf) over rows of x [R, C] (lightglue.py:143-148 ffn.1, ffn.2):
 * y = gelu(ln(x) * gamma + beta); saves mean / rstd [R] for the backward. */
int gf_ln_gelu_fwd(const void* x, const float* gamma, const float* beta, void* y,
                   float* mean, float* rstd, int R, int C, float eps, int dtype, void* stream);
/* Backward: dx [R,C]; dgamma/dbeta partial sums [nblk, C] fp32 (nblk = gf_ln_gelu_nblk(R)),
 * reduced by the caller. */
int gf_ln_gelu_nblk(int R);
int gf_ln_gelu_bwd(const void* x, const float* gamma, const float* beta, const float* mean,
                   const float* rstd, const void* dy, void* dx, float* dgamma_part,
                   float* dbeta_part, int R, int C, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GF_AMD_H */
