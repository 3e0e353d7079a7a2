// GlueStick line-matching head (gluefactory/models/matchers/gluestick.py:336-376 and log_double_softmax :772-783) on a
// DENSE [B, M, N] fp32 score matrix -- the line scores are not a factorised product (they are a max over the two
// endpoint pairings), so the point head's fused a b^T kernels do not apply.  The matrices are small (512 x 512 lines,
// 1024 x 1024 endpoints per pair): every kernel here is a coalesced HBM/L2 pass.
//   gf_rows_gather        out[b, e, :] = x[b, idx[b, e], :]                 (endpoint descriptors; bwd: gf_line_segsum)
//   gf_line_pair_scores   raw[a, c] = 1/2 max(S[2a,2c] + S[2a+1,2c+1], S[2a,2c+1] + S[2a+1,2c])   (:349-354) and its
//                         backward (the gradient goes to the pairing that won, recomputed from S)
//   gf_dense_rowcol       per-row and per-column reduction of a [B, M, N] view: log-sum-exp (the two softmax normalisers
//                         of :772-783 without the bin, which the host folds in with logaddexp) or plain sum (the row /
//                         column sums of the incoming gradient for the backward)
//   gf_dense_assign       out[b] = [[raw + rb_i + cb_j, br_i], [bc_j, corner]]   ([B, M+1, N+1], the bin-augmented
//                         log assignment; same argument roles as gf_assign_write)
//   gf_dense_assign_bwd   d raw = G - exp(raw - r_i) A_i - exp(raw - c_j) B_j   (autograd of the above; A, B = the
//                         gradient arriving at r and c, computed by the host from gf_dense_rowcol sums)
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ void rows_gather_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx, T* __restrict__ out,
                                   int E, int N, int D) {
    constexpr int VEC = 16 / sizeof(T);
    const int e = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x * VEC;
    if (c >= D) return;
    const int j = (int)idx[(size_t)b * E + e];
    *reinterpret_cast<u32x4*>(out + ((size_t)b * E + e) * D + c) = *reinterpret_cast<const u32x4*>(x + ((size_t)b * N + j) * D + c);
}

// one thread per line pair (a, c)
__global__ void pair_scores_fwd_kernel(const float* __restrict__ S, float* __restrict__ raw, int M, int N) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y, b = blockIdx.z;
    if (c >= N) return;
    const float* s0 = S + ((size_t)b * 2 * M + 2 * a) * 2 * N + 2 * c;
    const f32x2 top = *reinterpret_cast<const f32x2*>(s0), bot = *reinterpret_cast<const f32x2*>(s0 + 2 * N);
    raw[((size_t)b * M + a) * N + c] = 0.5f * fmaxf(top[0] + bot[1], top[1] + bot[0]);
}
__global__ void pair_scores_bwd_kernel(const float* __restrict__ S, const float* __restrict__ draw, float* __restrict__ dS,
                                       int M, int N) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y, b = blockIdx.z;
    if (c >= N) return;
    const size_t o = ((size_t)b * 2 * M + 2 * a) * 2 * N + 2 * c;
    const f32x2 top = *reinterpret_cast<const f32x2*>(S + o), bot = *reinterpret_cast<const f32x2*>(S + o + 2 * N);
    const float g = 0.5f * draw[((size_t)b * M + a) * N + c];
    const bool straight = top[0] + bot[1] >= top[1] + bot[0];      // ties: torch.maximum splits the gradient evenly; exact ties only arise
                                                                   // from duplicate endpoints, where both pairings sum to the same junction gradients
    const f32x2 dt = {straight ? g : 0.f, straight ? 0.f : g}, db = {straight ? 0.f : g, straight ? g : 0.f};
    *reinterpret_cast<f32x2*>(dS + o) = dt;
    *reinterpret_cast<f32x2*>(dS + o + 2 * N) = db;
}

// rows: one wave per row.  mode 0: log-sum-exp, mode 1: sum
__global__ __launch_bounds__(256) void dense_rows_kernel(const float* __restrict__ z, int64_t sb, int64_t ld, float* __restrict__ out,
                                                         int M, int N, int mode) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* p = z + (size_t)b * sb + (size_t)row * ld;
    if (mode == 1) {
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += p[j];
        s = wave_allsum(s);
        if (lane == 0) out[(size_t)b * M + row] = s;
        return;
    }
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) mx = fmaxf(mx, p[j]);
    mx = wave_allmax(mx);
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += fast_exp2((p[j] - mx) * GF_LOG2E);
    s = wave_allsum(s);
    if (lane == 0) out[(size_t)b * M + row] = mx + fast_log2(s) * GF_LN2;
}
// columns: 64 columns per workgroup, 4 row slices (one per wave), online max / sum, LDS merge
__global__ __launch_bounds__(256) void dense_cols_kernel(const float* __restrict__ z, int64_t sb, int64_t ld, float* __restrict__ out,
                                                         int M, int N, int mode) {
    __shared__ float sm[4][64], ss[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), b = blockIdx.y, w = threadIdx.x >> 6;
    const float* p = z + (size_t)b * sb + col;
    float mx = mode == 1 ? 0.f : -INFINITY, s = 0.f;
    if (col < N) {
        for (int i = w; i < M; i += 4) {
            const float v = p[(size_t)i * ld];
            if (mode == 1) s += v;
            else if (v > mx) { s = s * fast_exp2((mx - v) * GF_LOG2E) + 1.f; mx = v; }
            else s += fast_exp2((v - mx) * GF_LOG2E);
        }
    }
    sm[w][threadIdx.x & 63] = mx;
    ss[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && col < N) {
        const int l = threadIdx.x & 63;
        if (mode == 1) out[(size_t)b * N + col] = ss[0][l] + ss[1][l] + ss[2][l] + ss[3][l];
        else {
            const float m4 = fmaxf(fmaxf(sm[0][l], sm[1][l]), fmaxf(sm[2][l], sm[3][l]));
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += sm[k][l] == -INFINITY ? 0.f : ss[k][l] * fast_exp2((sm[k][l] - m4) * GF_LOG2E);
            out[(size_t)b * N + col] = m4 + fast_log2(t) * GF_LN2;
        }
    }
}

__global__ void dense_assign_kernel(const float* __restrict__ raw, const float* __restrict__ rb, const float* __restrict__ cb,
                                    const float* __restrict__ br, const float* __restrict__ bc, float corner,
                                    float* __restrict__ out, int M, int N) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j > N) return;
    float v;
    if (i < M) v = j < N ? raw[((size_t)b * M + i) * N + j] + rb[(size_t)b * M + i] + cb[(size_t)b * N + j] : br[(size_t)b * M + i];
    else v = j < N ? bc[(size_t)b * N + j] : corner;
    out[((size_t)b * (M + 1) + i) * (N + 1) + j] = v;
}
__global__ void dense_assign_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ r, const float* __restrict__ c,
                                        const float* __restrict__ A, const float* __restrict__ Bv, const float* __restrict__ G,
                                        float* __restrict__ draw, int M, int N) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= N) return;
    const size_t o = ((size_t)b * M + i) * N + j;
    const float x = raw[o];
    draw[o] = G[((size_t)b * (M + 1) + i) * (N + 1) + j] - fast_exp2((x - r[(size_t)b * M + i]) * GF_LOG2E) * A[(size_t)b * M + i]
              - fast_exp2((x - c[(size_t)b * N + j]) * GF_LOG2E) * Bv[(size_t)b * N + j];
}

}  // namespace

extern "C" int gf_rows_gather(const void* x, const int64_t* idx, void* out, int B, int E, int N, int D, int dtype, void* stream) {
    if (B <= 0 || E <= 0 || N <= 0 || D <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) {
        if (D % 4) return GF_ERR_ALIGN;
        rows_gather_kernel<float><<<dim3(E, B), dim3(((D / 4 + 63) / 64) * 64), 0, st>>>(static_cast<const float*>(x), idx, static_cast<float*>(out), E, N, D);
    } else if (dtype == GF_BF16) {
        if (D % 8) return GF_ERR_ALIGN;
        rows_gather_kernel<bf16_t><<<dim3(E, B), dim3(((D / 8 + 63) / 64) * 64), 0, st>>>(static_cast<const bf16_t*>(x), idx, static_cast<bf16_t*>(out), E, N, D);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_line_pair_scores(const float* S, const float* draw, float* out, int B, int M, int N, int backward, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    if (M > 65535 || B > 65535) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((N + 255) / 256, M, B);
    if (backward) pair_scores_bwd_kernel<<<grid, dim3(256), 0, st>>>(S, draw, out, M, N);
    else pair_scores_fwd_kernel<<<grid, dim3(256), 0, st>>>(S, out, M, N);
    return (int)hipGetLastError();
}

extern "C" int gf_dense_rowcol(const float* z, int64_t batch_stride, int64_t ld, float* rows, float* cols, int B, int M, int N,
                               int mode, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    if (B > 65535) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (rows) dense_rows_kernel<<<dim3((M + 3) / 4, B), dim3(256), 0, st>>>(z, batch_stride, ld, rows, M, N, mode);
    if (cols) dense_cols_kernel<<<dim3((N + 63) / 64, B), dim3(256), 0, st>>>(z, batch_stride, ld, cols, M, N, mode);
    return (int)hipGetLastError();
}

extern "C" int gf_dense_assign(const float* raw, const float* row_bias, const float* col_bias, const float* bin_row,
                               const float* bin_col, float corner, float* out, int B, int M, int N, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    if (M + 1 > 65535 || B > 65535) return GF_ERR_UNSUPPORTED;
    dense_assign_kernel<<<dim3((N + 256) / 256, M + 1, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        raw, row_bias, col_bias, bin_row, bin_col, corner, out, M, N);
    return (int)hipGetLastError();
}

extern "C" int gf_dense_assign_bwd(const float* raw, const float* r, const float* c, const float* A, const float* Bv,
                                   const float* G, float* draw, int B, int M, int N, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    if (M > 65535 || B > 65535) return GF_ERR_UNSUPPORTED;
    dense_assign_bwd_kernel<<<dim3((N + 255) / 256, M, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        raw, r, c, A, Bv, G, draw, M, N);
    return (int)hipGetLastError();
}
