// Assignment-head statistics from a CACHED similarity matrix (round 6).
//
// A LightGlue layer's deep-supervision head needs three reductions of S = md0 md1^T (lightglue.py:256-268, :81-94): the
// column log-sum-exp c, then the row log-sum-exp r + row arg-max (which needs c), then the column arg-max (which needs r).
// csrc/assignment.hip recomputes S on the matrix cores in each of the three DEPENDENT passes (88-108 us each at B = 32,
// N = 2048: MFMA-rate bound).  Here the first pass (rows_lse_kernel<..., STORE>) writes S once in fp16 -- 268 MB per layer,
// 54 us at 5 TB/s -- and the two later passes STREAM it (HBM bound) instead of recomputing it:
//   cached_rows_kernel : r_i = LSE_j S_ij and (max, arg max)_j of alpha S_ij + logsigmoid(z_j) - n_j      (one wave per row)
//   cached_cols_kernel : (max, arg max)_i of alpha S_ij + logsigmoid(z_i) - n_i, per 64-row block, contiguous reads;
//   cached_cols_merge  : the blocks' winners merged in ascending row order.
// Ties go to the LOWEST index, as in assignment.hip (and torch.max).  bf16 mode only: fp16 keeps 11 significant bits of the
// fp32-accumulated score (|error| <= 2^-11 |S|), below the bf16 rounding of the operands that produced it; the fp32 parity
// mode keeps the recomputing kernels.  The cache is [B, M, N] row-major, M = rows i (image 0), N = columns j (image 1).
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float logsigmoid_f(float z) { return fminf(z, 0.f) - log1pf(__expf(-fabsf(z))); }

// (value, index) of the better of two candidates, ties to the lower index
__device__ __forceinline__ void better(float& v, int& i, float ov, int oi) {
    const bool take = ov > v || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
}

// ---- rows: a wave walks CR_RPW consecutive rows; its lanes keep the biases of THEIR columns (16-byte chunks lane + 64 u) in
// registers for all of them, and the next row's chunks are in flight while a row is reduced.  N % 512 == 0, N <= 2048.
constexpr int CR_RPW = 16, CR_RPB = 4 * CR_RPW;
template <int NK, bool WITH_LSE>
__global__ __launch_bounds__(256) void cached_rows_kernel(const _Float16* __restrict__ S, const float* __restrict__ bz,
                                                          const float* __restrict__ bn, float alpha, float* __restrict__ lse,
                                                          float* __restrict__ rowmax, int64_t* __restrict__ rowarg,
                                                          int B, int M) {
    constexpr int N = NK * 512;
    const int nrb = (M + CR_RPB - 1) / CR_RPB;
    const int b = blockIdx.x / nrb, rb = blockIdx.x % nrb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float bias[NK][8];
#pragma unroll
    for (int u = 0; u < NK; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t j = (int64_t)b * N + 8 * (lane + 64 * u) + e;
            bias[u][e] = logsigmoid_f(bz[j]) - bn[j];
        }
    const int row0 = rb * CR_RPB + wave * CR_RPW;
    if (row0 >= M) return;
    const int nrow = min(CR_RPW, M - row0);
    const h16x8* sp = reinterpret_cast<const h16x8*>(S + ((int64_t)b * M + row0) * N);
    h16x8 cur[NK], nxt[NK];
#pragma unroll
    for (int u = 0; u < NK; ++u) cur[u] = sp[lane + 64 * u];
    for (int rr = 0; rr < nrow; ++rr) {
        const h16x8* np = sp + (int64_t)min(rr + 1, nrow - 1) * (N / 8);
#pragma unroll
        for (int u = 0; u < NK; ++u) nxt[u] = np[lane + 64 * u];
        float best = -INFINITY, mx = GF_NEG_BIG;
        int bidx = 0x7fffffff;
        float x[NK][8];
#pragma unroll
        for (int u = 0; u < NK; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[u][e] = (float)cur[u][e];
                const float val = fmaf(alpha, x[u][e], bias[u][e]);
                const bool gt = val > best;                                // ascending j inside a lane: strict > keeps the lowest
                best = gt ? val : best;
                bidx = gt ? 8 * (lane + 64 * u) + e : bidx;
                if (WITH_LSE) mx = fmaxf(mx, x[u][e]);
            }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) better(best, bidx, __shfl_xor(best, off), __shfl_xor(bidx, off));
        float tot = 0.f, mall = 0.f;
        if (WITH_LSE) {
            mall = wave_allmax(mx) * GF_LOG2E;
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) ps += fast_exp2(fmaf(x[u][e], GF_LOG2E, -mall));
            tot = wave_allsum(ps);
        }
        if (lane == 0) {
            const int64_t o = (int64_t)b * M + row0 + rr;
            rowmax[o] = best;
            rowarg[o] = bidx == 0x7fffffff ? 0 : bidx;
            if (WITH_LSE) lse[o] = (mall + fast_log2(tot)) * GF_LN2;
        }
#pragma unroll
        for (int u = 0; u < NK; ++u) cur[u] = nxt[u];
    }
}

// ---- columns: a workgroup = 64 consecutive rows x all N columns (N % 512 == 0, N <= 2048: a lane keeps N / 64 columns)
constexpr int CC_ROWS = 64;
template <int NK>                                                          // NK = N / 512 chunks of 8 columns per lane
__global__ __launch_bounds__(256) void cached_cols_kernel(const _Float16* __restrict__ S, const float* __restrict__ bz,
                                                          const float* __restrict__ bn, float alpha, float* __restrict__ pval,
                                                          int* __restrict__ pidx, int B, int M) {
    constexpr int N = NK * 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lv = reinterpret_cast<float*>(smem);                            // [3][N] winners of waves 1..3
    int* li = reinterpret_cast<int*>(lv + 3 * N);
    __shared__ float rbias[CC_ROWS];
    const int nblk = (M + CC_ROWS - 1) / CC_ROWS;
    const int b = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blk * CC_ROWS;
    if (threadIdx.x < CC_ROWS) {
        const int i = min(r0 + (int)threadIdx.x, M - 1);
        rbias[threadIdx.x] = logsigmoid_f(bz[(int64_t)b * M + i]) - bn[(int64_t)b * M + i];
    }
    __syncthreads();
    float best[NK][8];
    int bidx[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[k][e] = -INFINITY; bidx[k][e] = 0x7fffffff; }
    const int wr0 = r0 + wave * (CC_ROWS / 4);
    for (int rr = 0; rr < CC_ROWS / 4; rr += 2) {                          // two rows in flight
        h16x8 v[2][NK];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = min(wr0 + rr + t, M - 1);
            const h16x8* sp = reinterpret_cast<const h16x8*>(S + ((int64_t)b * M + i) * N);
#pragma unroll
            for (int k = 0; k < NK; ++k) v[t][k] = sp[lane + 64 * k];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = wr0 + rr + t;
            if (i >= M) continue;
            const float rb = rbias[i - r0];
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float val = fmaf(alpha, (float)v[t][k][e], rb);
                    const bool gt = val > best[k][e];                      // rows ascend: strict > keeps the lowest
                    best[k][e] = gt ? val : best[k][e];
                    bidx[k][e] = gt ? i : bidx[k][e];
                }
        }
    }
    // waves 1..3 park their winners; wave 0 merges them in ascending row order and writes the block's partial
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 8 * (lane + 64 * k) + e;
                lv[(wave - 1) * N + j] = best[k][e];
                li[(wave - 1) * N + j] = bidx[k][e];
            }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 8 * (lane + 64 * k) + e;
                float bv = best[k][e];
                int bi = bidx[k][e];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float ov = lv[w * N + j];
                    const bool gt = ov > bv;                               // later rows win only when strictly greater
                    bv = gt ? ov : bv;
                    bi = gt ? li[w * N + j] : bi;
                }
                best[k][e] = bv;
                bidx[k][e] = bi;
            }
        float* pv = pval + ((int64_t)b * nblk + blk) * N;
        int* pi = pidx + ((int64_t)b * nblk + blk) * N;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int j = 8 * (lane + 64 * k);
            *reinterpret_cast<f32x4*>(pv + j) = f32x4{best[k][0], best[k][1], best[k][2], best[k][3]};
            *reinterpret_cast<f32x4*>(pv + j + 4) = f32x4{best[k][4], best[k][5], best[k][6], best[k][7]};
#pragma unroll
            for (int e = 0; e < 8; ++e) pi[j + e] = bidx[k][e];
        }
    }
}

__global__ __launch_bounds__(256) void cached_cols_merge(const float* __restrict__ pval, const int* __restrict__ pidx, int nblk,
                                                         float* __restrict__ colmax, int64_t* __restrict__ colarg, int B, int N) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)B * N) return;
    const int b = (int)(t / N), j = (int)(t % N);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = 0; k < nblk; ++k) {
        const float ov = pval[((int64_t)b * nblk + k) * N + j];
        const bool gt = ov > bv;
        bv = gt ? ov : bv;
        bi = gt ? pidx[((int64_t)b * nblk + k) * N + j] : bi;
    }
    colmax[t] = bv;
    colarg[t] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

extern "C" int64_t gf_cached_cols_ws_bytes(int B, int M, int N) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    return (int64_t)B * ((M + CC_ROWS - 1) / CC_ROWS) * N * 8;
}

extern "C" int gf_cached_rows_lse_argmax(const void* s16, const float* bias_z, const float* bias_n, float alpha, float* lse,
                                         float* rowmax, int64_t* rowarg, int B, int M, int N, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || s16 == nullptr || bias_z == nullptr || bias_n == nullptr || rowmax == nullptr ||
        rowarg == nullptr) return GF_ERR_SHAPE;
    if (N % 512 || N > 2048) return GF_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(s16) & 15) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = B * ((M + CR_RPB - 1) / CR_RPB);
    const _Float16* S = reinterpret_cast<const _Float16*>(s16);
#define GF_CR(NK_)                                                                                                        \
    case NK_:                                                                                                             \
        if (lse) cached_rows_kernel<NK_, true><<<dim3(grid), dim3(256), 0, st>>>(S, bias_z, bias_n, alpha, lse, rowmax, rowarg, B, M); \
        else cached_rows_kernel<NK_, false><<<dim3(grid), dim3(256), 0, st>>>(S, bias_z, bias_n, alpha, lse, rowmax, rowarg, B, M);   \
        break;
    switch (N / 512) {
        GF_CR(1) GF_CR(2) GF_CR(3) GF_CR(4)
        default: return GF_ERR_UNSUPPORTED;
    }
#undef GF_CR
    return (int)hipGetLastError();
}

extern "C" int gf_cached_cols_argmax(const void* s16, const float* bias_z, const float* bias_n, float alpha, float* colmax,
                                     int64_t* colarg, void* ws, int B, int M, int N, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || s16 == nullptr || bias_z == nullptr || bias_n == nullptr || colmax == nullptr ||
        colarg == nullptr || ws == nullptr) return GF_ERR_SHAPE;
    if (N % 512 || N > 2048) return GF_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(s16) | reinterpret_cast<uintptr_t>(ws)) & 15) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nblk = (M + CC_ROWS - 1) / CC_ROWS;
    float* pval = reinterpret_cast<float*>(ws);
    int* pidx = reinterpret_cast<int*>(pval + (size_t)B * nblk * N);
    const _Float16* S = reinterpret_cast<const _Float16*>(s16);
    const size_t lds = (size_t)3 * N * 8;
#define GF_CC(NK_)                                                                                                        \
    case NK_: {                                                                                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cached_cols_kernel<NK_>),                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        if (e != hipSuccess) return (int)e;                                                                               \
        cached_cols_kernel<NK_><<<dim3(B * nblk), dim3(256), lds, st>>>(S, bias_z, bias_n, alpha, pval, pidx, B, M);      \
        break;                                                                                                            \
    }
    switch (N / 512) {
        GF_CC(1) GF_CC(2) GF_CC(3) GF_CC(4)
        default: return GF_ERR_UNSUPPORTED;
    }
#undef GF_CC
    const int64_t tot = (int64_t)B * N;
    cached_cols_merge<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(pval, pidx, nblk, colmax, colarg, B, N);
    return (int)hipGetLastError();
}
