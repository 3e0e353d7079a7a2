// Log-domain Sinkhorn optimal transport with dustbins, forward and backward (HBM/MALL-bound).
//
// Replaces gluefactory_nonfree/superglue.py:186-191 (log_sinkhorn_iterations) and the
// iteration part of :194-214 (log_optimal_transport).  Z is the [B, R=M+1, C=N+1] fp32
// coupling matrix (scores augmented with the bin score).  Per iteration
//     u_i = log_mu_i - LSE_j(Z_ij + v_j),      v_j = log_nu_j - LSE_i(Z_ij + u_i)
// the reference makes >= 6 full-matrix passes; here ONE pass: a workgroup pulls RB (<=16) whole
// rows of Z into LDS with coalesced loads, finishes the row log-sum-exp with wave-level
// reductions (new u), then sweeps the SAME LDS-resident rows column-wise to emit per-block
// column (max, sum) partials for the new v, which a tiny second kernel combines.  Only the
// iterates u^k, v^k are stored (2(N+1) floats per iteration) — no autograd tape of matrices.
// The launcher walks the batch in chunks that fit the 256 MB Infinity Cache so the T
// iterations of a chunk re-read Z from MALL instead of HBM.
//
// Backward (oracle/sinkhorn_oracle.py::backward_recurrence, verified against autograd):
//   ubar^k_i    = [k==T] rowsum(G)_i - sum_j exp(Z_ij + u^k_i + v^k_j - log_nu_j) vbar^k_j
//   vbar^{k-1}_j = - sum_i exp(Z_ij + u^k_i - log_mu_i + v^{k-1}_j) ubar^k_i
//   dZ_ij = G_ij - sum_k [ exp(Z_ij+u^k_i+v^k_j-log_nu_j) vbar^k_j + exp(Z_ij+u^k_i-log_mu_i+v^{k-1}_j) ubar^k_i ]
// i.e. T passes of the same one-read shape plus one final pass; every exponent is <= 0 up to
// rounding (Q, R are sub-stochastic), so no max-shift is needed in the reverse sweep.
#include <cstdlib>
#include "gf_common.h"
#include "gf_amd.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

struct Geo {
    int B, M, N, R, C, RB, nblk;
    float norm, lmu_last, lnu_last;   // log_mu = norm (i<M) | lmu_last ; log_nu = norm (j<N) | lnu_last
};
__device__ __forceinline__ float lmu(const Geo& g, int i) { return i < g.M ? g.norm : g.lmu_last; }
__device__ __forceinline__ float lnu(const Geo& g, int j) { return j < g.N ? g.norm : g.lnu_last; }

// Cooperative, fully coalesced pull of `n` contiguous floats (the RB rows of one block are adjacent
// in memory) into LDS with 16-byte loads/stores: the LDS image is shifted by (global offset mod 4)
// floats so that global-aligned <=> LDS-aligned; 8 independent loads per thread are in flight
// before the first LDS store (memory-level parallelism, one workgroup per CU).
#define SK_THREADS 512
__device__ __forceinline__ void pull_block(float* __restrict__ Zs, const float* __restrict__ g, int n, int shift) {
    const int tid = threadIdx.x;
    const int head = min(n, (4 - shift) & 3);            // scalars before the first aligned float4
    if (tid < head) Zs[shift + tid] = g[tid];
    const int nvec = (n - head) >> 2;
    const f32x4* gv = reinterpret_cast<const f32x4*>(g + head);
    f32x4* lv = reinterpret_cast<f32x4*>(Zs + shift + head);
    int i = tid;
    for (; i + 7 * SK_THREADS < nvec; i += 8 * SK_THREADS) {
        f32x4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = gv[i + k * SK_THREADS];
#pragma unroll
        for (int k = 0; k < 8; ++k) lv[i + k * SK_THREADS] = t[k];
    }
    for (; i < nvec; i += SK_THREADS) lv[i] = gv[i];
    const int tail0 = head + (nvec << 2);
    if (tid < n - tail0) Zs[shift + tail0 + tid] = g[tail0 + tid];
}

// ---- forward: rows -> u, column partials -----------------------------------------------------
// grid (nblk, Bc); v == nullptr means v = 0 (first iteration)
__global__ __launch_bounds__(SK_THREADS) void sk_rows_fwd(const float* __restrict__ Z, const float* __restrict__ v,
                                                          float* __restrict__ u, float* __restrict__ u_hist,
                                                          float* __restrict__ pm, float* __restrict__ ps, Geo g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Zraw = reinterpret_cast<float*>(smem);        // [4 + RB*C]
    float* vs = Zraw + 4 + (size_t)g.RB * g.C;           // [C]
    float* us = vs + g.C;                                // [RB]
    const int blk = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrows = min(g.RB, g.R - blk * g.RB);
    const size_t e0 = ((size_t)b * g.R + (size_t)blk * g.RB) * g.C;
    const int shift = (int)((reinterpret_cast<uintptr_t>(Z + e0) >> 2) & 3);   // float offset inside a 16-byte line
    pull_block(Zraw, Z + e0, nrows * g.C, shift);
    float* Zs = Zraw + shift;
    for (int j = threadIdx.x; j < g.C; j += SK_THREADS) vs[j] = v ? v[(size_t)b * g.C + j] : 0.f;
    __syncthreads();
    for (int r = wave; r < nrows; r += SK_THREADS / 64) {
        const int gi = blk * g.RB + r;
        const float* zs = Zs + (size_t)r * g.C;
        float mx = -INFINITY;
        for (int j = lane; j < g.C; j += 64) mx = fmaxf(mx, zs[j] + vs[j]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int j = lane; j < g.C; j += 64) s += __expf(zs[j] + vs[j] - mx);
        s = wave_sum(s);
        const float un = lmu(g, gi) - (mx + __logf(s));
        if (lane == 0) {
            us[r] = un;
            u[(size_t)b * g.R + gi] = un;
            u_hist[(size_t)b * g.R + gi] = un;
        }
    }
    __syncthreads();
    float* pmb = pm + ((size_t)b * g.nblk + blk) * g.C;
    float* psb = ps + ((size_t)b * g.nblk + blk) * g.C;
    for (int j = threadIdx.x; j < g.C; j += SK_THREADS) {
        float mx = -INFINITY;
        for (int r = 0; r < nrows; ++r) mx = fmaxf(mx, Zs[(size_t)r * g.C + j] + us[r]);
        float s = 0.f;
        for (int r = 0; r < nrows; ++r) s += __expf(Zs[(size_t)r * g.C + j] + us[r] - mx);
        pmb[j] = mx;
        psb[j] = s;
    }
}

// grid (ceil(C/64), Bc), 256 threads = 64 columns x 4 block-groups, combined through LDS
__global__ __launch_bounds__(256) void sk_cols_fwd(const float* __restrict__ pm, const float* __restrict__ ps,
                                                   float* __restrict__ v, float* __restrict__ v_hist, Geo g) {
    __shared__ float sm[4][64], ss[4][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cx, b = blockIdx.y;
    const int jc = min(j, g.C - 1);
    const float* pmb = pm + (size_t)b * g.nblk * g.C + jc;
    const float* psb = ps + (size_t)b * g.nblk * g.C + jc;
    float mx = -INFINITY, s = 0.f;
    for (int k = grp; k < g.nblk; k += 4) {
        const float m2 = pmb[(size_t)k * g.C], s2 = psb[(size_t)k * g.C];
        const float mn = fmaxf(mx, m2);
        s = s * __expf(mx - mn) + s2 * __expf(m2 - mn);
        mx = mn;
    }
    sm[grp][cx] = mx;
    ss[grp][cx] = s;
    __syncthreads();
    if (grp == 0 && j < g.C) {
        float M = fmaxf(fmaxf(sm[0][cx], sm[1][cx]), fmaxf(sm[2][cx], sm[3][cx]));
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) S += (sm[k][cx] == -INFINITY) ? 0.f : ss[k][cx] * __expf(sm[k][cx] - M);
        const float vn = lnu(g, j) - (M + __logf(S));
        v[(size_t)b * g.C + j] = vn;
        v_hist[(size_t)b * g.C + j] = vn;
    }
}

// out = Z + u + v - norm ; grid (ceil(C/256), R, Bc)
__global__ void sk_final_fwd(const float* __restrict__ Z, const float* __restrict__ u, const float* __restrict__ v,
                             float* __restrict__ out, Geo g) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= g.C) return;
    const size_t idx = ((size_t)b * g.R + i) * g.C + j;
    const float uu = u ? u[(size_t)b * g.R + i] : 0.f, vv = v ? v[(size_t)b * g.C + j] : 0.f;
    out[idx] = Z[idx] + uu + vv - g.norm;
}

// ---- backward: one reverse iteration -----------------------------------------------------------
// ubar_i = base_i - sum_j exp(Z_ij + u_i + (vk_j - lnu_j)) vbar_j ; column partials of
// sum_i exp(Z_ij + (u_i - lmu_i) + vprev_j) ubar_i
__global__ __launch_bounds__(SK_THREADS) void sk_rows_bwd(const float* __restrict__ Z, const float* __restrict__ uk,
                                                          const float* __restrict__ vk, const float* __restrict__ vprev,
                                                          const float* __restrict__ vbar, const float* __restrict__ base,
                                                          float* __restrict__ ubar_out, float* __restrict__ psum, Geo g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Zraw = reinterpret_cast<float*>(smem);
    float* as = Zraw + 4 + (size_t)g.RB * g.C;     // vk - lnu
    float* bs = as + g.C;                          // vbar
    float* ps_ = bs + g.C;                         // vprev
    float* us = ps_ + g.C;                         // [RB] u - lmu
    float* ubs = us + g.RB;                        // [RB] ubar
    const int blk = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrows = min(g.RB, g.R - blk * g.RB);
    const size_t e0 = ((size_t)b * g.R + (size_t)blk * g.RB) * g.C;
    const int shift = (int)((reinterpret_cast<uintptr_t>(Z + e0) >> 2) & 3);   // float offset inside a 16-byte line
    pull_block(Zraw, Z + e0, nrows * g.C, shift);
    float* Zs = Zraw + shift;
    for (int j = threadIdx.x; j < g.C; j += SK_THREADS) {
        as[j] = vk[(size_t)b * g.C + j] - lnu(g, j);
        bs[j] = vbar[(size_t)b * g.C + j];
        ps_[j] = vprev ? vprev[(size_t)b * g.C + j] : 0.f;
    }
    __syncthreads();
    for (int r = wave; r < nrows; r += SK_THREADS / 64) {
        const int gi = blk * g.RB + r;
        const float* zs = Zs + (size_t)r * g.C;
        const float ui = uk[(size_t)b * g.R + gi];
        float acc = 0.f;
        for (int j = lane; j < g.C; j += 64) acc += __expf(zs[j] + ui + as[j]) * bs[j];
        acc = wave_sum(acc);
        const float ub = (base ? base[(size_t)b * g.R + gi] : 0.f) - acc;
        if (lane == 0) {
            us[r] = ui - lmu(g, gi);
            ubs[r] = ub;
            ubar_out[(size_t)b * g.R + gi] = ub;
        }
    }
    __syncthreads();
    float* pb = psum + ((size_t)b * g.nblk + blk) * g.C;
    for (int j = threadIdx.x; j < g.C; j += SK_THREADS) {
        float acc = 0.f;
        const float vp = ps_[j];
        for (int r = 0; r < nrows; ++r) acc += __expf(Zs[(size_t)r * g.C + j] + us[r] + vp) * ubs[r];
        pb[j] = acc;
    }
}

__global__ __launch_bounds__(256) void sk_cols_bwd(const float* __restrict__ psum, float* __restrict__ vbar_out, Geo g) {
    __shared__ float ss[4][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cx, b = blockIdx.y;
    const int jc = min(j, g.C - 1);
    const float* pb = psum + (size_t)b * g.nblk * g.C + jc;
    float s = 0.f;
    for (int k = grp; k < g.nblk; k += 4) s += pb[(size_t)k * g.C];
    ss[grp][cx] = s;
    __syncthreads();
    if (grp == 0 && j < g.C) vbar_out[(size_t)b * g.C + j] = -(ss[0][cx] + ss[1][cx] + ss[2][cx] + ss[3][cx]);
}

// dZ = G - sum_k [...] ; thread = 1 column x 8 rows ; grid (ceil(C/256), ceil(R/8), Bc)
// u_hist/ubar_hist [T, B, R] (batch stride passed), v_hist/vbar_hist [T, B, C]
__global__ __launch_bounds__(256) void sk_final_bwd(const float* __restrict__ Z, const float* __restrict__ G,
                                                    const float* __restrict__ u_hist, const float* __restrict__ v_hist,
                                                    const float* __restrict__ ubar_hist, const float* __restrict__ vbar_hist,
                                                    float* __restrict__ gZ, int T, size_t ustride, size_t vstride,
                                                    size_t ubstride, size_t vbstride, Geo g) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i0 = blockIdx.y * 8, b = blockIdx.z;
    const int jc = min(j, g.C - 1);
    float z[8], acc[8], lm[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = min(i0 + r, g.R - 1);
        z[r] = Z[((size_t)b * g.R + i) * g.C + jc];
        lm[r] = lmu(g, i);
        acc[r] = 0.f;
    }
    const float ln = lnu(g, jc);
    for (int k = 1; k <= T; ++k) {
        const float a = v_hist[(size_t)(k - 1) * vstride + (size_t)b * g.C + jc] - ln;
        const float vb = vbar_hist[(size_t)(k - 1) * vbstride + (size_t)b * g.C + jc];
        const float vp = k >= 2 ? v_hist[(size_t)(k - 2) * vstride + (size_t)b * g.C + jc] : 0.f;
        const float* uk = u_hist + (size_t)(k - 1) * ustride + (size_t)b * g.R;
        const float* ubk = ubar_hist + (size_t)(k - 1) * ubstride + (size_t)b * g.R;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = min(i0 + r, g.R - 1);
            const float ui = uk[i], ub = ubk[i];
            acc[r] += __expf(z[r] + ui + a) * vb + __expf(z[r] + ui - lm[r] + vp) * ub;
        }
    }
    if (j < g.C) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (i0 + r < g.R) {
                const size_t idx = ((size_t)b * g.R + i0 + r) * g.C + j;
                gZ[idx] = G[idx] - acc[r];
            }
    }
}

const size_t LDS_BUDGET = 160 * 1024 - 512;

Geo make_geo(int B, int M, int N) {
    Geo g;
    g.B = B; g.M = M; g.N = N; g.R = M + 1; g.C = N + 1;
    // rows per block: bounded by LDS (RB rows + 4 column vectors), at most 16
    size_t rb = (LDS_BUDGET - 4 * (size_t)g.C * 4 - 512) / ((size_t)g.C * 4);
    static const int cap = getenv("GF_SK_RB") ? atoi(getenv("GF_SK_RB")) : 16;   // tuning knob
    g.RB = (int)(rb > (size_t)cap ? (size_t)cap : rb);
    g.nblk = g.RB > 0 ? (g.R + g.RB - 1) / g.RB : 0;
    g.norm = -logf((float)(M + N));
    g.lmu_last = logf((float)N) + g.norm;
    g.lnu_last = logf((float)M) + g.norm;
    return g;
}

int batch_chunk(const Geo& g) {
    // keep one chunk's Z (+ partials) inside the 256 MB Infinity Cache
    size_t per = (size_t)g.R * g.C * 4 + 2 * (size_t)g.nblk * g.C * 4;
    int ch = (int)((size_t)176 * 1024 * 1024 / per);
    return ch < 1 ? 1 : (ch > g.B ? g.B : ch);
}

size_t rows_lds(const Geo& g, bool bwd) {
    return ((size_t)g.RB * g.C + 4 + (bwd ? 3 : 1) * (size_t)g.C + 2 * (size_t)g.RB) * 4 + 64;
}

}  // namespace

extern "C" int64_t gf_sinkhorn_ws_bytes(int B, int M, int N, int iters) {
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0) return GF_ERR_SHAPE;
    Geo g = make_geo(B, M, N);
    if (g.RB < 1) return GF_ERR_UNSUPPORTED;
    size_t part = 2 * (size_t)B * g.nblk * g.C * 4;
    size_t cur = (size_t)B * (g.R + g.C) * 4;
    size_t hist = (size_t)(iters + 1) * B * ((size_t)g.R + g.C) * 4;   // ubar / vbar history (backward)
    return (int64_t)(part + cur + hist + 1024);
}

extern "C" int gf_sinkhorn_fwd(const float* Z, float* out, float* u_hist, float* v_hist, void* ws,
                               int B, int M, int N, int iters, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0) return GF_ERR_SHAPE;
    Geo g = make_geo(B, M, N);
    if (g.RB < 1) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* pm = reinterpret_cast<float*>(ws);
    float* ps = pm + (size_t)B * g.nblk * g.C;
    float* ucur = ps + (size_t)B * g.nblk * g.C;
    float* vcur = ucur + (size_t)B * g.R;
    const size_t lds = rows_lds(g, false);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sk_rows_fwd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int ch = batch_chunk(g);
    const size_t zs = (size_t)g.R * g.C;
    for (int b0 = 0; b0 < B; b0 += ch) {
        const int bc = (B - b0) < ch ? (B - b0) : ch;
        for (int it = 0; it < iters; ++it) {
            sk_rows_fwd<<<dim3(g.nblk, bc), SK_THREADS, lds, st>>>(
                Z + b0 * zs, it == 0 ? nullptr : vcur + (size_t)b0 * g.C, ucur + (size_t)b0 * g.R,
                u_hist + ((size_t)it * B + b0) * g.R, pm + (size_t)b0 * g.nblk * g.C,
                ps + (size_t)b0 * g.nblk * g.C, g);
            sk_cols_fwd<<<dim3((g.C + 63) / 64, bc), 256, 0, st>>>(
                pm + (size_t)b0 * g.nblk * g.C, ps + (size_t)b0 * g.nblk * g.C, vcur + (size_t)b0 * g.C,
                v_hist + ((size_t)it * B + b0) * g.C, g);
        }
        sk_final_fwd<<<dim3((g.C + 255) / 256, g.R, bc), 256, 0, st>>>(
            Z + b0 * zs, iters ? ucur + (size_t)b0 * g.R : nullptr, iters ? vcur + (size_t)b0 * g.C : nullptr,
            out + b0 * zs, g);
    }
    return (int)hipGetLastError();
}

extern "C" int gf_sinkhorn_bwd(const float* Z, const float* gout, const float* gsum_row, const float* gsum_col,
                               const float* u_hist, const float* v_hist, float* gZ, void* ws,
                               int B, int M, int N, int iters, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0) return GF_ERR_SHAPE;
    Geo g = make_geo(B, M, N);
    if (g.RB < 1) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* psum = reinterpret_cast<float*>(ws);
    float* skip = psum + 2 * (size_t)B * g.nblk * g.C + (size_t)B * (g.R + g.C);
    float* ubar_hist = skip;                                   // [iters, B, R]   (index k-1)
    float* vbar_hist = ubar_hist + (size_t)(iters + 1) * B * g.R;  // [iters+1, B, C] (index k, k = 0..T)
    const size_t zs = (size_t)g.R * g.C;
    if (iters == 0) {
        hipError_t e = hipMemcpyAsync(gZ, gout, (size_t)B * zs * 4, hipMemcpyDeviceToDevice, st);
        return (int)e;
    }
    const size_t lds = rows_lds(g, true);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sk_rows_bwd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    // vbar^T = colsum(G)
    e = hipMemcpyAsync(vbar_hist + (size_t)iters * B * g.C, gsum_col, (size_t)B * g.C * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    const int ch = batch_chunk(g);
    for (int b0 = 0; b0 < B; b0 += ch) {
        const int bc = (B - b0) < ch ? (B - b0) : ch;
        for (int k = iters; k >= 1; --k) {
            const float* uk = u_hist + ((size_t)(k - 1) * B + b0) * g.R;
            const float* vk = v_hist + ((size_t)(k - 1) * B + b0) * g.C;
            const float* vp = k >= 2 ? v_hist + ((size_t)(k - 2) * B + b0) * g.C : nullptr;
            sk_rows_bwd<<<dim3(g.nblk, bc), SK_THREADS, lds, st>>>(
                Z + b0 * zs, uk, vk, vp, vbar_hist + ((size_t)k * B + b0) * g.C,
                k == iters ? gsum_row + (size_t)b0 * g.R : nullptr,
                ubar_hist + ((size_t)(k - 1) * B + b0) * g.R, psum + (size_t)b0 * g.nblk * g.C, g);
            sk_cols_bwd<<<dim3((g.C + 63) / 64, bc), 256, 0, st>>>(
                psum + (size_t)b0 * g.nblk * g.C, vbar_hist + ((size_t)(k - 1) * B + b0) * g.C, g);
        }
        sk_final_bwd<<<dim3((g.C + 255) / 256, (g.R + 7) / 8, bc), 256, 0, st>>>(
            Z + b0 * zs, gout + b0 * zs, u_hist + (size_t)b0 * g.R, v_hist + (size_t)b0 * g.C,
            ubar_hist + (size_t)b0 * g.R, vbar_hist + (size_t)B * g.C + (size_t)b0 * g.C, gZ + b0 * zs, iters,
            (size_t)B * g.R, (size_t)B * g.C, (size_t)B * g.R, (size_t)B * g.C, g);
    }
    return (int)hipGetLastError();
}
