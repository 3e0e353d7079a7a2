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
// This generic LDS path serves N + 1 > 2304; smaller problems take the register-resident fast path below.
//
// Backward (oracle/sinkhorn_oracle.py::backward_recurrence, verified against autograd):
//   ubar^k_i    = [k==T] rowsum(G)_i - sum_j exp(Z_ij + u^k_i + v^k_j - log_nu_j) vbar^k_j
//   vbar^{k-1}_j = - sum_i exp(Z_ij + u^k_i - log_mu_i + v^{k-1}_j) ubar^k_i
//   dZ_ij = G_ij - sum_k [ exp(Z_ij+u^k_i+v^k_j-log_nu_j) vbar^k_j + exp(Z_ij+u^k_i-log_mu_i+v^{k-1}_j) ubar^k_i ]
// i.e. T passes of the same one-read shape plus one final pass; every exponent is <= 0 up to
// rounding (Q, R are sub-stochastic), so no max-shift is needed in the reverse sweep.
#include <atomic>

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
    int Cp;                           // fast path: row stride of the padded copy (C rounded up to 4)
    bool fast;                        // register-resident kernels (C <= 64*4*SKF_MAX_NS)
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


// =================================================================================================
// Fast path (C <= 2304): rows live in REGISTERS, one read of Z per iteration, no LDS staging
// =================================================================================================
// The generic kernels above spend their time moving every element global -> VGPR -> LDS (ds_write is the
// slowest LDS instruction) -> VGPR four times.  Here a wave owns SKF_RPW whole rows, one after the other: a
// row is NS float4 per lane (columns 4*(lane + 64 k) .. +3), loaded once with 16-byte coalesced loads from a
// padded, log2e-prescaled copy Zp [Bc, R, Cp] (Cp = C rounded up to 4, pad = -inf; the copy is made once per
// call, 1/T of the iteration traffic).  The next rows' loads are in flight while the current row is processed.
//
// One exponential per element and iteration: with ref_i = the previous u_i (log2 units, + SKF_SHIFT)
//     e_ij   = exp2(Zp_ij + v_j + ref_i)                 (<= nu_j 2^SHIFT: column-normalised by the last v)
//     rs_i   = sum_j e_ij          ->  u_i' = lmu_i - log2(rs_i) + ref_i           (the exact row update)
//     S_j   += e_ij * f_i,  f_i = 2^SHIFT mu_i / rs_i   (= exp2(Zp_ij + v_j + u_i' + SHIFT), <= mu_i 2^SHIFT)
//     v_j'   = v_j + lnu_j - log2(S_j) + SHIFT                                      (the exact column update)
// i.e. the row pass and the column pass share the exponential; no running maxima are needed because after a
// column (row) update every term is bounded by the column (row) marginal.  Only the very first row update
// (u = v = 0, nothing normalised yet) uses ref_i = -max_j Z_ij.  SKF_SHIFT = 64 moves the representable
// floor to a marginal of 2^-190; below that the sum is clamped (never NaN).
// Column sums are kept per lane in registers over the wave's rows, combined over the 4 waves of a workgroup
// through LDS once, and written as ONE partial row per 32 matrix rows (3 % of the Z traffic); a small second
// kernel finishes v.  The backward sweep has the same shape (see skf_bwd_iter).
#ifndef SK_CHUNK_MB
#define SK_CHUNK_MB 300     // bytes of one batch chunk (MB): measured, 16-pair chunks stream fastest
#endif
#ifndef SKF_RPW_V
#define SKF_RPW_V 8
#endif
#ifndef SKF_PF_V
#define SKF_PF_V 2
#endif
constexpr int SKF_RPW = SKF_RPW_V;             // rows per wave
constexpr int SKF_PF = SKF_PF_V;               // rows in flight ahead of the one being processed
constexpr int SKF_NB = SKF_PF + 1;             // register row buffers (ring, statically indexed)
constexpr int SKF_RPB = 4 * SKF_RPW;           // rows per workgroup (4 waves)
constexpr float SKF_SHIFT = 64.f;
constexpr int SKF_MAX_NS = 9;                  // C <= 2304

__device__ __forceinline__ f32x4 splat4(float x) { f32x4 v = {x, x, x, x}; return v; }

// Zp[b][i][4q..4q+3] = Z[b][i][..] * log2e, -inf past C.  One thread per float4 of Zp.
__global__ __launch_bounds__(256) void skf_prescale(const float* __restrict__ Z, float* __restrict__ Zp, Geo g, int rows_total) {
    const int nvec = g.Cp >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)rows_total * nvec) return;
    const size_t row = idx / nvec;
    const int q = (int)(idx - row * nvec);
    const float* src = Z + row * g.C + 4 * q;
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (4 * q + c < g.C) ? src[c] * GF_LOG2E : -INFINITY;
    *reinterpret_cast<f32x4*>(Zp + row * g.Cp + 4 * q) = o;
}

template <int NS>
__device__ __forceinline__ void skf_load_row(f32x4 (&z)[NS], const float* __restrict__ zrow, int lane, int nvec) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int q = lane + 64 * k;
        z[k] = q < nvec ? *reinterpret_cast<const f32x4*>(zrow + 4 * q) : splat4(-INFINITY);
    }
}

// workgroup-level sum of the per-wave column accumulators -> one partial row
template <int NS>
__device__ __forceinline__ void skf_store_partial(const f32x4 (&S)[NS], f32x4* red, float* __restrict__ prow, int lane,
                                                  int wave, int nvec) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int q = lane + 64 * k;
        if (q < nvec) red[wave * (NS * 64) + q] = S[k];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nvec; q += 256) {
        const f32x4 a = red[q], b = red[NS * 64 + q], c = red[2 * NS * 64 + q], d = red[3 * NS * 64 + q];
        *reinterpret_cast<f32x4*>(prow + 4 * q) = (a + b) + (c + d);
    }
}

// grid (nblk, Bc).  v2 [Bc, Cp] (log2 units), u2 [Bc, R] read (previous) and written (new) in place.
template <int NS, bool FIRST>
__global__ __launch_bounds__(256, 2) void skf_fwd_iter(const float* __restrict__ Zp, const float* __restrict__ v2,
                                                       float* __restrict__ u2, float* __restrict__ u_hist,
                                                       float* __restrict__ part, Geo g) {
    __shared__ f32x4 red[4 * NS * 64];
    const int blk = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nvec = g.Cp >> 2;
    const int row0 = blk * SKF_RPB + wave * SKF_RPW;
    const int nrows = min(SKF_RPW, g.R - row0);
    f32x4 vv[NS], S[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int q = lane + 64 * k;
        vv[k] = (!FIRST && q < nvec) ? *reinterpret_cast<const f32x4*>(v2 + (size_t)b * g.Cp + 4 * q) : splat4(0.f);
        S[k] = splat4(0.f);
    }
    if (nrows > 0) {
        const float* zrow = Zp + ((size_t)b * g.R + row0) * g.Cp;
        f32x4 zb[SKF_NB][NS];
#pragma unroll
        for (int p = 0; p < SKF_PF; ++p)
            if (p < nrows) skf_load_row<NS>(zb[p], zrow + (size_t)p * g.Cp, lane, nvec);
        for (int r0 = 0; r0 < nrows; r0 += SKF_NB) {
#pragma unroll
            for (int s_ = 0; s_ < SKF_NB; ++s_) {
                const int r = r0 + s_;
                if (r + SKF_PF < nrows)
                    skf_load_row<NS>(zb[(s_ + SKF_PF) % SKF_NB], zrow + (size_t)(r + SKF_PF) * g.Cp, lane, nvec);
                if (r >= nrows) continue;
                f32x4 (&e)[NS] = zb[s_];
                const int gi = row0 + r;
                float ref;
                if (FIRST) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int k = 0; k < NS; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c) mx = fmaxf(mx, e[k][c]);
                    ref = -wave_allmax(mx);
                } else {
                    ref = u2[(size_t)b * g.R + gi] + SKF_SHIFT;
                }
                float rs = 0.f;
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = fast_exp2(e[k][c] + vv[k][c] + ref);
                        e[k][c] = x;
                        rs += x;
                    }
                rs = fmaxf(wave_allsum(rs), 1.17549435e-38f);
                const float lmu2 = lmu(g, gi) * GF_LOG2E, l2 = fast_log2(rs);
                const float un = lmu2 - l2 + ref;
                const float f = fast_exp2(lmu2 - l2 + SKF_SHIFT);
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) S[k][c] = fmaf(e[k][c], f, S[k][c]);
                if (lane == 0) {
                    u2[(size_t)b * g.R + gi] = un;
                    u_hist[(size_t)b * g.R + gi] = un * GF_LN2;
                }
            }
        }
    }
    skf_store_partial<NS>(S, red, part + ((size_t)b * g.nblk + blk) * g.Cp, lane, wave, nvec);
}

// sum of the nblk partial rows of 64 columns: 4 groups of threads take every 4th partial row, LDS combines them
__device__ __forceinline__ float skf_colsum(const float* __restrict__ part, int b, int j, int cx, int grp, float (*ss)[64],
                                            const Geo& g) {
    const float* p = part + (size_t)b * g.nblk * g.Cp + j;
    float s0 = 0.f, s1 = 0.f;
    int k = grp;
    for (; k + 4 < g.nblk; k += 8) { s0 += p[(size_t)k * g.Cp]; s1 += p[(size_t)(k + 4) * g.Cp]; }
    if (k < g.nblk) s0 += p[(size_t)k * g.Cp];
    ss[grp][cx] = s0 + s1;
    __syncthreads();
    return (ss[0][cx] + ss[1][cx]) + (ss[2][cx] + ss[3][cx]);
}

// grid (Cp/64 rounded up, Bc), 256 threads: v2' = v2 + lnu - log2(sum of partials) + SHIFT
__global__ __launch_bounds__(256) void skf_cols_fwd(const float* __restrict__ part, float* __restrict__ v2,
                                                    float* __restrict__ v_hist, int first, Geo g) {
    __shared__ float ss[4][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6, b = blockIdx.y;
    const int j = blockIdx.x * 64 + cx, jc = min(j, g.Cp - 1);
    const float tot = skf_colsum(part, b, jc, cx, grp, ss, g);
    if (grp != 0 || j >= g.Cp) return;
    float vn = 0.f;                                   // pad columns: any finite value (Zp is -inf there)
    if (j < g.C) {
        const float S = fmaxf(tot, 1.17549435e-38f);
        vn = (first ? 0.f : v2[(size_t)b * g.Cp + j]) + lnu(g, j) * GF_LOG2E - fast_log2(S) + SKF_SHIFT;
        v_hist[(size_t)b * g.C + j] = vn * GF_LN2;
    }
    v2[(size_t)b * g.Cp + j] = vn;
}

// ---- backward iteration k: e_ij = exp(Z_ij + u^k_i + v^k_j - lnu_j) (<= 1, columns sum to 1) serves both sums:
//   ubar^k_i     = base_i - sum_j e_ij vbar^k_j
//   vbar^{k-1}_j = -c_j sum_i e_ij w_i,   w_i = ubar^k_i exp(-lmu_i),  c_j = exp(v^{k-1}_j - v^k_j + lnu_j)
// (exp(Z_ij + u^k_i - lmu_i + v^{k-1}_j) = e_ij exp(-lmu_i) c_j); c_j is applied by skf_cols_bwd.
template <int NS>
__global__ __launch_bounds__(256, 2) void skf_bwd_iter(const float* __restrict__ Zp, const float* __restrict__ uk,
                                                       const float* __restrict__ a2p, const float* __restrict__ vbp,
                                                       const float* __restrict__ base, float* __restrict__ ubar_out,
                                                       float* __restrict__ part, Geo g) {
    // a2p [Bc, Cp] = (v^k - lnu) log2e, vbp [Bc, Cp] = vbar^k, both zero in the pad columns (written by
    // skf_cols_bwd / skf_bwd_prep)
    __shared__ f32x4 red[4 * NS * 64];
    const int blk = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nvec = g.Cp >> 2;
    const int row0 = blk * SKF_RPB + wave * SKF_RPW;
    const int nrows = min(SKF_RPW, g.R - row0);
    f32x4 a2[NS], vb[NS], S[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int q = lane + 64 * k;
        a2[k] = q < nvec ? *reinterpret_cast<const f32x4*>(a2p + (size_t)b * g.Cp + 4 * q) : splat4(0.f);
        vb[k] = q < nvec ? *reinterpret_cast<const f32x4*>(vbp + (size_t)b * g.Cp + 4 * q) : splat4(0.f);
        S[k] = splat4(0.f);
    }
    if (nrows > 0) {
        const float* zrow = Zp + ((size_t)b * g.R + row0) * g.Cp;
        f32x4 zb[SKF_NB][NS];
#pragma unroll
        for (int p = 0; p < SKF_PF; ++p)
            if (p < nrows) skf_load_row<NS>(zb[p], zrow + (size_t)p * g.Cp, lane, nvec);
        for (int r0 = 0; r0 < nrows; r0 += SKF_NB) {
#pragma unroll
            for (int s_ = 0; s_ < SKF_NB; ++s_) {
                const int r = r0 + s_;
                if (r + SKF_PF < nrows)
                    skf_load_row<NS>(zb[(s_ + SKF_PF) % SKF_NB], zrow + (size_t)(r + SKF_PF) * g.Cp, lane, nvec);
                if (r >= nrows) continue;
                f32x4 (&e)[NS] = zb[s_];
                const int gi = row0 + r;
                const float u2 = uk[(size_t)b * g.R + gi] * GF_LOG2E;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = fast_exp2(e[k][c] + a2[k][c] + u2);
                        e[k][c] = x;
                        acc = fmaf(x, vb[k][c], acc);
                    }
                acc = wave_allsum(acc);
                const float ub = (base ? base[(size_t)b * g.R + gi] : 0.f) - acc;
                const float w = ub * fast_exp2(-lmu(g, gi) * GF_LOG2E);
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) S[k][c] = fmaf(e[k][c], w, S[k][c]);
                if (lane == 0) ubar_out[(size_t)b * g.R + gi] = ub;
            }
        }
    }
    skf_store_partial<NS>(S, red, part + ((size_t)b * g.nblk + blk) * g.Cp, lane, wave, nvec);
}

// grid (Cp/64 rounded up, Bc): vbar^{k-1}_j = -exp(v^{k-1}_j - v^k_j + lnu_j) * sum of partials   (v^0 = 0);
// also the padded inputs of the NEXT reverse iteration (k-1): a2p = (v^{k-1} - lnu) log2e, vbp = vbar^{k-1}
__global__ __launch_bounds__(256) void skf_cols_bwd(const float* __restrict__ part, const float* __restrict__ vk,
                                                    const float* __restrict__ vprev, float* __restrict__ vbar_out,
                                                    float* __restrict__ a2p, float* __restrict__ vbp, Geo g) {
    __shared__ float ss[4][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6, b = blockIdx.y;
    const int j = blockIdx.x * 64 + cx, jc = min(j, g.Cp - 1);
    const float tot = skf_colsum(part, b, jc, cx, grp, ss, g);
    if (grp != 0 || j >= g.Cp) return;
    float vbn = 0.f, a2n = 0.f;
    if (j < g.C) {
        const float vp = vprev ? vprev[(size_t)b * g.C + j] : 0.f;
        vbn = -__expf(vp - vk[(size_t)b * g.C + j] + lnu(g, j)) * tot;
        a2n = (vp - lnu(g, j)) * GF_LOG2E;
        vbar_out[(size_t)b * g.C + j] = vbn;
    }
    a2p[(size_t)b * g.Cp + j] = a2n;
    vbp[(size_t)b * g.Cp + j] = vbn;
}

// first reverse iteration (k = T): a2p from v^T, vbp = colsum(G)
__global__ __launch_bounds__(256) void skf_bwd_prep(const float* __restrict__ vT, const float* __restrict__ gsum_col,
                                                    float* __restrict__ a2p, float* __restrict__ vbp, Geo g) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= g.Cp) return;
    const bool ok = j < g.C;
    a2p[(size_t)b * g.Cp + j] = ok ? (vT[(size_t)b * g.C + j] - lnu(g, j)) * GF_LOG2E : 0.f;
    vbp[(size_t)b * g.Cp + j] = ok ? gsum_col[(size_t)b * g.C + j] : 0.f;
}

// ---- final gradient: dZ = G - sum_k [ e1^k vbar^k_j + e2^k ubar^k_i ] as ONE rank-2T product on the matrix cores.
// With the last iterates as reference, E_ij = exp(Z_ij + u^T_i + v^T_j - lnu_j) (<= 1: its columns sum to 1),
//   e1^k_ij vbar^k_j = E_ij * exp(u^k_i - u^T_i)                   * [exp(v^k_j - v^T_j) vbar^k_j]
//   e2^k_ij ubar^k_i = E_ij * [exp(u^k_i - u^T_i - lmu_i) ubar^k_i] * exp(v^{k-1}_j - v^T_j + lnu_j)
// so dZ = G - E o (P Q^T) with P [R, 2T], Q [C, 2T].  The product runs on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains); the differences of iterates are small (Sinkhorn contracts), the
// exponents are clamped to +-80 so that nothing can overflow.  2T is padded to a multiple of 16 with zeros.
// P/Q layout: [pairs, R or C, KP] row-major, k contiguous.
__device__ __forceinline__ float exp_clamped(float x) { return __expf(fminf(fmaxf(x, -80.f), 80.f)); }

// grid (ceil(max(R,C)/256), T, Bc): thread = one row (or column) of one iteration's two factor columns
__global__ __launch_bounds__(256) void skf_factors(const float* __restrict__ u_hist, const float* __restrict__ v_hist,
                                                   const float* __restrict__ ubar_hist, const float* __restrict__ vbar_hist,
                                                   float* __restrict__ P, float* __restrict__ Q, int T, int KP,
                                                   size_t ustride, size_t vstride, Geo g) {
    const int x = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y + 1, b = blockIdx.z;
    if (x < g.R) {
        const float uk = u_hist[(size_t)(k - 1) * ustride + (size_t)b * g.R + x];
        const float uT = u_hist[(size_t)(T - 1) * ustride + (size_t)b * g.R + x];
        const float ub = ubar_hist[(size_t)(k - 1) * ustride + (size_t)b * g.R + x];
        float* p = P + ((size_t)b * g.R + x) * KP + 2 * (k - 1);
        p[0] = exp_clamped(uk - uT);
        p[1] = exp_clamped(uk - uT - lmu(g, x)) * ub;
    }
    if (x < g.C) {
        const float vk = v_hist[(size_t)(k - 1) * vstride + (size_t)b * g.C + x];
        const float vT = v_hist[(size_t)(T - 1) * vstride + (size_t)b * g.C + x];
        const float vp = k >= 2 ? v_hist[(size_t)(k - 2) * vstride + (size_t)b * g.C + x] : 0.f;
        const float vb = vbar_hist[(size_t)k * vstride + (size_t)b * g.C + x];
        float* q = Q + ((size_t)b * g.C + x) * KP + 2 * (k - 1);
        q[0] = exp_clamped(vk - vT) * vb;
        q[1] = exp_clamped(vp - vT + lnu(g, x));
    }
    if (k == T) {                                   // zero the k padding once
        for (int c = 2 * T; c < KP; ++c) {
            if (x < g.R) P[((size_t)b * g.R + x) * KP + c] = 0.f;
            if (x < g.C) Q[((size_t)b * g.C + x) * KP + c] = 0.f;
        }
    }
}

// grid (ceil(C/128) * ceil(R/128), Bc), 256 threads: one wave = a 64 x 64 block (2 x 2 MFMA tiles) of the workgroup's 128 x 128,
// the factor fragments of the next k-step in flight under the current one's MFMAs.  (Round 6: was one 32 x 32 tile per wave with
// the loads in front of their MFMAs -- ablations of that form: 0.97 of its 1.46 ms per step were the product, against 0.34 ms at
// the exact-fp32 MFMA rate; prefetching took 0.15 ms off the call, sharing every fragment between two tiles another 0.12:
// 7.96 -> 7.71 ms backward at B = 32, T = 100, bit-identical -- the summation order over k is unchanged.)
__global__ __launch_bounds__(256) void skf_final_bwd(const float* __restrict__ Z, const float* __restrict__ G,
                                                     const float* __restrict__ P, const float* __restrict__ Q,
                                                     const float* __restrict__ uT, const float* __restrict__ vT,
                                                     float* __restrict__ gZ, int KP, Geo g) {
    const int ncb = (g.C + 127) / 128;
    const int b = blockIdx.y, rb = blockIdx.x / ncb, cb = blockIdx.x % ncb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i0 = rb * 128 + (wave >> 1) * 64, j0 = cb * 128 + (wave & 1) * 64;
    if (i0 >= g.R || j0 >= g.C) return;
    const float* prow[2];
    const float* qrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        prow[t] = P + ((size_t)b * g.R + min(i0 + 32 * t + l31, g.R - 1)) * KP + 8 * hi;
        qrow[t] = Q + ((size_t)b * g.C + min(j0 + 32 * t + l31, g.C - 1)) * KP + 8 * hi;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
    Frag<float> p0 = ld_frag8(prow[0]), p1 = ld_frag8(prow[1]), q0 = ld_frag8(qrow[0]), q1 = ld_frag8(qrow[1]);
    for (int s_ = 0; s_ < KP; s_ += 16) {
        const int sn = min(s_ + 16, KP - 16);                       // (the last step re-fetches itself: no branch)
        const Frag<float> p0n = ld_frag8(prow[0] + sn), p1n = ld_frag8(prow[1] + sn);
        const Frag<float> q0n = ld_frag8(qrow[0] + sn), q1n = ld_frag8(qrow[1] + sn);
        mma32(acc[0][0], p0, q0);
        mma32(acc[0][1], p0, q1);
        mma32(acc[1][0], p1, q0);
        mma32(acc[1][1], p1, q1);
        p0 = p0n; p1 = p1n; q0 = q0n; q1 = q1n;
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int j = j0 + 32 * tj + l31;
        if (j >= g.C) continue;
        const float cj = vT[(size_t)b * g.C + j] - lnu(g, j);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + 32 * ti + crow(r, hi);
                if (i < g.R) {
                    const size_t idx = ((size_t)b * g.R + i) * g.C + j;
                    const float E = __expf(Z[idx] + uT[(size_t)b * g.R + i] + cj);
                    gZ[idx] = G[idx] - E * acc[ti][tj][r];
                }
            }
    }
}

template <int NS> int skf_fwd_launch(const float* Zp, float* v2, float* u2, float* u_hist, float* v_hist, float* part,
                                     const Geo& g, int bc, int iters, hipStream_t st) {
    for (int it = 0; it < iters; ++it) {
        float* uh = u_hist + (size_t)it * g.B * g.R;
        if (it == 0) skf_fwd_iter<NS, true><<<dim3(g.nblk, bc), 256, 0, st>>>(Zp, v2, u2, uh, part, g);
        else skf_fwd_iter<NS, false><<<dim3(g.nblk, bc), 256, 0, st>>>(Zp, v2, u2, uh, part, g);
        skf_cols_fwd<<<dim3((g.Cp + 63) / 64, bc), 256, 0, st>>>(part, v2, v_hist + (size_t)it * g.B * g.C, it == 0, g);
    }
    return (int)hipGetLastError();
}
template <int NS> int skf_bwd_launch(const float* Zp, const float* u_hist, const float* v_hist, const float* gsum_row,
                                     const float* gsum_col, float* ubar_hist, float* vbar_hist, float* part, float* a2p,
                                     float* vbp, const Geo& g, int bc, int iters, hipStream_t st) {
    // all pointers already offset to the chunk's first pair; history strides are g.B * R (or C)
    skf_bwd_prep<<<dim3((g.Cp + 255) / 256, bc), 256, 0, st>>>(v_hist + (size_t)(iters - 1) * g.B * g.C, gsum_col, a2p, vbp, g);
    for (int k = iters; k >= 1; --k) {
        const float* uk = u_hist + (size_t)(k - 1) * g.B * g.R;
        const float* vk = v_hist + (size_t)(k - 1) * g.B * g.C;
        const float* vp = k >= 2 ? v_hist + (size_t)(k - 2) * g.B * g.C : nullptr;
        skf_bwd_iter<NS><<<dim3(g.nblk, bc), 256, 0, st>>>(Zp, uk, a2p, vbp, k == iters ? gsum_row : nullptr,
                                                           ubar_hist + (size_t)(k - 1) * g.B * g.R, part, g);
        skf_cols_bwd<<<dim3((g.Cp + 63) / 64, bc), 256, 0, st>>>(part, vk, vp, vbar_hist + (size_t)(k - 1) * g.B * g.C,
                                                                 a2p, vbp, g);
    }
    return (int)hipGetLastError();
}
const size_t LDS_BUDGET = 160 * 1024 - 512;

Geo make_geo(int B, int M, int N) {
    Geo g;
    g.B = B; g.M = M; g.N = N; g.R = M + 1; g.C = N + 1;
    g.Cp = (g.C + 3) & ~3;
    g.fast = (g.Cp >> 2) <= 64 * SKF_MAX_NS;
    if (g.fast) {
        g.RB = SKF_RPB;
    } else {
        // generic path: rows per block bounded by LDS (RB rows + 4 column vectors), at most 16
        size_t rb = (LDS_BUDGET - 4 * (size_t)g.C * 4 - 512) / ((size_t)g.C * 4);
        g.RB = (int)(rb > 16 ? 16 : rb);
    }
    g.nblk = g.RB > 0 ? (g.R + g.RB - 1) / g.RB : 0;
    g.norm = -logf((float)(M + N));
    g.lmu_last = logf((float)N) + g.norm;
    g.lnu_last = logf((float)M) + g.norm;
    return g;
}

// Pairs per chunk (balanced over the batch).  Measured on MI355X (tools/probe/time_sinkhorn.py, B=32, N=2048, T=100,
// forward ms): chunks of 1 / 2 / 4 / 8 / 16 / 32 pairs = 49.0 / 28.3 / 17.9 / 12.8 / 11.2 / 11.8 -- small chunks are
// launch/latency-bound and keeping a chunk under the Infinity Cache size (8 pairs = 134 MB) buys nothing: the row
// sweep streams at ~5 TB/s either way.  The chunk only bounds the padded copy / factor workspaces.
int batch_chunk(const Geo& g) {
    size_t per = g.fast ? (size_t)g.R * g.Cp * 4 + (size_t)g.nblk * g.Cp * 4
                        : (size_t)g.R * g.C * 4 + 2 * (size_t)g.nblk * g.C * 4;
    int ch = (int)((size_t)SK_CHUNK_MB * 1024 * 1024 / per);
    ch = ch < 1 ? 1 : (ch > g.B ? g.B : ch);
    const int nch = (g.B + ch - 1) / ch;
    return (g.B + nch - 1) / nch;
}

size_t rows_lds(const Geo& g, bool bwd) {
    return ((size_t)g.RB * g.C + 4 + (bwd ? 3 : 1) * (size_t)g.C + 2 * (size_t)g.RB) * 4 + 64;
}

// workspace carve (floats): [ partials | u cur | v cur | ubar hist | vbar hist | Zp (fast path) ]
constexpr int SKR_PART_CUS = 320;    // partial rows reserved for the resident path: 4 waves x this many CUs
struct Ws { float *part, *ucur, *vcur, *ubar_hist, *vbar_hist, *zp, *a2p, *vbp, *P, *Q; unsigned* ctr; int KP; size_t part_rows, total; };
Ws carve(void* ws, const Geo& g, int iters) {
    Ws w;
    float* p = reinterpret_cast<float*>(ws);
    const size_t wid = g.fast ? g.Cp : g.C;
    w.part_rows = 2 * (size_t)g.B * g.nblk;
    if (g.fast && w.part_rows < 4 * (size_t)SKR_PART_CUS) w.part_rows = 4 * (size_t)SKR_PART_CUS;   // resident path: one row per wave
    w.part = p;       p += w.part_rows * wid;
    w.ucur = p;       p += (size_t)g.B * g.R;
    w.vcur = p;       p += (size_t)g.B * wid;
    w.ubar_hist = p;  p += (size_t)(iters + 1) * g.B * g.R;
    w.vbar_hist = p;  p += (size_t)(iters + 1) * g.B * g.C;
    p += (4 - ((p - reinterpret_cast<float*>(ws)) & 3)) & 3;      // 16-byte aligned Zp
    w.zp = p;
    w.KP = (2 * iters + 15) & ~15;
    w.a2p = w.vbp = w.P = w.Q = nullptr;
    w.ctr = nullptr;
    if (g.fast) {
        const size_t ch = (size_t)batch_chunk(g);
        p += ch * g.R * g.Cp;
        w.a2p = p;  p += (size_t)g.B * g.Cp;
        w.vbp = p;  p += (size_t)g.B * g.Cp;
        w.P = p;    p += ch * g.R * w.KP;        // rank-2T factors of the final gradient (backward only)
        w.Q = p;    p += ch * g.C * w.KP;
        w.ctr = reinterpret_cast<unsigned*>(p);  p += 64;       // pair-barrier counters of the resident path
    }
    w.total = (size_t)(p - reinterpret_cast<float*>(ws)) * 4 + 1024;
    return w;
}

#include "sinkhorn_resident.h"

}  // namespace

// Host-only: the distribution the chip-resident path would use for this problem on a device of `ncu` compute units
// (out[8] = pairs per launch, workgroups per pair, waves per pair, rows per wave, waves with one more row, float4 columns per
// workgroup in the column phase, N / 256, LDS bytes).  Returns 1 and fills `out`, or 0 when the streaming kernels are used.
extern "C" int gf_sinkhorn_plan(int B, int M, int N, int ncu, int backward, int schedule, int64_t* out) {
    if (B <= 0 || M <= 0 || N <= 0 || ncu <= 0 || out == nullptr || (schedule & 3) == 3) return GF_ERR_SHAPE;
    const Geo g = make_geo(B, M, N);
    SkrPlan d;
    if (g.RB < 1 || !skr_plan(g, B, ncu, backward != 0, schedule & 3, d)) return 0;
    const Ws w = carve(nullptr, g, 1);
    if ((size_t)d.nw * d.bc > w.part_rows) return 0;
    const int64_t v[8] = {d.bc, d.wpp, d.nw, d.base, d.extra, d.cs, d.nsm, (int64_t)d.lds};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return 1;
}

extern "C" int64_t gf_sinkhorn_ws_bytes(int B, int M, int N, int iters) {
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0) return GF_ERR_SHAPE;
    Geo g = make_geo(B, M, N);
    if (g.RB < 1) return GF_ERR_UNSUPPORTED;
    return (int64_t)carve(nullptr, g, iters).total;
}

extern "C" int gf_sinkhorn_fwd(const float* Z, float* out, float* u_hist, float* v_hist, void* ws,
                               int B, int M, int N, int iters, int schedule, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0 || (schedule & 3) == 3) return GF_ERR_SHAPE;
    Geo g = make_geo(B, M, N);
    if (g.RB < 1) return GF_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(ws) & 15) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const Ws w = carve(ws, g, iters);
    SkrPlan rp;
    const bool resident = g.fast && iters > 0 && skr_plan(g, B, skr_cus(), false, schedule & 3, rp) &&
                          (size_t)rp.nw * rp.bc <= w.part_rows;
    const long long wait_ticks = resident ? skr_wait_ticks(schedule) : 0;
    const bool same_xcd_ok = resident && skr_same_xcd_allowed();
    const int ch = resident ? rp.bc : batch_chunk(g);
    const size_t zs = (size_t)g.R * g.C;
    const size_t lds = g.fast ? 0 : rows_lds(g, false);
    if (!g.fast) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sk_rows_fwd),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    for (int b0 = 0; b0 < B; b0 += ch) {
        const int bc = (B - b0) < ch ? (B - b0) : ch;
        if (g.fast) {
            if (iters > 0) {
                const size_t nv4 = (size_t)bc * g.R * (g.Cp >> 2);
                if (!resident) skf_prescale<<<dim3((unsigned)((nv4 + 255) / 256)), 256, 0, st>>>(Z + b0 * zs, w.zp, g, bc * g.R);
                if (resident) {                       // the chunk stays on the chip for all iterations; it is loaded from the
                    SkrArgs ra{};                     // couplings themselves and writes `out` from its last iteration (round 6)
                    ra.Zraw = Z + b0 * zs; ra.out = out + b0 * zs;
                    ra.Zp = w.zp; ra.part = w.part; ra.ctr = w.ctr;
                    ra.colA = w.a2p; ra.colB = w.vbp;                 // 16-byte aligned [B, Cp] scratch (free in the forward)
                    ra.u_hist = u_hist + (size_t)b0 * g.R; ra.v_hist = v_hist + (size_t)b0 * g.C;
                    ra.ustride = (size_t)B * g.R; ra.vstride = (size_t)B * g.C;
                    ra.iters = iters; ra.d = rp; ra.d.bc = bc; ra.g = g; ra.wait_ticks = wait_ticks; ra.safe_only = ((schedule >> 2) & 1) || !same_xcd_ok;
                    int rc = skr_launch<false>(ra, st);
                    if (rc) return rc;
                } else {
                const int ns = ((g.Cp >> 2) + 63) / 64;
                float* part = w.part + (size_t)b0 * g.nblk * g.Cp;
                float* u2 = w.ucur + (size_t)b0 * g.R;
                float* v2 = w.vcur + (size_t)b0 * g.Cp;
                float* uh = u_hist + (size_t)b0 * g.R;
                float* vh = v_hist + (size_t)b0 * g.C;
                int rc = [&]() -> int {
#define SKF_CALL_FWD(NSV) skf_fwd_launch<NSV>(w.zp, v2, u2, uh, vh, part, g, bc, iters, st)
                    switch (ns) {
                        case 1: return SKF_CALL_FWD(1); case 2: return SKF_CALL_FWD(2); case 3: return SKF_CALL_FWD(3);
                        case 4: return SKF_CALL_FWD(4); case 5: return SKF_CALL_FWD(5); case 6: return SKF_CALL_FWD(6);
                        case 7: return SKF_CALL_FWD(7); case 8: return SKF_CALL_FWD(8); default: return SKF_CALL_FWD(9);
                    }
#undef SKF_CALL_FWD
                }();
                if (rc) return rc;
                }
            }
        } else {
            float* pm = w.part;
            float* ps = pm + (size_t)B * g.nblk * g.C;
            for (int it = 0; it < iters; ++it) {
                sk_rows_fwd<<<dim3(g.nblk, bc), SK_THREADS, lds, st>>>(
                    Z + b0 * zs, it == 0 ? nullptr : w.vcur + (size_t)b0 * g.C, w.ucur + (size_t)b0 * g.R,
                    u_hist + ((size_t)it * B + b0) * g.R, pm + (size_t)b0 * g.nblk * g.C,
                    ps + (size_t)b0 * g.nblk * g.C, g);
                sk_cols_fwd<<<dim3((g.C + 63) / 64, bc), 256, 0, st>>>(
                    pm + (size_t)b0 * g.nblk * g.C, ps + (size_t)b0 * g.nblk * g.C, w.vcur + (size_t)b0 * g.C,
                    v_hist + ((size_t)it * B + b0) * g.C, g);
            }
        }
        // out = Z + u + v - norm with the final iterates (natural-log units = the last history entries); the resident kernel
        // has written it already
        if (!(g.fast && iters > 0 && resident))
        sk_final_fwd<<<dim3((g.C + 255) / 256, g.R, bc), 256, 0, st>>>(
            Z + b0 * zs, iters ? u_hist + ((size_t)(iters - 1) * B + b0) * g.R : nullptr,
            iters ? v_hist + ((size_t)(iters - 1) * B + b0) * g.C : nullptr, out + b0 * zs, g);
    }
    return (int)hipGetLastError();
}

extern "C" int gf_sinkhorn_bwd(const float* Z, const float* gout, const float* gsum_row, const float* gsum_col,
                               const float* u_hist, const float* v_hist, float* gZ, void* ws,
                               int B, int M, int N, int iters, int schedule, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0 || (schedule & 3) == 3) return GF_ERR_SHAPE;
    Geo g = make_geo(B, M, N);
    if (g.RB < 1) return GF_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(ws) & 15) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const Ws w = carve(ws, g, iters);
    float* ubar_hist = w.ubar_hist;                            // [iters, B, R]   (index k-1)
    float* vbar_hist = w.vbar_hist;                            // [iters+1, B, C] (index k, k = 0..T)
    const size_t zs = (size_t)g.R * g.C;
    if (iters == 0) {
        hipError_t e = gf_copy_f32(gZ, gout, (size_t)B * zs, st);
        return (int)e;
    }
    const size_t lds = g.fast ? 0 : rows_lds(g, true);
    hipError_t e = hipSuccess;
    if (!g.fast) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(sk_rows_bwd),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    // vbar^T = colsum(G)
    e = gf_copy_f32(vbar_hist + (size_t)iters * B * g.C, gsum_col, (size_t)B * g.C, st);
    if (e != hipSuccess) return (int)e;
    SkrPlan rp;
    const bool resident = g.fast && skr_plan(g, B, skr_cus(), true, schedule & 3, rp) &&
                          (size_t)rp.nw * rp.bc <= w.part_rows;
    const long long wait_ticks = resident ? skr_wait_ticks(schedule) : 0;
    const bool same_xcd_ok = resident && skr_same_xcd_allowed();
    const int ch = resident ? rp.bc : batch_chunk(g);
    for (int b0 = 0; b0 < B; b0 += ch) {
        const int bc = (B - b0) < ch ? (B - b0) : ch;
        if (g.fast) {
            const size_t nv4 = (size_t)bc * g.R * (g.Cp >> 2);
            skf_prescale<<<dim3((unsigned)((nv4 + 255) / 256)), 256, 0, st>>>(Z + b0 * zs, w.zp, g, bc * g.R);
            const int ns = ((g.Cp >> 2) + 63) / 64;
            float* part = w.part + (size_t)b0 * g.nblk * g.Cp;
            int rc = resident ? [&]() -> int {
                skf_bwd_prep<<<dim3((g.Cp + 255) / 256, bc), 256, 0, st>>>(
                    v_hist + ((size_t)(iters - 1) * B + b0) * g.C, gsum_col + (size_t)b0 * g.C, w.a2p, w.vbp, g);
                SkrArgs ra{};
                ra.Zp = w.zp; ra.part = w.part; ra.ctr = w.ctr;
                ra.colA = w.a2p; ra.colB = w.vbp;
                ra.u_hist = const_cast<float*>(u_hist) + (size_t)b0 * g.R;
                ra.v_hist = const_cast<float*>(v_hist) + (size_t)b0 * g.C;
                ra.base_row = gsum_row + (size_t)b0 * g.R;
                ra.ubar_hist = ubar_hist + (size_t)b0 * g.R; ra.vbar_hist = vbar_hist + (size_t)b0 * g.C;
                ra.ustride = (size_t)B * g.R; ra.vstride = (size_t)B * g.C;
                ra.iters = iters; ra.d = rp; ra.d.bc = bc; ra.g = g; ra.wait_ticks = wait_ticks; ra.safe_only = ((schedule >> 2) & 1) || !same_xcd_ok;
                return skr_launch<true>(ra, st);
            }() : [&]() -> int {
#define SKF_CALL_BWD(NSV) skf_bwd_launch<NSV>(w.zp, u_hist + (size_t)b0 * g.R, v_hist + (size_t)b0 * g.C,          \
                                              gsum_row + (size_t)b0 * g.R, gsum_col + (size_t)b0 * g.C,              \
                                              ubar_hist + (size_t)b0 * g.R, vbar_hist + (size_t)b0 * g.C, part,      \
                                              w.a2p + (size_t)b0 * g.Cp, w.vbp + (size_t)b0 * g.Cp, g, bc, iters, st)
                switch (ns) {
                    case 1: return SKF_CALL_BWD(1); case 2: return SKF_CALL_BWD(2); case 3: return SKF_CALL_BWD(3);
                    case 4: return SKF_CALL_BWD(4); case 5: return SKF_CALL_BWD(5); case 6: return SKF_CALL_BWD(6);
                    case 7: return SKF_CALL_BWD(7); case 8: return SKF_CALL_BWD(8); default: return SKF_CALL_BWD(9);
                }
#undef SKF_CALL_BWD
            }();
            if (rc) return rc;
        } else {
            float* psum = w.part;
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
        }
        if (g.fast) {
            skf_factors<<<dim3((max(g.R, g.C) + 255) / 256, iters, bc), 256, 0, st>>>(
                u_hist + (size_t)b0 * g.R, v_hist + (size_t)b0 * g.C, ubar_hist + (size_t)b0 * g.R,
                vbar_hist + (size_t)b0 * g.C, w.P, w.Q, iters, w.KP, (size_t)B * g.R, (size_t)B * g.C, g);
            skf_final_bwd<<<dim3(((g.C + 127) / 128) * ((g.R + 127) / 128), bc), 256, 0, st>>>(
                Z + b0 * zs, gout + b0 * zs, w.P, w.Q, u_hist + ((size_t)(iters - 1) * B + b0) * g.R,
                v_hist + ((size_t)(iters - 1) * B + b0) * g.C, gZ + b0 * zs, w.KP, g);
        } else {
            sk_final_bwd<<<dim3((g.C + 255) / 256, (g.R + 7) / 8, bc), 256, 0, st>>>(
                Z + b0 * zs, gout + b0 * zs, u_hist + (size_t)b0 * g.R, v_hist + (size_t)b0 * g.C,
                ubar_hist + (size_t)b0 * g.R, vbar_hist + (size_t)B * g.C + (size_t)b0 * g.C, gZ + b0 * zs, iters,
                (size_t)B * g.R, (size_t)B * g.C, (size_t)B * g.R, (size_t)B * g.C, g);
        }
    }
    return (int)hipGetLastError();
}
