// Memory-bound fused elementwise kernels of the transformer block (HBM-roofline class):
//   * rotary embedding of the q,k thirds of the fused Wqkv output, in place, + its backward
//     (reference: gluefactory/models/matchers/lightglue.py:42-49, 159-160);
//   * LayerNorm(affine) + exact GELU of the FFN hidden, forward and backward
//     (reference: lightglue.py:143-148, the ffn.1 / ffn.2 modules).
// One wave per row/token, 4 waves per workgroup, grid-stride; fp32 math, T in / T out.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load8(float (&x)[8], const bf16_t* p) {
    union { u32x4_ u; bf16_t e[8]; } v;
    v.u = *reinterpret_cast<const u32x4_*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)v.e[e];
}
__device__ __forceinline__ void load8(float (&x)[8], const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[4 + e] = b[e]; }
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&x)[8]) {
    union { u32x4_ u; bf16_t e[8]; } v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v.e[e] = (bf16_t)x[e];
    *reinterpret_cast<u32x4_*>(p) = v.u;
}
__device__ __forceinline__ void store8(float* p, const float (&x)[8]) {
    f32x4 a = {x[0], x[1], x[2], x[3]}, b = {x[4], x[5], x[6], x[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

// ---------------------------------------------------------------------------- rotary
// qkv: [B*N, 3, H, D]; lane -> (half = q|k, pair i); loop over heads.
template <typename T, bool INVERSE, bool WITH_DTHETA>
__global__ __launch_bounds__(256) void rotary_kernel(T* qkv, const T* yrot, const float* cs, float* dtheta,
                                                     const float* dtheta_base, int64_t tokens, int H, int D) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int npair = D / 2;
    for (int64_t tok = (int64_t)blockIdx.x * 4 + wave; tok < tokens; tok += (int64_t)gridDim.x * 4) {
        // generic in D: work items = 2 (q,k) * npair, strided over the 64 lanes
        float carry = 0.f;
        for (int w = lane; w < 2 * npair; w += 64) {
            const int half = w / npair, i = w % npair;
            const f32x2 csv = *reinterpret_cast<const f32x2*>(cs + tok * D + 2 * i);
            const float c = csv[0], s = INVERSE ? -csv[1] : csv[1];
            float acc = 0.f;
            T* base = qkv + tok * 3 * H * D + (int64_t)half * H * D + 2 * i;
            const T* ybase = WITH_DTHETA ? yrot + tok * 3 * H * D + (int64_t)half * H * D + 2 * i : nullptr;
            for (int h = 0; h < H; ++h) {
                float x0 = to_f32(base[h * D]), x1 = to_f32(base[h * D + 1]);
                if (WITH_DTHETA) {
                    float y0 = to_f32(ybase[h * D]), y1 = to_f32(ybase[h * D + 1]);
                    acc += x1 * y0 - x0 * y1;  // x = upstream gradient here
                }
                base[h * D] = from_f32<T>(x0 * c - x1 * s);
                base[h * D + 1] = from_f32<T>(x1 * c + x0 * s);
            }
            if (WITH_DTHETA) {
                // q and k halves of the same pair live in different lanes (or different w): atomics
                // would be non-deterministic, so reduce through the other half explicitly.
                // npair <= 32 -> lanes l and l+npair hold (q,i) and (k,i) when 2*npair <= 64;
                // npair == 64 (head_dim 128): THIS lane visits (q, i) at w = lane and (k, i) at w = lane + 64.
                float other = __shfl(acc, (lane + npair) & 63);
                if (2 * npair <= 64) {
                    if (half == 0) dtheta[tok * npair + i] = acc + other + (dtheta_base ? dtheta_base[tok * npair + i] : 0.f);
                } else {
                    if (half == 0) carry = acc;
                    else dtheta[tok * npair + i] = carry + acc + (dtheta_base ? dtheta_base[tok * npair + i] : 0.f);
                }
            }
        }
    }
}

// Vector form for D % 8 == 0 and 2*H*D <= 512 (LightGlue: H=4, D=64 -> exactly one wave per token): the q|k
// block of a token is 2*H*D contiguous elements; lane w owns elements [8w, 8w+8) = 4 rotation pairs of one
// head (one 16-byte load/store in bf16), cos/sin come as 8 consecutive floats.  The angle gradient sums
// over the 2H lanes that share the same channel chunk (lane bits >= log2(D/8)).
template <typename T, bool INVERSE, bool WITH_DTHETA>
__global__ __launch_bounds__(256) void rotary_vec_kernel(T* qkv, const T* yrot, const float* cs, float* dtheta,
                                                         const float* dtheta_base, int64_t tokens, int H, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = 2 * H * D / 8, cpd = D / 8;           // work items per token, chunks per head
    const bool active = lane < W;
    const int d0 = (lane % cpd) * 8;
    for (int64_t tok = (int64_t)blockIdx.x * 4 + wave; tok < tokens; tok += (int64_t)gridDim.x * 4) {
        float x[8], y[8], c[8];
        T* base = qkv + tok * 3 * H * D + lane * 8;
        if (active) {
            load8(x, base);
            if (WITH_DTHETA) load8(y, yrot + tok * 3 * H * D + lane * 8);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + tok * D + d0);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs + tok * D + d0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { c[e] = c0[e]; c[4 + e] = c1[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[e] = 0.f; y[e] = 0.f; c[e] = 0.f; }
        }
        float acc[4], o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float co = c[2 * i], si = INVERSE ? -c[2 * i + 1] : c[2 * i + 1];
            if (WITH_DTHETA) acc[i] = x[2 * i + 1] * y[2 * i] - x[2 * i] * y[2 * i + 1];   // x = upstream gradient
            o[2 * i] = x[2 * i] * co - x[2 * i + 1] * si;
            o[2 * i + 1] = x[2 * i + 1] * co + x[2 * i] * si;
        }
        if (active) store8(base, o);
        if (WITH_DTHETA) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                for (int off = cpd; off < 64; off <<= 1) acc[i] += __shfl_xor(acc[i], off);
            if (lane < cpd) {
                f32x4 v = {acc[0], acc[1], acc[2], acc[3]};
                // the angles are shared by every layer: the running sum of the layers behind this one rides along
                if (dtheta_base) v += *reinterpret_cast<const f32x4*>(dtheta_base + tok * (D / 2) + lane * 4);
                *reinterpret_cast<f32x4*>(dtheta + tok * (D / 2) + lane * 4) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------- LN + GELU
// Exact (erf) GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far inside the 1e-4 parity
// budget): ~12 VALU + one v_exp instead of libm erff's ~30-instruction polynomial ladder; the backward
// reuses the same exp(-z^2/2) for the Gaussian density term.
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& pdf) {
    const float x = fabsf(z) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * x);   // v_rcp_f32 (1 ulp); __frcp_rn is a full IEEE divide
    const float e = __expf(-x * x);                       // = exp(-z^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * e;
    cdf = 0.5f * (1.f + copysignf(erf_abs, z));
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float z) {
    float cdf, pdf;
    gelu_parts(z, cdf, pdf);
    return z * cdf;
}
__device__ __forceinline__ float gelu_grad(float z) {
    float cdf, pdf;
    gelu_parts(z, cdf, pdf);
    return cdf + z * pdf;
}

template <typename T> struct Chunk { static constexpr int VEC = 16 / sizeof(T); };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// FULL: C == NCH * 64 * VEC, every lane's chunk is inside the row (no exec-masked branches)
template <typename T, int NCH, bool FULL = false>
__device__ __forceinline__ void load_row(float (&x)[NCH][16 / sizeof(T)], const T* row, int C, int lane) {
    constexpr int VEC = 16 / sizeof(T);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        int col = (ch * 64 + lane) * VEC;
        if (FULL || col < C) {
            union { u32x4 u; T e[VEC]; } v;
            v.u = *reinterpret_cast<const u32x4*>(row + col);
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[ch][e] = to_f32(v.e[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[ch][e] = 0.f;
        }
    }
}
template <typename T, int NCH, bool FULL = false>
__device__ __forceinline__ void store_row(T* row, const float (&y)[NCH][16 / sizeof(T)], int C, int lane) {
    constexpr int VEC = 16 / sizeof(T);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        int col = (ch * 64 + lane) * VEC;
        if (FULL || col < C) {
            union { u32x4 u; T e[VEC]; } v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.e[e] = from_f32<T>(y[ch][e]);
            *reinterpret_cast<u32x4*>(row + col) = v.u;
        }
    }
}

template <typename T, int NCH, bool FULL>
__global__ __launch_bounds__(256) void ln_gelu_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                          float* mean, float* rstd, int R, int C, float eps) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][VEC], bt[NCH][VEC];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            int col = (ch * 64 + lane) * VEC + e;
            g[ch][e] = (FULL || col < C) ? gamma[col] : 0.f;
            bt[ch][e] = (FULL || col < C) ? beta[col] : 0.f;
        }
    // RPI rows per wave and iteration, the next ones prefetched
    constexpr int RPI = 1;       // 2 rows per iteration measured no faster
    const int stride = gridDim.x * 4 * RPI;
    const float inv_c = 1.f / C;
    int row = (blockIdx.x * 4 + wave) * RPI;
    float nxt[RPI][NCH][VEC];
#pragma unroll
    for (int r = 0; r < RPI; ++r)
        if (row + r < R) load_row<T, NCH, FULL>(nxt[r], x + (int64_t)(row + r) * C, C, lane);
    for (; row < R; row += stride) {
        float v[RPI][NCH][VEC];
#pragma unroll
        for (int r = 0; r < RPI; ++r)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[r][ch][e] = nxt[r][ch][e];
#pragma unroll
        for (int r = 0; r < RPI; ++r)
            if (row + stride + r < R) load_row<T, NCH, FULL>(nxt[r], x + (int64_t)(row + stride + r) * C, C, lane);
        // shifted single pass: sums of d = x - x0 and d^2 (x0 = the row's first element keeps the
        // cancellation of E[d^2] - E[d]^2 harmless)
        float x0[RPI], s[RPI], q[RPI];
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            x0[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[r][0][0]), 0));
            s[r] = 0.f;
            q[r] = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    int col = (ch * 64 + lane) * VEC + e;
                    float d = (FULL || col < C) ? v[r][ch][e] - x0[r] : 0.f;
                    s[r] += d;
                    q[r] += d * d;
                }
        }
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            s[r] = wave_allsum(s[r]);
            q[r] = wave_allsum(q[r]);
        }
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            if (row + r >= R) break;
            const float md = s[r] * inv_c;
            const float mu = x0[r] + md;
            const float rs = rsqrtf(fmaxf(q[r] * inv_c - md * md, 0.f) + eps);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    v[r][ch][e] = gelu_f((v[r][ch][e] - mu) * rs * g[ch][e] + bt[ch][e]);
            store_row<T, NCH, FULL>(y + (int64_t)(row + r) * C, v[r], C, lane);
            if (lane == 0) { mean[row + r] = mu; rstd[row + r] = rs; }
        }
    }
}

template <typename T, int NCH, bool FULL>
__global__ __launch_bounds__(256) void ln_gelu_bwd_kernel(const T* x, const float* gamma, const float* beta,
                                                          const float* mean, const float* rstd, const T* dy, T* dx,
                                                          float* dgamma_part, float* dbeta_part, int R, int C) {
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][NCH*64*VEC]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NCH][VEC], bt[NCH][VEC], dg[NCH][VEC], db[NCH][VEC];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            int col = (ch * 64 + lane) * VEC + e;
            g[ch][e] = (FULL || col < C) ? gamma[col] : 0.f;
            bt[ch][e] = (FULL || col < C) ? beta[col] : 0.f;
            dg[ch][e] = 0.f;
            db[ch][e] = 0.f;
        }
    const int stride = gridDim.x * 4;
    const float inv_c = 1.f / C;
    int row = blockIdx.x * 4 + wave;
    float nv[NCH][VEC], nd[NCH][VEC];
    if (row < R) {
        load_row<T, NCH, FULL>(nv, x + (int64_t)row * C, C, lane);
        load_row<T, NCH, FULL>(nd, dy + (int64_t)row * C, C, lane);
    }
    for (; row < R; row += stride) {
        float v[NCH][VEC], d[NCH][VEC];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[ch][e] = nv[ch][e]; d[ch][e] = nd[ch][e]; }
        if (row + stride < R) {                                                                   // prefetch
            load_row<T, NCH, FULL>(nv, x + (int64_t)(row + stride) * C, C, lane);
            load_row<T, NCH, FULL>(nd, dy + (int64_t)(row + stride) * C, C, lane);
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                int col = (ch * 64 + lane) * VEC + e;
                float xh = (v[ch][e] - mu) * rs;
                float z = xh * g[ch][e] + bt[ch][e];
                float dz = (FULL || col < C) ? d[ch][e] * gelu_grad(z) : 0.f;
                dg[ch][e] += dz * xh;
                db[ch][e] += dz;
                float dxh = dz * g[ch][e];
                v[ch][e] = xh;
                d[ch][e] = dxh;
                s1 += dxh;
                s2 += dxh * xh;
            }
        const float m1 = wave_allsum(s1) * inv_c, m2 = wave_allsum(s2) * inv_c;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[ch][e] = rs * (d[ch][e] - m1 - v[ch][e] * m2);
        store_row<T, NCH, FULL>(dx + (int64_t)row * C, d, C, lane);
    }
    // reduce the 4 waves' column partials through LDS, one partial row per block
    constexpr int W = NCH * 64 * VEC;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            int col = (ch * 64 + lane) * VEC + e;
            red[(wave * 2 + 0) * W + col] = dg[ch][e];
            red[(wave * 2 + 1) * W + col] = db[ch][e];
        }
    __syncthreads();
    for (int col = threadIdx.x; col < C; col += 256) {
        float a = 0.f, b2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 2) * W + col]; b2 += red[(w * 2 + 1) * W + col]; }
        dgamma_part[(int64_t)blockIdx.x * C + col] = a;
        dbeta_part[(int64_t)blockIdx.x * C + col] = b2;
    }
}

int ln_blocks(int R) {
    int nb = (R + 3) / 4;
    return nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
}

template <typename T> int ln_nch(int C) {
    constexpr int VEC = 16 / sizeof(T);
    if (C % VEC) return -1;
    int chunks = C / VEC;
    if (chunks <= 64) return 1;
    if (chunks <= 128) return 2;
    if (chunks <= 256) return 4;
    return -1;
}

template <typename T>
int ln_fwd_t(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int R, int C,
             float eps, hipStream_t st) {
    int nb = ln_blocks(R);
    const T* xp = reinterpret_cast<const T*>(x);
    T* yp = reinterpret_cast<T*>(y);
    constexpr int VEC = 16 / sizeof(T);
    const int nch = ln_nch<T>(C);
    const bool full = nch > 0 && C == nch * 64 * VEC;
#define GF_LN_FWD(N_, F_) ln_gelu_fwd_kernel<T, N_, F_><<<nb, 256, 0, st>>>(xp, gamma, beta, yp, mean, rstd, R, C, eps)
    switch (nch) {
        case 1: if (full) GF_LN_FWD(1, true); else GF_LN_FWD(1, false); break;
        case 2: if (full) GF_LN_FWD(2, true); else GF_LN_FWD(2, false); break;
        case 4: if (full) GF_LN_FWD(4, true); else GF_LN_FWD(4, false); break;
        default: return GF_ERR_UNSUPPORTED;
    }
#undef GF_LN_FWD
    return (int)hipGetLastError();
}

template <typename T>
int ln_bwd_t(const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
             const void* dy, void* dx, float* dgp, float* dbp, int R, int C, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    int nb = ln_blocks(R);
    const T* xp = reinterpret_cast<const T*>(x);
    const T* dyp = reinterpret_cast<const T*>(dy);
    T* dxp = reinterpret_cast<T*>(dx);
    int nch = ln_nch<T>(C);
    if (nch < 0) return GF_ERR_UNSUPPORTED;
    size_t lds = (size_t)8 * nch * 64 * VEC * sizeof(float);
    const bool full = C == nch * 64 * VEC;
#define GF_LN_BWD(N_, F_) ln_gelu_bwd_kernel<T, N_, F_><<<nb, 256, lds, st>>>(xp, gamma, beta, mean, rstd, dyp, dxp, dgp, dbp, R, C)
    switch (nch) {
        case 1: if (full) GF_LN_BWD(1, true); else GF_LN_BWD(1, false); break;
        case 2: if (full) GF_LN_BWD(2, true); else GF_LN_BWD(2, false); break;
        default: if (full) GF_LN_BWD(4, true); else GF_LN_BWD(4, false); break;
    }
#undef GF_LN_BWD
    return (int)hipGetLastError();
}

}  // namespace

// vector kernel: D % 8 == 0, one work item per lane, power-of-two chunk count per head (dtheta butterfly)
static bool rotary_vec_ok(int H, int D) {
    const int cpd = D / 8;
    return D % 8 == 0 && 2 * H * D / 8 <= 64 && (cpd & (cpd - 1)) == 0 && (64 % cpd) == 0;
}

extern "C" int gf_rotary_qk(void* qkv, const float* cs, int B, int N, int H, int D, int inverse,
                            int dtype, void* stream) {
    if (B <= 0 || N <= 0 || H <= 0 || D <= 0 || (D & 1)) return GF_ERR_SHAPE;
    if (dtype != GF_F32 && dtype != GF_BF16) return GF_ERR_DTYPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t tokens = (int64_t)B * N;
    int nb = (int)((tokens + 3) / 4 > 8192 ? 8192 : (tokens + 3) / 4);
    const bool vec = rotary_vec_ok(H, D);
#define GF_ROT(K, T, INV) K<T, INV, false><<<nb, 256, 0, st>>>((T*)qkv, nullptr, cs, nullptr, nullptr, tokens, H, D)
    if (dtype == GF_F32) {
        if (vec) { if (inverse) GF_ROT(rotary_vec_kernel, float, true); else GF_ROT(rotary_vec_kernel, float, false); }
        else { if (inverse) GF_ROT(rotary_kernel, float, true); else GF_ROT(rotary_kernel, float, false); }
    } else {
        if (vec) { if (inverse) GF_ROT(rotary_vec_kernel, bf16_t, true); else GF_ROT(rotary_vec_kernel, bf16_t, false); }
        else { if (inverse) GF_ROT(rotary_kernel, bf16_t, true); else GF_ROT(rotary_kernel, bf16_t, false); }
    }
#undef GF_ROT
    return (int)hipGetLastError();
}

extern "C" int gf_rotary_qk_bwd(void* dqkv, const void* qkv_rot, const float* cs, float* dtheta, const float* dtheta_base,
                                int B, int N, int H, int D, int dtype, void* stream) {
    if (B <= 0 || N <= 0 || H <= 0 || D <= 0 || (D & 1)) return GF_ERR_SHAPE;
    if (D > 64 && D != 128) return GF_ERR_UNSUPPORTED;  // the q/k pair reduction stays inside one wave pass (<= 64) or one lane (128)
    if (dtype != GF_F32 && dtype != GF_BF16) return GF_ERR_DTYPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t tokens = (int64_t)B * N;
    int nb = (int)((tokens + 3) / 4 > 8192 ? 8192 : (tokens + 3) / 4);
    const bool vec = rotary_vec_ok(H, D);
#define GF_ROTB(K, T) K<T, true, true><<<nb, 256, 0, st>>>((T*)dqkv, (const T*)qkv_rot, cs, dtheta, dtheta_base, tokens, H, D)
    if (dtype == GF_F32) { if (vec) GF_ROTB(rotary_vec_kernel, float); else GF_ROTB(rotary_kernel, float); }
    else { if (vec) GF_ROTB(rotary_vec_kernel, bf16_t); else GF_ROTB(rotary_kernel, bf16_t); }
#undef GF_ROTB
    return (int)hipGetLastError();
}

extern "C" int gf_ln_gelu_nblk(int R) { return ln_blocks(R); }

extern "C" int gf_ln_gelu_fwd(const void* x, const float* gamma, const float* beta, void* y,
                              float* mean, float* rstd, int R, int C, float eps, int dtype, void* stream) {
    if (R <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return ln_fwd_t<float>(x, gamma, beta, y, mean, rstd, R, C, eps, st);
    if (dtype == GF_BF16) return ln_fwd_t<bf16_t>(x, gamma, beta, y, mean, rstd, R, C, eps, st);
    return GF_ERR_DTYPE;
}

extern "C" int gf_ln_gelu_bwd(const void* x, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, const void* dy, void* dx, float* dgamma_part,
                              float* dbeta_part, int R, int C, int dtype, void* stream) {
    if (R <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return ln_bwd_t<float>(x, gamma, beta, mean, rstd, dy, dx, dgamma_part, dbeta_part, R, C, st);
    if (dtype == GF_BF16) return ln_bwd_t<bf16_t>(x, gamma, beta, mean, rstd, dy, dx, dgamma_part, dbeta_part, R, C, st);
    return GF_ERR_DTYPE;
}
