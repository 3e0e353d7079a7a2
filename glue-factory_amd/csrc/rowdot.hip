// Single-output linear heads: z[m] = x[m,:] . w + b  (matchability and token-confidence logits,
// reference lightglue.py:71,275-276,285-286: nn.Linear(dim, 1)).  The library serves these GEMVs and
// their gradients at ~80 us per call on [131072, 256]; they are pure streaming work:
//   fwd    one wave per row, 16-byte loads, butterfly reduction            (reads x once)
//   bwd    dx[m,:] = dz[m] * w (optional, elementwise)                     (writes dx once)
//          dw = sum_m dz[m] x[m,:], db = sum_m dz[m] via per-block column partials (reads x once)
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, float bias,
                                                         const float* __restrict__ bias_dev, float* __restrict__ z, int64_t M,
                                                         int C) {
    if (bias_dev != nullptr) bias += bias_dev[0];          // the module's bias parameter, read on the device (no host sync)
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        float acc = 0.f;
        for (int c = lane * VEC; c < C; c += 64 * VEC) {
            union { u32x4 u; T e[VEC]; } v;
            v.u = *reinterpret_cast<const u32x4*>(x + row * C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += to_f32(v.e[e]) * w[c + e];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) z[row] = acc + bias;
    }
}

// dx = dz * w (WITH_DX) and per-block partials of dw (column sums of dz[m] * x[m,:]) and db
template <typename T, bool WITH_DX>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dz,
                                                         const float* __restrict__ w, T* __restrict__ dx,
                                                         const T* __restrict__ base,
                                                         float* __restrict__ part, int64_t M, int C) {
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);          // [256][VEC + 1]
    const int cpr = C / VEC;
    const int cc = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    const int rows_per_iter = 256 / cpr;
    float acc[VEC], wv[VEC], bsum = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[e] = 0.f; wv[e] = w[cc * VEC + e]; }
    if (rl < rows_per_iter) {
        for (int64_t r = (int64_t)blockIdx.x * rows_per_iter + rl; r < M; r += (int64_t)gridDim.x * rows_per_iter) {
            const float g = dz[r];
            union { u32x4 u; T e[VEC]; } v, o, bs;
            v.u = *reinterpret_cast<const u32x4*>(x + r * C + cc * VEC);
            if (WITH_DX && base) bs.u = *reinterpret_cast<const u32x4*>(base + r * C + cc * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                acc[e] += g * to_f32(v.e[e]);
                if (WITH_DX) o.e[e] = from_f32<T>(base ? to_f32(bs.e[e]) + g * wv[e] : g * wv[e]);
            }
            if (WITH_DX) *reinterpret_cast<u32x4*>(dx + r * C + cc * VEC) = o.u;
            if (cc == 0) bsum += g;
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * (VEC + 1) + e] = acc[e];
    red[threadIdx.x * (VEC + 1) + VEC] = bsum;
    __syncthreads();
    for (int c = threadIdx.x; c <= C; c += 256) {          // column C = the bias slot
        float s = 0.f;
        if (c < C) {
            const int ccx = c / VEC, e = c % VEC;
            for (int l = 0; l < rows_per_iter; ++l) s += red[(l * cpr + ccx) * (VEC + 1) + e];
        } else {
            for (int l = 0; l < rows_per_iter; ++l) s += red[(l * cpr) * (VEC + 1) + VEC];
        }
        part[(int64_t)blockIdx.x * (C + 1) + c] = s;
    }
}

int rd_blocks(int64_t M) {
    int64_t nb = (M + 255) / 256;
    return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}

template <typename T> int rd_check(int C) {
    constexpr int VEC = 16 / sizeof(T);
    return (C % VEC == 0 && C / VEC <= 256) ? 0 : GF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int gf_rowdot_nblk(int M) { return rd_blocks(M); }

extern "C" int gf_rowdot_fwd(const void* x, const float* w, float bias, const float* bias_dev, float* z, int M, int C,
                             int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int nb = (int)(((int64_t)M + 3) / 4 > 4096 ? 4096 : ((int64_t)M + 3) / 4);
    if (dtype == GF_F32) {
        if (int e = rd_check<float>(C)) return e;
        rowdot_fwd_kernel<float><<<nb, 256, 0, st>>>(reinterpret_cast<const float*>(x), w, bias, bias_dev, z, M, C);
    } else if (dtype == GF_BF16) {
        if (int e = rd_check<bf16_t>(C)) return e;
        rowdot_fwd_kernel<bf16_t><<<nb, 256, 0, st>>>(reinterpret_cast<const bf16_t*>(x), w, bias, bias_dev, z, M, C);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_rowdot_bwd(const void* x, const float* dz, const float* w, void* dx, const void* base, float* part,
                             int M, int C, int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = rd_blocks(M);
    if (dtype == GF_F32) {
        if (int e = rd_check<float>(C)) return e;
        size_t lds = 256 * 5 * sizeof(float);
        if (dx) rowdot_bwd_kernel<float, true><<<nb, 256, lds, st>>>(reinterpret_cast<const float*>(x), dz, w, reinterpret_cast<float*>(dx), reinterpret_cast<const float*>(base), part, M, C);
        else rowdot_bwd_kernel<float, false><<<nb, 256, lds, st>>>(reinterpret_cast<const float*>(x), dz, w, nullptr, nullptr, part, M, C);
    } else if (dtype == GF_BF16) {
        if (int e = rd_check<bf16_t>(C)) return e;
        size_t lds = 256 * 9 * sizeof(float);
        if (dx) rowdot_bwd_kernel<bf16_t, true><<<nb, 256, lds, st>>>(reinterpret_cast<const bf16_t*>(x), dz, w, reinterpret_cast<bf16_t*>(dx), reinterpret_cast<const bf16_t*>(base), part, M, C);
        else rowdot_bwd_kernel<bf16_t, false><<<nb, 256, lds, st>>>(reinterpret_cast<const bf16_t*>(x), dz, w, nullptr, nullptr, part, M, C);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
