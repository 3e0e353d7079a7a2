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

// ---- two single-output heads on the SAME rows (a LightGlue layer's matchability and token-confidence logits,
// lightglue.py:71,275-276 / :285-286): x is read once for both.  z0 = x w0 + b0 carries an input gradient (dx = base + dz0 w0),
// z1 = x w1 + b1 does not (the token-confidence head sees desc.detach(), lightglue.py:81-94).
template <typename T>
__global__ __launch_bounds__(256) void rowdot2_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w0,
                                                          const float* __restrict__ w1, const float* __restrict__ b0,
                                                          const float* __restrict__ b1, float* __restrict__ z0,
                                                          float* __restrict__ z1, int64_t M, int C) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float bias0 = b0 ? b0[0] : 0.f, bias1 = b1 ? b1[0] : 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        float a0 = 0.f, a1 = 0.f;
        for (int c = lane * VEC; c < C; c += 64 * VEC) {
            union { u32x4 u; T e[VEC]; } v;
            v.u = *reinterpret_cast<const u32x4*>(x + row * C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f32(v.e[e]);
                a0 += xv * w0[c + e];
                a1 += xv * w1[c + e];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); }
        if (lane == 0) { z0[row] = a0 + bias0; z1[row] = a1 + bias1; }
    }
}

// part [nblk][2][C + 1]: column sums of (dz0 x, dz0) and (dz1 x, dz1) per block
template <typename T, bool WITH_DX>
__global__ __launch_bounds__(256) void rowdot2_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dz0,
                                                          const float* __restrict__ dz1, const float* __restrict__ w0,
                                                          T* __restrict__ dx, const T* __restrict__ base,
                                                          float* __restrict__ part, int64_t M, int C) {
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);          // [256][2 (VEC + 1)]
    constexpr int RW = 2 * (VEC + 1);
    const int cpr = C / VEC;
    const int cc = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    const int rows_per_iter = 256 / cpr;
    float acc0[VEC], acc1[VEC], wv[VEC], bs0 = 0.f, bs1 = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; wv[e] = w0[cc * VEC + e]; }
    if (rl < rows_per_iter) {
        for (int64_t r = (int64_t)blockIdx.x * rows_per_iter + rl; r < M; r += (int64_t)gridDim.x * rows_per_iter) {
            const float g0 = dz0[r], g1 = dz1[r];
            union { u32x4 u; T e[VEC]; } v, o, bs;
            v.u = *reinterpret_cast<const u32x4*>(x + r * C + cc * VEC);
            if (WITH_DX && base) bs.u = *reinterpret_cast<const u32x4*>(base + r * C + cc * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f32(v.e[e]);
                acc0[e] += g0 * xv;
                acc1[e] += g1 * xv;
                if (WITH_DX) o.e[e] = from_f32<T>(base ? to_f32(bs.e[e]) + g0 * wv[e] : g0 * wv[e]);
            }
            if (WITH_DX) *reinterpret_cast<u32x4*>(dx + r * C + cc * VEC) = o.u;
            if (cc == 0) { bs0 += g0; bs1 += g1; }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        red[threadIdx.x * RW + e] = acc0[e];
        red[threadIdx.x * RW + (VEC + 1) + e] = acc1[e];
    }
    red[threadIdx.x * RW + VEC] = bs0;
    red[threadIdx.x * RW + (VEC + 1) + VEC] = bs1;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (C + 1); i += 256) {     // head h, column c (c == C: the bias slot)
        const int h = i / (C + 1), c = i % (C + 1);
        float s = 0.f;
        if (c < C) {
            const int ccx = c / VEC, e = c % VEC;
            for (int l = 0; l < rows_per_iter; ++l) s += red[(l * cpr + ccx) * RW + h * (VEC + 1) + e];
        } else {
            for (int l = 0; l < rows_per_iter; ++l) s += red[(l * cpr) * RW + h * (VEC + 1) + VEC];
        }
        part[((int64_t)blockIdx.x * 2 + h) * (C + 1) + c] = s;
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


extern "C" int gf_rowdot2_fwd(const void* x, const float* w0, const float* w1, const float* b0, const float* b1, float* z0,
                              float* z1, int M, int C, int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int nb = (int)(((int64_t)M + 3) / 4 > 4096 ? 4096 : ((int64_t)M + 3) / 4);
    if (dtype == GF_F32) {
        if (int e = rd_check<float>(C)) return e;
        rowdot2_fwd_kernel<float><<<nb, 256, 0, st>>>(reinterpret_cast<const float*>(x), w0, w1, b0, b1, z0, z1, M, C);
    } else if (dtype == GF_BF16) {
        if (int e = rd_check<bf16_t>(C)) return e;
        rowdot2_fwd_kernel<bf16_t><<<nb, 256, 0, st>>>(reinterpret_cast<const bf16_t*>(x), w0, w1, b0, b1, z0, z1, M, C);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_rowdot2_bwd(const void* x, const float* dz0, const float* dz1, const float* w0, void* dx, const void* base,
                              float* part, int M, int C, int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = rd_blocks(M);
    if (dtype == GF_F32) {
        if (int e = rd_check<float>(C)) return e;
        size_t lds = 256 * 10 * sizeof(float);
        if (dx) rowdot2_bwd_kernel<float, true><<<nb, 256, lds, st>>>(reinterpret_cast<const float*>(x), dz0, dz1, w0, reinterpret_cast<float*>(dx), reinterpret_cast<const float*>(base), part, M, C);
        else rowdot2_bwd_kernel<float, false><<<nb, 256, lds, st>>>(reinterpret_cast<const float*>(x), dz0, dz1, w0, nullptr, nullptr, part, M, C);
    } else if (dtype == GF_BF16) {
        if (int e = rd_check<bf16_t>(C)) return e;
        size_t lds = 256 * 18 * sizeof(float);
        if (dx) rowdot2_bwd_kernel<bf16_t, true><<<nb, 256, lds, st>>>(reinterpret_cast<const bf16_t*>(x), dz0, dz1, w0, reinterpret_cast<bf16_t*>(dx), reinterpret_cast<const bf16_t*>(base), part, M, C);
        else rowdot2_bwd_kernel<bf16_t, false><<<nb, 256, lds, st>>>(reinterpret_cast<const bf16_t*>(x), dz0, dz1, w0, nullptr, nullptr, part, M, C);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
