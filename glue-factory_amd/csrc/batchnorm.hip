// BatchNorm1d(+ReLU) over channels-last activations [M = B*N, C], batch statistics in training.
//
// Replaces the BatchNorm1d+ReLU pairs of the SuperGlue / GlueStick MLPs (reference:
// gluefactory_nonfree/superglue.py:70-79, gluefactory/models/matchers/gluestick.py:465-474),
// which torch serves with channels-last reduction kernels at ~0.2 TB/s for these shapes.
// HBM-bound design: one column-sum pass (per-thread 16-byte column chunk, rows strided over the
// workgroup, LDS combine, per-block partials -> host-side tiny sum / all-reduce for SyncBN) and
// one fused normalise+affine+ReLU pass; the backward mirrors it (sum dz, sum dz*xhat, then dx).
// T in / T out, fp32 statistics.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef GF_BN_UNR
#define GF_BN_UNR 8           // rows of a thread in flight in the statistics passes (probe builds override it)
#endif

int bn_blocks(int M) {
    int nb = (M + 255) / 256;
    return nb < 1 ? 1 : (nb > 512 ? 512 : nb);
}

// MODE 0: part[blk][0][c] = sum x, part[blk][1][c] = sum x^2
// MODE 1: dz = dy * (relu ? z > 0 : 1); part[0] = sum dz, part[1] = sum dz * xhat
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_colsum_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ part, int M, int C, int relu) {
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);      // [256][2*VEC]
    const int cpr = C / VEC;                          // chunks per row
    const int cc = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    const int rows_per_iter = 256 / cpr;              // row lanes (threads beyond cpr*rows_per_iter idle)
    float a0[VEC], a1[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        a0[e] = 0.f; a1[e] = 0.f;
        if (MODE == 1) {
            int c = cc * VEC + e;
            mu[e] = mean[c]; rs[e] = rstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
        }
    }
    if (rl < rows_per_iter) {
        // UNR rows of this thread in flight at once: with one 16-byte load per thread and iteration the kernel was latency-
        // bound at ~2 TB/s (512 workgroups x 4 KB)
        constexpr int UNR = GF_BN_UNR;
        const int64_t stride = (int64_t)gridDim.x * rows_per_iter;
        for (int64_t r0 = (int64_t)blockIdx.x * rows_per_iter + rl; r0 < M; r0 += UNR * stride) {
            union Chunk { u32x4 u; T e[VEC]; };
            Chunk v[UNR], d[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int64_t r = r0 + j * stride;
                const int64_t rc = r < M ? r : r0;                       // (clamped load, masked below)
                v[j].u = *reinterpret_cast<const u32x4*>(x + rc * C + cc * VEC);
                if (MODE == 1) d[j].u = *reinterpret_cast<const u32x4*>(dy + rc * C + cc * VEC);
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                if (r0 + j * stride >= M) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float xv = to_f32(v[j].e[e]);
                    if (MODE == 0) {
                        a0[e] += xv;
                        a1[e] += xv * xv;
                    } else {
                        float xh = (xv - mu[e]) * rs[e];
                        float z = xh * ga[e] + be[e];
                        float dz = (relu && z <= 0.f) ? 0.f : to_f32(d[j].e[e]);
                        a0[e] += dz;
                        a1[e] += dz * xh;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        red[threadIdx.x * 2 * VEC + e] = a0[e];
        red[threadIdx.x * 2 * VEC + VEC + e] = a1[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int ccx = c / VEC, e = c % VEC;
        float s0 = 0.f, s1 = 0.f;
        for (int l = 0; l < rows_per_iter; ++l) {
            const int t = l * cpr + ccx;
            s0 += red[t * 2 * VEC + e];
            s1 += red[t * 2 * VEC + VEC + e];
        }
        part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s0;
        part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
}

// MODE 0 (forward):  y = act(xhat * gamma + beta)
// MODE 1 (backward): dx = gamma * rstd * (dz - m1 - xhat * m2), dz = dy * (z > 0), m1/m2 per-channel means
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ m1, const float* __restrict__ m2,
                                                       T* __restrict__ out, int64_t M, int C, int relu) {
    constexpr int VEC = 16 / sizeof(T);
    const int cpr = C / VEC;
    const int64_t total = M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cc = (int)(i % cpr);
        union { u32x4 u; T e[VEC]; } v, d, o;
        v.u = *reinterpret_cast<const u32x4*>(x + i * VEC);
        if (MODE == 1) d.u = *reinterpret_cast<const u32x4*>(dy + i * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = cc * VEC + e;
            const float xh = (to_f32(v.e[e]) - mean[c]) * rstd[c];
            const float z = xh * gamma[c] + beta[c];
            if (MODE == 0) {
                o.e[e] = from_f32<T>((relu && z <= 0.f) ? 0.f : z);
            } else {
                const float dz = (relu && z <= 0.f) ? 0.f : to_f32(d.e[e]);
                o.e[e] = from_f32<T>(gamma[c] * rstd[c] * (dz - m1[c] - xh * m2[c]));
            }
        }
        *reinterpret_cast<u32x4*>(out + i * VEC) = o.u;
    }
}

template <typename T>
int bn_check(int C) {
    constexpr int VEC = 16 / sizeof(T);
    return (C % VEC == 0 && C / VEC <= 256) ? 0 : GF_ERR_UNSUPPORTED;
}

template <typename T, int MODE>
int launch_colsum(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, float* part, int M, int C, int relu, hipStream_t st) {
    if (int e = bn_check<T>(C)) return e;
    constexpr int VEC = 16 / sizeof(T);
    size_t lds = 256 * 2 * VEC * sizeof(float);
    bn_colsum_kernel<T, MODE><<<bn_blocks(M), 256, lds, st>>>(reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy),
                                                              mean, rstd, gamma, beta, part, M, C, relu);
    return (int)hipGetLastError();
}
template <typename T, int MODE>
int launch_apply(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, const float* m1, const float* m2, void* out, int M, int C, int relu, hipStream_t st) {
    if (int e = bn_check<T>(C)) return e;
    constexpr int VEC = 16 / sizeof(T);
    int64_t total = (int64_t)M * (C / VEC);
    int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    bn_apply_kernel<T, MODE><<<nb, 256, 0, st>>>(reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), mean, rstd,
                                                 gamma, beta, m1, m2, reinterpret_cast<T*>(out), M, C, relu);
    return (int)hipGetLastError();
}


// Everything between the per-block partial sums and the apply pass, in one launch per direction (was ~15 tiny
// tensor kernels per BatchNorm call: sum over blocks, divisions, clamp, rsqrt, the running-statistics update).
// fwd: part [nblk][2][C] = (sum x, sum x^2) -> mean, var (biased), rstd; running_mean / running_var (may be NULL)
//      updated in place with `momentum` and the unbiased variance, as torch.nn.BatchNorm1d does.
// (256 threads = 32 channels x 8 block groups: the <= 512 partials of a channel are summed 8-way in parallel, then
// through LDS -- a single thread per channel walking 512 dependent loads took ~100 us)
__device__ __forceinline__ void bn_block_sums(const float* __restrict__ part, int nblk, int C, int c, float& s0, float& s1) {
    __shared__ float red[2][8][32];
    const int cl = threadIdx.x & 31, kg = threadIdx.x >> 5;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    if (c < C) {
        int k = kg;
        // eight partial rows in flight per thread (the two-row loop below waits for its loads every time round: nblk / 16
        // dependent round trips, which is ALL this kernel's time); same two chains per statistic, same order
        for (; k + 56 < nblk; k += 64) {
            float x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0[u] = part[(int64_t)(k + 8 * u) * 2 * C + c];
                x1[u] = part[(int64_t)(k + 8 * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += x0[u]; a1 += x1[u]; b0 += x0[u + 1]; b1 += x1[u + 1]; }
        }
        for (; k + 8 < nblk; k += 16) {
            a0 += part[(int64_t)k * 2 * C + c];
            a1 += part[(int64_t)k * 2 * C + C + c];
            b0 += part[(int64_t)(k + 8) * 2 * C + c];
            b1 += part[(int64_t)(k + 8) * 2 * C + C + c];
        }
        if (k < nblk) {
            a0 += part[(int64_t)k * 2 * C + c];
            a1 += part[(int64_t)k * 2 * C + C + c];
        }
    }
    red[0][kg][cl] = a0 + b0;
    red[1][kg][cl] = a1 + b1;
    __syncthreads();
    s0 = s1 = 0.f;
    if (kg == 0) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { s0 += red[0][g][cl]; s1 += red[1][g][cl]; }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ part, int nblk, int C, float n, float eps,
                                                              float momentum, float* __restrict__ mean, float* __restrict__ var,
                                                              float* __restrict__ rstd, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    float s0, s1;
    bn_block_sums(part, nblk, C, c, s0, s1);
    if (threadIdx.x >= 32 || c >= C) return;
    const float m = s0 / n;
    const float v = fmaxf(s1 / n - m * m, 0.f);
    mean[c] = m;
    var[c] = v;
    rstd[c] = rsqrtf(v + eps);
    if (run_mean) {
        run_mean[c] = run_mean[c] * (1.f - momentum) + m * momentum;
        run_var[c] = run_var[c] * (1.f - momentum) + v * (n / fmaxf(n - 1.f, 1.f)) * momentum;
    }
}
// The running statistics take the batch statistics of `sets` forward calls AGAIN (mvr = [sets][3][C]: mean, biased var, rstd
// as bn_finalize_fwd_kernel wrote them), set after set -- what an activation-checkpointed reference does to its BatchNorm
// buffers when the backward re-runs the forward (superglue.py:160-169, gluestick.py:724-757).
__global__ __launch_bounds__(256) void bn_replay_running_kernel(const float* __restrict__ mvr, int sets, int C, float n, float momentum,
                                                                float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                const float* __restrict__ skip) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C || (skip != nullptr && !(*skip == 0.f))) return;      // (a NaN flag skips too)
    float rm = run_mean[c], rv = run_var[c];
    const float unbias = n / fmaxf(n - 1.f, 1.f);
    for (int h = 0; h < sets; ++h) {
        rm = rm * (1.f - momentum) + mvr[(size_t)(3 * h) * C + c] * momentum;
        rv = rv * (1.f - momentum) + mvr[(size_t)(3 * h + 1) * C + c] * unbias * momentum;
    }
    run_mean[c] = rm;
    run_var[c] = rv;
}
// bwd: part = (sum dz, sum dz * xhat) -> dbeta, dgamma (the sums) and m1 = sum dz / n, m2 = sum dz xhat / n
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ part, int nblk, int C, float inv_n,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                              float* __restrict__ m1, float* __restrict__ m2) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    float s0, s1;
    bn_block_sums(part, nblk, C, c, s0, s1);
    if (threadIdx.x >= 32 || c >= C) return;
    dbeta[c] = s0;
    dgamma[c] = s1;
    m1[c] = s0 * inv_n;
    m2[c] = s1 * inv_n;
}


// ---- SyncBatchNorm over several statistics sets with ONE exchange (superglue.py:70-79 called once per image, train.py:338):
// the block sums of all `sets` calls are packed into one buffer -- [sets][2][C] sums, then the `sets` row counts -- which the
// host all-reduces across ranks ONCE; the finalize kernels below read the reduced sums and the reduced counts from it.
__global__ __launch_bounds__(256) void bn_pack_sums_kernel(const float* __restrict__ part, int sets, int nblk, int C, float n_local,
                                                           float* __restrict__ packed, float* __restrict__ local) {
    const int h = blockIdx.y;
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    float s0, s1;
    bn_block_sums(part + (size_t)h * nblk * 2 * C, nblk, C, c, s0, s1);
    if (threadIdx.x == 0 && blockIdx.x == 0) packed[(size_t)sets * 2 * C + h] = n_local;
    if (threadIdx.x >= 32 || c >= C) return;
    packed[(size_t)(2 * h) * C + c] = s0;
    packed[(size_t)(2 * h + 1) * C + c] = s1;
    if (local) {
        local[(size_t)(2 * h) * C + c] = s0;
        local[(size_t)(2 * h + 1) * C + c] = s1;
    }
}
// mean / biased var / rstd of every set from the (reduced) packed sums and counts; the running statistics take the sets'
// updates one after the other, as `sets` consecutive module calls would apply them
__global__ __launch_bounds__(256) void bn_finalize_sets_fwd_kernel(const float* __restrict__ packed, int sets, int C, float eps,
                                                                   float momentum, float* __restrict__ mvr,
                                                                   float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float rm = run_mean ? run_mean[c] : 0.f, rv = run_var ? run_var[c] : 0.f;
    for (int h = 0; h < sets; ++h) {
        const float n = packed[(size_t)sets * 2 * C + h];
        const float m = packed[(size_t)(2 * h) * C + c] / n;
        const float v = fmaxf(packed[(size_t)(2 * h + 1) * C + c] / n - m * m, 0.f);
        mvr[(size_t)(3 * h) * C + c] = m;
        mvr[(size_t)(3 * h + 1) * C + c] = v;
        mvr[(size_t)(3 * h + 2) * C + c] = rsqrtf(v + eps);
        rm = rm * (1.f - momentum) + m * momentum;
        rv = rv * (1.f - momentum) + v * (n / fmaxf(n - 1.f, 1.f)) * momentum;
    }
    if (run_mean) {
        run_mean[c] = rm;
        run_var[c] = rv;
    }
}
// m1 = sum dz / n, m2 = sum dz xhat / n of every set from the reduced packed sums and the forward's (global) counts
__global__ __launch_bounds__(256) void bn_finalize_sets_bwd_kernel(const float* __restrict__ packed, const float* __restrict__ counts,
                                                                   int sets, int C, float* __restrict__ m12) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    for (int h = 0; h < sets; ++h) {
        const float inv_n = 1.f / counts[h];
        m12[(size_t)(2 * h) * C + c] = packed[(size_t)(2 * h) * C + c] * inv_n;
        m12[(size_t)(2 * h + 1) * C + c] = packed[(size_t)(2 * h + 1) * C + c] * inv_n;
    }
}
// bn_replay_running_kernel with the row count of every set on the device (the global count of a SyncBatchNorm call)
__global__ __launch_bounds__(256) void bn_replay_running_n_kernel(const float* __restrict__ mvr, const float* __restrict__ counts, int sets,
                                                                  int C, float momentum, float* __restrict__ run_mean,
                                                                  float* __restrict__ run_var, const float* __restrict__ skip) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C || (skip != nullptr && !(*skip == 0.f))) return;
    float rm = run_mean[c], rv = run_var[c];
    for (int h = 0; h < sets; ++h) {
        const float n = counts[h];
        rm = rm * (1.f - momentum) + mvr[(size_t)(3 * h) * C + c] * momentum;
        rv = rv * (1.f - momentum) + mvr[(size_t)(3 * h + 1) * C + c] * (n / fmaxf(n - 1.f, 1.f)) * momentum;
    }
    run_mean[c] = rm;
    run_var[c] = rv;
}

}  // namespace

extern "C" int gf_bn_nblk(int M) { return bn_blocks(M); }

extern "C" int gf_bn_pack_sums(const float* part, int sets, int nblk, int C, float n_local, float* packed, float* local_copy,
                               void* stream) {
    if (sets <= 0 || sets > 65535 || nblk <= 0 || C <= 0 || n_local < 0.f || part == nullptr || packed == nullptr) return GF_ERR_SHAPE;
    bn_pack_sums_kernel<<<dim3((C + 31) / 32, sets), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        part, sets, nblk, C, n_local, packed, local_copy);
    return (int)hipGetLastError();
}

extern "C" int gf_bn_finalize_sets_fwd(const float* packed, int sets, int C, float eps, float momentum, float* mvr,
                                       float* run_mean, float* run_var, void* stream) {
    if (sets <= 0 || C <= 0 || packed == nullptr || mvr == nullptr) return GF_ERR_SHAPE;
    if ((run_mean == nullptr) != (run_var == nullptr)) return GF_ERR_SHAPE;
    bn_finalize_sets_fwd_kernel<<<dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        packed, sets, C, eps, momentum, mvr, run_mean, run_var);
    return (int)hipGetLastError();
}

extern "C" int gf_bn_finalize_sets_bwd(const float* packed, const float* counts, int sets, int C, float* m12, void* stream) {
    if (sets <= 0 || C <= 0 || packed == nullptr || counts == nullptr || m12 == nullptr) return GF_ERR_SHAPE;
    bn_finalize_sets_bwd_kernel<<<dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        packed, counts, sets, C, m12);
    return (int)hipGetLastError();
}

extern "C" int gf_bn_replay_running_n(const float* mvr, const float* counts, int sets, int C, float momentum, float* run_mean,
                                      float* run_var, const float* skip, void* stream) {
    if (sets <= 0 || C <= 0 || mvr == nullptr || counts == nullptr || run_mean == nullptr || run_var == nullptr) return GF_ERR_SHAPE;
    bn_replay_running_n_kernel<<<dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        mvr, counts, sets, C, momentum, run_mean, run_var, skip);
    return (int)hipGetLastError();
}

extern "C" int gf_bn_stats(const void* x, float* part, int M, int C, int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_colsum<float, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, part, M, C, 0, st);
    if (dtype == GF_BF16) return launch_colsum<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, part, M, C, 0, st);
    return GF_ERR_DTYPE;
}

extern "C" int gf_bn_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, void* y, int M, int C, int relu, int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_apply<float, 0>(x, nullptr, mean, rstd, gamma, beta, nullptr, nullptr, y, M, C, relu, st);
    if (dtype == GF_BF16) return launch_apply<bf16_t, 0>(x, nullptr, mean, rstd, gamma, beta, nullptr, nullptr, y, M, C, relu, st);
    return GF_ERR_DTYPE;
}

extern "C" int gf_bn_bwd_stats(const void* x, const void* dy, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, float* part, int M, int C, int relu,
                               int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_colsum<float, 1>(x, dy, mean, rstd, gamma, beta, part, M, C, relu, st);
    if (dtype == GF_BF16) return launch_colsum<bf16_t, 1>(x, dy, mean, rstd, gamma, beta, part, M, C, relu, st);
    return GF_ERR_DTYPE;
}

extern "C" int gf_bn_bwd_dx(const void* x, const void* dy, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, const float* m1, const float* m2,
                            void* dx, int M, int C, int relu, int dtype, void* stream) {
    if (M <= 0 || C <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_apply<float, 1>(x, dy, mean, rstd, gamma, beta, m1, m2, dx, M, C, relu, st);
    if (dtype == GF_BF16) return launch_apply<bf16_t, 1>(x, dy, mean, rstd, gamma, beta, m1, m2, dx, M, C, relu, st);
    return GF_ERR_DTYPE;
}

extern "C" int gf_bn_finalize_fwd(const float* part, int nblk, int C, float n, float eps, float momentum,
                                  float* mean, float* var, float* rstd, float* run_mean, float* run_var, void* stream) {
    if (nblk <= 0 || C <= 0 || n <= 0.f) return GF_ERR_SHAPE;
    if ((run_mean == nullptr) != (run_var == nullptr)) return GF_ERR_SHAPE;
    bn_finalize_fwd_kernel<<<dim3((C + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        part, nblk, C, n, eps, momentum, mean, var, rstd, run_mean, run_var);
    return (int)hipGetLastError();
}

extern "C" int gf_bn_replay_running(const float* mvr, int sets, int C, float n, float momentum, float* run_mean,
                                    float* run_var, const float* skip, void* stream) {
    if (sets <= 0 || C <= 0 || n <= 0.f || mvr == nullptr || run_mean == nullptr || run_var == nullptr) return GF_ERR_SHAPE;
    bn_replay_running_kernel<<<dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        mvr, sets, C, n, momentum, run_mean, run_var, skip);
    return (int)hipGetLastError();
}

extern "C" int gf_bn_finalize_bwd(const float* part, int nblk, int C, float n, float* dbeta, float* dgamma,
                                  float* m1, float* m2, void* stream) {
    if (nblk <= 0 || C <= 0 || n <= 0.f) return GF_ERR_SHAPE;
    bn_finalize_bwd_kernel<<<dim3((C + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        part, nblk, C, 1.f / n, dbeta, dgamma, m1, m2);
    return (int)hipGetLastError();
}
