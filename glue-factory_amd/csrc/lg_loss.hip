// Fused deep-supervision loss of LightGlue (reference: lightglue.py:598-657 `loss`, utils/losses.py:6-73
// NLL with dustbins, lightglue.py:81-94 TokenConfidence.loss) on the statistics of one layer's assignment
// head, WITHOUT the dense log-assignment matrix and without the ~60 tiny tensor ops per layer the same
// math costs as host-side tensor algebra:
//   A_ij = 2 md0_i.md1_j - r_i - c_j + lz0_i + lz1_j   (lz = logsigmoid(z)),  A_i,N = logsigmoid(-z0_i), ...
//   acc[b] = { sum_{(i,j) positive} A_ij,  sum_i neg0_i A_i,N + sum_j neg1_j A_M,j,
//              sum_i bce(t0_i, tgt0_i),    sum_j bce(t1_j, tgt1_j) }
// with tgt = (layer arg-max incl. dustbin == final arg-max).  The caller divides by the counts and
// mixes the layers (a handful of [L,B] tensor ops for the whole step).
// Backward: per-token gradients (dz, dt) are dense one-pass kernels; positives add their sparse terms
// with atomics (any COO list is accepted, duplicates included).
#include "gf_common.h"
#include "gf_amd.h"

namespace {

__device__ __forceinline__ float logsig(float x) { return fminf(x, 0.f) - log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one wave per chunk of POS_CHUNK consecutive positives, FOUR at a time (one per 16-lane row: each lane 4-element slices
// of the two descriptor rows, the dot product through the row's DPP ladder): the index -> address -> row -> reduce chain
// of a positive is ~1.5 us of latency, 16 of them in sequence per wave made this kernel 47 us for 33 MB of reads.  The
// per-image sums stay in registers while the batch index stays the same (the lists are sorted by it) -> a few atomics
// per wave.
constexpr int POS_CHUNK = 16;
template <typename T>
__global__ __launch_bounds__(256) void loss_pos_fwd_kernel(const T* __restrict__ md0, const T* __restrict__ md1,
                                                           const float* __restrict__ z0, const float* __restrict__ z1,
                                                           const float* __restrict__ r, const float* __restrict__ c,
                                                           const int64_t* __restrict__ pb, const int64_t* __restrict__ pi,
                                                           const int64_t* __restrict__ pj, int64_t P,
                                                           float* __restrict__ acc, int M, int N, int D) {
    const int lane = threadIdx.x & 63, row = lane >> 4, gl = lane & 15;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * POS_CHUNK;
    int64_t cur = -1;
    float run = 0.f;
#pragma unroll
    for (int it = 0; it < POS_CHUNK / 4; ++it) {
        const int64_t p = p0 + it * 4 + row;                  // (rows walk the chunk interleaved: still ascending per row)
        const bool live = p < P;
        const int64_t b = live ? pb[p] : 0, i = live ? pi[p] : 0;
        int64_t j = live ? pj[p] : -1;
        const bool has = j >= 0;                              // fixed-length list: -1 = "row has no positive"
        j = has ? j : 0;
        const T* a = md0 + (b * M + i) * D;
        const T* q = md1 + (b * N + j) * D;
        float dot = 0.f;
        for (int d = gl * 4; d < D; d += 64) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dot += to_f32(a[d + e]) * to_f32(q[d + e]);
        }
        dot = row16_allsum(dot);
        const float v = 2.f * dot - r[b * M + i] - c[b * N + j] + logsig(z0[b * M + i]) + logsig(z1[b * N + j]);
        if (has) {
            if (b != cur) {
                if (cur >= 0 && gl == 0) atomicAdd(acc + 4 * cur, run);
                cur = b;
                run = 0.f;
            }
            run += v;
        }
    }
    // the four rows of a wave almost always end on the same image: one atomic instead of four
    const int c0 = __builtin_amdgcn_readlane((int)cur, 0), c1 = __builtin_amdgcn_readlane((int)cur, 16);
    const int c2 = __builtin_amdgcn_readlane((int)cur, 32), c3 = __builtin_amdgcn_readlane((int)cur, 48);
    if (c0 >= 0 && c0 == c1 && c0 == c2 && c0 == c3) {
        const int rb = __builtin_bit_cast(int, run);
        const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rb, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(rb, 16)) +
                          __builtin_bit_cast(float, __builtin_amdgcn_readlane(rb, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(rb, 48));
        if (lane == 0) atomicAdd(acc + 4 * c0, tot);
    } else if (cur >= 0 && gl == 0) {
        atomicAdd(acc + 4 * cur, run);
    }
}

// one block per (image, batch element): dustbin NLL terms and token-confidence BCE of that image's tokens
__global__ __launch_bounds__(256) void loss_tok_fwd_kernel(const float* __restrict__ z0, const float* __restrict__ z1,
                                                           const float* __restrict__ r, const float* __restrict__ c,
                                                           const float* __restrict__ neg0, const float* __restrict__ neg1,
                                                           const float* __restrict__ t0, const float* __restrict__ t1,
                                                           const float* __restrict__ v0, const int64_t* __restrict__ a0,
                                                           const float* __restrict__ v1, const int64_t* __restrict__ a1,
                                                           const int64_t* __restrict__ fin0, const int64_t* __restrict__ fin1,
                                                           float* __restrict__ tgt0, float* __restrict__ tgt1,
                                                           float* __restrict__ acc, int B, int M, int N) {
    const int img = blockIdx.x / B, b = blockIdx.x % B;
    const int n = img ? N : M, other = img ? M : N;
    const int64_t base = (int64_t)b * n;
    const float* z = (img ? z1 : z0) + base;
    const float* nrm = (img ? c : r) + base;
    const float* neg = (img ? neg1 : neg0) + base;
    const float* t = img ? t1 : t0;
    const float* v = img ? v1 : v0;
    const int64_t* a = img ? a1 : a0;
    const int64_t* fin = img ? fin1 : fin0;
    float* tgt = img ? tgt1 : tgt0;
    float sneg = 0.f, sbce = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float zz = z[i];
        const float lz = logsig(zz), bin = lz - zz;          // logsigmoid(-z) = logsigmoid(z) - z
        sneg += bin * neg[i];
        if (t != nullptr) {
            const float mx = v[base + i] - nrm[i] + lz;      // row maximum of the core of the log assignment
            const int64_t full = bin > mx ? (int64_t)other : a[base + i];
            const float y = full == fin[base + i] ? 1.f : 0.f;
            tgt[base + i] = y;
            const float tt = t[base + i];
            sbce += fmaxf(tt, 0.f) - tt * y + log1pf(__expf(-fabsf(tt)));
        }
    }
    __shared__ float red[2][4];
    sneg = wave_sum(sneg);
    sbce = wave_sum(sbce);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sneg; red[1][wave] = sbce; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc + 4 * b + 1, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        if (t != nullptr) acc[4 * b + 2 + img] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// dense per-token gradients; also clears gr / gc, which the positives then accumulate into
__global__ __launch_bounds__(256) void loss_tok_bwd_kernel(const float* __restrict__ z0, const float* __restrict__ z1,
                                                           const float* __restrict__ neg0, const float* __restrict__ neg1,
                                                           const float* __restrict__ t0, const float* __restrict__ t1,
                                                           const float* __restrict__ tgt0, const float* __restrict__ tgt1,
                                                           const float* __restrict__ gacc,
                                                           float* __restrict__ dz0, float* __restrict__ dz1,
                                                           float* __restrict__ dt0, float* __restrict__ dt1,
                                                           float* __restrict__ gr, float* __restrict__ gc,
                                                           int B, int M, int N) {
    const int img = blockIdx.x / B, b = blockIdx.x % B;
    const int n = img ? N : M;
    const int64_t base = (int64_t)b * n;
    const float* z = (img ? z1 : z0) + base;
    const float* neg = (img ? neg1 : neg0) + base;
    const float* t = img ? t1 : t0;
    const float* tgt = img ? tgt1 : tgt0;
    float* dz = (img ? dz1 : dz0) + base;
    float* dt = img ? dt1 : dt0;
    float* gn = (img ? gc : gr) + base;
    const float gneg = gacc[4 * b + 1], gbce = gacc[4 * b + 2 + img];
    for (int i = threadIdx.x; i < n; i += 256) {
        dz[i] = -gneg * neg[i] * sigm(z[i]);                 // d logsigmoid(-z) / dz = -sigmoid(z)
        gn[i] = 0.f;
        if (t != nullptr) dt[base + i] = gbce * (sigm(t[base + i]) - tgt[base + i]);
    }
}

__global__ __launch_bounds__(256) void loss_pos_bwd_scalar_kernel(const float* __restrict__ z0, const float* __restrict__ z1,
                                                                  const int64_t* __restrict__ pb, const int64_t* __restrict__ pi,
                                                                  const int64_t* __restrict__ pj, int64_t P,
                                                                  const float* __restrict__ gacc,
                                                                  float* __restrict__ dz0, float* __restrict__ dz1,
                                                                  float* __restrict__ gr, float* __restrict__ gc, int M, int N) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    if (pj[p] < 0) return;
    const int64_t b = pb[p], f0 = b * M + pi[p], f1 = b * N + pj[p];
    const float g = gacc[4 * b];
    atomicAdd(gr + f0, -g);
    atomicAdd(gc + f1, -g);
    atomicAdd(dz0 + f0, g * sigm(-z0[f0]));                  // d logsigmoid(z) / dz = sigmoid(-z)
    atomicAdd(dz1 + f1, g * sigm(-z1[f1]));
}

__device__ __forceinline__ void atomic_add2(float* p, float a, float b) {
    atomicAdd(p, a);
    atomicAdd(p + 1, b);
}
__device__ __forceinline__ void atomic_add2(bf16_t* p, float a, float b) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(bf16_t)a, (bf16_t)b};
    __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) s16x2*)p, __builtin_bit_cast(s16x2, v));
}

// dmd0[b,i,:] += 2 g md1[b,j,:],  dmd1[b,j,:] += 2 g md0[b,i,:]   (one wave per positive)
template <typename T>
__global__ __launch_bounds__(256) void loss_pos_bwd_rows_kernel(const T* __restrict__ md0, const T* __restrict__ md1,
                                                                const int64_t* __restrict__ pb, const int64_t* __restrict__ pi,
                                                                const int64_t* __restrict__ pj, int64_t P,
                                                                const float* __restrict__ gacc, T* __restrict__ dmd0,
                                                                T* __restrict__ dmd1, int M, int N, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P || pj[p] < 0) return;
    const int64_t b = pb[p], o0 = (b * M + pi[p]) * D, o1 = (b * N + pj[p]) * D;
    const float g2 = 2.f * gacc[4 * b];
    for (int d = lane * 2; d < D; d += 128) {
        atomic_add2(dmd0 + o0 + d, g2 * to_f32(md1[o1 + d]), g2 * to_f32(md1[o1 + d + 1]));
        atomic_add2(dmd1 + o1 + d, g2 * to_f32(md0[o0 + d]), g2 * to_f32(md0[o0 + d + 1]));
    }
}

}  // namespace

extern "C" int gf_lg_loss_fwd(const void* md0, const void* md1, const float* z0, const float* z1,
                              const float* r, const float* c,
                              const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int64_t P,
                              const float* neg0, const float* neg1,
                              const float* t0, const float* t1,
                              const float* v0, const int64_t* a0, const float* v1, const int64_t* a1,
                              const int64_t* fin0, const int64_t* fin1,
                              float* tgt0, float* tgt1, float* acc,
                              int B, int M, int N, int D, int dtype, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || P < 0) return GF_ERR_SHAPE;
    if (D <= 0 || D % 4) return GF_ERR_ALIGN;
    if (dtype != GF_F32 && dtype != GF_BF16) return GF_ERR_DTYPE;
    if ((t0 == nullptr) != (t1 == nullptr)) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipError_t e = gf_zero_f32(acc, (size_t)4 * B, st)) return (int)e;
    if (P > 0) {
        const dim3 grid((unsigned)((P + 4 * POS_CHUNK - 1) / (4 * POS_CHUNK)));
        if (dtype == GF_BF16)
            loss_pos_fwd_kernel<bf16_t><<<grid, dim3(256), 0, st>>>(
                static_cast<const bf16_t*>(md0), static_cast<const bf16_t*>(md1), z0, z1, r, c, pos_b, pos_i, pos_j, P,
                acc, M, N, D);
        else
            loss_pos_fwd_kernel<float><<<grid, dim3(256), 0, st>>>(
                static_cast<const float*>(md0), static_cast<const float*>(md1), z0, z1, r, c, pos_b, pos_i, pos_j, P,
                acc, M, N, D);
        if (int e = (int)hipGetLastError()) return e;
    }
    loss_tok_fwd_kernel<<<dim3(2 * B), dim3(256), 0, st>>>(z0, z1, r, c, neg0, neg1, t0, t1, v0, a0, v1, a1, fin0, fin1,
                                                           tgt0, tgt1, acc, B, M, N);
    return (int)hipGetLastError();
}

extern "C" int gf_lg_loss_bwd_tokens(const float* z0, const float* z1, const float* neg0, const float* neg1,
                                     const float* t0, const float* t1, const float* tgt0, const float* tgt1,
                                     const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int64_t P,
                                     const float* gacc, float* dz0, float* dz1, float* dt0, float* dt1,
                                     float* gr, float* gc, int B, int M, int N, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || P < 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    loss_tok_bwd_kernel<<<dim3(2 * B), dim3(256), 0, st>>>(z0, z1, neg0, neg1, t0, t1, tgt0, tgt1, gacc, dz0, dz1, dt0,
                                                           dt1, gr, gc, B, M, N);
    if (int e = (int)hipGetLastError()) return e;
    if (P > 0) {
        loss_pos_bwd_scalar_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st>>>(
            z0, z1, pos_b, pos_i, pos_j, P, gacc, dz0, dz1, gr, gc, M, N);
        if (int e = (int)hipGetLastError()) return e;
    }
    return 0;
}

extern "C" int gf_lg_loss_bwd_rows(const void* md0, const void* md1,
                                   const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int64_t P,
                                   const float* gacc, void* dmd0, void* dmd1,
                                   int B, int M, int N, int D, int dtype, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || P < 0) return GF_ERR_SHAPE;
    if (D <= 0 || D % 4) return GF_ERR_ALIGN;
    if (P == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((P + 3) / 4));
    if (dtype == GF_BF16)
        loss_pos_bwd_rows_kernel<bf16_t><<<grid, dim3(256), 0, st>>>(
            static_cast<const bf16_t*>(md0), static_cast<const bf16_t*>(md1), pos_b, pos_i, pos_j, P, gacc,
            static_cast<bf16_t*>(dmd0), static_cast<bf16_t*>(dmd1), M, N, D);
    else if (dtype == GF_F32)
        loss_pos_bwd_rows_kernel<float><<<grid, dim3(256), 0, st>>>(
            static_cast<const float*>(md0), static_cast<const float*>(md1), pos_b, pos_i, pos_j, P, gacc,
            static_cast<float*>(dmd0), static_cast<float*>(dmd1), M, N, D);
    else
        return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
