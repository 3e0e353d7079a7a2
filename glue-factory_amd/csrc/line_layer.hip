// GlueStick line message passing (gluefactory/models/matchers/gluestick.py:589-691): every line endpoint e reads the
// descriptor of its junction idx[e] and of the junction at the other end of its line (e ^ 1), an MLP turns
// [own | other | endpoint encoding] into an update, and every junction takes the MEAN (or, with line attention,
// the weighted sum) of the updates of the endpoints that sit on it.  The reference does this with gather / flip /
// cat / scatter_reduce on [B, D, 2 Nl] tensors; here:
//   line_csr     per image: endpoints grouped by junction (stable counting sort: order[], seg[]), built ONCE per
//                forward -- the junction graph is the same for all line layers and for the backward;
//   line_gather  msg[e] = [x[idx[e]] | x[idx[e^1]] | enc[e]]   (the MLP's input rows, channels-last)
//   line_segsum  out[j] = base[j] + scale_j * sum_{e in seg(j)} (s0[e] + s1[e^1])   (deterministic segment sums: no
//                atomics).  It is the forward aggregation (s0 = updates, scale = 1/count or 1, base = x: the
//                residual add is fused) and the backward of the gather (s0 = d msg[:, :D], s1 = d msg[:, D:2D]);
//   line_expand  d upd[e] = scale_{idx[e]} * g[idx[e]]          (backward of the aggregation)
// All are O(E D) streaming kernels on [B, E <= 4096, D] tensors; T = float or bf16 (fp32 accumulation).
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one workgroup per image; E <= 4096 endpoints, N <= 8192 junctions (LDS: 4 E + 4 (N + 1) bytes)
__global__ __launch_bounds__(256) void line_csr_kernel(const int64_t* __restrict__ idx, int* __restrict__ order,
                                                       int* __restrict__ seg, int E, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* key = reinterpret_cast<int*>(smem);            // [E] junction of endpoint e
    int* cnt = key + E;                                 // [N + 1]
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j <= N; j += 256) cnt[j] = 0;
    for (int e = threadIdx.x; e < E; e += 256) key[e] = (int)idx[(size_t)b * E + e];
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) atomicAdd(&cnt[key[e]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {                             // exclusive scan (N <= 8192: a few microseconds, once per forward)
        int run = 0;
        for (int j = 0; j <= N; ++j) { const int c = cnt[j]; cnt[j] = run; run += c; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j <= N; j += 256) seg[(size_t)b * (N + 1) + j] = cnt[j];
    // stable position inside the segment: number of earlier endpoints on the same junction
    for (int e = threadIdx.x; e < E; e += 256) {
        const int k = key[e];
        int rank = 0;
        for (int f = 0; f < e; ++f) rank += key[f] == k;
        order[(size_t)b * E + cnt[k] + rank] = e;
    }
}

// grid (E, B); D / VEC threads: msg[b, e, :] = [x[b, idx[e]] | x[b, idx[e ^ 1]] | enc[b, e]]
template <typename T>
__global__ void line_gather_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx, const T* __restrict__ enc,
                                   T* __restrict__ msg, int E, int N, int D) {
    constexpr int VEC = 16 / sizeof(T);
    const int e = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x * VEC;
    if (c >= D) return;
    const int j0 = (int)idx[(size_t)b * E + e], j1 = (int)idx[(size_t)b * E + (e ^ 1)];
    T* out = msg + ((size_t)b * E + e) * 3 * D;
    *reinterpret_cast<u32x4*>(out + c) = *reinterpret_cast<const u32x4*>(x + ((size_t)b * N + j0) * D + c);
    *reinterpret_cast<u32x4*>(out + D + c) = *reinterpret_cast<const u32x4*>(x + ((size_t)b * N + j1) * D + c);
    *reinterpret_cast<u32x4*>(out + 2 * D + c) = *reinterpret_cast<const u32x4*>(enc + ((size_t)b * E + e) * D + c);
}

// grid (N, B); D / 4 threads (4 channels each): out[b, j] = base[b, j] + scale * sum_{e in seg} (s0[e] + s1[e ^ 1])
// mode: 0 sum, 1 mean (scale = 1 / count)
template <typename T>
__global__ void line_segsum_kernel(const T* __restrict__ s0, int64_t ld0, const T* __restrict__ s1, int64_t ld1,
                                   const int* __restrict__ order, const int* __restrict__ seg,
                                   const T* __restrict__ base, T* __restrict__ out, int E, int N, int D, int mode) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x * 4;
    if (c >= D) return;
    const int lo = seg[(size_t)b * (N + 1) + j], hi = seg[(size_t)b * (N + 1) + j + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lo; k < hi; ++k) {
        const int e = order[(size_t)b * E + k];
        const T* p0 = s0 + ((size_t)b * E + e) * ld0 + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += to_f32(p0[i]);
        if (s1) {
            const T* p1 = s1 + ((size_t)b * E + (e ^ 1)) * ld1 + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += to_f32(p1[i]);
        }
    }
    const float scale = (mode == 1 && hi > lo) ? 1.f / (float)(hi - lo) : 1.f;
    const size_t o = ((size_t)b * N + j) * D + c;
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = acc[i] * scale + (base ? to_f32(base[o + i]) : 0.f);
    st4(out + o, r[0], r[1], r[2], r[3]);
}

// grid (E, B); D / 4 threads: d[b, e] = scale_{idx[e]} * g[b, idx[e]]
template <typename T>
__global__ void line_expand_kernel(const T* __restrict__ g, const int64_t* __restrict__ idx, const int* __restrict__ seg,
                                   T* __restrict__ d, int E, int N, int D, int mode) {
    const int e = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x * 4;
    if (c >= D) return;
    const int j = (int)idx[(size_t)b * E + e];
    const int cnt = seg[(size_t)b * (N + 1) + j + 1] - seg[(size_t)b * (N + 1) + j];
    const float scale = mode == 1 ? 1.f / (float)max(cnt, 1) : 1.f;
    const T* p = g + ((size_t)b * N + j) * D + c;
    st4(d + ((size_t)b * E + e) * D + c, to_f32(p[0]) * scale, to_f32(p[1]) * scale, to_f32(p[2]) * scale,
        to_f32(p[3]) * scale);
}

}  // namespace

extern "C" int gf_line_csr(const int64_t* idx, int* order, int* seg, int B, int E, int N, void* stream) {
    if (B <= 0 || E <= 0 || N <= 0) return GF_ERR_SHAPE;
    if (E > 4096 || N > 8192 || (E & 1)) return GF_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(E + N + 1) * 4;
    line_csr_kernel<<<dim3(B), 256, lds, reinterpret_cast<hipStream_t>(stream)>>>(idx, order, seg, E, N);
    return (int)hipGetLastError();
}

extern "C" int gf_line_gather(const void* x, const int64_t* idx, const void* enc, void* msg, int B, int E, int N, int D,
                              int dtype, void* stream) {
    if (B <= 0 || E <= 0 || N <= 0 || D <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_BF16) {
        if (D % 8 || D > 8 * 1024) return GF_ERR_UNSUPPORTED;
        line_gather_kernel<bf16_t><<<dim3(E, B), D / 8, 0, st>>>(static_cast<const bf16_t*>(x), idx, static_cast<const bf16_t*>(enc),
                                                               static_cast<bf16_t*>(msg), E, N, D);
    } else if (dtype == GF_F32) {
        if (D % 4 || D > 4 * 1024) return GF_ERR_UNSUPPORTED;
        line_gather_kernel<float><<<dim3(E, B), D / 4, 0, st>>>(static_cast<const float*>(x), idx, static_cast<const float*>(enc),
                                                              static_cast<float*>(msg), E, N, D);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_line_segsum(const void* s0, int64_t ld0, const void* s1, int64_t ld1, const int* order, const int* seg,
                              const void* base, void* out, int B, int E, int N, int D, int mode, int dtype, void* stream) {
    if (B <= 0 || E <= 0 || N <= 0 || D <= 0) return GF_ERR_SHAPE;
    if (D % 4 || D > 4096 || ld0 % 4 || (s1 && ld1 % 4)) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_BF16)
        line_segsum_kernel<bf16_t><<<dim3(N, B), D / 4, 0, st>>>(static_cast<const bf16_t*>(s0), ld0, static_cast<const bf16_t*>(s1), ld1,
                                                               order, seg, static_cast<const bf16_t*>(base),
                                                               static_cast<bf16_t*>(out), E, N, D, mode);
    else if (dtype == GF_F32)
        line_segsum_kernel<float><<<dim3(N, B), D / 4, 0, st>>>(static_cast<const float*>(s0), ld0, static_cast<const float*>(s1), ld1,
                                                              order, seg, static_cast<const float*>(base),
                                                              static_cast<float*>(out), E, N, D, mode);
    else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_line_expand(const void* g, const int64_t* idx, const int* seg, void* d, int B, int E, int N, int D,
                              int mode, int dtype, void* stream) {
    if (B <= 0 || E <= 0 || N <= 0 || D <= 0) return GF_ERR_SHAPE;
    if (D % 4 || D > 4096) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_BF16)
        line_expand_kernel<bf16_t><<<dim3(E, B), D / 4, 0, st>>>(static_cast<const bf16_t*>(g), idx, seg, static_cast<bf16_t*>(d), E, N, D, mode);
    else if (dtype == GF_F32)
        line_expand_kernel<float><<<dim3(E, B), D / 4, 0, st>>>(static_cast<const float*>(g), idx, seg, static_cast<float*>(d), E, N, D, mode);
    else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
