// Small batched GEMM with arbitrary element strides (so transposed operands need no copy):
//     C[b][i][j] = alpha * sum_k A[b][i][k] B[b][k][j]        fp32 accumulation, exact-fp32 MFMA for fp32 operands.
// Used where the path multiplies two ACTIVATION tensors outside the fused kernels:
//   * the GlueStick line head (gluestick.py:336-376): endpoint scores G0 G1^T and their gradients dS G1, dS^T G0;
//   * the fp32 parity mode of the assignment-head backward (lightglue.py:256-290 autograd): d md0 = dS md1,
//     d md1 = dS^T md0 -- so that mode has no library product either (bf16: the fused gf_head_bwd).
// These are <= 1 GFLOP per pair (1 % of a GlueStick step): one 64 x 64 tile per 4-wave workgroup, 16-deep k-steps
// staged through LDS as k-contiguous rows for both operands, whatever their memory orientation.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

struct BgemmParams {
    const void* a; const void* b; void* c;
    int batch, M, N, K;
    int64_t sab, sai, sak, sbb, sbk, sbj, scb, sci, scj;
    float alpha;
};

constexpr int BG_LD = 16 + 4;       // LDS row stride in elements: 16 k-values + pad (rows stay 16-byte aligned)

// tile of `rows` x 16 (row index r, k index k): element at base[r * sr + k * sk], zero outside [0,R) x [0,K)
template <typename T>
__device__ __forceinline__ void stage_tile(T* lds, const T* base, int64_t sr, int64_t sk, int r0, int R, int k0, int K) {
    // thread mapping follows the unit stride so that a wave touches whole cache lines
    if (sk == 1) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, k = e & 15;
            const bool ok = r0 + r < R && k0 + k < K;
            lds[r * BG_LD + k] = ok ? base[(int64_t)(r0 + r) * sr + (k0 + k)] : from_f32<T>(0.f);
        }
    } else {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int k = e >> 6, r = e & 63;
            const bool ok = r0 + r < R && k0 + k < K;
            lds[r * BG_LD + k] = ok ? base[(int64_t)(r0 + r) * sr + (int64_t)(k0 + k) * sk] : from_f32<T>(0.f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bgemm_kernel(BgemmParams p) {
    __shared__ __attribute__((aligned(16))) T As[64 * BG_LD];
    __shared__ __attribute__((aligned(16))) T Bs[64 * BG_LD];
    const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wi = wave >> 1, wj = wave & 1;
    const T* A = static_cast<const T*>(p.a) + b * p.sab;
    const T* B = static_cast<const T*>(p.b) + b * p.sbb;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        stage_tile<T>(As, A, p.sai, p.sak, i0, p.M, k0, p.K);
        stage_tile<T>(Bs, B, p.sbj, p.sbk, j0, p.N, k0, p.K);
        __syncthreads();
        mma32(acc, ld_frag8(As + (wi * 32 + l31) * BG_LD + 8 * hi), ld_frag8(Bs + (wj * 32 + l31) * BG_LD + 8 * hi));
        __syncthreads();
    }
    T* C = static_cast<T*>(p.c) + b * p.scb;
    const int j = j0 + wj * 32 + l31;
    if (j < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + wi * 32 + crow(r, hi);
            if (i < p.M) C[(int64_t)i * p.sci + (int64_t)j * p.scj] = from_f32<T>(acc[r] * p.alpha);
        }
    }
}

}  // namespace

extern "C" int gf_bgemm(const void* a, const void* b, void* c, int batch, int M, int N, int K,
                        const int64_t* a_strides, const int64_t* b_strides, const int64_t* c_strides,
                        float alpha, int dtype, void* stream) {
    if (batch <= 0 || M <= 0 || N <= 0 || K <= 0) return GF_ERR_SHAPE;
    if (batch > 65535) return GF_ERR_UNSUPPORTED;
    BgemmParams p;
    p.a = a; p.b = b; p.c = c; p.batch = batch; p.M = M; p.N = N; p.K = K; p.alpha = alpha;
    p.sab = a_strides[0]; p.sai = a_strides[1]; p.sak = a_strides[2];
    p.sbb = b_strides[0]; p.sbk = b_strides[1]; p.sbj = b_strides[2];
    p.scb = c_strides[0]; p.sci = c_strides[1]; p.scj = c_strides[2];
    const dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) bgemm_kernel<float><<<grid, dim3(256), 0, st>>>(p);
    else if (dtype == GF_BF16) bgemm_kernel<bf16_t><<<grid, dim3(256), 0, st>>>(p);
    else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
