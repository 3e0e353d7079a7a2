// y = x W^T + bias (+ residual) for the tall-and-skinny linear layers of the matcher blocks
// (reference: nn.Linear calls of gluefactory/models/matchers/lightglue.py:131-221,271-290; M = B*N tokens
// is 1e5, the weight is at most 768 x 512).  bf16 in / bf16 out, fp32 accumulation, fp32 bias.
// STATUS: numerically verified (tests/test_gpu_kernels.py) but 1.4x slower than the tuned library GEMM at these
// shapes (44 vs 31 us at 131072 x 256 x 256), so the host code keeps the library by default
// (GF_AMD_HIP_GEMM=1 opts in); kept as the base for a persistent, epilogue-fused version.
//
// One workgroup = 4 waves computes a 128 x 128 output tile; both operands are ROW reads of row-major
// matrices (x[m][k] and W[n][k] share the contraction index as their contiguous one), so no transposition
// is needed anywhere.  K is walked in chunks of 32: a stage holds 128 rows x 64 B of x and of W, written by
// LDS-DMA (global_load_lds_dwordx4) into a 4-stage ring (three chunks in flight, counted vmcnt, one raw
// barrier per chunk).  Rows are 64 B = four 16-byte chunks, XOR-swizzled by (row >> 2) & 3: conflict-free
// for ds_read_b128 of one chunk from 16 different rows.  The epilogue parks the fp32 tile in the ring's
// LDS and writes whole rows: + bias, + residual (the "x +" of the block, or the first half of the FFN's
// concatenated input), one rounding to bf16, 8-byte coalesced stores.
#include <type_traits>
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void g_lds_void;
typedef const __attribute__((address_space(1))) void g_glb_void;

constexpr int G_KC = 32;                       // k per chunk
constexpr int G_TILE = 128 * G_KC * 2;         // bytes of one operand tile (128 rows x 64 B)
constexpr int G_STAGE = 2 * G_TILE;
constexpr int G_NSTAGE = 4;

template <int OFF> __device__ __forceinline__ u32x4 g_rd128(unsigned a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void g_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename V> __device__ __forceinline__ void g_tie(V& v) { asm volatile("" : "+v"(v)); }

struct GemmParams {
    const bf16_t* x; const bf16_t* w; const float* bias; const bf16_t* res; bf16_t* y;
    int M, N, K;
    int64_t ldx, ldw, ldr, ldy;
};

template <int SO>
__device__ __forceinline__ void g_chunk(f32x16 (&acc)[2][2], unsigned aA, unsigned aB) {
    // [i or j][ks]: row block (+2048 B = 32 rows) and k-step (address bit 5)
    u32x4 fa[2][2], fb[2][2];
    fa[0][0] = g_rd128<SO>(aA);                 fa[0][1] = g_rd128<SO>(aA ^ 32u);
    fa[1][0] = g_rd128<SO + 2048>(aA);          fa[1][1] = g_rd128<SO + 2048>(aA ^ 32u);
    fb[0][0] = g_rd128<SO + G_TILE>(aB);        fb[0][1] = g_rd128<SO + G_TILE>(aB ^ 32u);
    fb[1][0] = g_rd128<SO + G_TILE + 2048>(aB); fb[1][1] = g_rd128<SO + G_TILE + 2048>(aB ^ 32u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) { g_tie(fa[i][k]); g_tie(fb[i][k]); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k)          // the two k-steps chained on one accumulator
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][k]),
                                                                    __builtin_bit_cast(bf16x8, fb[j][k]), acc[i][j], 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nt = p.N / 128, mt = (p.M + 127) / 128;
    const int id = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (id / nt) * 128, n0 = (id % nt) * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- DMA descriptors: pieces 2*wave, 2*wave+1 of each tile; a piece = 16 rows x 64 B
    const int prow = lane >> 2;                                     // row inside the piece
    const int plog = (lane & 3) ^ ((lane >> 4) & 3);                // logical 16-byte chunk this lane fetches
    auto issue = [&](int kc, int stage) {
        char* sb = smem + stage * G_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = 2 * wave + i;
            const int row = 16 * piece + prow;
            const int64_t xr = min(m0 + row, p.M - 1);
            __builtin_amdgcn_global_load_lds((g_glb_void*)(p.x + xr * p.ldx + kc * G_KC + plog * 8),
                                             (g_lds_void*)(sb + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((g_glb_void*)(p.w + (int64_t)(n0 + row) * p.ldw + kc * G_KC + plog * 8),
                                             (g_lds_void*)(sb + G_TILE + piece * 1024), 16, 0, 0);
        }
    };
    const int nchunk = p.K / G_KC;
#pragma unroll
    for (int c = 0; c < G_NSTAGE - 1; ++c)
        if (c < nchunk) issue(c, c);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane read addresses (stage 0, k-step 0, first row block): row r, logical chunk hi, swizzle (r >> 2) & 3
    auto rd_addr = [&](int r) { return lds0 + (unsigned)(r * 64 + ((hi ^ ((r >> 2) & 3)) * 16)); };
    const unsigned aA = rd_addr(64 * wm + l31), aB = rd_addr(64 * wn + l31);

    auto step = [&](int c, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        if (c + 1 >= nchunk) g_wait_vm<0>();                      // chunk c landed (this wave's pieces)
        else if (c + 2 >= nchunk) g_wait_vm<4>();
        else g_wait_vm<8>();
        __builtin_amdgcn_s_barrier();                             // ... everyone's; the stage of chunk c-1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (c + G_NSTAGE - 1 < nchunk) issue(c + G_NSTAGE - 1, (ST + G_NSTAGE - 1) % G_NSTAGE);
        g_chunk<ST * G_STAGE>(acc, aA, aB);
    };
    for (int c = 0; c < nchunk; c += G_NSTAGE) {
        step(c, std::integral_constant<int, 0>{});
        if (c + 1 < nchunk) step(c + 1, std::integral_constant<int, 1>{});
        if (c + 2 < nchunk) step(c + 2, std::integral_constant<int, 2>{});
        if (c + 3 < nchunk) step(c + 3, std::integral_constant<int, 3>{});
    }

    // ---- epilogue: fp32 tile -> LDS (the ring is free after the barrier) -> whole rows out
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem);                    // [128][128] fp32 = 64 KB
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(64 * wm + 32 * i + crow(r, hi)) * 128 + 64 * wn + 32 * j + l31] = acc[i][j][r];
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int idx = threadIdx.x + 256 * it;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        const int m = m0 + row;
        if (m < p.M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(ct + row * 128 + c4);
            if (p.bias) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c4);
                v += b4;
            }
            if (p.res) {
                const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(p.res + (int64_t)m * p.ldr + n0 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
            }
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(p.y + (int64_t)m * p.ldy + n0 + c4) = o;
        }
    }
}

}  // namespace

extern "C" int gf_linear_fwd(const void* x, const void* w, const float* bias, const void* res, void* y,
                             int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldr, int64_t ldy, int dtype,
                             void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return GF_ERR_SHAPE;
    if (dtype != GF_BF16) return GF_ERR_DTYPE;
    if (N % 128 || K % G_KC) return GF_ERR_UNSUPPORTED;
    if (ldx % 8 || ldw % 8 || ldy % 4 || (res && ldr % 4)) return GF_ERR_ALIGN;
    GemmParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(w); p.bias = bias;
    p.res = static_cast<const bf16_t*>(res); p.y = static_cast<bf16_t*>(y);
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldr = ldr; p.ldy = ldy;
    const size_t lds = (size_t)G_NSTAGE * G_STAGE;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int total = ((M + 127) / 128) * (N / 128);
    gemm_nt_kernel<<<dim3(total), 256, lds, reinterpret_cast<hipStream_t>(stream)>>>(p);
    return (int)hipGetLastError();
}
