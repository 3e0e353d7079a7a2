// Batched GEMM with arbitrary element strides (so transposed operands need no copy):
//     C[b][i][j] = alpha * sum_k A[b][i][k] B[b][k][j]        fp32 accumulation, exact-fp32 MFMA for fp32 operands.
// Used where the path multiplies two ACTIVATION tensors outside the fused kernels:
//   * the GlueStick line head (gluestick.py:336-376): endpoint scores G0 G1^T and their gradients dS G1, dS^T G0;
//   * the dense-gradient backward of the materialised log assignment (SuperGlue's Sinkhorn / GlueStick's double
//     softmax differentiate THROUGH the matrix: d md0 = G md1, d md1 = G^T md0; superglue.py:283-289, gluestick.py:336-347);
//   * the fp32 parity mode of the LightGlue assignment-head backward (lightglue.py:256-290 autograd; bf16: gf_head_bwd).
// One 128 x 128 output tile per 4-wave workgroup (each wave 64 x 64 = 2 x 2 MFMA tiles), k-steps of 128 bytes staged
// through LDS in the operand's OWN orientation (16-byte loads and stores either way; a row-contiguous operand is
// transposed by the fragment reads: ds_read_b64_tr_b16 / 32-bit reads); the global loads of step t+1 are in flight
// (registers) while step t is multiplied.  Scalar guarded loads on ragged edges and for operands without a unit stride.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

struct BgemmParams {
    const void* a; const void* b; void* c;
    int batch, M, N, K;
    int64_t sab, sai, sak, sbb, sbk, sbj, scb, sci, scj;
    float alpha;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T> struct BG {
    static constexpr int VEC = 16 / sizeof(T);           // elements per 16-byte chunk
    static constexpr int BK = 128 / sizeof(T);           // k-step: 128-byte row segments (bf16 64, fp32 32)
    static constexpr int LDK = BK + VEC;                 // k-contiguous image [128 rows][LDK]
    static constexpr int LDR = sizeof(T) == 2 ? 144 : 132;
                                                         // row-contiguous image [BK k][LDR]: bf16 144 (288 B = 8 banks
                                                         // past a bank row: the 4 k-rows of a transposing read do
                                                         // not collide), fp32 132
    static constexpr int ITEMS = 128 * BK / VEC / 256;   // 16-byte chunks per thread and operand tile (4)
    static constexpr int TILE = 128 * LDK > BK * LDR ? 128 * LDK : BK * LDR;    // elements of LDS per operand
};

// Operand tile: 128 rows x BK k; element (r, k) at base[r * sr + k * sk]; zero outside [0, R) x [0, K).
// KC: the operand is k-contiguous in memory (or has no unit stride at all: scalar loads), chunks run along k and the
// LDS image is [row][k]; !KC: row-contiguous, chunks run along rows, image [k][row] (no transposition on the way in:
// the fragment reads transpose -- ds_read_b64_tr_b16 for bf16, plain 32-bit reads for fp32).
template <typename T>
struct TileRegs { u32x4 v[BG<T>::ITEMS]; };

template <typename T, bool KC>
__device__ __forceinline__ void chunk_pos(int it, int& r, int& k) {
    using G = BG<T>;
    const int c = threadIdx.x + 256 * it;
    if (KC) { r = c / (G::BK / G::VEC); k = (c % (G::BK / G::VEC)) * G::VEC; }
    else { k = c / (128 / G::VEC); r = (c % (128 / G::VEC)) * G::VEC; }
}

template <typename T, bool KC>
__device__ __forceinline__ void tile_load(TileRegs<T>& rg, const T* base, int64_t sr, int64_t sk, int r0, int R, int k0, int K,
                                          bool vec_ok) {
    using G = BG<T>;
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
        int r, k;
        chunk_pos<T, KC>(it, r, k);
        const T* p = base + (int64_t)(r0 + r) * sr + (int64_t)(k0 + k) * sk;
        const bool inside = KC ? (r0 + r < R && k0 + k + G::VEC <= K) : (k0 + k < K && r0 + r + G::VEC <= R);
        if (vec_ok && inside) {
            rg.v[it] = *reinterpret_cast<const u32x4*>(p);
        } else {
            union { u32x4 u; T e[G::VEC]; } x;
#pragma unroll
            for (int e = 0; e < G::VEC; ++e) {
                const int rr = KC ? r : r + e, kk = KC ? k + e : k;
                const bool ok = r0 + rr < R && k0 + kk < K;
                x.e[e] = ok ? base[(int64_t)(r0 + rr) * sr + (int64_t)(k0 + kk) * sk] : from_f32<T>(0.f);
            }
            rg.v[it] = x.u;
        }
    }
}

template <typename T, bool KC>
__device__ __forceinline__ void tile_store(const TileRegs<T>& rg, T* lds) {
    using G = BG<T>;
#pragma unroll
    for (int it = 0; it < G::ITEMS; ++it) {
        int r, k;
        chunk_pos<T, KC>(it, r, k);
        *reinterpret_cast<u32x4*>(KC ? lds + r * G::LDK + k : lds + k * G::LDR + r) = rg.v[it];
    }
}

// fragment of rows rowbase + (lane & 31), k = 16 ks + 8 hi .. + 7 of the staged tile
template <bool KC>
__device__ __forceinline__ Frag<bf16_t> tile_frag(const bf16_t* lds, int rowbase, int ks, int lane) {
    using G = BG<bf16_t>;
    const int l31 = lane & 31, hi = lane >> 5;
    if (KC) return ld_frag8(lds + (rowbase + l31) * G::LDK + ks * 16 + 8 * hi);
    // [k][row] image: each 16-lane group reads a [4 k][16 rows] block, lane i supplying k-row i >> 2, rows 4 (i & 3) .. + 3,
    // and receives row i of the block with its 4 k values
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const bf16_t* p = lds + (ks * 16 + 8 * hi + (i >> 2)) * G::LDR + rowbase + 16 * g1 + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * G::LDR));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = up;
    Frag<bf16_t> f;
    f.v = u.v;
    return f;
}
template <bool KC>
__device__ __forceinline__ Frag<float> tile_frag(const float* lds, int rowbase, int ks, int lane) {
    using G = BG<float>;
    const int l31 = lane & 31, hi = lane >> 5;
    if (KC) return ld_frag8(lds + (rowbase + l31) * G::LDK + ks * 16 + 8 * hi);
    Frag<float> f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = lds[(ks * 16 + 8 * hi + e) * G::LDR + rowbase + l31];
    return f;
}

template <typename T, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void bgemm_kernel(BgemmParams p) {
    using G = BG<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + G::TILE;
    const int b = blockIdx.z, i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wi = wave >> 1, wj = wave & 1;
    const T* A = static_cast<const T*>(p.a) + b * p.sab;
    const T* B = static_cast<const T*>(p.b) + b * p.sbb;
    // 16-byte loads: unit stride along the chunk axis, the other stride and the base multiples of the chunk
    const bool avec = (AKC ? (p.sak == 1 && p.sai % G::VEC == 0) : (p.sai == 1 && p.sak % G::VEC == 0))
                      && reinterpret_cast<size_t>(A) % 16 == 0;
    const bool bvec = (BKC ? (p.sbk == 1 && p.sbj % G::VEC == 0) : (p.sbj == 1 && p.sbk % G::VEC == 0))
                      && reinterpret_cast<size_t>(B) % 16 == 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    TileRegs<T> ra, rb;
    tile_load<T, AKC>(ra, A, p.sai, p.sak, i0, p.M, 0, p.K, avec);
    tile_load<T, BKC>(rb, B, p.sbj, p.sbk, j0, p.N, 0, p.K, bvec);
    for (int k0 = 0; k0 < p.K; k0 += G::BK) {
        tile_store<T, AKC>(ra, As);
        tile_store<T, BKC>(rb, Bs);
        __syncthreads();
        if (k0 + G::BK < p.K) {                              // next step's loads fly under this step's MFMAs
            tile_load<T, AKC>(ra, A, p.sai, p.sak, i0, p.M, k0 + G::BK, p.K, avec);
            tile_load<T, BKC>(rb, B, p.sbj, p.sbk, j0, p.N, k0 + G::BK, p.K, bvec);
        }
#pragma unroll
        for (int ks = 0; ks < G::BK / 16; ++ks) {
            Frag<T> af[2], bf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                af[x] = tile_frag<AKC>(As, wi * 64 + x * 32, ks, lane);
                bf[x] = tile_frag<BKC>(Bs, wj * 64 + x * 32, ks, lane);
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) mma32(acc[x][y], af[x], bf[y]);
        }
        __syncthreads();
    }
    T* C = static_cast<T*>(p.c) + b * p.scb;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int j = j0 + wj * 64 + y * 32 + l31;
            if (j >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * 64 + x * 32 + crow(r, hi);
                if (i < p.M) C[(int64_t)i * p.sci + (int64_t)j * p.scj] = from_f32<T>(acc[x][y][r] * p.alpha);
            }
        }
}

template <typename T>
void bgemm_launch(const BgemmParams& p, hipStream_t st) {
    const dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, p.batch);
    const size_t lds = 2 * BG<T>::TILE * sizeof(T);
    const bool akc = !(p.sai == 1 && p.sak != 1), bkc = !(p.sbj == 1 && p.sbk != 1);
    if (akc && bkc) bgemm_kernel<T, true, true><<<grid, dim3(256), lds, st>>>(p);
    else if (akc) bgemm_kernel<T, true, false><<<grid, dim3(256), lds, st>>>(p);
    else if (bkc) bgemm_kernel<T, false, true><<<grid, dim3(256), lds, st>>>(p);
    else bgemm_kernel<T, false, false><<<grid, dim3(256), lds, st>>>(p);
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
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) bgemm_launch<float>(p, st);
    else if (dtype == GF_BF16) bgemm_launch<bf16_t>(p, st);
    else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
