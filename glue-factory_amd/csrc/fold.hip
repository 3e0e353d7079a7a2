// Folding a linear layer into the one that consumes it (weights only; no activation tensor is touched).
//
// LightGlue's blocks compute  h = ffn.0(cat[x, out_proj(ctx)])  (lightglue.py:131-163; to_out in the cross block,
// :166-221), SuperGlue / GlueStick  mlp.0(cat[x, merge(ctx)])  (superglue.py:137-160): no non-linearity sits between the
// two linears, so
//     h = W0a x + W0b (Wo ctx + bo) + b0 = [W0a | W0b Wo] cat[x, ctx] + (b0 + W0b bo).
// Per block and step that removes a [M, 256] <- [M, 256] GEMM, its input-gradient GEMM and its weight-gradient reduction
// over M = 131072 tokens (and the [M, 256] message tensor) at the price of three 512 x 256 x 256 products on the WEIGHTS:
//   gf_fold_linear_fwd   Wc = W0[:, c0:c0+K] Wo[:, cperm],  bc = b0 + W0[:, c0:c0+K] bo   -- table-driven, every block of
//                        the model in ONE launch, ahead of the per-step cast launch (gf_multi_cast_transpose), which
//                        stacks [W0a | Wc] into the compute-dtype weight of the two-source GEMM;
//   gf_fold_linear_bwd   from the gradient g of the stacked weight and gb of the folded bias:
//                        dW0 = [g_a | g_c Wo[:, cperm]^T + gb bo^T],  dWo[:, cperm] = W0b^T g_c,  dbo = W0b^T gb  (db0 = gb).
// Exact fp32 FMA arithmetic (the compute-dtype rounding happens once, in the cast launch).
#include "gf_common.h"
#include "gf_amd.h"

namespace {

struct FoldEntry {
    const float* W0;      // [R, ldw0] fp32; the folded half is columns [c0, c0 + K)
    const float* Wo;      // [K, N] fp32, row-major
    const float* b0;      // [R] or NULL
    const float* bo;      // [K] or NULL
    const int* cperm;     // output column c of Wc reads column cperm[c] of Wo (NULL: c)
    float* Wc;            // [R, N] out
    float* bc;            // [R] out (NULL: no bias)
    int R, K, N, ldw0, c0;
    int tile0;            // first 64 x 64 tile of this entry in the launch (an entry has ceil(R/64) * (tiles_x + 1) tiles)
    int tiles_x;          // ceil(N / 64)
    int pad_;
};
static_assert(sizeof(FoldEntry) == 88, "host table layout (ops.precast)");

// 64 x 64 output tile by 256 threads (16 x 16, 4 x 4 outputs each); a(i, k) / b(k, j) return 0 outside their ranges.
// AI / BJ: the operand is contiguous in memory along its OUTPUT index (i / j) rather than along the reduction index k --
// the thread -> element mapping of the tile loads follows the contiguous direction.
// The products are tiny (67 MFLOP) and LATENCY-bound: 64-deep k tiles (32 independent global loads per thread in flight)
// and the next tile's loads issued before the current tile's FMAs (register prefetch) -- 16-deep tiles without prefetch
// measured 83 us per backward launch, i.e. one exposed memory round trip per k step.
constexpr int FK = 64;
// TS x TS output tile (TS = 64: 4 x 4 outputs per thread, TS = 32: 2 x 2).  The backward uses 32: 64 x 64 tiles made 84
// workgroups of 20 us of plain FMA work each -- a third of the chip for 42 us per launch; 32 x 32 tiles fill it.
template <int TS, bool AI, bool BJ, class FA, class FB>
__device__ __forceinline__ void mmT(float (&acc)[TS / 16][TS / 16], FA a, FB b, int KK, float (*As)[68], float (*Bs)[68]) {
    constexpr int R = TS / 16, NQ = TS * FK / 256, SH = TS == 64 ? 6 : 5, MK = TS - 1;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = 0.f;
    float ra[NQ], rb[NQ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = t + 256 * q;                 // TS * FK elements of each operand tile
            ra[q] = AI ? a(e & MK, k0 + (e >> SH)) : a(e >> 6, k0 + (e & 63));
            rb[q] = BJ ? b(k0 + (e >> SH), e & MK) : b(k0 + (e & 63), e >> 6);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < KK; k0 += FK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = t + 256 * q;
            if (AI) As[e >> SH][e & MK] = ra[q]; else As[e & 63][e >> 6] = ra[q];
            if (BJ) Bs[e >> SH][e & MK] = rb[q]; else Bs[e & 63][e >> 6] = rb[q];
        }
        __syncthreads();
        if (k0 + FK < KK) fetch(k0 + FK);
#pragma unroll 16
        for (int k = 0; k < FK; ++k) {
            float av[R], bv[R];
#pragma unroll
            for (int i = 0; i < R; ++i) av[i] = As[k][ty * R + i];
#pragma unroll
            for (int j = 0; j < R; ++j) bv[j] = Bs[k][tx * R + j];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
}
template <bool AI, bool BJ, class FA, class FB>
__device__ __forceinline__ void mm64(float (&acc)[4][4], FA a, FB b, int KK, float (*As)[68], float (*Bs)[68]) {
    mmT<64, AI, BJ>(acc, a, b, KK, As, Bs);
}

// tiles of one entry: ceil(R / 64) x (ceil(N / 64) + 1) -- the extra tile column of each row block computes the folded bias
// as one more "output column" of the same product (B = bo)
__global__ __launch_bounds__(256) void fold_fwd_kernel(const FoldEntry* __restrict__ tab, int n) {
    __shared__ float As[FK][68], Bs[FK][68];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FoldEntry e = tab[lo];
    const int tl = blockIdx.x - e.tile0;
    const int tnx = e.tiles_x + 1;
    const int r0 = (tl / tnx) * 64, tn = tl % tnx;
    const float* A = e.W0 + e.c0;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc[4][4];
    auto fa = [&](int i, int k) { return (r0 + i < e.R && k < e.K) ? A[(size_t)(r0 + i) * e.ldw0 + k] : 0.f; };
    if (tn == e.tiles_x) {                                   // bias tile
        if (e.bc == nullptr) return;
        mm64<false, true>(acc, fa, [&](int k, int j) { return (j == 0 && k < e.K && e.bo) ? e.bo[k] : 0.f; }, e.K, As, Bs);
        if (tx == 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + ty * 4 + i;
                if (r < e.R) e.bc[r] = acc[i][0] + (e.b0 ? e.b0[r] : 0.f);
            }
        return;
    }
    const int n0 = tn * 64;
    mm64<false, true>(acc, fa,
                      [&](int k, int j) {
                          if (k >= e.K || n0 + j >= e.N) return 0.f;
                          return e.Wo[(size_t)k * e.N + (e.cperm ? e.cperm[n0 + j] : n0 + j)];
                      },
                      e.K, As, Bs);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty * 4 + i, c = n0 + tx * 4 + j;
            if (r < e.R && c < e.N) e.Wc[(size_t)r * e.N + c] = acc[i][j];
        }
}

struct FoldBwd {
    const float* g;       // [R, c0 + N] gradient of the stacked weight [W0a | Wc]
    const float* gb;      // [R] gradient of the folded bias (NULL: none)
    const float* W0; const float* Wo; const float* bo; const int* cperm;
    float* dW0;           // [R, ldw0]
    float* dWo;           // [K, N]
    float* dbo;           // [K] (NULL when there is no bo)
    int R, K, N, ldw0, c0;
    int tiles_a, tiles_b, tiles_c;      // copy part | dW0b | dWo (+ dbo)
};

constexpr int BT = 32;                                      // backward tile
__global__ __launch_bounds__(256) void fold_bwd_kernel(const FoldBwd p) {
    __shared__ float As[FK][68], Bs[FK][68];
    const int ldg = p.c0 + p.N;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    int tl = blockIdx.x;
    if (tl < p.tiles_a) {                                   // dW0[:, :c0] = g[:, :c0]   (64 x 64 copy tiles)
        const int tcx = (p.c0 + 63) / 64, r0 = (tl / tcx) * 64, c0_ = (tl % tcx) * 64;
        for (int e = threadIdx.x; e < 4096; e += 256) {
            const int r = r0 + (e >> 6), c = c0_ + (e & 63);
            if (r < p.R && c < p.c0) p.dW0[(size_t)r * p.ldw0 + c] = p.g[(size_t)r * ldg + c];
        }
        return;
    }
    tl -= p.tiles_a;
    float acc[2][2];
    const float* gc = p.g + p.c0;
    if (tl < p.tiles_b) {                                   // dW0[r, c0 + k] = sum_c g_c[r, c] Wo[k, cp(c)] + gb[r] bo[k]
        const int tkx = (p.K + BT - 1) / BT, r0 = (tl / tkx) * BT, k0 = (tl % tkx) * BT;
        mmT<BT, false, false>(acc,
                              [&](int i, int c) { return (r0 + i < p.R && c < p.N) ? gc[(size_t)(r0 + i) * ldg + c] : 0.f; },
                              [&](int c, int j) {
                                  if (c >= p.N || k0 + j >= p.K) return 0.f;
                                  return p.Wo[(size_t)(k0 + j) * p.N + (p.cperm ? p.cperm[c] : c)];
                              },
                              p.N, As, Bs);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = r0 + ty * 2 + i, k = k0 + tx * 2 + j;
                if (r < p.R && k < p.K) {
                    float v = acc[i][j];
                    if (p.gb && p.bo) v = fmaf(p.gb[r], p.bo[k], v);
                    p.dW0[(size_t)r * p.ldw0 + p.c0 + k] = v;
                }
            }
        return;
    }
    tl -= p.tiles_b;
    {   // dWo[k, cp(c)] = sum_r W0[r, c0 + k] g_c[r, c]; the extra tile column: dbo[k] = sum_r W0[r, c0 + k] gb[r]
        const int tnx = (p.N + BT - 1) / BT + 1, k0 = (tl / tnx) * BT, tn = tl % tnx;
        const float* A = p.W0 + p.c0;
        auto fa = [&](int i, int r) { return (k0 + i < p.K && r < p.R) ? A[(size_t)r * p.ldw0 + k0 + i] : 0.f; };
        if (tn == tnx - 1) {
            if (p.dbo == nullptr) return;
            mmT<BT, true, true>(acc, fa, [&](int r, int j) { return (j == 0 && r < p.R && p.gb) ? p.gb[r] : 0.f; }, p.R, As, Bs);
            if (tx == 0)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (k0 + ty * 2 + i < p.K) p.dbo[k0 + ty * 2 + i] = acc[i][0];
            return;
        }
        const int n0 = tn * BT;
        mmT<BT, true, true>(acc, fa, [&](int r, int j) { return (r < p.R && n0 + j < p.N) ? gc[(size_t)r * ldg + n0 + j] : 0.f; },
                            p.R, As, Bs);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + ty * 2 + i, c = n0 + tx * 2 + j;
                if (k < p.K && c < p.N) p.dWo[(size_t)k * p.N + (p.cperm ? p.cperm[c] : c)] = acc[i][j];
            }
    }
}

}  // namespace

extern "C" int gf_fold_entry_bytes(void) { return (int)sizeof(FoldEntry); }

extern "C" int gf_fold_linear_fwd(const void* table, int n_entries, int total_tiles, void* stream) {
    if (n_entries <= 0 || total_tiles <= 0) return GF_ERR_SHAPE;
    fold_fwd_kernel<<<dim3(total_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        static_cast<const FoldEntry*>(table), n_entries);
    return (int)hipGetLastError();
}

extern "C" int gf_fold_linear_bwd(const float* g, const float* gb, const float* W0, const float* Wo, const float* bo,
                                  const int* cperm, float* dW0, float* dWo, float* dbo, int R, int K, int N, int ldw0,
                                  int c0, void* stream) {
    if (R <= 0 || K <= 0 || N <= 0 || c0 < 0 || ldw0 < c0 + K) return GF_ERR_SHAPE;
    if (!g || !W0 || !Wo || !dW0 || !dWo) return GF_ERR_SHAPE;
    FoldBwd p;
    p.g = g; p.gb = gb; p.W0 = W0; p.Wo = Wo; p.bo = bo; p.cperm = cperm;
    p.dW0 = dW0; p.dWo = dWo; p.dbo = bo ? dbo : nullptr;
    p.R = R; p.K = K; p.N = N; p.ldw0 = ldw0; p.c0 = c0;
    p.tiles_a = ((R + 63) / 64) * ((c0 + 63) / 64);             // copy tiles 64 x 64, product tiles BT x BT
    p.tiles_b = ((R + BT - 1) / BT) * ((K + BT - 1) / BT);
    p.tiles_c = ((K + BT - 1) / BT) * ((N + BT - 1) / BT + 1);
    fold_bwd_kernel<<<dim3(p.tiles_a + p.tiles_b + p.tiles_c), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(p);
    return (int)hipGetLastError();
}
