// Weight / bias gradient of a linear layer over the token axis:
//     dW[n][k] = sum_m dY[m][n] X[m][k],   db[n] = sum_m dY[m][n]        (M = B*N tokens ~ 1.3e5)
// i.e. autograd's grad of F.linear (reference: every nn.Linear of lightglue.py:131-221, 271-290).
// The library GEMM handles this "tiny output, huge reduction" shape poorly (0.16-0.26 ms per
// call, 2/3 of all GEMM time of the step), so it is a hand-written split-M MFMA kernel:
// each workgroup owns a 128x128 output tile and one slice of M, streams dY / X row tiles
// through LDS TRANSPOSED (token axis contiguous, so MFMA fragments are 16-byte reads),
// double-buffered with register prefetch, accumulates in fp32, and writes its partial tile to
// a workspace; a second kernel sums the slices (deterministic, no atomics) into fp32 dW / db.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct DwLay {
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int LDT = 64 + VEC;            // transposed tile stride (16-byte aligned rows)
    static constexpr int TILE = 128 * LDT;          // elements: [128 cols][64 tokens]
    static constexpr int CPR = 128 / VEC;           // 16-byte chunks per 128-wide row slice
    static constexpr int ITEMS = 32 * CPR / 256;    // (row pair, chunk) items per thread
};
template <typename T> struct PairT;
template <> struct PairT<bf16_t> { typedef bf16x2 type; };
template <> struct PairT<float> { typedef f32x2 type; };

template <typename T> struct DwRegs { u32x4 a[DwLay<T>::ITEMS], b[DwLay<T>::ITEMS]; };

// 8-element (16-byte) blocks of a transposed row are XOR-swizzled by the row's chunk index so that
// the coalesced staging order (consecutive lanes = consecutive 16-byte chunks of one source row)
// spreads its transposed 4-byte stores over the LDS banks; fragment reads stay 16-byte.
__device__ __forceinline__ int dswz(int d, int pos) { return pos ^ (((d >> 3) & 7) << 3); }

// rows [m0, m0+64) x cols [c0, c0+128) of a row-major [M, ld] matrix (zero beyond M / ncols)
template <typename T>
__device__ __forceinline__ void dw_load(DwRegs<T>& rg, const T* g, int64_t ld, int m0, int mend, int c0, int ncols) {
    using L = DwLay<T>;
#pragma unroll
    for (int i = 0; i < L::ITEMS; ++i) {
        int it = threadIdx.x + 256 * i;
        int cc = it % L::CPR, p = it / L::CPR;
        int r0 = m0 + 2 * p, r1 = r0 + 1, col = c0 + cc * L::VEC;
        u32x4 z = {0, 0, 0, 0};
        bool okc = col < ncols;
        rg.a[i] = (okc && r0 < mend) ? *reinterpret_cast<const u32x4*>(g + (int64_t)r0 * ld + col) : z;
        rg.b[i] = (okc && r1 < mend) ? *reinterpret_cast<const u32x4*>(g + (int64_t)r1 * ld + col) : z;
    }
}
template <typename T>
__device__ __forceinline__ void dw_store(const DwRegs<T>& rg, T* ldsT) {
    using L = DwLay<T>;
    typedef typename PairT<T>::type pair_t;
#pragma unroll
    for (int i = 0; i < L::ITEMS; ++i) {
        int it = threadIdx.x + 256 * i;
        int cc = it % L::CPR, p = it / L::CPR;
        union { u32x4 u; T e[L::VEC]; } x, y;
        x.u = rg.a[i];
        y.u = rg.b[i];
#pragma unroll
        for (int e = 0; e < L::VEC; ++e) {
            pair_t pr = {x.e[e], y.e[e]};
            const int d = cc * L::VEC + e;
            *reinterpret_cast<pair_t*>(ldsT + d * L::LDT + dswz(d, 2 * p)) = pr;
        }
    }
}

// grid: (ntile_n * ntile_k, nslice); partial [nslice][Nout][K] fp32, bias partial [nslice][Nout]
template <typename T>
__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           float* __restrict__ part, float* __restrict__ bpart,
                                                           int M, int Nout, int K, int rows_per_slice) {
    using L = DwLay<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lds = reinterpret_cast<T*>(smem);           // 2 buffers x (A^T tile | B^T tile)
    const int ntk = (K + 127) / 128;
    const int tn = blockIdx.x / ntk, tk = blockIdx.x % ntk;
    const int slice = blockIdx.y;
    const int m_begin = slice * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;       // 2 x 2 waves, each 64 (n) x 64 (k)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[L::ITEMS][L::VEC];
#pragma unroll
    for (int i = 0; i < L::ITEMS; ++i)
#pragma unroll
        for (int e = 0; e < L::VEC; ++e) bsum[i][e] = 0.f;

    DwRegs<T> ra, rb;
    auto add_bias = [&]() {
        if (tk == 0) {
#pragma unroll
            for (int i = 0; i < L::ITEMS; ++i) {
                union { u32x4 u; T e[L::VEC]; } x0, x1;
                x0.u = ra.a[i];
                x1.u = ra.b[i];
#pragma unroll
                for (int e = 0; e < L::VEC; ++e) bsum[i][e] += to_f32(x0.e[e]) + to_f32(x1.e[e]);
            }
        }
    };
    const int nchunk = (m_end - m_begin + 63) / 64;
    if (nchunk > 0) {
        dw_load<T>(ra, dy, Nout, m_begin, m_end, tn * 128, Nout);
        dw_load<T>(rb, x, K, m_begin, m_end, tk * 128, K);
        add_bias();
        dw_store<T>(ra, lds);
        dw_store<T>(rb, lds + L::TILE);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const T* At = lds + (c & 1) * 2 * L::TILE;
        const T* Bt = At + L::TILE;
        if (c + 1 < nchunk) {
            dw_load<T>(ra, dy, Nout, m_begin + (c + 1) * 64, m_end, tn * 128, Nout);
            dw_load<T>(rb, x, K, m_begin + (c + 1) * 64, m_end, tk * 128, K);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {      // 64 tokens = 4 k-steps of 16
            Frag<T> af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int dn = wn * 64 + i * 32 + l31, dk = wk * 64 + i * 32 + l31;
                af[i] = ld_frag8(At + dn * L::LDT + dswz(dn, 16 * s + 8 * hi));
                bf[i] = ld_frag8(Bt + dk * L::LDT + dswz(dk, 16 * s + 8 * hi));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32(acc[i][j], af[i], bf[j]);
        }
        if (c + 1 < nchunk) {
            add_bias();
            T* nb = lds + ((c + 1) & 1) * 2 * L::TILE;
            dw_store<T>(ra, nb);
            dw_store<T>(rb, nb + L::TILE);
        }
        __syncthreads();
    }
    // C[n][k]: lane column = k (l31), rows n = crow(r, hi)
    float* pp = part + (int64_t)slice * Nout * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = tk * 128 + wk * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = tn * 128 + wn * 64 + i * 32 + crow(r, hi);
                if (nn < Nout && kk < K) pp[(int64_t)nn * K + kk] = acc[i][j][r];
            }
        }
    if (tk == 0) {
        // chunk cc = it % CPR is fixed per thread across chunks; its 32 row pairs p = it / CPR are spread
        // over the whole workgroup -> combine through LDS (reused after the last barrier)
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < L::ITEMS; ++i) {
            int it = threadIdx.x + 256 * i;
#pragma unroll
            for (int e = 0; e < L::VEC; ++e) red[it * L::VEC + e] = bsum[i][e];
        }
        __syncthreads();
        for (int col = threadIdx.x; col < 128; col += 256) {
            const int cc = col / L::VEC, e = col % L::VEC;
            float v = 0.f;
            for (int p = 0; p < 32; ++p) v += red[(p * L::CPR + cc) * L::VEC + e];
            if (tn * 128 + col < Nout) bpart[(int64_t)slice * Nout + tn * 128 + col] = v;
        }
    }
}

__global__ void linear_dw_reduce(const float* __restrict__ part, const float* __restrict__ bpart,
                                 float* __restrict__ dw, float* __restrict__ db, int nslice, int64_t nw, int Nout) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) {
        float s = 0.f;
        for (int k = 0; k < nslice; ++k) s += part[(int64_t)k * nw + i];
        dw[i] = s;
    } else if (db && i < nw + Nout) {
        int n = (int)(i - nw);
        float s = 0.f;
        for (int k = 0; k < nslice; ++k) s += bpart[(int64_t)k * Nout + n];
        db[n] = s;
    }
}

struct DwPlan { int ntile, nslice, rows; };
DwPlan plan(int M, int Nout, int K) {
    DwPlan p;
    p.ntile = ((Nout + 127) / 128) * ((K + 127) / 128);
    int want = (640 + p.ntile - 1) / p.ntile;                 // ~2.5 workgroups per CU in total
    int rows = (M + want - 1) / want;
    rows = ((rows + 63) / 64) * 64;
    if (rows < 256) rows = 256;
    p.rows = rows;
    p.nslice = (M + rows - 1) / rows;
    return p;
}

template <typename T>
int launch_dw(const void* dy, const void* x, float* dw, float* db, void* ws, int M, int Nout, int K, hipStream_t st) {
    using L = DwLay<T>;
    DwPlan p = plan(M, Nout, K);
    float* part = reinterpret_cast<float*>(ws);
    float* bpart = part + (int64_t)p.nslice * Nout * K;
    size_t lds = 4 * (size_t)L::TILE * sizeof(T);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_dw_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    linear_dw_kernel<T><<<dim3(p.ntile, p.nslice), 256, lds, st>>>(
        reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(x), part, bpart, M, Nout, K, p.rows);
    int64_t nw = (int64_t)Nout * K;
    int64_t tot = nw + (db ? Nout : 0);
    linear_dw_reduce<<<dim3((unsigned)((tot + 255) / 256)), 256, 0, st>>>(part, bpart, dw, db, p.nslice, nw, Nout);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int64_t gf_linear_dw_ws_bytes(int M, int Nout, int K) {
    if (M <= 0 || Nout <= 0 || K <= 0) return GF_ERR_SHAPE;
    DwPlan p = plan(M, Nout, K);
    return ((int64_t)p.nslice * Nout * K + (int64_t)p.nslice * Nout) * 4 + 256;
}

extern "C" int gf_linear_dw(const void* dy, const void* x, float* dw, float* db, void* ws,
                            int M, int Nout, int K, int dtype, void* stream) {
    if (M <= 0 || Nout <= 0 || K <= 0) return GF_ERR_SHAPE;
    const int align = dtype == GF_BF16 ? 8 : 4;
    if (Nout % align || K % align) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_dw<float>(dy, x, dw, db, ws, M, Nout, K, st);
    if (dtype == GF_BF16) return launch_dw<bf16_t>(dy, x, dw, db, ws, M, Nout, K, st);
    return GF_ERR_DTYPE;
}
