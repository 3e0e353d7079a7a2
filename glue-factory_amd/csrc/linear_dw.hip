// Weight / bias gradient of a linear layer over the token axis:
//     dW[n][k] = sum_m dY[m][n] X[m][k],   db[n] = sum_m dY[m][n]        (M = B*N tokens ~ 1.3e5)
// i.e. autograd's grad of F.linear (reference: every nn.Linear of lightglue.py:131-221, 271-290).
// The library GEMM handles this "tiny output, huge reduction" shape poorly (0.16-0.26 ms per
// call, 2/3 of all GEMM time of the step), so it is a hand-written split-M MFMA kernel:
// each workgroup owns a 128x128 output tile and one slice of M, streams dY / X row tiles
// through LDS TRANSPOSED (token axis contiguous, so MFMA fragments are 16-byte reads),
// double-buffered with register prefetch, accumulates in fp32, and writes its partial tile to
// a workspace; a second kernel sums the slices (deterministic, no atomics) into fp32 dW / db.
#include <cstdlib>
#include <type_traits>
#include "gf_common.h"
#include "gf_amd.h"

#ifndef GF_DW_WG          // workgroup slots of one dW round (two per CU); probe builds override it
#define GF_DW_WG 512
#endif

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
    // XCD-aware order: block b runs on XCD b % 8; all output tiles of one M-slice are given to the
    // same XCD back to back, so the dY / X row slabs they share are served by that XCD's L2 instead of
    // being re-read from HBM once per tile (4x the traffic at 512x512).
    const int ntk = (K + 127) / 128;
    const int ntile = ((Nout + 127) / 128) * ntk;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % ntile;
    const int slice = (j / ntile) * 8 + xcd;
    const int tn = tile / ntk, tk = tile % ntk;
    const int m_begin = slice * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    if (m_begin >= M) return;
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

// ---------------------------------------------------------------------------------------------
// bf16 fast path: ROW-MAJOR LDS tiles (16-byte coalesced global loads -> conflict-free 16-byte LDS
// stores, no transposing stores at all) read back through gfx950's transposing LDS load
// ds_read_b64_tr_b16.  Semantics (probed on hardware, tools/probe/tr.hip): inside each group of 16
// lanes, lane s supplies the address of 4 consecutive bf16; output lane i receives, for j = 0..3,
// element (i & 3) of source lane (i >> 2) + 4 j.  With source lane s pointing at
// tile[m0 + (s >> 2)][c0 + 4 (s & 3)], output lane i holds column c0 + i at rows m0 .. m0 + 3:
// exactly the "4 consecutive tokens of my column" an MFMA fragment needs.
// ---------------------------------------------------------------------------------------------
constexpr int TR_LD = 128 + 32;                 // row stride (bf16): 320 B -> rows 16 banks apart (of 64)
constexpr int TR_TILE = 64 * TR_LD;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2 ds_read_tr16(const bf16_t* p) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}

struct TrRegs { u32x4 v[4]; unsigned ok; };     // 64 x 128 bf16 tile = 1024 chunks / 256 threads

// Loads are UNCONDITIONAL (indices clamped into the matrix, validity kept as bits and applied when
// the registers are stashed): predicated loads become exec-masked branches, which both serialise
// issue and make the compiler fall back to s_waitcnt vmcnt(0), i.e. no prefetch across iterations.
__device__ __forceinline__ void tr_load(TrRegs& rg, const bf16_t* g, int64_t ld, int m0, int mend, int c0, int ncols) {
    rg.ok = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int it = threadIdx.x + 256 * i;
        int cc = it & 15, r = it >> 4;
        int row = m0 + r, col = c0 + cc * 8;
        rg.ok |= (unsigned)(row < mend && col < ncols) << i;
        rg.v[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)min(row, mend - 1) * ld + min(col, ncols - 8));
    }
}
__device__ __forceinline__ void tr_mask(TrRegs& rg) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned keep = (rg.ok >> i) & 1u ? 0xffffffffu : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) rg.v[i][e] &= keep;
    }
}
__device__ __forceinline__ void tr_store(const TrRegs& rg, bf16_t* tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int it = threadIdx.x + 256 * i;
        int cc = it & 15, r = it >> 4;
        *reinterpret_cast<u32x4*>(tile + r * TR_LD + cc * 8) = rg.v[i];
    }
}

__global__ __launch_bounds__(256, 2) void linear_dw_tr_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                              float* __restrict__ part, float* __restrict__ bpart,
                                                              int M, int Nout, int K, int rows_per_slice) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* lds = reinterpret_cast<bf16_t*>(smem);          // 2 buffers x (dY tile | X tile), row-major
    // XCD-aware order: block b runs on XCD b % 8; all output tiles of one M-slice are given to the
    // same XCD back to back, so the dY / X row slabs they share are served by that XCD's L2 instead of
    // being re-read from HBM once per tile (4x the traffic at 512x512).
    const int ntk = (K + 127) / 128;
    const int ntile = ((Nout + 127) / 128) * ntk;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % ntile;
    const int slice = (j / ntile) * 8 + xcd;
    const int tn = tile / ntk, tk = tile % ntk;
    const int m_begin = slice * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    if (m_begin >= M) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    // per-lane part of the transposing-read address: row (s >> 2) + 8 hi, column 16 (g & 1) + 4 (s & 3)
    const int s16 = lane & 15, g1 = (lane >> 4) & 1;
    const int lane_off = ((s16 >> 2) + 8 * hi) * TR_LD + 16 * g1 + 4 * (s16 & 3);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i][e] = 0.f;

    // Global loads run TWO chunks ahead of the MFMAs (two register sets, statically indexed through an
    // unroll-by-2), the LDS image one chunk ahead.
    TrRegs ra0, rb0, ra1, rb1;
    auto load = [&](TrRegs& a_, TrRegs& b_, int c) {
        tr_load(a_, dy, Nout, m_begin + c * 64, m_end, tn * 128, Nout);
        tr_load(b_, x, K, m_begin + c * 64, m_end, tk * 128, K);
    };
    auto stash = [&](TrRegs& a_, TrRegs& b_, int c) {
        tr_mask(a_);
        tr_mask(b_);
        if (tk == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                union { u32x4 u; bf16_t e[8]; } t;
                t.u = a_.v[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[i][e] += (float)t.e[e];
            }
        }
        bf16_t* nb = lds + (c & 1) * 2 * TR_TILE;
        tr_store(a_, nb);
        tr_store(b_, nb + TR_TILE);
    };
    // Fragment reads of k-step ks+1 are issued before the MFMAs of k-step ks (two register sets), so the
    // LDS latency of the (compiler-invisible) transposing reads hides behind 4 MFMAs instead of stalling.
    struct FragSet { u32x2 a[2][2], b[2][2]; };
    auto issue = [&](FragSet& f, const bf16_t* At, const bf16_t* Bt, int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* pa = At + (16 * ks) * TR_LD + wn * 64 + i * 32 + lane_off;
            const bf16_t* pb = Bt + (16 * ks) * TR_LD + wk * 64 + i * 32 + lane_off;
            f.a[i][0] = ds_read_tr16(pa);
            f.a[i][1] = ds_read_tr16(pa + 4 * TR_LD);
            f.b[i][0] = ds_read_tr16(pb);
            f.b[i][1] = ds_read_tr16(pb + 4 * TR_LD);
        }
    };
    auto wait_all = [&](FragSet& f) {
        // asm loads are invisible to the compiler's waitcnt bookkeeping: tie every destination to the wait
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]),
                       "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])
                     :: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](const FragSet& f) {
        Frag<bf16_t> af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            union { u32x2 u[2]; bf16x8 v; } ta, tb;
            ta.u[0] = f.a[i][0]; ta.u[1] = f.a[i][1];
            tb.u[0] = f.b[i][0]; tb.u[1] = f.b[i][1];
            af[i].v = ta.v;
            bf[i].v = tb.v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma32(acc[i][j], af[i], bf[j]);
    };
    auto compute = [&](int c) {
        const bf16_t* At = lds + (c & 1) * 2 * TR_TILE;
        const bf16_t* Bt = At + TR_TILE;
        FragSet f0, f1;
        issue(f0, At, Bt, 0);
        wait_all(f0);
        issue(f1, At, Bt, 1);
        mfmas(f0);
        wait_all(f1);
        issue(f0, At, Bt, 2);
        mfmas(f1);
        wait_all(f0);
        issue(f1, At, Bt, 3);
        mfmas(f0);
        wait_all(f1);
        mfmas(f1);
    };
    const int nchunk = (m_end - m_begin + 63) / 64;
    if (nchunk > 0) {
        load(ra0, rb0, 0);
        if (nchunk > 1) load(ra1, rb1, 1);
        stash(ra0, rb0, 0);
    }
    __syncthreads();
    // top of iteration c: LDS[c&1] holds chunk c; register set (c+1)&1 holds chunk c+1 (maybe in flight)
    for (int c = 0; c < nchunk; c += 2) {
        if (c + 2 < nchunk) load(ra0, rb0, c + 2);
        compute(c);
        if (c + 1 < nchunk) stash(ra1, rb1, c + 1);
        __syncthreads();
        if (c + 1 < nchunk) {
            if (c + 3 < nchunk) load(ra1, rb1, c + 3);
            compute(c + 1);
            if (c + 2 < nchunk) stash(ra0, rb0, c + 2);
            __syncthreads();
        }
    }
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
        // column chunk cc = tid & 15 is fixed per thread; its rows are spread over the workgroup
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e)
            red[threadIdx.x * 8 + e] = bsum[0][e] + bsum[1][e] + bsum[2][e] + bsum[3][e];
        __syncthreads();
        for (int col = threadIdx.x; col < 128; col += 256) {
            const int cc = col >> 3, e = col & 7;
            float v = 0.f;
            for (int r = 0; r < 16; ++r) v += red[(r * 16 + cc) * 8 + e];
            if (tn * 128 + col < Nout) bpart[(int64_t)slice * Nout + tn * 128 + col] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16, Nout % 128 == 0 and K % 128 == 0: LDS-DMA staging.  The register-staged kernel above spends more
// LDS-pipe time on its staging stores (8 ds_write_b128 per thread and chunk: ~415 cycles per workgroup-chunk
// at 79 B/clk) than on the transposing reads the MFMAs need (256 cycles), and the LDS pipe is what bounds it.
// Here `global_load_lds_dwordx4` writes the tiles (32 tokens x 128 channels of dY and of X per stage,
// unpadded rows, 16-byte chunks XOR-swizzled by (row & 3) << 2 so the 4-row transposing reads hit distinct
// bank groups) into a 4-stage ring, counted vmcnt, one raw barrier per chunk; the bias gradient is a third
// MFMA against a fragment of ones.
// ---------------------------------------------------------------------------------------------
constexpr int DM_ROWS = 32, DM_TILE = DM_ROWS * 256, DM_STAGE = 2 * DM_TILE, DM_NSTAGE = 4;
typedef __attribute__((address_space(3))) void dm_lds_void;
typedef const __attribute__((address_space(1))) void dm_glb_void;

template <int OFF> __device__ __forceinline__ u32x2 dm_rdtr(unsigned a) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void dm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename V> __device__ __forceinline__ void dm_tie(V& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ bf16x8 dm_frag(u32x2 lo, u32x2 hi) {
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int SO, bool BIAS>
__device__ __forceinline__ void dm_chunk(f32x16 (&acc)[2][2], f32x16 (&bacc)[2], unsigned aA, unsigned aB) {
    // 32 tokens = 2 k-steps: [ks][i][half] for the dY^T (A) and X^T (B) fragments; i toggles address bit 6
    u32x2 fa[2][2][2], fb[2][2][2];
#define GF_RD(ks) \
    fa[ks][0][0] = dm_rdtr<SO + ks * 4096>(aA);            fa[ks][0][1] = dm_rdtr<SO + ks * 4096 + 1024>(aA); \
    fa[ks][1][0] = dm_rdtr<SO + ks * 4096>(aA ^ 64u);      fa[ks][1][1] = dm_rdtr<SO + ks * 4096 + 1024>(aA ^ 64u); \
    fb[ks][0][0] = dm_rdtr<SO + DM_TILE + ks * 4096>(aB);  fb[ks][0][1] = dm_rdtr<SO + DM_TILE + ks * 4096 + 1024>(aB); \
    fb[ks][1][0] = dm_rdtr<SO + DM_TILE + ks * 4096>(aB ^ 64u); fb[ks][1][1] = dm_rdtr<SO + DM_TILE + ks * 4096 + 1024>(aB ^ 64u);
    GF_RD(0)
    GF_RD(1)
#undef GF_RD
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (ks == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) { dm_tie(fa[ks][i][0]); dm_tie(fa[ks][i][1]); dm_tie(fb[ks][i][0]); dm_tie(fb[ks][i][1]); }
        const bf16x8 a0 = dm_frag(fa[ks][0][0], fa[ks][0][1]), a1 = dm_frag(fa[ks][1][0], fa[ks][1][1]);
        const bf16x8 b0 = dm_frag(fb[ks][0][0], fb[ks][0][1]), b1 = dm_frag(fb[ks][1][0], fb[ks][1][1]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        if (BIAS) {
            bacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ones, bacc[0], 0, 0, 0);
            bacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ones, bacc[1], 0, 0, 0);
        }
    }
}

// x: the X source of THIS k-tile, already offset to its first column; ldx: that source's row stride (two-source calls:
// tiles left of K1 read x1, the others x2 -- the virtual concatenation [x1 | x2] is never built)
template <bool BIAS>
__device__ __forceinline__ void dm_body(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, int ldx,
                                        float* __restrict__ pp, float* __restrict__ bp, int M, int Nout, int K, int m_begin,
                                        int m_end, int tn, int tk) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5, s16 = lane & 15, g1 = (lane >> 4) & 1;
    const int wn = wave >> 1, wk = wave & 1;

    // ---- DMA descriptors: pieces 2*wave, 2*wave+1 of each tile; a piece = 4 rows x 256 B
    const int prow = lane >> 4, pphys = lane & 15;
    const int plog = pphys ^ ((prow & 3) << 2);
    const bf16_t* gA = dy + (int64_t)tn * 128 + plog * 8;
    const bf16_t* gB = x + plog * 8;
    auto issue = [&](int c, int stage) {
        char* sb = smem + stage * DM_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = 2 * wave + i;
            const int64_t row = min(m_begin + c * DM_ROWS + 4 * piece + prow, M - 1);
            __builtin_amdgcn_global_load_lds((dm_glb_void*)(gA + row * Nout), (dm_lds_void*)(sb + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((dm_glb_void*)(gB + row * ldx), (dm_lds_void*)(sb + DM_TILE + piece * 1024), 16, 0, 0);
        }
    };
    const int nchunk = (m_end - m_begin + DM_ROWS - 1) / DM_ROWS;
#pragma unroll
    for (int c = 0; c < DM_NSTAGE - 1; ++c)
        if (c < nchunk) issue(c, c);

    f32x16 acc[2][2], bacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // per-lane transposing-read addresses (stage 0, k-step 0, first half): row 8 hi + (s16 >> 2), 8-byte piece
    // (s16 & 1) of logical 16-byte chunk 8 w + 2 g1 + ((s16 & 3) >> 1), XOR-swizzled by (row & 3) << 2
    const int rr = (s16 >> 2) & 3;
    auto rd_addr = [&](int w) {
        const int chunk = (8 * w + 2 * g1 + ((s16 & 3) >> 1)) ^ (rr << 2);
        return lds0 + (unsigned)((8 * hi + (s16 >> 2)) * 256 + chunk * 16 + (s16 & 1) * 8);
    };
    const unsigned aA = rd_addr(wn), aB = rd_addr(wk);

    auto step = [&](int c, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        if (c + 1 >= nchunk) dm_wait_vm<0>();                     // chunk c landed (this wave's pieces)
        else if (c + 2 >= nchunk) dm_wait_vm<4>();
        else dm_wait_vm<8>();
        __builtin_amdgcn_s_barrier();                             // ... everyone's; the stage of chunk c-1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (c + DM_NSTAGE - 1 < nchunk) issue(c + DM_NSTAGE - 1, (ST + DM_NSTAGE - 1) % DM_NSTAGE);
        const int valid = m_end - (m_begin + c * DM_ROWS);
        if (valid < DM_ROWS) {                                    // ragged last chunk: rows past m_end count as 0
            for (int i = threadIdx.x; i < 2 * DM_ROWS * 16; i += 256) {
                const int row = (i >> 4) & (DM_ROWS - 1);
                if (row >= valid) *reinterpret_cast<u32x4*>(smem + ST * DM_STAGE + (i >> 9) * DM_TILE + row * 256 + (i & 15) * 16) = u32x4{0, 0, 0, 0};
            }
            __syncthreads();
        }
        dm_chunk<ST * DM_STAGE, BIAS>(acc, bacc, aA, aB);
    };
    for (int c = 0; c < nchunk; c += DM_NSTAGE) {
        step(c, std::integral_constant<int, 0>{});
        if (c + 1 < nchunk) step(c + 1, std::integral_constant<int, 1>{});
        if (c + 2 < nchunk) step(c + 2, std::integral_constant<int, 2>{});
        if (c + 3 < nchunk) step(c + 3, std::integral_constant<int, 3>{});
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = tk * 128 + wk * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = tn * 128 + wn * 64 + i * 32 + crow(r, hi);
                pp[(int64_t)nn * K + kk] = acc[i][j][r];
            }
        }
    if (BIAS && wk == 0 && l31 == 0) {          // every column of the ones-product holds the row sums
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bp[tn * 128 + wn * 64 + i * 32 + crow(r, hi)] = bacc[i][r];
    }
}

__global__ __launch_bounds__(256, 2) void linear_dw_dma_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ x2, int K1,
                                                               float* __restrict__ part, float* __restrict__ bpart,
                                                               int M, int Nout, int K, int rows_per_slice) {
    const int ntk = K / 128;
    const int ntile = (Nout / 128) * ntk;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % ntile;
    const int slice = (j / ntile) * 8 + xcd;
    const int tn = tile / ntk, tk = tile % ntk;
    const int m_begin = slice * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    if (m_begin >= M) return;
    float* pp = part + (int64_t)slice * Nout * K;
    float* bp = bpart + (int64_t)slice * Nout;
    const bf16_t* xs = x + (int64_t)tk * 128;
    int ldx = K;
    if (x2 != nullptr) {                         // x = [x1 (K1 columns) | x2 (K - K1 columns)], K1 % 128 == 0
        if (tk * 128 < K1) { ldx = K1; }
        else { xs = x2 + (int64_t)(tk * 128 - K1); ldx = K - K1; }
    }
    if (tk == 0) dm_body<true>(dy, xs, ldx, pp, bp, M, Nout, K, m_begin, m_end, tn, tk);
    else dm_body<false>(dy, xs, ldx, pp, bp, M, Nout, K, m_begin, m_end, tn, tk);
}

// Sum of the per-slice partials.  A workgroup covers 64 float4 columns with FOUR slice groups (one per wave,
// slices k = g mod 4) that meet in LDS: a 256x256 output with 64 slices is 256 workgroups of 16-deep
// chains instead of 64 workgroups of 64-deep ones (the kernel is latency-, not bandwidth-bound).
// Blocks past the weight columns reduce the bias partials the same way.  Deterministic (no atomics).
__global__ __launch_bounds__(256) void linear_dw_reduce(const float* __restrict__ part, const float* __restrict__ bpart,
                                                        float* __restrict__ dw, float* __restrict__ db, int nslice,
                                                        int64_t nw, int Nout, int nwb) {
    __shared__ f32x4 red[3][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const bool is_bias = (int)blockIdx.x >= nwb;
    const int64_t ncol4 = is_bias ? Nout / 4 : nw / 4;
    const int64_t j = (int64_t)(is_bias ? blockIdx.x - nwb : blockIdx.x) * 64 + c;
    const float* src = is_bias ? bpart : part;
    const int64_t stride = is_bias ? Nout : nw;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    if (j < ncol4) {
        int k = g;
        // eight slices in flight per thread (a loop of load pairs waits for every pair: nslice / 8 dependent round trips in a
        // kernel that is nothing but latency); same two chains, same order
        for (; k + 28 < nslice; k += 32) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 4 * u) * stride + 4 * j);
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += x[u]; a1 += x[u + 1]; }
        }
        for (; k + 4 < nslice; k += 8) {
            a0 += *reinterpret_cast<const f32x4*>(src + (int64_t)k * stride + 4 * j);
            a1 += *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 4) * stride + 4 * j);
        }
        if (k < nslice) a0 += *reinterpret_cast<const f32x4*>(src + (int64_t)k * stride + 4 * j);
    }
    a0 += a1;
    if (g > 0) red[g - 1][c] = a0;
    __syncthreads();
    if (g == 0 && j < ncol4) {
        a0 = (a0 + red[0][c]) + (red[1][c] + red[2][c]);
        *reinterpret_cast<f32x4*>((is_bias ? db : dw) + 4 * j) = a0;
    }
}

struct DwPlan { int ntile, nslice, rows; };
DwPlan plan(int M, int Nout, int K) {
    DwPlan p;
    p.ntile = ((Nout + 127) / 128) * ((K + 127) / 128);
    // One full round of two workgroups per CU (512 slots): slices = floor(512 / tiles), so no workgroup is
    // left for a second, nearly empty round (12 tiles x 43 slices = 516 workgroups ran 27 % slower).
    // Measured on MI355X at M = 131072 with the 4-group reduce.  (-DGF_DW_WG=... in probe builds.)
    const int total_wg = GF_DW_WG;
    // slices are dealt to the 8 XCDs round-robin: keep (slices per XCD) x tiles within that XCD's 64 slots
    int per_xcd = (total_wg / 8) / p.ntile;
    if (per_xcd < 1) per_xcd = 1;
    int want = 8 * per_xcd;
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
    linear_dw_kernel<T><<<dim3(p.ntile * ((p.nslice + 7) / 8) * 8), 256, lds, st>>>(
        reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(x), part, bpart, M, Nout, K, p.rows);
    int64_t nw = (int64_t)Nout * K;
    const int nwb = (int)((nw / 4 + 63) / 64), nbb = db ? (Nout / 4 + 63) / 64 : 0;
    linear_dw_reduce<<<dim3(nwb + nbb), 256, 0, st>>>(part, bpart, dw, db, p.nslice, nw, Nout, nwb);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int64_t gf_linear_dw_ws_bytes(int M, int Nout, int K) {
    if (M <= 0 || Nout <= 0 || K <= 0) return GF_ERR_SHAPE;
    DwPlan p = plan(M, Nout, K);
    return ((int64_t)p.nslice * Nout * K + (int64_t)p.nslice * Nout) * 4 + 256;
}

int launch_dw_tr(const void* dy, const void* x, float* dw, float* db, void* ws, int M, int Nout, int K, hipStream_t st,
                 const void* x2 = nullptr, int K1 = 0) {
    DwPlan p = plan(M, Nout, K);
    float* part = reinterpret_cast<float*>(ws);
    float* bpart = part + (int64_t)p.nslice * Nout * K;
    if (Nout % 128 == 0 && K % 128 == 0) {
        const size_t dlds = (size_t)DM_NSTAGE * DM_STAGE;
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_dw_dma_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds);
        if (e2 != hipSuccess) return (int)e2;
        linear_dw_dma_kernel<<<dim3(p.ntile * ((p.nslice + 7) / 8) * 8), 256, dlds, st>>>(
            reinterpret_cast<const bf16_t*>(dy), reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(x2), K1,
            part, bpart, M, Nout, K, p.rows);
        if (int e3 = (int)hipGetLastError()) return e3;
        int64_t nw2 = (int64_t)Nout * K;
        const int nwb2 = (int)((nw2 / 4 + 63) / 64), nbb2 = db ? (Nout / 4 + 63) / 64 : 0;
        linear_dw_reduce<<<dim3(nwb2 + nbb2), 256, 0, st>>>(part, bpart, dw, db, p.nslice, nw2, Nout, nwb2);
        return (int)hipGetLastError();
    }
    if (x2 != nullptr) return GF_ERR_UNSUPPORTED;          // (two sources: 128-multiple shapes only, checked by the caller)
    size_t lds = 4 * (size_t)TR_TILE * sizeof(bf16_t);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_dw_tr_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    linear_dw_tr_kernel<<<dim3(p.ntile * ((p.nslice + 7) / 8) * 8), 256, lds, st>>>(
        reinterpret_cast<const bf16_t*>(dy), reinterpret_cast<const bf16_t*>(x), part, bpart, M, Nout, K, p.rows);
    int64_t nw = (int64_t)Nout * K;
    const int nwb = (int)((nw / 4 + 63) / 64), nbb = db ? (Nout / 4 + 63) / 64 : 0;
    linear_dw_reduce<<<dim3(nwb + nbb), 256, 0, st>>>(part, bpart, dw, db, p.nslice, nw, Nout, nwb);
    return (int)hipGetLastError();
}

extern "C" int gf_linear_dw2(const void* dy, const void* x1, const void* x2, int K1, float* dw, float* db, void* ws,
                             int M, int Nout, int K, int dtype, void* stream) {
    if (M <= 0 || Nout <= 0 || K <= 0 || K1 <= 0 || K1 >= K) return GF_ERR_SHAPE;
    if (dtype != GF_BF16 || Nout % 128 || K1 % 128 || (K - K1) % 128 || x1 == nullptr || x2 == nullptr) return GF_ERR_UNSUPPORTED;
    return launch_dw_tr(dy, x1, dw, db, ws, M, Nout, K, reinterpret_cast<hipStream_t>(stream), x2, K1);
}

extern "C" int gf_linear_dw(const void* dy, const void* x, float* dw, float* db, void* ws,
                            int M, int Nout, int K, int dtype, void* stream) {
    if (M <= 0 || Nout <= 0 || K <= 0) return GF_ERR_SHAPE;
    const int align = dtype == GF_BF16 ? 8 : 4;
    if (Nout % align || K % align) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_dw<float>(dy, x, dw, db, ws, M, Nout, K, st);
    if (dtype == GF_BF16) return launch_dw_tr(dy, x, dw, db, ws, M, Nout, K, st);
    return GF_ERR_DTYPE;
}
