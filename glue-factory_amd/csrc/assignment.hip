// Assignment head kernels: row/column log-sum-exp, arg-max, materialisation and backward of the
// double-softmax log assignment with dustbins, all streaming S = a b^T through MFMA tiles so
// that no N x N similarity tensor is ever read from HBM.
//
// Replaces (reference) gluefactory/models/matchers/lightglue.py:256-268
// (sigmoid_log_double_softmax), :278-287 (MatchAssignment: the `sim` einsum), :293-309
// (filter_matches), :81-94 (TokenConfidence.loss arg-maxes) and the dense passes of
// gluefactory/models/utils/losses.py:6-73; gluestick.py:772-783 uses the same kernels with a
// bin column bias.
//
// Common structure: one workgroup = 4 waves owns 128 rows of the "owner" matrix (fragments kept
// in registers: D/16 k-steps), and streams the other matrix through LDS in 64-row tiles, double-buffered
// with register prefetch (stream_tiles: one barrier per tile; -20 % vs. the single-buffered form).  The
// tile is computed as C[tile_row][owner] so every reduction over the streamed axis is
// lane-local (see gf_common.h).  Kernels that write N x N data make the owner the CONTIGUOUS
// (column) index of the output so a half-wave writes 32 consecutive elements of one row.
#include <cstdlib>
#include "gf_common.h"
#include "gf_amd.h"

#ifndef GF_WRITE_SPLIT    // workgroups per column block of gf_assign_write (disjoint row ranges)
#define GF_WRITE_SPLIT 4
#endif
#ifndef GF_WRITE_ABL      // timing probes of gf_assign_write only: 1 no MFMA, 2 no stores, 4 no exp
#define GF_WRITE_ABL 0
#endif
#ifndef GF_BWD_SPLIT      // workgroups per column block of the dual-softmax backward (probe builds override it)
#define GF_BWD_SPLIT 4
#endif

namespace {

template <typename T, int D> struct ALay {
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int CPR = D / VEC;
    static constexpr int LDR = D + VEC;
    static constexpr int TILE = 64 * LDR;  // elements
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Double-buffered stream of 64-row tiles [s_begin, s_end) of the streamed matrix through LDS: the global loads
// of tile t+1 are in flight (registers) while tile t is consumed, ONE barrier per tile.  Loads are
// unconditional (rows clamped to Ns-1) so the compiler keeps counted waits.  `bias(si, v0, v1)` supplies two
// per-row floats staged next to the tile (every thread evaluates it on a clamped row; 64 threads store).
// body(tile, vec0, vec1, s0) consumes one tile.
template <typename T, int D, typename Bias, typename Body>
__device__ __forceinline__ void stream_tiles(T* tiles, float* vecs, const T* othp, int s_begin, int s_end, int Ns,
                                             Bias&& bias, Body&& body) {
    using L = ALay<T, D>;
    constexpr int NCH = 64 * L::CPR / 256;      // 16-byte chunks per thread and tile
    if (s_begin >= s_end) return;
    u32x4 rg[NCH];
    float bv0 = 0.f, bv1 = 0.f;
    auto load = [&](int s0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + 256 * i;
            const int r = c / L::CPR, cc = c % L::CPR;
            rg[i] = *reinterpret_cast<const u32x4*>(othp + (int64_t)min(s0 + r, Ns - 1) * D + cc * L::VEC);
        }
        bias(s0 + (int)(threadIdx.x & 63), bv0, bv1);
    };
    auto store = [&](int buf) {
        T* t = tiles + buf * L::TILE;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + 256 * i;
            const int r = c / L::CPR, cc = c % L::CPR;
            *reinterpret_cast<u32x4*>(t + r * L::LDR + cc * L::VEC) = rg[i];
        }
        if (threadIdx.x < 64) {
            vecs[buf * 128 + threadIdx.x] = bv0;
            vecs[buf * 128 + 64 + threadIdx.x] = bv1;
        }
    };
    load(s_begin);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int s0 = s_begin; s0 < s_end; s0 += 64, buf ^= 1) {
        const bool more = s0 + 64 < s_end;
        if (more) load(s0 + 64);
        body(tiles + buf * L::TILE, vecs + buf * 128, vecs + buf * 128 + 64, s0);
        if (more) store(buf ^ 1);
        __syncthreads();
    }
}

template <typename T, int D>
__device__ __forceinline__ void load_owner(Frag<T> (&f)[D / 16], const T* rowptr, int hi) {
#pragma unroll
    for (int s = 0; s < D / 16; ++s) f[s] = ld_frag8(rowptr + 16 * s + 8 * hi);
}

template <typename T, int D>
__device__ __forceinline__ void mma_tile(f32x16& acc, const T* lds, int i0, const Frag<T> (&f)[D / 16],
                                         int l31, int hi) {
    using L = ALay<T, D>;
    const T* base = lds + (i0 + l31) * L::LDR + 8 * hi;
#pragma unroll
    for (int s = 0; s < D / 16; ++s) mma32(acc, ld_frag8(base + 16 * s), f[s]);
}

struct HeadParams {
    const void* own;   // owner matrix  [B, No, D]
    const void* oth;   // streamed matrix [B, Ns, D]
    int B, No, Ns;
    const float* sbias;  // per streamed row  [B, Ns] or null
    const float* obias;  // per owner row     [B, No] or null
    float alpha;
    // outputs / extra inputs (kernel specific)
    float* f0; float* f1; int64_t* i0;
    const float* g0; const float* g1; const float* g2; const float* g3;
    const float* G; int64_t ldg; float galpha; float corner;
    void* out;
    int nsplit;          // workgroups per owner block along the streamed dimension (kernels with disjoint outputs)
};

#define GF_HEAD_PROLOGUE(T, D)                                                                   \
    using L = ALay<T, D>;                                                                         \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                   \
    T* tiles = reinterpret_cast<T*>(smem);                                                        \
    float* vecs = reinterpret_cast<float*>(tiles + 2 * L::TILE);                                  \
    const int nob = (p.No + 127) / 128;                                                           \
    const int nsp = p.nsplit > 1 ? p.nsplit : 1;                                                  \
    const int lb_ = xcd_remap(blockIdx.x, nob * p.B * nsp);                                       \
    const int split = lb_ % nsp, lb = lb_ / nsp;                                                  \
    const int ob = lb % nob, b = lb / nob;                                                        \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                   \
    const int l31 = lane & 31, hi = lane >> 5;                                                    \
    const int orow = ob * 128 + wave * 32 + l31;                                                  \
    const int old_ = min(orow, p.No - 1);                                                         \
    const T* ownp = reinterpret_cast<const T*>(p.own) + (int64_t)b * p.No * D;                    \
    const T* othp = reinterpret_cast<const T*>(p.oth) + (int64_t)b * p.Ns * D;                    \
    Frag<T> of[D / 16];                                                                           \
    load_owner<T, D>(of, ownp + (int64_t)old_ * D, hi);                                           \
    (void)split;

// Running (max, argmax) of one lane.  A lane visits its streamed rows in ASCENDING index order (tiles ascend, and inside a
// tile the offset 32 kb + 8 g + e ascends with (kb, g, e); + 4 hi is fixed per lane), so a strict > keeps the lowest index
// of a tie: one compare + two selects per score, the offset an inline constant; the tile's winner meets the running one
// once per tile, and the full tie rule is only needed where the two half-wave partners meet.
struct TileArg {
    float v; int off;
    __device__ __forceinline__ void reset() { v = -INFINITY; off = 0; }
    __device__ __forceinline__ void see(float x, int off_const) {
        const bool gt = x > v;
        v = gt ? x : v;
        off = gt ? off_const : off;
    }
    __device__ __forceinline__ void merge(int s0, int hi, float& best, int& bidx) const {
        const bool gt = v > best;
        best = gt ? v : best;
        bidx = gt ? s0 + 4 * hi + off : bidx;
    }
};

// lse[b, owner] = log sum_s exp(own . oth_s + sbias_s)
template <typename T, int D, bool HAS_BIAS>
__global__ __launch_bounds__(256) void rows_lse_kernel(HeadParams p) {
    GF_HEAD_PROLOGUE(T, D)
    float m = GF_NEG_BIG, lsum = 0.f;
    const float* sb = p.sbias ? p.sbias + (int64_t)b * p.Ns : nullptr;
    auto bias = [&](int si, float& v0, float& v1) {
        const float x = sb ? sb[min(si, p.Ns - 1)] * GF_LOG2E : 0.f;
        v0 = si < p.Ns ? x : -INFINITY;
        v1 = 0.f;
    };
    auto body = [&](const T* tile, const float* vec0, const float*, int s0) {
        f32x16 s[2];
        float mx = GF_NEG_BIG;
        // without a bias (LightGlue's double softmax) the scores stay RAW: the maximum is taken on them and log2(e) and the
        // shift ride in the fma in front of exp2 -- max3, fma, exp2, add = 3.5 instructions per score instead of 5
        const bool plain = !HAS_BIAS && s0 + 64 <= p.Ns;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            mma_tile<T, D>(s[kb], tile, kb * 32, of, l31, hi);
            if (plain) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 b4 = *reinterpret_cast<const f32x4*>(vec0 + kb * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = s[kb][4 * g + e] * GF_LOG2E + b4[e];
                        s[kb][4 * g + e] = x;
                        mx = fmaxf(mx, x);
                    }
                }
            }
        }
        if (plain) mx *= GF_LOG2E;
        mx = fmaxf(mx, xhalf(mx));
        const float mnew = fmaxf(m, mx);
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ps += plain ? fast_exp2(fmaf(s[kb][r], GF_LOG2E, -mnew)) : fast_exp2(s[kb][r] - mnew);
        lsum = lsum * fast_exp2(m - mnew) + ps;
        m = mnew;
    };
    stream_tiles<T, D>(tiles, vecs, othp, 0, p.Ns, p.Ns, bias, body);
    lsum += xhalf(lsum);
    if (orow < p.No && hi == 0) p.f0[(int64_t)b * p.No + orow] = (m + fast_log2(lsum)) * GF_LN2;
}

// max / argmax over streamed rows of alpha * own.oth_s + sbias_s
template <typename T, int D>
__global__ __launch_bounds__(256) void rows_argmax_kernel(HeadParams p) {
    GF_HEAD_PROLOGUE(T, D)
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const float* sb = p.sbias ? p.sbias + (int64_t)b * p.Ns : nullptr;
    auto bias = [&](int si, float& v0, float& v1) {
        const float x = sb ? sb[min(si, p.Ns - 1)] : 0.f;
        v0 = si < p.Ns ? x : -INFINITY;
        v1 = 0.f;
    };
    auto body = [&](const T* tile, const float* vec0, const float*, int s0) {
        TileArg ta;
        ta.reset();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            mma_tile<T, D>(s, tile, kb * 32, of, l31, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(vec0 + kb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ta.see(p.alpha * s[4 * g + e] + b4[e], kb * 32 + 8 * g + e);
                }
            }
        }
        ta.merge(s0, hi, best, bidx);
    };
    stream_tiles<T, D>(tiles, vecs, othp, 0, p.Ns, p.Ns, bias, body);
    float ob_ = xhalf(best);
    int oi = __shfl_xor(bidx, 32);
    if (ob_ > best || (ob_ == best && oi < bidx)) { best = ob_; bidx = oi; }
    if (orow < p.No && hi == 0) {
        p.f0[(int64_t)b * p.No + orow] = best;
        p.i0[(int64_t)b * p.No + orow] = (bidx == 0x7fffffff) ? 0 : bidx;
    }
}

// One pass over the streamed rows for BOTH statistics of a LightGlue head row:
//   lse   = log sum_s exp(own . oth_s)                                     (WITH_LSE; -> f1)
//   max / argmax over s of alpha * own . oth_s + logsigmoid(g0_s) - g1_s   (-> f0, i0)
// (g0 = matchability logits of the streamed side, g1 = its normaliser from the previous pass.)
template <typename T, int D, bool WITH_LSE>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void rows_lse_argmax_kernel(HeadParams p) {     // bf16: two waves per SIMD
    GF_HEAD_PROLOGUE(T, D)
    float m = GF_NEG_BIG, lsum = 0.f;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const float* zb = p.g0 + (int64_t)b * p.Ns;
    const float* nb = p.g1 + (int64_t)b * p.Ns;
    auto bias = [&](int si, float& v0, float& v1) {
        const int sc = min(si, p.Ns - 1);
        const float z = zb[sc];
        const float x = fminf(z, 0.f) - log1pf(__expf(-fabsf(z))) - nb[sc];
        v0 = si < p.Ns ? x : -INFINITY;
        v1 = si < p.Ns ? 0.f : -INFINITY;                         // rows past Ns never enter the lse
    };
    auto body = [&](const T* tile, const float* vec0, const float* vec1, int s0) {
        f32x16 s[2];
        float mx = GF_NEG_BIG;
        TileArg ta;
        ta.reset();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            mma_tile<T, D>(s[kb], tile, kb * 32, of, l31, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(vec0 + kb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float raw = s[kb][4 * g + e];
                    ta.see(p.alpha * raw + b4[e], kb * 32 + 8 * g + e);
                    if (WITH_LSE) mx = fmaxf(mx, raw);            // the lse works on the RAW scores (see rows_lse_kernel)
                }
            }
        }
        ta.merge(s0, hi, best, bidx);
        if (WITH_LSE) {
            if (s0 + 64 > p.Ns) {                                 // ragged last tile
                mx = GF_NEG_BIG;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v4 = *reinterpret_cast<const f32x4*>(vec1 + kb * 32 + 8 * g + 4 * hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s[kb][4 * g + e] += v4[e];
                            mx = fmaxf(mx, s[kb][4 * g + e]);
                        }
                    }
            }
            mx *= GF_LOG2E;
            mx = fmaxf(mx, xhalf(mx));
            const float mnew = fmaxf(m, mx);
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += fast_exp2(fmaf(s[kb][r], GF_LOG2E, -mnew));
            lsum = lsum * fast_exp2(m - mnew) + ps;
            m = mnew;
        }
    };
    stream_tiles<T, D>(tiles, vecs, othp, 0, p.Ns, p.Ns, bias, body);
    float ob_ = xhalf(best);
    int oi = __shfl_xor(bidx, 32);
    if (ob_ > best || (ob_ == best && oi < bidx)) { best = ob_; bidx = oi; }
    if (WITH_LSE) lsum += xhalf(lsum);
    if (orow < p.No && hi == 0) {
        p.f0[(int64_t)b * p.No + orow] = best;
        p.i0[(int64_t)b * p.No + orow] = (bidx == 0x7fffffff) ? 0 : bidx;
        if (WITH_LSE) p.f1[(int64_t)b * p.No + orow] = (m + fast_log2(lsum)) * GF_LN2;
    }
}

// out[b, s, o] = alpha * oth_s . own_o + sbias_s + obias_o  (+ dustbin row / column / corner)
// owner = column index of `out` ([B, Ns+1, No+1]); streamed = row index.
template <typename T, int D>
__global__ __launch_bounds__(256) void assign_write_kernel(HeadParams p) {
    GF_HEAD_PROLOGUE(T, D)
#ifdef GF_WRITE_LDPAD      // timing probe only (tools/probe/time_assign.py): rows padded to a multiple of GF_WRITE_LDPAD floats
    const int64_t ldo = (p.No + 1 + GF_WRITE_LDPAD - 1) / GF_WRITE_LDPAD * GF_WRITE_LDPAD;
    float* out = reinterpret_cast<float*>(p.out) + (int64_t)b * (p.Ns + 1) * ldo;
#else
    float* out = reinterpret_cast<float*>(p.out) + (int64_t)b * (p.Ns + 1) * (p.No + 1);
    const int64_t ldo = p.No + 1;
#endif
    const float ocb = p.obias ? p.obias[(int64_t)b * p.No + old_] : 0.f;
    const float* sb = p.sbias ? p.sbias + (int64_t)b * p.Ns : nullptr;
    auto bias = [&](int si, float& v0, float& v1) {
        v0 = sb ? sb[min(si, p.Ns - 1)] : 0.f;
        v1 = 0.f;
    };
    float esum = 0.f;       // sum of exp(out) over this lane's entries (rows < Ns, all columns): the "row_norm" statistic
    auto body = [&](const T* tile, const float* vec0, const float*, int s0) {
        TileArg ta;
        ta.reset();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            if (!(GF_WRITE_ABL & 1)) mma_tile<T, D>(s, tile, kb * 32, of, l31, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(vec0 + kb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int si = s0 + kb * 32 + 8 * g + 4 * hi + e;
                    if (si < p.Ns && orow < p.No) {
                        const float v = p.alpha * s[4 * g + e] + b4[e] + ocb;
                        if (!(GF_WRITE_ABL & 2)) out[(int64_t)si * ldo + orow] = v;
                        else if (v == 12345.678f) out[0] = v;
                        if (p.f0 && !(GF_WRITE_ABL & 4)) esum += fast_exp2(v * GF_LOG2E);
                    }
                }
            }
        }
    };
    // the streamed rows are divided among nsplit workgroups (disjoint row ranges of the same column panel): 512 workgroups of
    // 2048 x 128 outputs each left three quarters of the chip's store queues idle (0.21 ms = 2.5 TB/s at B=32, N=2048)
    const int per = ((p.Ns + 63) / 64 + nsp - 1) / nsp * 64;
    const int s_begin = split * per, s_end = min(p.Ns, s_begin + per);
    if (s_begin < s_end) stream_tiles<T, D>(tiles, vecs, othp, s_begin, s_end, p.Ns, bias, body);
    // dustbins: last row for the owned columns, last column by the first block, corner once
    if (split == 0 && orow < p.No && hi == 0) out[(int64_t)p.Ns * ldo + orow] = p.g1 ? p.g1[(int64_t)b * p.No + orow] : 0.f;
    if (ob == 0 && split == 0) {
        for (int si = threadIdx.x; si < p.Ns; si += 256) {
            const float v = p.g0 ? p.g0[(int64_t)b * p.Ns + si] : 0.f;
            out[(int64_t)si * ldo + p.No] = v;
            if (p.f0) esum += fast_exp2(v * GF_LOG2E);
        }
        if (threadIdx.x == 0) out[(int64_t)p.Ns * ldo + p.No] = p.corner;
    }
    if (p.f0) {
        esum = wave_allsum(esum);
        if ((threadIdx.x & 63) == 0) atomicAdd(p.f0 + b, esum);
    }
}

// dS[b, s, o] = galpha * G[b,s,o] + exp(S - r_s) * gr_s + exp(S - c_o) * gc_o   (S = oth_s . own_o)
template <typename T, int D>
__global__ __launch_bounds__(256) void dual_softmax_bwd_kernel(HeadParams p) {
    GF_HEAD_PROLOGUE(T, D)
    T* dS = reinterpret_cast<T*>(p.out) + (int64_t)b * p.Ns * p.No;
    const float c2 = p.g1[(int64_t)b * p.No + old_] * GF_LOG2E;   // column normaliser of the owner
    const float gco = p.g3[(int64_t)b * p.No + old_];
    const float* Gb = p.G ? p.G + (int64_t)b * (p.Ns + 1) * p.ldg : nullptr;
    // the streamed rows are divided among nsplit workgroups (disjoint dS tiles): more waves in flight for a
    // kernel that is bound by store issue and latency, not by its MFMA work
    const int per = ((p.Ns + 63) / 64 + nsp - 1) / nsp * 64;
    const int s_begin = split * per, s_end = min(p.Ns, s_begin + per);
    const float* rb = p.g0 + (int64_t)b * p.Ns;
    const float* grb = p.g2 + (int64_t)b * p.Ns;
    auto bias = [&](int si, float& v0, float& v1) {
        const int sc = min(si, p.Ns - 1);
        const float r_ = rb[sc] * GF_LOG2E, g_ = grb[sc];
        v0 = si < p.Ns ? r_ : INFINITY;      // r_s
        v1 = si < p.Ns ? g_ : 0.f;           // gr_s
    };
    auto body = [&](const T* tile, const float* vec0, const float* vec1, int s0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            mma_tile<T, D>(s, tile, kb * 32, of, l31, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 r4 = *reinterpret_cast<const f32x4*>(vec0 + kb * 32 + 8 * g + 4 * hi);
                f32x4 g4 = *reinterpret_cast<const f32x4*>(vec1 + kb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int si = s0 + kb * 32 + 8 * g + 4 * hi + e;
                    if (si < p.Ns && orow < p.No) {
                        float x = s[4 * g + e] * GF_LOG2E;
                        float v = fast_exp2(x - r4[e]) * g4[e] + fast_exp2(x - c2) * gco;
                        if (Gb) v += p.galpha * Gb[(int64_t)si * p.ldg + orow];
                        dS[(int64_t)si * p.No + orow] = from_f32<T>(v);
                    }
                }
            }
        }
    };
    stream_tiles<T, D>(tiles, vecs, othp, s_begin, s_end, p.Ns, bias, body);
}

__global__ void filter_matches_kernel(const float* max0, const int64_t* arg0, const int64_t* arg1, float th,
                                      int64_t* m0, int64_t* m1, float* s0, float* s1, int B, int M, int N) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)B * (M + N);
    if (t >= tot) return;
    int b = t / (M + N);
    int k = t % (M + N);
    const int64_t* a0 = arg0 + (int64_t)b * M;
    const int64_t* a1 = arg1 + (int64_t)b * N;
    if (k < M) {
        int64_t j = a0[k];
        bool mutual = (a1[j] == k);
        float sc = mutual ? __expf(max0[(int64_t)b * M + k]) : 0.f;
        s0[(int64_t)b * M + k] = sc;
        m0[(int64_t)b * M + k] = (mutual && sc > th) ? j : -1;
    } else {
        int j = k - M;
        int64_t i = a1[j];
        bool mutual = (a0[i] == j);
        float sc = mutual ? __expf(max0[(int64_t)b * M + i]) : 0.f;
        s1[(int64_t)b * N + j] = sc;
        m1[(int64_t)b * N + j] = (mutual && sc > th) ? i : -1;
    }
}

template <typename T, int D> size_t head_lds() { return 2 * ALay<T, D>::TILE * sizeof(T) + 256 * sizeof(float); }

template <typename K> int set_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

enum { K_LSE, K_ARGMAX, K_WRITE, K_BWD, K_LSEARG, K_ZNARG };

template <typename T, int D> int launch_td(int which, const HeadParams& p, hipStream_t st) {
    const int total = ((p.No + 127) / 128) * p.B * (p.nsplit > 1 ? p.nsplit : 1);
    const size_t lds = head_lds<T, D>();
#define GF_LAUNCH(kern)                                             \
    {                                                               \
        if (int e = set_lds(kern<T, D>, lds)) return e;             \
        kern<T, D><<<dim3(total), dim3(256), lds, st>>>(p);         \
        return (int)hipGetLastError();                              \
    }
    switch (which) {
        case K_LSE:
            if (p.sbias) {
                if (int e = set_lds(rows_lse_kernel<T, D, true>, lds)) return e;
                rows_lse_kernel<T, D, true><<<dim3(total), dim3(256), lds, st>>>(p);
            } else {
                if (int e = set_lds(rows_lse_kernel<T, D, false>, lds)) return e;
                rows_lse_kernel<T, D, false><<<dim3(total), dim3(256), lds, st>>>(p);
            }
            return (int)hipGetLastError();
        case K_ARGMAX: GF_LAUNCH(rows_argmax_kernel)
        case K_WRITE: GF_LAUNCH(assign_write_kernel)
        case K_BWD: GF_LAUNCH(dual_softmax_bwd_kernel)
        default: break;
    }
#undef GF_LAUNCH
#define GF_LAUNCH2(flag)                                                              \
    {                                                                                 \
        if (int e = set_lds(rows_lse_argmax_kernel<T, D, flag>, lds)) return e;       \
        rows_lse_argmax_kernel<T, D, flag><<<dim3(total), dim3(256), lds, st>>>(p);   \
        return (int)hipGetLastError();                                                \
    }
    if (which == K_LSEARG) GF_LAUNCH2(true)
    GF_LAUNCH2(false)
#undef GF_LAUNCH2
}

template <typename T> int launch_t(int which, const HeadParams& p, int D, hipStream_t st) {
    switch (D) {
        case 64: return launch_td<T, 64>(which, p, st);
        case 128: return launch_td<T, 128>(which, p, st);
        case 256: return launch_td<T, 256>(which, p, st);
        default: return GF_ERR_UNSUPPORTED;
    }
}

int launch(int which, const HeadParams& p, int D, int dtype, void* stream) {
    if (p.B <= 0 || p.No <= 0 || p.Ns <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_t<float>(which, p, D, st);
    if (dtype == GF_BF16) return launch_t<bf16_t>(which, p, D, st);
    return GF_ERR_DTYPE;
}

}  // namespace

extern "C" int gf_rows_lse(const void* a, const void* b, const float* colbias, float* lse,
                           int B, int M, int N, int D, int dtype, void* stream) {
    HeadParams p = {};
    p.own = a; p.oth = b; p.B = B; p.No = M; p.Ns = N; p.sbias = colbias; p.f0 = lse;
    return launch(K_LSE, p, D, dtype, stream);
}

extern "C" int gf_rows_argmax(const void* a, const void* b, const float* colbias, float alpha,
                              float* rowmax, int64_t* rowarg,
                              int B, int M, int N, int D, int dtype, void* stream) {
    HeadParams p = {};
    p.own = a; p.oth = b; p.B = B; p.No = M; p.Ns = N; p.sbias = colbias; p.alpha = alpha;
    p.f0 = rowmax; p.i0 = rowarg;
    return launch(K_ARGMAX, p, D, dtype, stream);
}

extern "C" int gf_rows_lse_argmax(const void* a, const void* b, const float* bias_z, const float* bias_n,
                                  float alpha, float* lse, float* rowmax, int64_t* rowarg,
                                  int B, int M, int N, int D, int dtype, void* stream) {
    if (bias_z == nullptr || bias_n == nullptr || rowmax == nullptr || rowarg == nullptr) return GF_ERR_SHAPE;
    HeadParams p = {};
    p.own = a; p.oth = b; p.B = B; p.No = M; p.Ns = N; p.g0 = bias_z; p.g1 = bias_n; p.alpha = alpha;
    p.f0 = rowmax; p.i0 = rowarg; p.f1 = lse;
    return launch(lse ? K_LSEARG : K_ZNARG, p, D, dtype, stream);
}

extern "C" int gf_assign_write(const void* a, const void* b, const float* rowbias, const float* colbias,
                               const float* bin_col, const float* bin_row, float alpha, float corner,
                               float* out, float* expsum, int B, int M, int N, int D, int dtype, void* stream) {
    // owner = columns (b rows), streamed = rows (a rows)
    HeadParams p = {};
    p.own = b; p.oth = a; p.B = B; p.No = N; p.Ns = M; p.sbias = rowbias; p.obias = colbias;
    p.alpha = alpha; p.corner = corner; p.g0 = bin_col; p.g1 = bin_row; p.out = out; p.f0 = expsum;
    p.nsplit = GF_WRITE_SPLIT;
    if (expsum)
        if (hipError_t e = gf_zero_f32(expsum, (size_t)B, reinterpret_cast<hipStream_t>(stream))) return (int)e;
    return launch(K_WRITE, p, D, dtype, stream);
}

extern "C" int gf_dual_softmax_bwd(const void* a, const void* b, const float* r, const float* c,
                                   const float* gr, const float* gc, const float* G, int64_t ldg,
                                   float galpha, void* dS, int B, int M, int N, int D, int dtype,
                                   void* stream) {
    HeadParams p = {};
    p.own = b; p.oth = a; p.B = B; p.No = N; p.Ns = M;
    p.g0 = r; p.g1 = c; p.g2 = gr; p.g3 = gc; p.G = G; p.ldg = ldg; p.galpha = galpha; p.out = dS;
    p.nsplit = GF_BWD_SPLIT;                // 227 -> 172 us at B=32, N=2048 (1 -> 4 workgroups per column block)
    return launch(K_BWD, p, D, dtype, stream);
}

extern "C" int gf_filter_matches(const float* max0, const int64_t* arg0, const int64_t* arg1, float th,
                                 int64_t* m0, int64_t* m1, float* s0, float* s1,
                                 int B, int M, int N, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    int64_t tot = (int64_t)B * (M + N);
    int blocks = (int)((tot + 255) / 256);
    filter_matches_kernel<<<dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
        max0, arg0, arg1, th, m0, m1, s0, s1, B, M, N);
    return (int)hipGetLastError();
}
