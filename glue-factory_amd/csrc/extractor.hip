// Memory-bound tails of the frozen SuperPoint extractor (the convolutions themselves stay on the stock
// library, by design).  Reference: gluefactory/models/extractors/superpoint_open.py
//   * VGGBlock = Conv2d -> ReLU -> BatchNorm2d(eval) [-> MaxPool2d(2,2)]  (:37-75, :98-111)
//       one pass over the channels-last activation instead of bias-add, clamp, batch-norm and pool passes;
//   * simple_nms (:19-34): five 2r+1 max-pools and a dozen elementwise passes over the [B,H,W] score map
//       become one kernel working on a register tile (64 columns x 48 + 10r rows per wave) with a 5r halo.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct V16 { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void ld_vec(float (&x)[16 / sizeof(T)], const T* p) {
    union { u32x4 u; T e[16 / sizeof(T)]; } v;
    v.u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) x[e] = to_f32(v.e[e]);
}
template <typename T> __device__ __forceinline__ void st_vec(T* p, const float (&x)[16 / sizeof(T)]) {
    union { u32x4 u; T e[16 / sizeof(T)]; } v;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) v.e[e] = from_f32<T>(x[e]);
    *reinterpret_cast<u32x4*>(p) = v.u;
}

// y[p, c] = act(x[p, c] + bias[c]) * scale[c] + shift[c]   (channels-last: c fastest; in place when y == x)
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void bias_act_bn_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int64_t nvec, int C) {
    constexpr int VEC = V16<T>::N;
    const int cv = C / VEC;                                   // vectors per pixel
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * VEC;
        float v[VEC];
        ld_vec<T>(v, x + i * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float t = v[e] + bias[c0 + e];
            if (RELU) t = fmaxf(t, 0.f);
            v[e] = t * scale[c0 + e] + shift[c0 + e];
        }
        st_vec<T>(y + i * VEC, v);
    }
}

// same followed by MaxPool2d(2, 2): x [B,H,W,C] -> y [B,H/2,W/2,C]
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void bias_act_bn_pool_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                               const float* __restrict__ bias, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int64_t nvec_out,
                                                               int H, int W, int C) {
    constexpr int VEC = V16<T>::N;
    const int cv = C / VEC, Wo = W / 2, Ho = H / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_out; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * VEC;
        int64_t p = i / cv;
        const int xo = (int)(p % Wo);
        p /= Wo;
        const int yo = (int)(p % Ho);
        const int64_t b = p / Ho;
        const T* src = x + (((b * H + 2 * yo) * W + 2 * xo) * (int64_t)C + c0);
        float best[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) best[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[VEC];
                ld_vec<T>(v, src + ((int64_t)dy * W + dx) * C);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float t = v[e] + bias[c0 + e];
                    if (RELU) t = fmaxf(t, 0.f);
                    best[e] = fmaxf(best[e], t * scale[c0 + e] + shift[c0 + e]);
                }
            }
        st_vec<T>(y + i * VEC, best);
    }
}

// ---- fused simple_nms ----------------------------------------------------------------------------
// out = where(M, s, 0) with  M0 = (s == P(s));  twice: supp = P(M) > 0, ss = where(supp, 0, s),
// M |= (ss == P(ss)) & ~supp;  P = max over the (2r+1)^2 window clipped to the image.
// Every pooling is separable; pixels outside the image are -inf for the score pools and "no maximum" for the mask
// pools, which is what clipping the window means.
// ---- first VGG block on a 1-channel image ---------------------------------------------------------------------
// backbone.0.0 (superpoint_open.py:98-100: Conv2d(1, C, 3, padding=1) -> ReLU -> BatchNorm2d(eval)) is not
// GEMM-shaped -- 9 multiply-adds per output value -- but its OUTPUT is the largest tensor of the extractor
// (B x H x W x C: 8.6 GB for 64 images of 1024^2 in bf16).  Through the library it costs a convolution that writes
// that tensor plus the tail pass that reads and rewrites it; here ONE kernel reads the image and writes the finished
// channels-last activation once.  A workgroup owns a 32 x 32 pixel tile (fp32 copy of the 34 x 34 input window in
// LDS, zero outside the image); lane = (pixel column, group of 8 channels) keeps its 8 x 9 weights and the
// per-channel bias / scale / shift in registers; a wave's store covers 8 pixels x 128 contiguous bytes (C = 64 bf16).
// The products are fp32 on the T-rounded image and weights (what the library computes from the same operands);
// bias, ReLU and the batch-norm affine act on the fp32 sum -- one rounding.
#ifndef C1_TY
#define C1_TY 64            // tile rows and 32-pixel passes per row of a workgroup (tools/probe/time_conv1.py, 64 x 1024^2: 32x1 1.67 ms, 64x1 1.59, 128x1 1.60, 256x1 1.63, 8x4 1.75, 2x16 1.81)
#endif
#ifndef C1_NX
#define C1_NX 1
#endif
template <typename T, bool RELU, int C>
__global__ __launch_bounds__(256) void conv1_fused_kernel(const T* __restrict__ img, const T* __restrict__ wgt,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ out, int H, int W) {
    constexpr int CG = C / 8;                 // lanes per pixel (8 channels each)
    constexpr int PX = 256 / CG;              // pixels per pass of the workgroup (32)
    constexpr int TY = C1_TY, NX = C1_NX, TW = PX * NX;      // tile: TY rows x NX passes of PX pixels
    __shared__ float tile[(TY + 2) * (TW + 2)];
    const int b = blockIdx.z, y0 = blockIdx.y * TY, x0 = blockIdx.x * TW;
    const T* im = img + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < (TY + 2) * (TW + 2); i += 256) {
        const int y = y0 - 1 + i / (TW + 2), x = x0 - 1 + i % (TW + 2);
        tile[i] = (y >= 0 && y < H && x >= 0 && x < W) ? to_f32(im[(int64_t)y * W + x]) : 0.f;
    }
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
    if constexpr (sizeof(T) == 2) {
        // bf16: the nine products of a channel as five v_dot2c_f32_bf16 (pairs of taps; image and weights ARE bf16, the products
        // are exact and the sums fp32 as before) on the bias as the initial value: 7 VALU instructions per output value where
        // the fma form took 12.  Measured at 64 x 1024^2 (8.6 GB written; the box fills at 6.9 TB/s = 1.25 ms): 1.75 -> 1.67 ms,
        // with 64-row tiles 1.59 ms = 5.4 TB/s -- the kernel is bound by its store stream, not by these instructions
        typedef __bf16 c1_bf16x2 __attribute__((ext_vector_type(2)));
        c1_bf16x2 wp[8][5];
        float bi[8], sc[8], sh[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const T* wc = wgt + (8 * cg + c) * 9;
#pragma unroll
            for (int j = 0; j < 4; ++j) wp[c][j] = c1_bf16x2{wc[2 * j], wc[2 * j + 1]};
            wp[c][4] = c1_bf16x2{wc[8], (T)0.f};
            bi[c] = bias[8 * cg + c];
            sc[c] = scale[8 * cg + c];
            sh[c] = shift[8 * cg + c];
        }
        __syncthreads();
        for (int it = 0; it < TY * NX; ++it) {
            const int ly = it / NX, lx = (it % NX) * PX + px;
            const int y = y0 + ly, x = x0 + lx;
            if (y >= H) break;
            float v[10];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[3 * dy + dx] = tile[(ly + dy) * (TW + 2) + lx + dx];
            v[9] = 0.f;
            c1_bf16x2 vp[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) vp[j] = c1_bf16x2{(T)v[2 * j], (T)v[2 * j + 1]};     // exact: the tile holds bf16 values
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float a = bi[c];
#pragma unroll
                for (int j = 0; j < 5; ++j) a = __builtin_amdgcn_fdot2_f32_bf16(wp[c][j], vp[j], a, false);
                if (RELU) a = fmaxf(a, 0.f);
                o[c] = fmaf(a, sc[c], sh[c]);
            }
            if (x < W) st_vec<T>(out + (((int64_t)b * H + y) * W + x) * C + 8 * cg, o);
        }
    } else {
        float w[8][9], bi[8], sc[8], sh[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) w[c][t] = to_f32(wgt[(8 * cg + c) * 9 + t]);
            bi[c] = bias[8 * cg + c];
            sc[c] = scale[8 * cg + c];
            sh[c] = shift[8 * cg + c];
        }
        __syncthreads();
        for (int it = 0; it < TY * NX; ++it) {
            const int ly = it / NX, lx = (it % NX) * PX + px;
            const int y = y0 + ly, x = x0 + lx;
            if (y >= H) break;
            float v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[3 * dy + dx] = tile[(ly + dy) * (TW + 2) + lx + dx];
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) a = fmaf(w[c][t], v[t], a);
                a += bi[c];
                if (RELU) a = fmaxf(a, 0.f);
                o[c] = fmaf(a, sc[c], sh[c]);
            }
            if (x < W) {
                T* dst = out + (((int64_t)b * H + y) * W + x) * C + 8 * cg;
                st4(dst, o[0], o[1], o[2], o[3]);
                st4(dst + 4, o[4], o[5], o[6], o[7]);
            }
        }
    }
}

// ---- register-resident NMS ------------------------------------------------------------------------------------------
// (Round 2's kernel kept five planes of a (64 + 10 R)^2 tile in 124 KB of LDS -- one workgroup per CU, ~20 barriers per
// tile -- and ran at 1/4 of its own LDS-bandwidth bound: 1.46 ms on 64 x 1024^2; this one 0.52 ms.)  A WAVE owns a strip of 64 image columns (lane = column)
// and a tile of NR = NMS_RT + 10 R rows of it in REGISTERS (sc[NR]): a (2R+1)^2 max-pool is a horizontal pass over lanes
// (window maximum by doubling: m2, m4, .. by __shfl_down, then two shifted copies) and a vertical pass over the register
// array (the same doubling, compile-time indices, in place); the two pools of the 0/1 masks are bit operations on a
// 128-bit row mask per lane (dilation = ORs of shifts, across lanes ORs of shuffles: ~40 instructions for the whole tile
// instead of 9 per pixel).  Every pool invalidates R more columns / rows at each side: after the 5 pools of simple_nms
// (superpoint_open.py:19-34) the inner 64 - 10 R columns and NMS_RT rows are exact, ties included (plain float ==).
// No LDS, no barriers; waves are independent.
#ifndef NMS_RT_V
#define NMS_RT_V 48             // 48 rows + 30 halo rows at R = 3: 216 VGPRs (R = 4: 247) -> two waves per SIMD;
#endif                          // 64 needs 308 (one wave: 0.73 ms instead of 0.52 on 64 x 1024^2), 32 wastes half the rows on halo
#ifndef NMS_WPS
#define NMS_WPS 2
#endif
constexpr int NMS_RT = NMS_RT_V;           // output rows per wave tile

struct RowMask {                           // one bit per tile row of this lane's column
    unsigned long long lo, hi;
    __device__ __forceinline__ RowMask shl(int k) const { return {lo << k, (hi << k) | (lo >> (64 - k))}; }
    __device__ __forceinline__ RowMask shr(int k) const { return {(lo >> k) | (hi << (64 - k)), hi >> k}; }
    __device__ __forceinline__ RowMask operator|(const RowMask& o) const { return {lo | o.lo, hi | o.hi}; }
};
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src_lane) {
    const int a = __shfl((int)(unsigned)v, src_lane), b = __shfl((int)(unsigned)(v >> 32), src_lane);
    return ((unsigned long long)(unsigned)b << 32) | (unsigned)a;
}

// plain v_max_f32: fmaxf() in IEEE mode canonicalises every operand that comes out of a shuffle or a load first (one more
// v_max per operand); the scores are never NaN here and -inf orders correctly
__device__ __forceinline__ float vmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float bperm(int byte_addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

template <int R, int NR>
__device__ __forceinline__ void pool_rows_cols(float (&p)[NR], int lane) {
    constexpr int Wn = 2 * R + 1;
    constexpr int P = Wn >= 8 ? 8 : (Wn >= 4 ? 4 : 2);       // largest power of two <= window
    constexpr int OFF = P - 1 - R;                            // window [i-R, i+R] = m_P[i-R] U m_P[i-OFF], m_P[i] = max x[i .. i+P-1]
    // horizontal (across lanes), RB rows at a time so that RB shuffles are in flight together; a clamped source lane puts
    // garbage only into the columns this pool invalidates anyway
    constexpr int RB = 8;
    const int a1 = min(lane + 1, 63) * 4, a2 = min(lane + 2, 63) * 4, a4 = min(lane + 4, 63) * 4;
    const int ar = max(lane - R, 0) * 4, ao = max(lane - OFF, 0) * 4;
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += RB) {
        float t[RB];
#pragma unroll
        for (int sft = 1; sft < P; sft <<= 1) {
            const int ad = sft == 1 ? a1 : (sft == 2 ? a2 : a4);
#pragma unroll
            for (int j = 0; j < RB; ++j) if (r0 + j < NR) t[j] = bperm(ad, p[r0 + j]);
#pragma unroll
            for (int j = 0; j < RB; ++j) if (r0 + j < NR) p[r0 + j] = vmax(p[r0 + j], t[j]);
        }
        float u[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) if (r0 + j < NR) { t[j] = bperm(ar, p[r0 + j]); if (OFF != 0) u[j] = bperm(ao, p[r0 + j]); }
#pragma unroll
        for (int j = 0; j < RB; ++j) if (r0 + j < NR) p[r0 + j] = vmax(t[j], OFF != 0 ? u[j] : p[r0 + j]);
    }
    // vertical (register array), in place: doubling upwards, then the two shifted copies from the bottom row down
#pragma unroll
    for (int sft = 1; sft < P; sft <<= 1)
#pragma unroll
        for (int r = 0; r + sft < NR; ++r) p[r] = vmax(p[r], p[r + sft]);
#pragma unroll
    for (int r = NR - 1; r >= 0; --r) p[r] = vmax(p[r - R >= 0 ? r - R : 0], p[r - OFF >= 0 ? r - OFF : 0]);
}

template <int R>
__device__ __forceinline__ RowMask dilate(RowMask m, int lane) {
    RowMask v = m;
#pragma unroll
    for (int k = 1; k <= R; ++k) v = v | m.shl(k) | m.shr(k);
    RowMask h = v;
#pragma unroll
    for (int k = 1; k <= R; ++k) {
        const int up = max(lane - k, 0), dn = min(lane + k, 63);
        // (a clamped source repeats an edge column: only columns that this pool invalidates anyway are affected)
        h.lo |= shfl64(v.lo, up) | shfl64(v.lo, dn);
        h.hi |= shfl64(v.hi, up) | shfl64(v.hi, dn);
    }
    return h;
}

// bit r of the result = (a[r] == b[r]), built 32 rows at a time in plain 32-bit registers
template <int NR>
__device__ __forceinline__ RowMask eq_mask(const float (&a)[NR], const float (&b)[NR]) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < NR; ++r) w[r >> 5] |= (a[r] == b[r]) ? (1u << (r & 31)) : 0u;
    return {((unsigned long long)w[1] << 32) | w[0], ((unsigned long long)w[3] << 32) | w[2]};
}

// bit r of the result = (a[r] > 0)
template <int NR>
__device__ __forceinline__ RowMask pos_mask(const float (&a)[NR]) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < NR; ++r) w[r >> 5] |= (a[r] > 0.f) ? (1u << (r & 31)) : 0u;
    return {((unsigned long long)w[1] << 32) | w[0], ((unsigned long long)w[3] << 32) | w[2]};
}
__device__ __forceinline__ RowMask row_range(int lo, int hi) {      // bits [lo, hi), 0 <= lo, hi <= 128
    auto below = [](int n) -> RowMask {                              // bits [0, n)
        if (n <= 0) return RowMask{0ull, 0ull};
        if (n >= 128) return RowMask{~0ull, ~0ull};
        if (n >= 64) return RowMask{~0ull, n == 64 ? 0ull : (~0ull >> (128 - n))};
        return RowMask{~0ull >> (64 - n), 0ull};
    };
    const RowMask a = below(hi), b = below(lo);
    return RowMask{a.lo & ~b.lo, a.hi & ~b.hi};
}

// COMPACT: instead of the dense map, the surviving maxima with a positive score (never in the border) go to a per-image
// candidate list (score, flat pixel index): what a top-k over the dense map can select before it runs into the zeros.
// Every wave tile owns a fixed segment of nms_seg(R) slots (>= the number of points more than R apart that fit its
// 64 - 10 R columns x NMS_RT rows), filled in (column, row) order: deterministic, no atomics; unused slots keep the
// caller's fill value, entries past the segment are dropped (possible only on plateaus of exactly tied scores).
constexpr int nms_seg(int R) {
    const int n = ((NMS_RT + R) / (R + 1)) * ((64 - 10 * R + R) / (R + 1));
    int p2 = 32;
    while (p2 < n) p2 <<= 1;
    return p2;
}
template <int R, bool COMPACT>
__global__ __launch_bounds__(256, NMS_WPS) void nms_reg_kernel(const float* __restrict__ s, float* __restrict__ out, int H, int W,
                                                         int border, int strips, int tiles, int* __restrict__ cand_idx) {
    constexpr int HALO = 5 * R, WOUT = 64 - 2 * HALO, NR = NMS_RT + 2 * HALO;
    static_assert(WOUT > 0 && NR <= 128, "radius too large for a 64-lane strip / a 128-bit row mask");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave;
    if (t >= tiles) return;
    const int strip = t % strips, ty = t / strips;
    const int x = strip * WOUT - HALO + lane, y0 = ty * NMS_RT - HALO;
    const bool xin = x >= 0 && x < W;
    const float* img = s + (int64_t)blockIdx.y * H * W + min(max(x, 0), W - 1);
    // rows of this tile inside the image: bits [rlo, rhi) -- pixels outside carry -inf and never enter a mask
    RowMask inside = row_range(max(0, -y0), min(NR, H - y0));
    if (!xin) inside = RowMask{0ull, 0ull};
    float sc[NR], p[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int yc = min(max(y0 + r, 0), H - 1);                 // clamped, branch-free load; masked below
        sc[r] = img[(int64_t)yc * W];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool in = ((r < 64 ? inside.lo >> r : inside.hi >> (r - 64)) & 1ull) != 0;
        sc[r] = in ? sc[r] : -INFINITY;
        p[r] = sc[r];
    }
    pool_rows_cols<R, NR>(p, lane);
    RowMask mk = eq_mask<NR>(sc, p);
    mk.lo &= inside.lo; mk.hi &= inside.hi;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        RowMask su = dilate<R>(mk, lane);
        su.lo &= inside.lo; su.hi &= inside.hi;                    // outside stays -inf (never "suppressed to 0")
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool sr = ((r < 64 ? su.lo >> r : su.hi >> (r - 64)) & 1ull) != 0;
            p[r] = sr ? 0.f : sc[r];                                                 // supp_scores
        }
        pool_rows_cols<R, NR>(p, lane);
        const RowMask e = eq_mask<NR>(sc, p);                      // un-suppressed pixels: supp_scores == scores
        mk.lo |= e.lo & ~su.lo & inside.lo;
        mk.hi |= e.hi & ~su.hi & inside.hi;
    }
    if (COMPACT) {
        // candidates of this lane: final maxima in the tile's own rows, outside the border, with a positive score
        const bool col_ok = lane >= HALO && lane < 64 - HALO && xin && !(border > 0 && (x < border || x >= W - border));
        RowMask c = row_range(max(HALO, border - y0), min(HALO + NMS_RT, H - border - y0));
        const RowMask pos = pos_mask<NR>(sc);
        c.lo &= mk.lo & pos.lo; c.hi &= mk.hi & pos.hi;
        if (!col_ok) c = RowMask{0ull, 0ull};
        const int mine = __popcll(c.lo) + __popcll(c.hi);
        int incl = mine;                                              // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o_ = __shfl_up(incl, d);
            if (lane >= d) incl += o_;
        }
        constexpr int SEG = nms_seg(R);
        int slot = incl - mine;                                        // position inside this tile's segment
        float* cs_ = out + ((int64_t)blockIdx.y * tiles + t) * SEG;
        int* ci_ = cand_idx + ((int64_t)blockIdx.y * tiles + t) * SEG;
#pragma unroll
        for (int r = HALO; r < HALO + NMS_RT; ++r) {
            if ((r < 64 ? c.lo >> r : c.hi >> (r - 64)) & 1ull) {
                if (slot < SEG) { cs_[slot] = sc[r]; ci_[slot] = (y0 + r) * W + x; }
                ++slot;
            }
        }
        return;
    }
    if (lane < HALO || lane >= 64 - HALO || !xin) return;
    float* o = out + (int64_t)blockIdx.y * H * W + x;
    const bool xb = border > 0 && (x < border || x >= W - border);
#pragma unroll
    for (int r = HALO; r < HALO + NMS_RT; ++r) {
        const int y = y0 + r;
        if (y >= H) break;
        const bool m = ((r < 64 ? mk.lo >> r : mk.hi >> (r - 64)) & 1ull) != 0;
        float v = m ? sc[r] : 0.f;
        if (xb || (border > 0 && (y < border || y >= H - border))) v = -1.f;
        o[(int64_t)y * W] = v;
    }
}

template <int R> int nms_launch(const float* s, float* out, int B, int H, int W, int border, hipStream_t st,
                                int* cand_idx = nullptr) {
    constexpr int WOUT = 64 - 10 * R;
    const int strips = (W + WOUT - 1) / WOUT, tiles = strips * ((H + NMS_RT - 1) / NMS_RT);
    if (B > 65535) return GF_ERR_UNSUPPORTED;
    const dim3 grid((tiles + 3) / 4, B);
    if (cand_idx) nms_reg_kernel<R, true><<<grid, dim3(256), 0, st>>>(s, out, H, W, border, strips, tiles, cand_idx);
    else nms_reg_kernel<R, false><<<grid, dim3(256), 0, st>>>(s, out, H, W, border, strips, tiles, nullptr);
    return (int)hipGetLastError();
}

// ---- descriptor sampling -------------------------------------------------------------------------
// sample_descriptors (superpoint_open.py:10-16) fused with the per-pixel L2 normalisation of the dense map
// (:149): out[b,n,:] = normalize( sum_k w_k * normalize(map[b, y_k, x_k, :]) ) over the 4 bilinear corners
// (align_corners=False, zero padding), x_pix = (kp_x + 0.5) / s - 0.5.  One wave per keypoint; the
// channels-last map makes every corner one contiguous C-vector.
template <typename T>
__global__ __launch_bounds__(256) void sample_desc_kernel(const T* __restrict__ map, const float* __restrict__ kpts,
                                                          float* __restrict__ out, int64_t total, int N, int h, int w,
                                                          int C, float inv_s) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= total) return;
    const int64_t b = k / N;
    const float px = (kpts[2 * k] + 0.5f) * inv_s - 0.5f, py = (kpts[2 * k + 1] + 0.5f) * inv_s - 0.5f;
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx = px - fx, wy = py - fy;
    const int nch = C / 64;                       // channels per lane (C % 64 == 0, <= 8 per lane)
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
        const float wgt = ((c & 1) ? wx : 1.f - wx) * ((c >> 1) ? wy : 1.f - wy);
        if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;          // wave-uniform
        const T* src = map + ((b * h + yy) * (int64_t)w + xx) * C + lane * nch;
        float v[8], ss = 0.f;
        for (int e = 0; e < nch; ++e) { v[e] = to_f32(src[e]); ss += v[e] * v[e]; }
        ss = wave_allsum(ss);
        const float sc = wgt / fmaxf(sqrtf(ss), 1e-12f);
        for (int e = 0; e < nch; ++e) acc[e] += sc * v[e];
    }
    float ss = 0.f;
    for (int e = 0; e < nch; ++e) ss += acc[e] * acc[e];
    ss = wave_allsum(ss);
    const float sc = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    float* dst = out + k * C + lane * nch;
    for (int e = 0; e < nch; ++e) dst[e] = acc[e] * sc;
}

// ---- detector head tail ---------------------------------------------------------------------------------------------
// superpoint_open.py:141-147: scores = softmax(detector(features), dim 1)[:, :-1] re-arranged from [B, 64, h, w] cells
// of 8 x 8 pixels into the [B, 8h, 8w] score map -- together with the tail of detector.1 itself (Conv2d(256, 65, 1) bias
// and its eval BatchNorm: 65 channels, not a multiple of the tail kernel's vector width).  One thread per cell: the
// workgroup's 256 cells x 65 channels are contiguous in the channels-last convolution output (256 * 65 elements = a
// multiple of 16 bytes) and come in through LDS with 16-byte loads; fp32 softmax (max, exp, sum, divide, as torch's);
// the cell's 64 probabilities leave as 8 rows of two float4.
template <typename T>
__global__ __launch_bounds__(256) void detector_scores_kernel(const T* __restrict__ y, const float* __restrict__ bias,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ scores, int64_t cells, int h, int w, int relu) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    T* tile = reinterpret_cast<T*>(dsm);
    const int64_t c0 = (int64_t)blockIdx.x * 256;
    const int ncell = (int)min((int64_t)256, cells - c0);
    const int nchunk = ncell * 65 * (int)sizeof(T) / 16, tail0 = nchunk * 16 / (int)sizeof(T), nel = ncell * 65;
    const u32x4* src = reinterpret_cast<const u32x4*>(y + c0 * 65);
    for (int i = threadIdx.x; i < nchunk; i += 256) reinterpret_cast<u32x4*>(dsm)[i] = src[i];
    for (int i = tail0 + threadIdx.x; i < nel; i += 256) tile[i] = y[c0 * 65 + i];          // (partial last workgroup only)
    __syncthreads();
    if ((int)threadIdx.x >= ncell) return;
    float v[65];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 65; ++c) {
        float a = to_f32(tile[threadIdx.x * 65 + c]) + bias[c];
        if (relu) a = fmaxf(a, 0.f);
        v[c] = fmaf(a, scale[c], shift[c]);
        m = fmaxf(m, v[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 65; ++c) { v[c] = expf(v[c] - m); sum += v[c]; }
    const int64_t cell = c0 + threadIdx.x;
    const int x = (int)(cell % w), yy = (int)((cell / w) % h);
    const int64_t b = cell / ((int64_t)w * h);
    float* o = scores + (b * h * 8 + (int64_t)yy * 8) * ((int64_t)w * 8) + (int64_t)x * 8;
#pragma unroll
    for (int dy = 0; dy < 8; ++dy) {
        st4(o + (int64_t)dy * w * 8, v[dy * 8] / sum, v[dy * 8 + 1] / sum, v[dy * 8 + 2] / sum, v[dy * 8 + 3] / sum);
        st4(o + (int64_t)dy * w * 8 + 4, v[dy * 8 + 4] / sum, v[dy * 8 + 5] / sum, v[dy * 8 + 6] / sum, v[dy * 8 + 7] / sum);
    }
}

}  // namespace

extern "C" int gf_bias_act_bn_nhwc(const void* x, void* y, const float* bias, const float* scale, const float* shift,
                                   int B, int H, int W, int C, int relu, int pool, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return GF_ERR_SHAPE;
    if (dtype != GF_F32 && dtype != GF_BF16) return GF_ERR_DTYPE;
    const int vec = dtype == GF_BF16 ? 8 : 4;
    if (C % vec) return GF_ERR_ALIGN;
    if (pool && ((H | W) & 1)) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)B * (pool ? (H / 2) * (int64_t)(W / 2) : (int64_t)H * W);
    const int64_t nvec = npix * (C / vec);
    const int64_t want = (nvec + 255) / 256;
    const int nb = (int)(want > 16384 ? 16384 : want);
#define GF_BAB(T, RELU)                                                                                              \
    if (pool)                                                                                                        \
        bias_act_bn_pool_kernel<T, RELU><<<nb, 256, 0, st>>>((const T*)x, (T*)y, bias, scale, shift, nvec, H, W, C); \
    else                                                                                                             \
        bias_act_bn_kernel<T, RELU><<<nb, 256, 0, st>>>((const T*)x, (T*)y, bias, scale, shift, nvec, C);
    if (dtype == GF_BF16) { if (relu) { GF_BAB(bf16_t, true) } else { GF_BAB(bf16_t, false) } }
    else { if (relu) { GF_BAB(float, true) } else { GF_BAB(float, false) } }
#undef GF_BAB
    return (int)hipGetLastError();
}

extern "C" int gf_detector_scores(const void* y, const float* bias, const float* scale, const float* shift, float* scores,
                                  int B, int h, int w, int relu, int dtype, void* stream) {
    if (B <= 0 || h <= 0 || w <= 0) return GF_ERR_SHAPE;
    if ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(scores)) & 15) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t cells = (int64_t)B * h * w;
    const dim3 grid((unsigned)((cells + 255) / 256));
    if (dtype == GF_BF16) {
        detector_scores_kernel<bf16_t><<<grid, 256, 256 * 65 * sizeof(bf16_t), st>>>((const bf16_t*)y, bias, scale, shift, scores, cells, h, w, relu);
    } else if (dtype == GF_F32) {
        const int lds = 256 * 65 * (int)sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(detector_scores_kernel<float>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        detector_scores_kernel<float><<<grid, 256, lds, st>>>((const float*)y, bias, scale, shift, scores, cells, h, w, relu);
    } else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_nms_scores(const float* scores, float* out, int B, int H, int W, int radius, int border,
                             void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || border < 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (radius) {
        case 1: return nms_launch<1>(scores, out, B, H, W, border, st);
        case 2: return nms_launch<2>(scores, out, B, H, W, border, st);
        case 3: return nms_launch<3>(scores, out, B, H, W, border, st);
        case 4: return nms_launch<4>(scores, out, B, H, W, border, st);
        default: return GF_ERR_UNSUPPORTED;
    }
}

template <int R> int nms_cap(int H, int W) {
    constexpr int WOUT = 64 - 10 * R;
    return ((W + WOUT - 1) / WOUT) * ((H + NMS_RT - 1) / NMS_RT) * nms_seg(R);
}
extern "C" int gf_nms_candidates_cap(int H, int W, int radius) {
    switch (radius) {
        case 1: return nms_cap<1>(H, W);
        case 2: return nms_cap<2>(H, W);
        case 3: return nms_cap<3>(H, W);
        case 4: return nms_cap<4>(H, W);
        default: return GF_ERR_UNSUPPORTED;
    }
}
extern "C" int gf_nms_candidates(const float* scores, float* cand_scores, int* cand_idx, int B, int H, int W, int radius,
                                 int border, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || border < 0) return GF_ERR_SHAPE;
    if ((int64_t)H * W > 0x7fffffff) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (radius) {
        case 1: return nms_launch<1>(scores, cand_scores, B, H, W, border, st, cand_idx);
        case 2: return nms_launch<2>(scores, cand_scores, B, H, W, border, st, cand_idx);
        case 3: return nms_launch<3>(scores, cand_scores, B, H, W, border, st, cand_idx);
        case 4: return nms_launch<4>(scores, cand_scores, B, H, W, border, st, cand_idx);
        default: return GF_ERR_UNSUPPORTED;
    }
}

extern "C" int gf_sample_descriptors(const void* map, const float* kpts, float* out, int B, int N, int h, int w, int C,
                                     int stride, int dtype, void* stream) {
    if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || stride <= 0) return GF_ERR_SHAPE;
    if (C <= 0 || C % 64 || C > 512) return GF_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * N;
    const dim3 grid((unsigned)((total + 3) / 4));
    if (dtype == GF_BF16)
        sample_desc_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)map, kpts, out, total, N, h, w, C, 1.f / stride);
    else if (dtype == GF_F32)
        sample_desc_kernel<float><<<grid, 256, 0, st>>>((const float*)map, kpts, out, total, N, h, w, C, 1.f / stride);
    else
        return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int gf_conv1_bias_act_bn(const void* img, const void* w, const float* bias, const float* scale,
                                    const float* shift, void* out, int B, int H, int W, int C, int relu, int dtype,
                                    void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return GF_ERR_SHAPE;
    if (C != 64) return GF_ERR_UNSUPPORTED;
    if (dtype != GF_F32 && dtype != GF_BF16) return GF_ERR_DTYPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((W + 32 * C1_NX - 1) / (32 * C1_NX), (H + C1_TY - 1) / C1_TY, B);
#define GF_C1(T, RELU) conv1_fused_kernel<T, RELU, 64><<<grid, 256, 0, st>>>((const T*)img, (const T*)w, bias, scale, shift, (T*)out, H, W)
    if (dtype == GF_BF16) { if (relu) GF_C1(bf16_t, true); else GF_C1(bf16_t, false); }
    else { if (relu) GF_C1(float, true); else GF_C1(float, false); }
#undef GF_C1
    return (int)hipGetLastError();
}
