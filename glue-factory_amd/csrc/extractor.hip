// Memory-bound tails of the frozen SuperPoint extractor (the convolutions themselves stay on the stock
// library, by design).  Reference: gluefactory/models/extractors/superpoint_open.py
//   * VGGBlock = Conv2d -> ReLU -> BatchNorm2d(eval) [-> MaxPool2d(2,2)]  (:37-75, :98-111)
//       one pass over the channels-last activation instead of bias-add, clamp, batch-norm and pool passes;
//   * simple_nms (:19-34): five 2r+1 max-pools and a dozen elementwise passes over the [B,H,W] score map
//       become one kernel working on an LDS tile with a 5r halo.
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
// A workgroup produces a TS x TS output tile from a (TS + 10r)^2 input tile held in LDS (three float planes
// and two byte masks: 151 KB at r = 4); every pooling is separable (row pass into a scratch plane, column
// pass back) with register windows.  Pixels outside the image are -inf for the
// score pools and "no maximum" for the mask pools, which is what clipping the window means.
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
template <typename T, bool RELU, int C>
__global__ __launch_bounds__(256) void conv1_fused_kernel(const T* __restrict__ img, const T* __restrict__ wgt,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ out, int H, int W) {
    constexpr int CG = C / 8;                 // lanes per pixel (8 channels each)
    constexpr int PX = 256 / CG;              // pixels per workgroup row
    constexpr int TY = 32;
    __shared__ float tile[(TY + 2) * (PX + 2)];
    const int b = blockIdx.z, y0 = blockIdx.y * TY, x0 = blockIdx.x * PX;
    const T* im = img + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < (TY + 2) * (PX + 2); i += 256) {
        const int y = y0 - 1 + i / (PX + 2), x = x0 - 1 + i % (PX + 2);
        tile[i] = (y >= 0 && y < H && x >= 0 && x < W) ? to_f32(im[(int64_t)y * W + x]) : 0.f;
    }
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
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
    const int x = x0 + px;
    for (int ly = 0; ly < TY; ++ly) {
        const int y = y0 + ly;
        if (y >= H) break;
        float v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[3 * dy + dx] = tile[(ly + dy) * (PX + 2) + px + dx];
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
            if constexpr (sizeof(T) == 2) {
                st_vec<T>(dst, o);
            } else {
                st4(dst, o[0], o[1], o[2], o[3]);
                st4(dst + 4, o[4], o[5], o[6], o[7]);
            }
        }
    }
}

#ifndef NMS_THREADS_V
#define NMS_THREADS_V 1024
#endif
constexpr int NMS_TS = 64, NMS_SEG = 8;
// The tile's three fp32 planes fill most of a CU's LDS (one workgroup per CU), so the workgroup itself has to bring
// the waves that hide the LDS latency: 16 waves (4 per SIMD) instead of 4 measured 3.8 -> 1.x ms on 64 x 1024^2.
constexpr int NMS_THREADS = NMS_THREADS_V;

// 1-D running maximum of radius R over `n` lines of length `len`: each work item produces NMS_SEG consecutive
// outputs of one line from NMS_SEG + 2R inputs held in registers (1.75 LDS reads per output at R = 3
// instead of 2R + 1).  ALONG_X: lines are rows (stride 1 inside a line), else columns (stride TW).
template <int R, int TW, bool ALONG_X>
__device__ __forceinline__ void max1d(const float* __restrict__ src, float* __restrict__ dst) {
    constexpr int NSEG = (TW + NMS_SEG - 1) / NMS_SEG;
    for (int it = threadIdx.x; it < NSEG * TW; it += NMS_THREADS) {
        // consecutive work items walk the direction that is contiguous in LDS (conflict-free)
        const int line = ALONG_X ? it / NSEG : it % TW;
        const int seg = ALONG_X ? it % NSEG : it / TW;
        const int p0 = seg * NMS_SEG;
        float v[NMS_SEG + 2 * R];
#pragma unroll
        for (int k = 0; k < NMS_SEG + 2 * R; ++k) {
            const int p = p0 - R + k;
            v[k] = (p >= 0 && p < TW) ? src[ALONG_X ? line * TW + p : p * TW + line] : -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < NMS_SEG; ++k) {
            float m = v[k];
#pragma unroll
            for (int j = 1; j <= 2 * R; ++j) m = fmaxf(m, v[k + j]);
            const int p = p0 + k;
            if (p < TW) dst[ALONG_X ? line * TW + p : p * TW + line] = m;
        }
    }
}

template <int R>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ s, float* __restrict__ out, int H, int W,
                                                  int border) {
    constexpr int HALO = 5 * R, TW = NMS_TS + 2 * HALO, NP = TW * TW;
    extern __shared__ float smem[];
    float* sc = smem;                 // scores, -inf outside the image
    float* pa = sc + NP;              // plane being pooled (in: map, out: pooled map)
    float* pt = pa + NP;              // row-pass scratch
    unsigned char* mk = reinterpret_cast<unsigned char*>(pt + NP);   // current maxima
    unsigned char* sp = mk + NP;                                       // suppressed
    const int b = blockIdx.z, ty0 = blockIdx.y * NMS_TS - HALO, tx0 = blockIdx.x * NMS_TS - HALO;
    const float* img = s + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < NP; i += NMS_THREADS) {
        const int y = ty0 + i / TW, x = tx0 + i % TW;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const float v = in ? img[(int64_t)y * W + x] : -INFINITY;
        sc[i] = v;
        pa[i] = v;
    }
    __syncthreads();
    auto pool = [&]() {                                      // pa <- (2R+1)^2 max of pa
        max1d<R, TW, true>(pa, pt);
        __syncthreads();
        max1d<R, TW, false>(pt, pa);
        __syncthreads();
    };
    pool();
    for (int i = threadIdx.x; i < NP; i += NMS_THREADS) {
        const float v = sc[i];
        const bool m = v != -INFINITY && v == pa[i];
        mk[i] = m;
        pa[i] = v == -INFINITY ? -INFINITY : (m ? 1.f : 0.f);
    }
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
        pool();                                              // dilated maxima
        for (int i = threadIdx.x; i < NP; i += NMS_THREADS) {
            const bool su = pa[i] > 0.f;
            sp[i] = su;
            const float v = sc[i];
            pa[i] = v == -INFINITY ? -INFINITY : (su ? 0.f : v);            // supp_scores
        }
        __syncthreads();
        pool();
        for (int i = threadIdx.x; i < NP; i += NMS_THREADS) {
            const float v = sc[i];
            const bool su = sp[i];
            const float ss = su ? 0.f : v;
            const bool m = mk[i] | (v != -INFINITY && ss == pa[i] && !su);
            mk[i] = m;
            pa[i] = v == -INFINITY ? -INFINITY : (m ? 1.f : 0.f);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < NMS_TS * NMS_TS; i += NMS_THREADS) {
        const int ly = i / NMS_TS, lx = i % NMS_TS;
        const int y = blockIdx.y * NMS_TS + ly, x = blockIdx.x * NMS_TS + lx;
        if (y < H && x < W) {
            const int j = (ly + HALO) * TW + lx + HALO;
            float v = mk[j] ? sc[j] : 0.f;
            if (border > 0 && (y < border || x < border || y >= H - border || x >= W - border)) v = -1.f;
            out[(int64_t)b * H * W + (int64_t)y * W + x] = v;
        }
    }
}

template <int R> size_t nms_lds() {
    constexpr int TW = NMS_TS + 10 * R;
    return (size_t)TW * TW * (3 * sizeof(float) + 2) + 16;
}

template <int R> int nms_launch(const float* s, float* out, int B, int H, int W, int border, hipStream_t st) {
    const size_t lds = nms_lds<R>();
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((W + NMS_TS - 1) / NMS_TS, (H + NMS_TS - 1) / NMS_TS, B);
    nms_kernel<R><<<grid, dim3(NMS_THREADS), lds, st>>>(s, out, H, W, border);
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
    const dim3 grid((W + 31) / 32, (H + 31) / 32, B);
#define GF_C1(T, RELU) conv1_fused_kernel<T, RELU, 64><<<grid, 256, 0, st>>>((const T*)img, (const T*)w, bias, scale, shift, (T*)out, H, W)
    if (dtype == GF_BF16) { if (relu) GF_C1(bf16_t, true); else GF_C1(bf16_t, false); }
    else { if (relu) GF_C1(float, true); else GF_C1(float, false); }
#undef GF_C1
    return (int)hipGetLastError();
}
