// 3x3 convolution, 64 -> 64 channels, channels-last bf16, with the whole VGG-block tail fused:
//     out = [maxpool2x2] ( relu(conv3x3(x, w) + bias) * scale + shift )
// i.e. backbone.0.1, backbone.1.0, backbone.1.1 of SuperPoint-open (gluefactory/models/extractors/
// superpoint_open.py:37-75, 98-105: Conv2d(64,64,3,padding=1) -> ReLU -> BatchNorm2d(eval) [-> MaxPool2d(2,2)]).
// These three layers are 15 of the extractor's 20 ms of library convolution at 64 x 1024^2 (the first alone 10 ms:
// 4.9 TFLOP over an 8.6 GB input) and each is followed by a tail pass over the same tensor; here the activation is
// read once and the finished (pooled) activation written once.
//
// Implicit GEMM on the matrix cores, "weights stationary":
//   * a workgroup = 4 waves (one per SIMD, the whole 512-register file each) owns an 8 x 32 pixel output tile and
//     walks tiles persistently; wave w owns tile rows 2w, 2w+1;
//   * the 64 x 576 weight matrix lives in REGISTERS for the whole kernel (A operands, [tap][k-step] x 2 output-
//     channel blocks = 288 VGPRs, loaded once from a [tap][cout][cin] copy);
//   * the input window (10 x 34 pixels x 128 B) arrives by LDS-DMA into a double buffer -- the next tile's window
//     streams in while this tile's 144 MFMAs per wave run -- pixel-major, its eight 16-byte chunks swizzled by
//     (pixel >> 1) & 7: the B-operand read (32 consecutive pixels of one row, one chunk) is conflict-free; halo
//     pixels outside the image are zeroed after the DMA landed (border tiles only);
//   * output channels on the MFMA i axis (registers), pixels on lanes: a lane ends with 64 channels of two pixels;
//     bias / ReLU / BN affine are lane-local, the bf16 result goes through an LDS tile so that the store (and the
//     2x2 max-pool) is done by the whole workgroup in whole 128-byte pixels.
#include "gf_common.h"
#include "gf_amd.h"

#ifndef C3_ABL          // probe builds only (tools/probe): bit 0 no tail+store, 1 no DMA in the loop, 2 no MFMA loop, 3 no global store
#define C3_ABL 0
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void c3_lds_void;
typedef const __attribute__((address_space(1))) void c3_glb_void;

constexpr int C3_TH = 8, C3_TW = 32;                       // output tile
constexpr int C3_IW = C3_TW + 2, C3_IH = C3_TH + 2;        // input window
constexpr int C3_NPIX = C3_IW * C3_IH;                     // 340 pixels
constexpr int C3_PIECES = 44;                              // 1-KiB DMA pieces (8 pixels each) per window
constexpr int C3_INBUF = C3_PIECES * 1024;                 // bytes of one input buffer
constexpr int C3_OUT = C3_TH * C3_TW * 128;                // bytes of the output staging tile

template <int OFF> __device__ __forceinline__ u32x4 c3_rd128(unsigned a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void c3_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename V> __device__ __forceinline__ void c3_tie(V& v) { asm volatile("" : "+v"(v)); }

struct C3Params {
    const bf16_t* x; const bf16_t* w;        // x [B,H,W,64]; w [9][64 cout][64 cin]
    const float* bias; const float* scale; const float* shift;
    bf16_t* y;                               // [B,H,W,64] or pooled [B,H/2,W/2,64]
    int B, H, W, relu;
};

template <bool POOL>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(C3Params p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    char* outb = smem + 2 * C3_INBUF;                       // output staging tile [8][32 px][128 B], chunk-swizzled
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int tx_n = p.W / C3_TW, ty_n = p.H / C3_TH;
    const int tiles = tx_n * ty_n * p.B;

    // ---- weights: A operands, lane = output channel 32 nt + l31, elements cin 16 ks + 8 hi .. + 8 of tap t
    bf16x8 wf[2][9][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                wf[nt][t][ks] = *reinterpret_cast<const bf16x8*>(p.w + ((t * 64 + 32 * nt + l31) * 64 + 16 * ks + 8 * hi));
    // epilogue constants [bias | scale | shift][64] in LDS (read per tile: registers are full of weights)
    float* cst = reinterpret_cast<float*>(smem + 2 * C3_INBUF + C3_OUT);
    if (threadIdx.x < 64) {
        cst[threadIdx.x] = p.bias[threadIdx.x];
        cst[64 + threadIdx.x] = p.scale[threadIdx.x];
        cst[128 + threadIdx.x] = p.shift[threadIdx.x];
    }

    // ---- input window DMA: piece q = LDS chunk positions [64 q, 64 q + 64) = pixels 8 q .. 8 q + 7; the lane for chunk
    // position (pixel pp, slot c') fetches logical chunk c' ^ ((pp >> 1) & 7) of that pixel (swizzle on the source).
    auto issue = [&](int tile, int buf) {
        const int b = tile / (tx_n * ty_n), r = tile % (tx_n * ty_n);
        const int y0 = (r / tx_n) * C3_TH, x0 = (r % tx_n) * C3_TW;
        const bf16_t* img = p.x + (int64_t)b * p.H * p.W * 64;
#pragma unroll
        for (int i = 0; i < C3_PIECES / 4; ++i) {
            const int q = wave + 4 * i;
            const int pp = min(q * 8 + (lane >> 3), C3_NPIX - 1);
            const int iy = pp / C3_IW, ix = pp - iy * C3_IW;
            const int gy = min(max(y0 - 1 + iy, 0), p.H - 1), gx = min(max(x0 - 1 + ix, 0), p.W - 1);
            const int c = (lane & 7) ^ ((pp >> 1) & 7);
            __builtin_amdgcn_global_load_lds((c3_glb_void*)(img + ((int64_t)gy * p.W + gx) * 64 + c * 8),
                                             (c3_lds_void*)(smem + buf * C3_INBUF + q * 1024), 16, 0, 0);
        }
    };
    // halo pixels outside the image -> 0 (after the DMA of every wave landed)
    auto zero_halo = [&](int tile, int buf) {
        const int r = tile % (tx_n * ty_n);
        const int y0 = (r / tx_n) * C3_TH, x0 = (r % tx_n) * C3_TW;
        for (int pp = threadIdx.x; pp < C3_NPIX; pp += 256) {
            const int iy = pp / C3_IW, ix = pp - iy * C3_IW;
            const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
            if (gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) {
                u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4*>(smem + buf * C3_INBUF + pp * 128 + c * 16) = z;
            }
        }
    };
    auto is_border = [&](int tile) {
        const int r = tile % (tx_n * ty_n);
        const int ty = r / tx_n, tx = r % tx_n;
        return ty == 0 || tx == 0 || ty == ty_n - 1 || tx == tx_n - 1;
    };

    int tile = blockIdx.x;
    if (tile >= tiles) return;
    issue(tile, 0);
    c3_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (is_border(tile)) {
        zero_halo(tile, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    int buf = 0;
    for (; tile < tiles; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (next < tiles && !(C3_ABL & 2)) issue(next, buf ^ 1);

        // ---- 144 MFMAs: acc[nt][r] (32 channels x 32 pixels of tile row 2 wave + r)
        f32x16 acc[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[nt][r][k] = 0.f;
        // B operand of (tap, ks, r): pixel pp = (2 wave + r + dy) * 34 + l31 + dx, chunk 2 ks + hi, swizzled:
        // byte address = pp * 128 + (((hi ^ v) * 16) ^ (32 ks)), v = (pp >> 1) & 7
        auto base = [&](int t, int r) {
            const int pp = (2 * wave + r + t / 3) * C3_IW + l31 + t % 3;
            return lds0 + (unsigned)(buf * C3_INBUF + pp * 128 + ((hi ^ ((pp >> 1) & 7)) << 4));
        };
        u32x4 ra[4], rb[4];                                      // [ks], double buffer over (tap, row) groups
        if (!(C3_ABL & 4)) {
            const unsigned a = base(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ra[ks] = c3_rd128<0>(a ^ (32u * ks));
        }
#pragma unroll
        for (int g = 0; g < ((C3_ABL & 4) ? 0 : 18); ++g) {
            const int t = g >> 1, r = g & 1;
            u32x4 (&cur)[4] = (g & 1) ? rb : ra;
            u32x4 (&nxt)[4] = (g & 1) ? ra : rb;
            if (g + 1 < 18) {
                const unsigned a = base((g + 1) >> 1, (g + 1) & 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) nxt[ks] = c3_rd128<0>(a ^ (32u * ks));
                c3_wait_lgkm<4>();                               // this group's four fragments landed
            } else {
                c3_wait_lgkm<0>();
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c3_tie(cur[ks]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc[nt][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][t][ks], __builtin_bit_cast(bf16x8, cur[ks]), acc[nt][r], 0, 0, 0);
        }

        // ---- tail on the fp32 sums, bf16 into the staging tile: pixel (2 wave + r, l31), channels 8-byte groups
        if (C3_ABL & 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { asm volatile("" ::"v"(acc[0][r]), "v"(acc[1][r])); }
        } else
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int opx = (2 * wave + r) * C3_TW + l31;
            char* orow = outb + opx * 128;
            const int v = (opx >> 1) & 7;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * nt + 8 * g + 4 * hi;
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(cst + c0);
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(cst + 64 + c0);
                    const f32x4 ch = *reinterpret_cast<const f32x4*>(cst + 128 + c0);
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = acc[nt][r][4 * g + e] + cb[e];
                        if (p.relu) a = fmaxf(a, 0.f);
                        o[e] = fmaf(a, cs[e], ch[e]);
                    }
                    // channels 32 nt + 8 g + 4 hi + e -> 16-byte chunk 4 nt + g (swizzled), half hi
                    st4(reinterpret_cast<bf16_t*>(orow + (((4 * nt + g) ^ v) << 4) + 8 * hi), o[0], o[1], o[2], o[3]);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // staging tile complete; input buffer `buf` free

        // ---- store (whole 128-byte pixels, coalesced) [+ 2x2 max-pool]
        if (!(C3_ABL & 9)) {
            const int b = tile / (tx_n * ty_n), r_ = tile % (tx_n * ty_n);
            const int y0 = (r_ / tx_n) * C3_TH, x0 = (r_ % tx_n) * C3_TW;
            if (!POOL) {
                bf16_t* dst = p.y + (((int64_t)b * p.H + y0) * p.W + x0) * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int u = threadIdx.x + 256 * i;             // chunk index: row i, pixel u/8 % 32, chunk u % 8
                    const int opx = u >> 3, c = u & 7;
                    const u32x4 val = *reinterpret_cast<const u32x4*>(outb + opx * 128 + ((c ^ ((opx >> 1) & 7)) << 4));
                    *reinterpret_cast<u32x4*>(dst + ((int64_t)(opx >> 5) * p.W + (opx & 31)) * 64 + c * 8) = val;
                }
            } else {
                bf16_t* dst = p.y + (((int64_t)b * (p.H / 2) + y0 / 2) * (p.W / 2) + x0 / 2) * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int u = threadIdx.x + 256 * i;             // pooled chunk: row u / 128, pixel u / 8 % 16, chunk u % 8
                    const int py = u >> 7, px = (u >> 3) & 15, c = u & 7;
                    float m[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int opx = (2 * py + dy) * C3_TW + 2 * px + dx;
                            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(outb + opx * 128 + ((c ^ ((opx >> 1) & 7)) << 4));
#pragma unroll
                            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)vv[e]);
                        }
                    const bf16x8 o = {(bf16_t)m[0], (bf16_t)m[1], (bf16_t)m[2], (bf16_t)m[3],
                                      (bf16_t)m[4], (bf16_t)m[5], (bf16_t)m[6], (bf16_t)m[7]};
                    *reinterpret_cast<bf16x8*>(dst + ((int64_t)py * (p.W / 2) + px) * 64 + c * 8) = o;
                }
            }
        }
        // ---- next window landed?  (vmcnt retires in order: the stores above were issued after the DMA)
        if (next < tiles) {
            if (POOL) c3_wait_vm<2>(); else c3_wait_vm<8>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone's DMA landed; staging tile read
            if (is_border(next)) {
                zero_halo(next, buf ^ 1);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
}

}  // namespace

extern "C" int gf_conv3x3_c64(const void* x, const void* w, const float* bias, const float* scale, const float* shift,
                              void* y, int B, int H, int W, int relu, int pool, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return GF_ERR_SHAPE;
    if (dtype != GF_BF16) return GF_ERR_DTYPE;
    if (H % C3_TH || W % C3_TW) return GF_ERR_UNSUPPORTED;
    C3Params p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(w); p.bias = bias; p.scale = scale; p.shift = shift;
    p.y = static_cast<bf16_t*>(y); p.B = B; p.H = H; p.W = W; p.relu = relu;
    const size_t lds = 2 * C3_INBUF + C3_OUT + 3 * 64 * sizeof(float);
    const int tiles = (W / C3_TW) * (H / C3_TH) * B;
    const int grid = tiles < 256 ? tiles : 256;            // one persistent workgroup per CU
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (pool) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        conv3x3_c64_kernel<true><<<dim3(grid), 256, lds, st>>>(p);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        conv3x3_c64_kernel<false><<<dim3(grid), 256, lds, st>>>(p);
    }
    return (int)hipGetLastError();
}
