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
//     channel blocks = 288 registers, loaded once from a [tap][cout][cin] copy, their register file pinned by hand);
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
constexpr int C3_OUT = C3_TH * C3_TW * 128;                // bytes of one output staging tile
constexpr int C3_STAGE = 2 * C3_INBUF;                     // LDS map: 2 input windows | 2 staging tiles | epilogue constants
constexpr int C3_CST = C3_STAGE + 2 * C3_OUT;
constexpr int C3_LDS = C3_CST + 3 * 64 * 4;
#ifndef C3_AGPR_FRAGS_V
#define C3_AGPR_FRAGS_V 48
#endif
constexpr int C3_AGPR_FRAGS = C3_AGPR_FRAGS_V;   // weight fragments kept in AGPRs (4 registers each)

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
    bf16_t* y;                               // [B,H,W,64] or pooled [B,H/2,W/2,64]; pixel stride ldy >= 64 elements (a 64-channel
    int ldy;                                 // slice of a wider channels-last tensor: one half of a 64 -> 128 block's output)
    int B, H, W, relu;
};

template <bool POOL, bool RELU>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(C3Params p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int tx_n = p.W / C3_TW, ty_n = p.H / C3_TH;
    const int tiles = tx_n * ty_n * p.B;
#ifndef C3_SKEW32
#define C3_SKEW32 5         // (3 ... 17 measure the same, 4.88-4.94 ms; skews of the 16-tile rows of the 512^2 blocks: no gain)
#endif
    const int skew = tx_n % 32 == 0 ? C3_SKEW32 : 0;

    // ---- weights: A operands, lane = output channel 32 nt + l31, elements cin 16 ks + 8 hi .. + 8 of tap t
    // (all 72 fragments = 288 registers stay resident: 192 in the AGPRs next to the 64 accumulator registers, 96 in VGPRs)
    bf16x8 wf[2][9][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.w + ((t * 64 + 32 * nt + l31) * 64 + 16 * ks + 8 * hi));
                {
                    wf[nt][t][ks] = v;
                    // pin the register file of every fragment for the whole kernel: 48 fragments (192 registers) in the AGPRs
                    // next to the 64 accumulator registers, the other 20 in VGPRs -- MFMA reads either directly.  Left to
                    // itself the allocator "spills" weights to AGPRs and copies them back in front of every use (237
                    // v_accvgpr_read / _mov per tile).
                    if ((nt * 9 + t) * 4 + ks < C3_AGPR_FRAGS) asm volatile("" : "+a"(wf[nt][t][ks]));
                    else asm volatile("" : "+v"(wf[nt][t][ks]));
                }
            }
    // epilogue constants [bias | scale | shift][64] in LDS (read per tile: registers are full of weights)
    float* cst = reinterpret_cast<float*>(smem + C3_CST);
    if (threadIdx.x < 64) {
        cst[threadIdx.x] = p.bias[threadIdx.x];
        cst[64 + threadIdx.x] = p.scale[threadIdx.x];
        cst[128 + threadIdx.x] = p.shift[threadIdx.x];
    }

    struct Tile { int b, y0, x0; };
    auto coords = [&](int tile) {
        Tile t;
        t.b = tile / (tx_n * ty_n);
        const int r = tile - t.b * (tx_n * ty_n);
        const int ty = r / tx_n;
        // tile columns skewed by the tile row when a tile row is a multiple of 32 tiles (W = 1024: 128 KB per pixel row): with the
        // plain order workgroup j -- XCD j % 8 -- works on tile column j % 32 for ever, i.e. every XCD reads the same 4 KB blocks
        // of every row; measured 5.35 -> 4.92 ms at 64 x 1024^2 (and 4.74 on a 33-tile row), neutral or slightly worse for the
        // 16- and 8-tile rows of the later blocks (tools/probe/time_conv3x3_shapes.py, -DC3_ORDER=...)
        const int tx = (r - ty * tx_n + skew * ty) % tx_n;
        t.y0 = ty * C3_TH; t.x0 = tx * C3_TW;
        return t;
    };
    // ---- input window DMA: piece q = wave + 4 i = LDS chunk positions [64 q, 64 q + 64) = window pixels 8 q .. 8 q + 7; the lane
    // for chunk position (pixel pp, slot c') fetches logical chunk c' ^ ((pp >> 1) & 7) of that pixel (swizzle on the source).
    // `buffer_load ... lds` through a descriptor of the tile's IMAGE: window rows above / below the image are out of range
    // and land as zeros (the halo); window columns left / right of it alias neighbouring rows and are zeroed after landing
    // (zero_halo, border tiles only).  The pieces of a window are issued in order and piece i + 1 is piece i moved on by 32
    // window pixels: + 32 * 128 bytes, + one image row - 34 pixels where the pixel index wraps into the next window row; the
    // swizzled chunk is the same for every piece of a lane ((32 i) >> 1 & 7 == 0).  Two registers of running state and 4
    // instructions per piece instead of ~18 (every instruction of this kernel costs wall time: DESIGN.md section 5).
    const int pc_pp0 = wave * 8 + (lane >> 3);                                    // piece 0: window row 0, pixel pp0 < 34
    const int pc_lane0 = pc_pp0 * 128 + ((lane & 7) ^ ((pc_pp0 >> 1) & 7)) * 16;
    const int pc_wrap = (p.W - C3_IW) * 128;
    int pc_off = 0, pc_ix = 0;
    auto image_rsrc = [&](const Tile& t) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)t.b * p.H * p.W * 64), 0, p.H * p.W * 128, 0x00020000);
    };
    auto issue_piece = [&](__amdgpu_buffer_rsrc_t rs, const Tile& t, int buf, int i) {
        if (i == 0) { pc_ix = pc_pp0; pc_off = ((t.y0 - 1) * p.W + t.x0 - 1) * 128 + pc_lane0; }     // (may be negative: out of range)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (c3_lds_void*)(smem + buf * C3_INBUF + (wave + 4 * i) * 1024), 16, pc_off, 0, 0, 0);
        const bool wrap = pc_ix >= C3_IW - 32;
        pc_off += 32 * 128 + (wrap ? pc_wrap : 0);
        pc_ix += wrap ? 32 - C3_IW : 32;
    };
    // halo pixels outside the image -> 0 (after the DMA of every wave landed)
    auto zero_halo = [&](const Tile& t, int buf) {
        for (int pp = threadIdx.x; pp < C3_NPIX; pp += 256) {
            const int iy = pp / C3_IW, ix = pp - iy * C3_IW;
            const int gy = t.y0 - 1 + iy, gx = t.x0 - 1 + ix;
            if (gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) {
                u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4*>(smem + buf * C3_INBUF + pp * 128 + c * 16) = z;
            }
        }
    };
    auto is_border = [&](const Tile& t) {
        return t.y0 == 0 || t.x0 == 0 || t.y0 == p.H - C3_TH || t.x0 == p.W - C3_TW;
    };

    // ---- finished tiles leave through a double-buffered staging tile ([8][32 px][128 B], chunks swizzled by
    // (px >> 1) & 7): tile i is written to staging[i & 1] by its owner lanes right after its MFMA loop and goes
    // to HBM -- whole 128-byte pixels, [2x2 max-pooled] -- from INSIDE the MFMA loop of tile i + 1.
    constexpr int NSTORE = POOL ? 4 : 8;                    // store steps per thread and tile
    // element offset of this thread's 16-byte chunk inside its tile (step 0) -- a per-lane constant; step k adds whole
    // output rows (one, or two pooled ones), the tile's origin is uniform: one 32-bit add per store instead of ~8 instructions
    const int st_off0 = POOL ? ((int)(threadIdx.x >> 7) * (p.W / 2) + ((int)(threadIdx.x >> 3) & 15)) * p.ldy + ((int)threadIdx.x & 7) * 8
                             : ((int)(threadIdx.x >> 3)) * p.ldy + ((int)threadIdx.x & 7) * 8;
    auto store_addr = [&](const Tile& t, int k) -> bf16_t* {
        if (!POOL) {
            bf16_t* base = p.y + (((int64_t)t.b * p.H + t.y0 + k) * p.W + t.x0) * p.ldy;                   // (uniform)
            return base + st_off0;
        }
        bf16_t* base = p.y + (((int64_t)t.b * (p.H / 2) + t.y0 / 2 + 2 * k) * (p.W / 2) + t.x0 / 2) * p.ldy;   // (uniform)
        return base + st_off0;
    };
    // LDS byte address of (staging buffer sb, pixel opx, logical chunk c)
    auto stage_at = [&](int sb, int opx, int c) { return C3_STAGE + sb * C3_OUT + opx * 128 + ((c ^ ((opx >> 1) & 7)) << 4); };
    u32x4 sv[POOL ? 2 : 1];                                 // staging values in flight (read one MFMA group ahead)
    float pm[POOL ? 8 : 1];                                 // pool: running max of the 2x2 window (upper pixel row first)
    // non-pool: step k = chunk k.  pool: step k = (chunk k >> 1, pixel row k & 1): two reads, running max, store at odd k
    auto store_read = [&](int sb, int k) {
        if (!POOL) {
            const int u = threadIdx.x + 256 * k;
            sv[0] = c3_rd128<0>(lds0 + stage_at(sb, u >> 3, u & 7));
        } else {
            const int u = threadIdx.x + 256 * (k >> 1);
            const int py = u >> 7, px = (u >> 3) & 15, c = u & 7;
#pragma unroll
            for (int d = 0; d < 2; ++d) sv[d] = c3_rd128<0>(lds0 + stage_at(sb, (2 * py + (k & 1)) * C3_TW + 2 * px + d, c));
        }
    };
    auto store_write = [&](const Tile& t, int k) {          // the values of store_read(.., k) have landed
        if (!POOL) {
            c3_tie(sv[0]);
            if (!(C3_ABL & 8)) *reinterpret_cast<u32x4*>(store_addr(t, k)) = sv[0];
        } else {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                c3_tie(sv[d]);
                const bf16x8 vv = __builtin_bit_cast(bf16x8, sv[d]);
#pragma unroll
                for (int e = 0; e < 8; ++e) pm[e] = (d || (k & 1)) ? fmaxf(pm[e], (float)vv[e]) : (float)vv[e];
            }
            if (k & 1) {
                const bf16x8 o = {(bf16_t)pm[0], (bf16_t)pm[1], (bf16_t)pm[2], (bf16_t)pm[3],
                                  (bf16_t)pm[4], (bf16_t)pm[5], (bf16_t)pm[6], (bf16_t)pm[7]};
                if (!(C3_ABL & 8)) *reinterpret_cast<bf16x8*>(store_addr(t, k >> 1)) = o;
            }
        }
    };

    unsigned ab[4][3];
#pragma unroll
    for (int ro = 0; ro < 4; ++ro)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int pp = (2 * wave + ro) * C3_IW + l31 + dx;
            ab[ro][dx] = lds0 + (unsigned)(pp * 128 + ((hi ^ ((pp >> 1) & 7)) << 4));
        }
    const int T = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // my tiles: blockIdx.x + i gridDim.x
    if (T <= 0) return;
    Tile cur = coords(blockIdx.x), prev = cur;
    {
        const __amdgpu_buffer_rsrc_t rs0 = image_rsrc(cur);
#pragma unroll
        for (int i = 0; i < C3_PIECES / 4; ++i) issue_piece(rs0, cur, 0, i);
    }
    c3_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (is_border(cur)) {
        zero_halo(cur, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    for (int i = 0; i < T; ++i) {
        const int buf = i & 1;
        // No branches inside the MFMA loop: the last tile re-fetches its own window (unused), and tile 0 "stores" the
        // still-unwritten staging buffer to its OWN output pixels, which the same threads overwrite with the real
        // values one phase later (same thread, same address: program order holds).
        const bool has_next = i + 1 < T;
        const Tile nxt = coords(blockIdx.x + (has_next ? i + 1 : i) * gridDim.x);
        const __amdgpu_buffer_rsrc_t nxt_rs = image_rsrc(nxt);

        // ---- 144 MFMAs: acc[nt][r] (32 channels x 32 pixels of tile row 2 wave + r); between them the next window's
        // DMA pieces are issued and the previous tile's staging buffer is drained to HBM
        f32x16 acc[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[nt][r][k] = 0.f;
        // B operand of (tap, ks, r): pixel pp = (2 wave + r + dy) * 34 + l31 + dx, chunk 2 ks + hi, swizzled:
        // byte address = pp * 128 + (((hi ^ v) * 16) ^ (32 ks)), v = (pp >> 1) & 7
        // (only 4 window rows x 3 dx distinct addresses per lane: row = r + dy)
        auto base = [&](int t, int r) { return ab[r + t / 3][t % 3] + (unsigned)(buf * C3_INBUF); };
        u32x4 ra[4], rb[4];                                      // [ks], double buffer over (tap, row) groups
        if (!(C3_ABL & 4)) {
            const unsigned a = base(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ra[ks] = c3_rd128<0>(a ^ (32u * ks));
        }
#pragma unroll
        for (int g = 0; g < ((C3_ABL & 4) ? 0 : 18); ++g) {
            const int t = g >> 1, r = g & 1;
            u32x4 (&cur_)[4] = (g & 1) ? rb : ra;
            u32x4 (&nxt_)[4] = (g & 1) ? ra : rb;
            // store chunk k of the previous tile: its staging read is issued here (ahead of this group's B reads, so the
            // counted wait below covers it) and goes to HBM after this group's MFMAs
            const int k = POOL ? ((g & 3) == 1 && g < 16 ? g >> 2 : -1) : (g >= 1 && g <= 8 ? g - 1 : -1);
            if (k >= 0) store_read(buf ^ 1, k);
            if (g + 1 < 18) {
                const unsigned a = base((g + 1) >> 1, (g + 1) & 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) nxt_[ks] = c3_rd128<0>(a ^ (32u * ks));
                c3_wait_lgkm<4>();                               // this group's four fragments (and older reads) landed
            } else {
                c3_wait_lgkm<0>();
            }
            if (g < C3_PIECES / 4 && !(C3_ABL & 2)) issue_piece(nxt_rs, nxt, buf ^ 1, g);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c3_tie(cur_[ks]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[nt][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][t][ks], __builtin_bit_cast(bf16x8, cur_[ks]), acc[nt][r], 0, 0, 0);
                }
            if (k >= 0) store_write(prev, k);
        }

        // ---- tail on the fp32 sums, bf16 into staging[buf]: pixel (2 wave + r, l31), channels in 8-byte groups
        if (C3_ABL & 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { asm volatile("" ::"v"(acc[0][r]), "v"(acc[1][r])); }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * nt + 8 * g + 4 * hi;
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(cst + c0);
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(cst + 64 + c0);
                    const f32x4 ch = *reinterpret_cast<const f32x4*>(cst + 128 + c0);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int opx = (2 * wave + r) * C3_TW + l31;
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float a = acc[nt][r][4 * g + e] + cb[e];
                            if (RELU) a = fmaxf(a, 0.f);
                            o[e] = fmaf(a, cs[e], ch[e]);
                        }
                        // channels 32 nt + 8 g + 4 hi + e -> 16-byte chunk 4 nt + g (swizzled), half hi
                        st4(reinterpret_cast<bf16_t*>(smem + stage_at(buf, opx, 4 * nt + g) + 8 * hi), o[0], o[1], o[2], o[3]);
                    }
                }
        }
        // one barrier per tile: staging[buf] complete, staging[buf ^ 1] drained, window[buf] consumed, and (after
        // each wave waited for its own pieces) window[buf ^ 1] landed
        c3_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (has_next && is_border(nxt)) {
            zero_halo(nxt, buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        prev = cur; cur = nxt;
    }
    // ---- drain: the last tile's staging buffer
    {
        const int sb = (T - 1) & 1;
#pragma unroll
        for (int k = 0; k < NSTORE; ++k) {
            store_read(sb, k);
            c3_wait_lgkm<0>();
            store_write(prev, k);
        }
    }
}

}  // namespace

extern "C" int gf_conv3x3_c64_ld(const void* x, const void* w, const float* bias, const float* scale, const float* shift,
                                 void* y, int64_t ldy, int B, int H, int W, int relu, int pool, int dtype, void* stream);
extern "C" int gf_conv3x3_c64(const void* x, const void* w, const float* bias, const float* scale, const float* shift,
                              void* y, int B, int H, int W, int relu, int pool, int dtype, void* stream) {
    return gf_conv3x3_c64_ld(x, w, bias, scale, shift, y, 64, B, H, W, relu, pool, dtype, stream);
}
extern "C" int gf_conv3x3_c64_ld(const void* x, const void* w, const float* bias, const float* scale, const float* shift,
                                 void* y, int64_t ldy, int B, int H, int W, int relu, int pool, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return GF_ERR_SHAPE;
    if (dtype != GF_BF16) return GF_ERR_DTYPE;
    if (H % C3_TH || W % C3_TW) return GF_ERR_UNSUPPORTED;
    if ((int64_t)H * W * 128 >= (1ll << 31)) return GF_ERR_UNSUPPORTED;      // 32-bit byte offsets inside one image (buffer loads)
    if (ldy < 64 || ldy % 8) return GF_ERR_ALIGN;                            // whole 16-byte chunks of a pixel
    if ((int64_t)H * W * ldy >= (1ll << 31)) return GF_ERR_UNSUPPORTED;      // (32-bit element offsets inside one output image)
    C3Params p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(w); p.bias = bias; p.scale = scale; p.shift = shift;
    p.y = static_cast<bf16_t*>(y); p.ldy = (int)ldy; p.B = B; p.H = H; p.W = W; p.relu = relu;
    const size_t lds = C3_LDS;
    const int tiles = (W / C3_TW) * (H / C3_TH) * B;
    const int grid = tiles < 256 ? tiles : 256;            // one persistent workgroup per CU
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    void (*k)(C3Params) = pool ? (relu ? conv3x3_c64_kernel<true, true> : conv3x3_c64_kernel<true, false>)
                               : (relu ? conv3x3_c64_kernel<false, true> : conv3x3_c64_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    k<<<dim3(grid), 256, lds, st>>>(p);
    return (int)hipGetLastError();
}
