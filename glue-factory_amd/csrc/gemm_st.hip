// y = [x0 | x1] W^T + bias (+ residual), bf16, for the shapes that dominate the matcher blocks (K = 256 / 512, N a
// multiple of 256, M a multiple of 64): the "weights stationary, activations streamed" counterpart of gemm_ws.hip.
// Same reference lines (every nn.Linear forward and input-gradient GEMM of gluefactory/models/matchers/lightglue.py:
// 131-221,271-290 and the Conv1d(k=1) layers of superglue.py:70-160 / gluestick.py:465-586).
//
// gemm_ws keeps a wave's 32 activation rows in registers and walks the weight through LDS; its workgroups live as
// long as the kernel and every one of them is a serial chain (rows in -> fragments -> per weight slice: barrier, MFMAs,
// accumulators out through LDS): counters show both the matrix and the vector pipe ~85-90 % idle at 3 TB/s.  Here
//   * a workgroup is 8 waves and owns 256 output channels: wave w keeps the [32 x K] weight block of its 32 channels in
//     REGISTERS (64 / 128 VGPRs) for the whole kernel;
//   * the activation rows stream through a 3-stage LDS-DMA ring in 32 KiB tiles (64 rows at K = 256, 32 at K = 512),
//     two tiles ahead of the MFMAs, 16-byte chunks XOR-swizzled by (row & 15) on the source side so that the B-operand
//     ds_read_b128 (32 rows, one chunk) is conflict-free; every wave reads every row -- the same LDS traffic per MFMA as
//     gemm_ws, but no staging registers, no per-slice barrier, one barrier per tile;
//   * outputs are produced "swapped" (channel on the MFMA i axis = registers, row on j = lane): bias and residual are
//     lane-local; a v_permlane32_swap pairs the two half-rows of a lane and its partner into 8 consecutive channels, so
//     residual loads and stores are 16 bytes per lane.  Loads (residual), LDS-DMA pieces
//     and stores retire through ONE in-order counter on this part, so their program order is fixed by hand
//     (inline asm) and every wait is a counted vmcnt: residual(t), DMA(t + 2), MFMAs(t), stores(t).
#include <type_traits>
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void gs_lds_void;
typedef const __attribute__((address_space(1))) void gs_glb_void;

constexpr int GS_TILE = 32768;                // bytes of one activation tile
#ifndef GS_NSTAGE_
#define GS_NSTAGE_ 3                         // ring depth (probe builds: 2 = two workgroups per CU, 4 = three tiles ahead)
#endif
#ifndef GS_WGS
#define GS_WGS 256                           // resident workgroups the grid is sized for
#endif
constexpr int GS_NSTAGE = GS_NSTAGE_, GS_PD = GS_NSTAGE - 1;   // stages, prefetch distance in tiles

struct GsParams {
    const bf16_t* x0; const bf16_t* x1; const bf16_t* w; const float* bias; const bf16_t* res; const bf16_t* res2; bf16_t* y;
    const float* cs;          // [M, 64] (cos, sin) per channel pair of a 64-wide head (ROT kernels)
    int M, N;
    int64_t ld0, ld1, ldw, ldr, ldr2, ldy;
};

template <int OFF> __device__ __forceinline__ u32x4 gs_rd128(unsigned a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void gs_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void gs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename V> __device__ __forceinline__ void gs_tie(V& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ u32x4 gs_ld128(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// (a store of more than 64 bits reads its data registers over several cycles: the wait states a compiler-issued store
// would get are part of the asm, or the next VALU write to those registers clobbers lanes 12-15 of every row)
__device__ __forceinline__ void gs_st128(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
// The accumulator layout gives a lane 4 consecutive channels of a row per register group g (8 g + 4 hi + e); its partner
// lane ^ 32 holds the other 4 of that 8-channel group.  v_permlane32_swap trades the upper half's copy of group g0 for
// the lower half's copy of group g1: afterwards lanes 0-31 hold the 8 channels of g0, lanes 32-63 those of g1, in
// ascending order {a, b} -- 16-byte instead of 8-byte global accesses.  The same swap undoes it.
__device__ __forceinline__ void gs_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

// KF = K / 16 (16 or 32); TWO: the K columns come half from x0, half from x1; RES: number of residual inputs (0, 1, or 2:
// where a loss head's parked gradient meets a block's residual gradient, both ride in the input-gradient GEMM instead of
// being added by a separate 3-pass kernel first); ROT: rotary epilogue (lightglue.py:42-49,159-160) on every channel of the
// launch, head dim 64
template <int KF, bool TWO, int RES, bool ROT>
__global__ __launch_bounds__(512, 2) void gemm_st_kernel(GsParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int K = 16 * KF, ROWB = 2 * K, CPR = ROWB / 16;     // bytes / 16-byte chunks of one activation row
    constexpr int TR = GS_TILE / ROWB, NRB = TR / 32;             // rows, 32-row blocks per tile
    constexpr int NR = RES * 2 * NRB + (ROT ? 4 * NRB : 0), NS = 2 * NRB;   // residual + (cos, sin) loads / stores per tile and wave
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int ch0 = blockIdx.y * 256 + wave * 32;                 // this wave's 32 output channels
    const int ntile = p.M / TR;
    const int T = (ntile - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // my tiles: blockIdx.x + i gridDim.x
    if (T <= 0) return;

    // ---- weights (A operands): lane = channel ch0 + l31, k-step s = elements [16 s + 8 hi, + 8)
    bf16x8 wf[KF];
#pragma unroll
    for (int s = 0; s < KF; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(p.w + (int64_t)(ch0 + l31) * p.ldw + 16 * s + 8 * hi);
    float bias16[16];                                             // channels ch0 + crow(r, hi)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias16[r] = p.bias ? p.bias[ch0 + crow(r, hi)] : 0.f;
    gs_wait_vm<0>();                                              // (compiler-issued loads above: nothing of them in flight below)

    // ---- DMA of tile `ti` (clamped: the tail re-fetches the last tile, nobody reads it) into `stage`: 32 pieces of 1 KiB,
    // four per wave; chunk position lin = 64 piece + lane of the tile is row lin / CPR, slot lin % CPR, and holds the
    // row's logical chunk slot ^ (row & 15)
    auto issue = [&](int i, int stage) {
        const int ti = (int)blockIdx.x + min(i, T - 1) * (int)gridDim.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = 4 * wave + j;
            const int lin = 64 * piece + lane;
            const int row = lin / CPR, c = (lin % CPR) ^ (row & 15);
            const int64_t grow = (int64_t)ti * TR + row;
            const bf16_t* src = (TWO && c >= CPR / 2) ? p.x1 + grow * p.ld1 + (c - CPR / 2) * 8 : p.x0 + grow * p.ld0 + c * 8;
            __builtin_amdgcn_global_load_lds((gs_glb_void*)src, (gs_lds_void*)(smem + stage * GS_TILE + piece * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < GS_PD; ++t) issue(t, t);

    // B-operand read addresses: row 32 rb + l31 (rb as immediate), chunk (2 s + hi) ^ (l31 & 15); s = 8 sh + sl with
    // sh as an immediate offset of 256 bytes (the swizzle only touches the low four chunk bits)
    unsigned a0[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) a0[sl] = lds0 + (unsigned)(l31 * ROWB + (((2 * sl + hi) ^ (l31 & 15)) << 4));
    // this lane's output / residual accesses: row l31 of a 32-row block, the 8 channels of group 2 j + hi (j = 0, 1)
    const int64_t ycol = ch0 + 8 * hi;

    for (int i = 0; i < T; ++i) {
        const int stage = i % GS_NSTAGE;
        const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)i * gridDim.x) * TR;
        // tile i landed?  in-order counter: behind its pieces sit stores(i-2), residual(i-1), DMA(i+1), stores(i-1)
        // (tile i < PD came with the prologue: PD-1-i later prologue tiles and i whole tiles of traffic sit behind it;
        //  steady state: stores(i-PD) and PD-1 whole tiles)
        constexpr int PER = NR + 4 + NS;
        if (i >= GS_PD) gs_wait_vm<NS + (GS_PD - 1) * PER>();
        else if (i == 0) gs_wait_vm<(GS_PD - 1) * 4>();
        else if (i == 1) gs_wait_vm<(GS_PD > 1 ? (GS_PD - 2) * 4 + PER : 0)>();
        else gs_wait_vm<(GS_PD > 2 ? (GS_PD - 3) * 4 + 2 * PER : 0)>();
        __builtin_amdgcn_s_barrier();                             // ... everyone's pieces; the stage of tile i-1 is free
        __builtin_amdgcn_sched_barrier(0);
        u32x4 rr[NRB][2], rq[NRB][2];
        if (RES >= 1) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int j = 0; j < 2; ++j) rr[rb][j] = gs_ld128(p.res + (row0 + 32 * rb + l31) * p.ldr + ycol + 16 * j);
        }
        if (RES == 2) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int j = 0; j < 2; ++j) rq[rb][j] = gs_ld128(p.res2 + (row0 + 32 * rb + l31) * p.ldr2 + ycol + 16 * j);
        }
        u32x4 cc[NRB][4];                                          // (cos, sin) of this lane's channel pairs, per register group
        if (ROT) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    cc[rb][g] = gs_ld128(p.cs + (row0 + 32 * rb + l31) * 64 + (wave & 1) * 32 + 8 * g + 4 * hi);
        }
        issue(i + GS_PD, (i + GS_PD) % GS_NSTAGE);
        unsigned a[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) a[sl] = a0[sl] + stage * GS_TILE;

        f32x16 acc[NRB];
        // fragments four at a time (group G = k-steps 4 G .. 4 G + 3), two groups in flight (the LDS counter holds 15)
        u32x4 xa[4], xb[4];
        auto rd = [&](u32x4 (&dst)[4], auto RB, auto G) {
            constexpr int rb = decltype(RB)::value, g = decltype(G)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = gs_rd128<rb * 32 * ROWB + ((4 * g) / 8) * 256>(a[(4 * g + q) & 7]);
        };
        auto mma = [&](u32x4 (&src)[4], auto RB, auto G) {
            constexpr int rb = decltype(RB)::value, g = decltype(G)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gs_tie(src[q]);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[4 * g + q], __builtin_bit_cast(bf16x8, src[q]), acc[rb], 0, 0, 0);
            }
        };
        auto block = [&](auto RB) {
            constexpr int rb = decltype(RB)::value, NG = KF / 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = bias16[r];
            rd(xa, RB, std::integral_constant<int, 0>{});
            rd(xb, RB, std::integral_constant<int, 1>{});
            auto step = [&](u32x4 (&buf)[4], auto G) {
                constexpr int g = decltype(G)::value;
                if constexpr (g < NG) {
                    if constexpr (g == NG - 1) gs_wait_lgkm<0>(); else gs_wait_lgkm<4>();
                    mma(buf, RB, G);
                    if constexpr (g + 2 < NG) rd(buf, RB, std::integral_constant<int, g + 2>{});
                }
            };
            step(xa, std::integral_constant<int, 0>{}); step(xb, std::integral_constant<int, 1>{});
            step(xa, std::integral_constant<int, 2>{}); step(xb, std::integral_constant<int, 3>{});
            step(xa, std::integral_constant<int, 4>{}); step(xb, std::integral_constant<int, 5>{});
            step(xa, std::integral_constant<int, 6>{}); step(xb, std::integral_constant<int, 7>{});
        };
        block(std::integral_constant<int, 0>{});
        if constexpr (NRB > 1) block(std::integral_constant<int, 1>{});
        // ---- epilogue: (+ residual), bf16, 16-byte stores; residual(i) sits in front of DMA(i+2) only
        if (RES || ROT) gs_wait_vm<4>();
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            bf16_t* yp = p.y + (row0 + 32 * rb + l31) * p.ldy + ycol;
#pragma unroll
            for (int j = 0; j < 2; ++j) {                         // register groups g0 = 2 j, g1 = 2 j + 1
                float v[2][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[0][e] = acc[rb][8 * j + e]; v[1][e] = acc[rb][8 * j + 4 + e]; }
                if (ROT) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        gs_tie(cc[rb][2 * j + t]);
                        const f32x4 c = __builtin_bit_cast(f32x4, cc[rb][2 * j + t]);
                        const float o0 = v[t][0] * c[0] - v[t][1] * c[1], o1 = v[t][1] * c[0] + v[t][0] * c[1];
                        const float o2 = v[t][2] * c[2] - v[t][3] * c[3], o3 = v[t][3] * c[2] + v[t][2] * c[3];
                        v[t][0] = o0; v[t][1] = o1; v[t][2] = o2; v[t][3] = o3;
                    }
                }
                auto add_res = [&](u32x4& src) {
                    gs_tie(src);
                    unsigned a0_ = src[0], a1_ = src[1], b0_ = src[2], b1_ = src[3];
                    gs_swap(a0_, b0_);                            // back to "4 channels of g0 / 4 of g1 per lane"
                    gs_swap(a1_, b1_);
                    const bf16x4 r0 = __builtin_bit_cast(bf16x4, u32x2{a0_, a1_}), r1 = __builtin_bit_cast(bf16x4, u32x2{b0_, b1_});
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[0][e] += (float)r0[e]; v[1][e] += (float)r1[e]; }
                };
                if (RES >= 1) add_res(rr[rb][j]);
                if (RES == 2) add_res(rq[rb][j]);
                const bf16x4 o0 = {(bf16_t)v[0][0], (bf16_t)v[0][1], (bf16_t)v[0][2], (bf16_t)v[0][3]};
                const bf16x4 o1 = {(bf16_t)v[1][0], (bf16_t)v[1][1], (bf16_t)v[1][2], (bf16_t)v[1][3]};
                const u32x2 p0 = __builtin_bit_cast(u32x2, o0), p1 = __builtin_bit_cast(u32x2, o1);
                unsigned q0 = p0[0], q1 = p0[1], q2 = p1[0], q3 = p1[1];
                gs_swap(q0, q2);
                gs_swap(q1, q3);
                gs_st128(yp + 16 * j, u32x4{q0, q1, q2, q3});
            }
        }
    }
    gs_wait_vm<0>();                                              // the re-fetched tail tiles and the last stores
}

template <int KF, bool TWO, int RES, bool ROT = false>
int gs_launch(const GsParams& p, hipStream_t st) {
    constexpr int TR = GS_TILE / (32 * KF);
    const size_t lds = (size_t)GS_NSTAGE * GS_TILE;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_st_kernel<KF, TWO, RES, ROT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int ntile = p.M / TR, ncg = p.N / 256;
    int gx = GS_WGS / ncg;                                    // one workgroup per CU over all channel groups
    if (gx < 1) gx = 1;
    if (gx > ntile) gx = ntile;
    gemm_st_kernel<KF, TWO, RES, ROT><<<dim3(gx, ncg), dim3(512), lds, st>>>(p);
    return (int)hipGetLastError();
}

}  // namespace

// Internal (not part of the C ABI): called by gf_gemm first; GF_ERR_UNSUPPORTED = "use the register-resident kernel".
int gf_gemm_stream_try(const void* x0, const void* x1, const void* w, const float* bias, const void* res, void* y,
                       const float* cs, int rot_n, int M, int N, int K0, int K1, int64_t ld0, int64_t ld1, int64_t ldw,
                       int64_t ldr, int64_t ldy, hipStream_t st, const void* res2, int64_t ldr2) {
    const int K = K0 + K1;
    if ((K != 256 && K != 512) || N % 256 || M % 64 || (K1 && K1 != K0)) return GF_ERR_UNSUPPORTED;
    if (cs && (rot_n % 256 || rot_n <= 0 || rot_n > N || res || K1)) return GF_ERR_UNSUPPORTED;
    if (res2 && (!res || cs)) return GF_ERR_UNSUPPORTED;
    GsParams p;
    p.x0 = static_cast<const bf16_t*>(x0); p.x1 = static_cast<const bf16_t*>(x1); p.w = static_cast<const bf16_t*>(w);
    p.bias = bias; p.res = static_cast<const bf16_t*>(res); p.res2 = static_cast<const bf16_t*>(res2);
    p.y = static_cast<bf16_t*>(y); p.cs = cs;
    p.M = M; p.N = N; p.ld0 = ld0; p.ld1 = ld1; p.ldw = ldw; p.ldr = ldr; p.ldr2 = ldr2; p.ldy = ldy;
    if (cs) {       // rotated channel groups first, the rest (the v third of a fused qkv projection) as a plain launch
        p.N = rot_n;
        if (int e = K == 256 ? gs_launch<16, false, 0, true>(p, st) : gs_launch<32, false, 0, true>(p, st)) return e;
        if (N == rot_n) return 0;
        p.N = N - rot_n; p.w += (int64_t)rot_n * ldw; p.y += rot_n; p.cs = nullptr;
        if (p.bias) p.bias += rot_n;
        return K == 256 ? gs_launch<16, false, 0>(p, st) : gs_launch<32, false, 0>(p, st);
    }
    if (res2) return K1 ? (K == 256 ? gs_launch<16, true, 2>(p, st) : gs_launch<32, true, 2>(p, st))
                        : (K == 256 ? gs_launch<16, false, 2>(p, st) : gs_launch<32, false, 2>(p, st));
#define GS_GO(KF) (K1 ? (res ? gs_launch<KF, true, 1>(p, st) : gs_launch<KF, true, 0>(p, st)) \
                      : (res ? gs_launch<KF, false, 1>(p, st) : gs_launch<KF, false, 0>(p, st)))
    return K == 256 ? GS_GO(16) : GS_GO(32);
#undef GS_GO
}
