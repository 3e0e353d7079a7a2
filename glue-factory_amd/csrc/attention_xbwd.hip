// Backward of LightGlue's BIDIRECTIONAL cross attention as one kernel per image side (bf16).
//
// The cross block uses ONE similarity for both directions (lightglue.py:203-216):
//     sim = qk0 qk1^T,   m0 = softmax_j(sim) v1,   m1 = softmax_i(sim^T) v0.
// Run as two independent attentions (gf_attn_bwd_acc twice) the backward recomputes sim and dP in four kernels: 14 MFMA
// products per score tile (dQ kernel: S, dP, dQ; dK/dV kernel: S, dP, dV, dK; twice).  Here the image that OWNS a token
// block (32 tokens per wave, on the MFMA lane axis) streams the other image's qk / dO / v tiles once and takes from the one
// score tile everything that ends in its own gradients -- with X the owner, Y the streamed image, t / u their tokens:
//     s[u,t]   = qkY[u] . qkX[t]                                    (shared logits)
//     P_Y[u,t] = exp(s - lse_Y[u])      dP_Y[u,t] = dO_Y[u] . v_X[t]      direction "Y queries X":  dS_Y = P_Y (dP_Y - delta_Y[u])
//     P_X[t,u] = exp(s - lse_X[t])      dP_X[t,u] = dO_X[t] . v_Y[u]      direction "X queries Y":  dS_X = P_X (dP_X - delta_X[t])
//     d v_X[t]  += sum_u P_Y[u,t] dO_Y[u]
//     d qk_X[t] += sum_u (dS_Y[u,t] + dS_X[t,u]) qk_Y[u]            (qk_X is key in one direction and query in the other)
// = 5 products per tile and side, 10 per pair of sides instead of 14; no accumulation between launches (every gradient
// row has exactly one owner), no dQ / dK split.  The per-token statistics of BOTH directions (-lse log2e / r, -delta) come
// from a small pass over o and dO in front (attn_stats_kernel): the streamed side's enter as the initial values of the S and
// dP_Y accumulators (per-register vectors from LDS, as in the dK/dV kernel), the owner's as a lane constant / a C-operand
// splat.
// One wave = 32 owner tokens, 4 waves per workgroup, ONE workgroup per CU (the five products need ~300 registers: one wave
// per SIMD, up to 512 unified registers); streamed tiles of 64 tokens x {qk, dO, v} + statistics arrive by
// `global_load ... lds` in a 3-stage ring (attention.hip: chunk swizzle, ds_read_b128 row reads, ds_read_b64_tr_b16
// transposed reads).
#include "gf_common.h"
#include "gf_amd.h"
#include "attn_common.h"

namespace gfattn {
namespace {

#ifndef XB_ABL
#define XB_ABL 0           // timing-only ablations (wrong results; tools/probe/time_xbwd.py): 1 no exponentials, 2 no output MFMAs,
#endif                     // 4 no dP_X product, 8 no transposed LDS reads, 16 no statistics pass, 32 no DMA
constexpr int XB_STATS = 3 * FT_TILE;                 // per wave: 16 lse | 16 delta | duplicates (256 B)
constexpr int XB_STAGE = 3 * FT_TILE + 1024;
constexpr int XB_NSTAGE = 3;

struct XbParams {
    const bf16_t* qk; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
    bf16_t* dqk; bf16_t* dv;
    const float* lse;       // [B2, H, N]
    float* stat;            // [2, B2, H, N] workspace: -lse log2(e) / rr | -delta
    int B2, pair, H, N;     // image b attends to (and is attended by) image (b + pair) mod B2
    int64_t sqb, sqn, sqh, svb, svn, svh, sob, son, soh, sdob, sdon, sdoh, sdqb, sdqn, sdqh, sdvb, sdvn, sdvh;
    float scale, p2, rr;
};

// stat[0] = -lse log2(e) / rr, stat[1] = -sum_d o dO per (image, head, token): one 16-lane group per head of a token
__global__ __launch_bounds__(256) void attn_stats_kernel(XbParams p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);                  // token index over B2 * N
    if (row >= p.B2 * p.N) return;
    const int b = row / p.N, n = row % p.N;
    const int lane = threadIdx.x & 63, h = lane >> 4, c = (lane & 15) * 4;
    if (h >= p.H) return;
    const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(p.o + b * p.sob + n * p.son + h * p.soh + c);
    const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(p.dout + b * p.sdob + n * p.sdon + h * p.sdoh + c);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += (float)o4[e] * (float)d4[e];
    s = row16_allsum(s);
    if ((lane & 15) == 0) {
        const int64_t i = ((int64_t)b * p.H + h) * p.N + n;
        p.stat[i] = -p.lse[i] * GF_LOG2E / p.rr;
        p.stat[(int64_t)p.B2 * p.H * p.N + i] = -s;
    }
}

// One wave per SIMD: nothing but this wave's own instruction stream can fill the matrix pipe while the exponentials run.  A
// tile (two half tiles of 32 streamed tokens) is therefore processed as  scores<0>, scores<1>, update<0>, update<1>:  the 12
// MFMAs of the second half tile's scores execute under the first half tile's exponentials, and the 8 MFMAs of update<0>
// under the exponentials of update<1> (measured against the straight order scores, update, scores, update: see DESIGN.md).
struct XbHalf { f32x16 sy, dpy, dpx, lv; };

// the three score-side products of half tile QB: sy = s - lse_Y, dpy = dP_Y - delta_Y, dpx = dP_X - delta_X, and lv = lse_Y - lse_X
// (s - lse_X = sy + lv: a subtract and an add per score; running the S product a second time from a -lse_X splat instead --
// four more MFMAs, no VALU -- measured 1.8 ms per step SLOWER: the kernel is not bound by its VALU issue slots alone)
template <int QB>
__device__ __forceinline__ void xb_scores(XbHalf& o, const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], const bf16x8 (&dof)[4],
                                          const f32x16& ndx, float nlx, const unsigned (&aR)[4], unsigned aS) {
    f32x4 l4[4], d4[4];
#define GF_ST(g) l4[g] = __builtin_bit_cast(f32x4, lds_rd128<(2 * QB + (g >> 1)) * 256 + 32 * (g & 1)>(aS)); \
                 d4[g] = __builtin_bit_cast(f32x4, lds_rd128<(2 * QB + (g >> 1)) * 256 + 32 * (g & 1) + 64>(aS));
    GF_ST(0) GF_ST(1) GF_ST(2) GF_ST(3)
#undef GF_ST
    u32x4 qa[4], da[4], va[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qa[s] = lds_rd128<QB * 4096>(aR[s]);
    wait_lgkm<4>();
#pragma unroll
    for (int g = 0; g < 4; ++g) {                                   // the streamed side's statistics ARE the initial values
        tie(l4[g]);
        tie(d4[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o.sy[4 * g + e] = l4[g][e];
            o.lv[4 * g + e] = nlx - l4[g][e];                       // s_X = s_Y + (-lse_X) - (-lse_Y)  (exponent units)
            o.dpy[4 * g + e] = d4[g][e];
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) da[s] = lds_rd128<FT_TILE + QB * 4096>(aR[s]);
    // (the k-steps of a product stay chained on ONE accumulator: interleaving the three chains measured 6 % slower)
    wait_lgkm<4>();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tie(qa[s]);
        mma16(o.sy, as_frag(qa[s]), kf[s]);                         // s[u][t] - lse_Y[u]
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) va[s] = lds_rd128<2 * FT_TILE + QB * 4096>(aR[s]);
    wait_lgkm<4>();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tie(da[s]);
        mma16(o.dpy, as_frag(da[s]), vf[s]);                        // dP_Y[u][t] - delta_Y[u]
    }
    wait_lgkm<0>();
#pragma unroll
    for (int s = 0; s < 4; ++s) tie(va[s]);
#if XB_ABL & 4
    o.dpx = ndx;
#pragma unroll
    for (int r = 0; r < 4; ++r) o.dpx[r] += __builtin_bit_cast(float, va[r][0]) * 1e-30f;
#else
    o.dpx = mma16c(as_frag(va[0]), dof[0], ndx);                    // dP_X[t][u] - delta_X[t]
#pragma unroll
    for (int s = 1; s < 4; ++s) mma16(o.dpx, as_frag(va[s]), dof[s]);
#endif
}

// exponentials, D = dS_Y + dS_X, and the two output products of half tile QB
template <int QB, bool PRE, typename Mid>
__device__ __forceinline__ void xb_update(XbHalf& o, f32x16 (&dqk)[2], f32x16 (&dv)[2], const unsigned (&aT)[4], float c, Mid&& mid) {
    // transposed operands: [t][db] -> rows 16t + 4hi + {0..3} (lo) and + 8 (hi half), columns db*32 + l31
    u32x2 dot[2][2][2], qt[2][2][2];
#define GF_TR(dst, base, t, db) dst[t][db][0] = lds_rdtr<base + QB * 4096 + t * 2048>(aT[db]); \
                                dst[t][db][1] = lds_rdtr<base + QB * 4096 + t * 2048 + 1024>(aT[2 + db]);
    GF_TR(dot, FT_TILE, 0, 0) GF_TR(dot, FT_TILE, 0, 1) GF_TR(dot, FT_TILE, 1, 0) GF_TR(dot, FT_TILE, 1, 1)
    mid();                                                          // DMA issue rides in the VALU gap
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#if XB_ABL & 1
        const float py = o.sy[r] * 0.5f, px = (o.sy[r] + o.lv[r]) * 0.25f;
#else
        const float py = fast_exp2(PRE ? o.sy[r] : o.sy[r] * c);
        const float px = fast_exp2(PRE ? o.sy[r] + o.lv[r] : (o.sy[r] + o.lv[r]) * c);
#endif
        o.sy[r] = py;
        o.dpy[r] = fmaf(px, o.dpx[r], py * o.dpy[r]);               // D = dS_Y + dS_X overwrites dP_Y
    }
    wait_lgkm<0>();
    {
        const bf16x8 pf0 = cvt_frag(o.sy, 0), pf1 = cvt_frag(o.sy, 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) { tie(dot[0][db][0]); tie(dot[0][db][1]); tie(dot[1][db][0]); tie(dot[1][db][1]); }
#if XB_ABL & 2
#pragma unroll
        for (int db = 0; db < 2; ++db)
            dv[db][0] += (float)pf0[0] + (float)pf1[0] + __builtin_bit_cast(float, dot[0][db][0][0]) * 1e-30f + __builtin_bit_cast(float, dot[1][db][1][0]) * 1e-30f;
#else
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            mma16(dv[db], as_frag(dot[0][db][0], dot[0][db][1]), pf0);
            mma16(dv[db], as_frag(dot[1][db][0], dot[1][db][1]), pf1);
        }
#endif
    }
    GF_TR(qt, 0, 0, 0) GF_TR(qt, 0, 0, 1) GF_TR(qt, 0, 1, 0) GF_TR(qt, 0, 1, 1)      // (late: 16 registers fewer across the exponentials)
#undef GF_TR
    {
        const bf16x8 pf0 = cvt_frag(o.dpy, 0), pf1 = cvt_frag(o.dpy, 1);
        wait_lgkm<0>();
#pragma unroll
        for (int db = 0; db < 2; ++db) { tie(qt[0][db][0]); tie(qt[0][db][1]); tie(qt[1][db][0]); tie(qt[1][db][1]); }
#if XB_ABL & 2
#pragma unroll
        for (int db = 0; db < 2; ++db)
            dqk[db][0] += (float)pf0[0] + (float)pf1[0] + __builtin_bit_cast(float, qt[0][db][0][0]) * 1e-30f + __builtin_bit_cast(float, qt[1][db][1][0]) * 1e-30f;
#else
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            mma16(dqk[db], as_frag(qt[0][db][0], qt[0][db][1]), pf0);
            mma16(dqk[db], as_frag(qt[1][db][0], qt[1][db][1]), pf1);
        }
#endif
    }
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_xbwd_bf16_kernel(XbParams p) {
    constexpr int PPW = 2;                                           // 1-KiB DMA pieces per wave and matrix
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int ntb = (p.N + 127) / 128;
    const int total = ntb * p.H * p.B2;
    const int lb = xcd_remap(blockIdx.x, total);
    const int tb = lb % ntb, h = (lb / ntb) % p.H, b = lb / (ntb * p.H);
    const int by = (b + p.pair) % p.B2;                              // the streamed image
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5, s16 = lane & 15, half = (lane >> 4) & 1;
    const int trow = tb * 128 + wave * 32 + l31;
    const int tld = min(trow, p.N - 1);

    const bf16_t* qkx = p.qk + b * p.sqb + h * p.sqh;
    const bf16_t* vx = p.v + b * p.svb + h * p.svh;
    const bf16_t* dox = p.dout + b * p.sdob + h * p.sdoh;
    const bf16_t* qky = p.qk + by * p.sqb + h * p.sqh;
    const bf16_t* vy = p.v + by * p.svb + h * p.svh;
    const bf16_t* doy = p.dout + by * p.sdob + h * p.sdoh;
    const int64_t plane = (int64_t)p.B2 * p.H * p.N;
    const float* lsey = p.stat + ((int64_t)by * p.H + h) * p.N;      // stat[0] of the streamed image
    const float* dely = lsey + plane;                                // stat[1]

    // ---- DMA: chunk (PPW wave + i) * 64 + lane of a tile -> row, swizzled source column.  Per-lane state = one 32-bit element
    // offset per piece and matrix (the tile advance rides in the uniform base): registers are the scarce resource here
    int oq[PPW], od[PPW], ov[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int drow = (PPW * wave + i) * 8 + (lane >> 3);
        const int dcol = ((lane & 7) ^ fswz(drow)) * 8;
        oq[i] = drow * (int)p.sqn + dcol;
        od[i] = drow * (int)p.sdon + dcol;
        ov[i] = drow * (int)p.svn + dcol;
    }
    const float* statp = ((lane & 16) ? dely : lsey) + 16 * wave + s16;
    const int64_t qstep = 64 * p.sqn, dostep = 64 * p.sdon, vstep = 64 * p.svn;
    // part 0: qk pieces + statistics + dO pieces, part 1: v pieces (issued in the VALU gaps of the two half tiles)
    auto issue_part = [&](int part, int t, int stage) {
        char* sb = smem + stage * XB_STAGE;
        const bf16_t* tq = qky + t * qstep;
        const bf16_t* td = doy + t * dostep;
        const bf16_t* tv = vy + t * vstep;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (part == 0) {
                dma16(tq + oq[i], sb + (PPW * wave + i) * 1024);
                dma16(td + od[i], sb + FT_TILE + (PPW * wave + i) * 1024);
            } else {
                dma16(tv + ov[i], sb + 2 * FT_TILE + (PPW * wave + i) * 1024);
            }
        }
        if (part == 0) dma4(statp + t * 64, sb + XB_STATS + wave * 256);
    };
    constexpr int VM_TILE = 3 * PPW + 1;                             // vector-memory operations of one tile per wave
    const int nt = p.N / 64;                                         // (N % 64 == 0: checked by the launcher)
    issue_part(0, 0, 0); issue_part(1, 0, 0);
    issue_part(0, min(1, nt - 1), 1); issue_part(1, min(1, nt - 1), 1);

    const float c = PRE ? 1.f : p.rr;
    bf16x8 kf[4], vf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const bf16x8*>(qkx + (int64_t)tld * p.sqn + 16 * s + 8 * hi);
        if (p.p2 != 1.f) kf[s] = scale_frag(kf[s], p.p2);
        vf[s] = *reinterpret_cast<const bf16x8*>(vx + (int64_t)tld * p.svn + 16 * s + 8 * hi);
        dof[s] = *reinterpret_cast<const bf16x8*>(dox + (int64_t)tld * p.sdon + 16 * s + 8 * hi);
    }
    const int64_t sx = ((int64_t)b * p.H + h) * p.N + tld;
    const float nlx = p.stat[sx], ndx_ = p.stat[plane + sx];
    f32x16 dqk[2], dv[2], ndx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dqk[0][r] = 0.f; dqk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; ndx[r] = ndx_; }

    // ---- per-lane LDS read addresses (see attn_bwd_dkv_bf16_kernel), advanced IN PLACE from stage to stage
    unsigned aR[4], aT[4];
    {
        const unsigned rb = l31 * 128 + 16 * (hi ^ fswz(l31));
#pragma unroll
        for (int s = 0; s < 4; ++s) aR[s] = lds0 + (rb ^ (32 * s));
        const int bq = s16 >> 3;
        const unsigned tbs = (4 * hi + (s16 >> 2)) * 128 + 8 * (s16 & 1) + 16 * ((2 * half + ((s16 & 3) >> 1)) ^ (4 * bq + hi));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int db = 0; db < 2; ++db) aT[2 * u + db] = lds0 + (tbs ^ (32 * u) ^ (64 * db));
    }
    unsigned aS = lds0 + XB_STATS + 16 * hi;

    tie(ndx);
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        wait_vm<VM_TILE>();                                          // tile t landed (this wave's share; tile t + 1 may be in flight)
        __builtin_amdgcn_s_barrier();                                // ... everyone's; the stage of tile t - 1 is free
        __builtin_amdgcn_sched_barrier(0);
        const int nstage = stage == 0 ? 2 : stage - 1;
        const int tn = min(t + 2, nt - 1);                           // tiles past the end re-fetch the last one (constant wait count)
        XbHalf h0;
        xb_scores<0>(h0, kf, vf, dof, ndx, nlx, aR, aS);
        xb_update<0, PRE>(h0, dqk, dv, aT, c, [&] { issue_part(0, tn, nstage); });
        xb_scores<1>(h0, kf, vf, dof, ndx, nlx, aR, aS);
        xb_update<1, PRE>(h0, dqk, dv, aT, c, [&] { issue_part(1, tn, nstage); });
        const int step = stage == 2 ? -2 * XB_STAGE : XB_STAGE;      // ring: the read addresses follow the stage
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] += step; aT[i] += step; }
        aS += step;
        stage = stage == 2 ? 0 : stage + 1;
    }
    wait_vm<0>();                                                    // the re-fetched tail tiles
    if (trow < p.N) {
        bf16_t* dqp = p.dqk + b * p.sdqb + h * p.sdqh + (int64_t)trow * p.sdqn;
        bf16_t* dvp = p.dv + b * p.sdvb + h * p.sdvh + (int64_t)trow * p.sdvn;
        store_row<bf16_t, 64>(dqp, dqk, p.scale, hi);
        store_row<bf16_t, 64>(dvp, dv, 1.f, hi);
    }
}

}  // namespace
}  // namespace gfattn

extern "C" int gf_attn_cross_bwd(const void* qk, const void* v, const void* o, const void* dout, const float* lse, float* stat,
                                 void* dqk, void* dv, int B2, int pair, int H, int N, int D,
                                 const int64_t* qk_strides, const int64_t* v_strides, const int64_t* o_strides,
                                 const int64_t* do_strides, const int64_t* dqk_strides, const int64_t* dv_strides,
                                 float scale, int dtype, void* stream) {
    using namespace gfattn;
    if (D != 64 || dtype != GF_BF16) return GF_ERR_UNSUPPORTED;
    if (B2 <= 0 || H <= 0 || N <= 0 || pair <= 0 || pair >= B2) return GF_ERR_SHAPE;
    if (H > 4 || N % 64) return GF_ERR_UNSUPPORTED;                  // (the caller runs gf_attn_bwd_acc twice instead)
    const int64_t* all[6] = {qk_strides, v_strides, o_strides, do_strides, dqk_strides, dv_strides};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j)
            if (all[i][j] % 8) return GF_ERR_ALIGN;
    XbParams p = {};
    p.qk = static_cast<const bf16_t*>(qk); p.v = static_cast<const bf16_t*>(v); p.o = static_cast<const bf16_t*>(o);
    p.dout = static_cast<const bf16_t*>(dout); p.dqk = static_cast<bf16_t*>(dqk); p.dv = static_cast<bf16_t*>(dv);
    p.lse = lse; p.stat = stat;
    p.B2 = B2; p.pair = pair; p.H = H; p.N = N;
    p.sqb = qk_strides[0]; p.sqn = qk_strides[1]; p.sqh = qk_strides[2];
    p.svb = v_strides[0]; p.svn = v_strides[1]; p.svh = v_strides[2];
    p.sob = o_strides[0]; p.son = o_strides[1]; p.soh = o_strides[2];
    p.sdob = do_strides[0]; p.sdon = do_strides[1]; p.sdoh = do_strides[2];
    p.sdqb = dqk_strides[0]; p.sdqn = dqk_strides[1]; p.sdqh = dqk_strides[2];
    p.sdvb = dv_strides[0]; p.sdvn = dv_strides[1]; p.sdvh = dv_strides[2];
    p.scale = scale;
    host_split_scale(scale, p.p2, p.rr);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if !(XB_ABL & 16)
    attn_stats_kernel<<<dim3((unsigned)(((int64_t)B2 * N + 3) / 4)), dim3(256), 0, st>>>(p);
#endif
    const size_t lds = XB_NSTAGE * XB_STAGE;
    void (*const kern[2])(XbParams) = {attn_xbwd_bf16_kernel<false>, attn_xbwd_bf16_kernel<true>};
    static unsigned long long attr_set = 0;                         // function attributes are per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
        for (auto k : kern) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        if (dev < 64) attr_set |= 1ull << dev;
    }
    const int total = ((N + 127) / 128) * H * B2;
    kern[p.rr == 1.f ? 1 : 0]<<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}
