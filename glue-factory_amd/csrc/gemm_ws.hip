// y = [x0 | x1] W^T (+ bias) (+ residual) (+ rotary) for the tall-and-skinny linear layers of the matcher blocks:
// every nn.Linear / Conv1d(k=1) forward of gluefactory/models/matchers/lightglue.py:131-221,271-290,
// gluefactory_nonfree/superglue.py:70-160, gluefactory/models/matchers/gluestick.py:465-586, and -- with the
// transposed weight -- every input-gradient GEMM of their backward.  M = B*N tokens is ~1e5, the weight is at most
// 768 x 512: the layer is HBM-bound on x and y (17-34 us at 8 TB/s for 131072 rows), the MFMA work (7-14 us at the
// bf16 peak) has to hide under it.
//
// Structure ("x stationary, W streamed"):
//   * a wave owns 32*R token rows; their K-long activations sit in REGISTERS as MFMA B-operand fragments.  They are
//     fetched with fully coalesced 16-byte loads (16 lanes per 256-byte row segment) and turned into fragments
//     through a wave-private LDS scratch (XOR-swizzled, conflict-free both ways) -- row-per-lane "fragment shaped"
//     global accesses touch 32 cache lines per instruction and measured 1.5x slower end to end.  The two-source form
//     [x0 | x1] is the FFN input cat[x, message] (lightglue.py:163) without the concatenated tensor;
//   * the weight streams through LDS in slices of whole output rows (double buffered, register staged: the next
//     slice's loads are issued before the current slice's MFMAs and written to LDS after them -- one barrier per
//     slice); rows are XOR-swizzled in 16-byte chunks so the A-operand ds_read_b128 of 16 different rows is
//     conflict-free;
//   * scores are produced transposed (output channel on the MFMA i axis = registers, token row on j = lane); every
//     two 32-channel tiles the fp32 accumulators go through the same wave-private scratch and leave as whole
//     128-byte (bf16) lines, 8 rows per store instruction: bias / residual / rotary are applied there on 8
//     consecutive channels per lane, one rounding;
//   * two workgroups per CU (one of 8 waves at K = 512) run out of phase, so one group's loads / stores overlap the
//     other's MFMAs (each group has its own barriers).
// The rotary epilogue (lightglue.py:42-49,159-160) rotates channel pairs of the q and k thirds of the fused Wqkv
// output by the cached (cos, sin) table while the tile is still in fp32.
// T = float runs the same code on v_mfma_f32_32x32x2_f32 (exact fp32, parity mode).
#include "gf_common.h"
#include "gf_amd.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GwParams {
    const void* x0; const void* x1; const void* w; const float* bias; const void* res; void* y;
    const float* cs;          // [M, 64] interleaved (cos, sin) per token, or null
    int rot_n;                // output channels [0, rot_n) are rotated (head dim 64)
    int M, N, K0, K1;
    int64_t ld0, ld1, ldw, ldr, ldy;
};

#ifndef GW_R256
#define GW_R256 1                             // 32-row blocks per wave at K = 256 (bf16)
#endif
#ifndef GW_NW256
#define GW_NW256 8                            // waves per workgroup at K = 256 (bf16); 8 measured 4-6 % faster than 4 at N = 256, 512
#endif
#ifndef GW_GRID
#define GW_GRID 0                             // > 0: persistent grid of that many workgroups
#endif
#ifndef GW_ABL
#define GW_ABL 0                              // probe ablations: 1 no global stores, 2 x rows all = row 0 (cached), 3 no MFMA
#endif
#ifndef GW_TRACE
#define GW_TRACE 0
#endif
#if GW_TRACE
__device__ unsigned long long gw_trace_buf[16 * 4096];
#define GW_STAMP(i) do { if (lane == 0 && (wave == 0)) gw_trace_buf[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define GW_STAMP(i)
#endif
constexpr int GW_SCRATCH = 8192;              // wave-private LDS scratch: 32 rows x 256 B

// 16-byte chunk c of row `row` (cpr chunks per row, a power of two) sits at chunk position swz(row, c) of its LDS row:
// conflict-free for ds_read_b128 / ds_write_b128 of one logical chunk from 16 (or 8) different rows
__device__ __forceinline__ int gw_swz(int row, int c, int cpr) {
    return cpr >= 16 ? (c ^ (row & 15)) : (c ^ ((row >> (cpr == 8 ? 1 : (cpr == 4 ? 2 : 3))) & (cpr - 1)));
}

// KF = K / 16 (K is a compile-time power of two, 32 .. 512), R = 32-row blocks per wave, TWO: K split in two
// equal halves read from x0 and x1, NW waves per workgroup
template <typename T, int KF, int R, bool TWO, int NW>
__global__ __launch_bounds__(64 * NW, 2) void gemm_ws_kernel(GwParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = 16 / (int)sizeof(T);               // elements per 16-byte chunk
    constexpr int K = 16 * KF;
    constexpr int ROWB = K * (int)sizeof(T);               // bytes of one weight / activation row
    constexpr int CPR = ROWB / 16;                         // 16-byte chunks per row
    constexpr int SLICE = ROWB * 32 > 16384 ? ROWB * 32 : 16384;   // bytes of one weight slice (two resident)
    constexpr int SLN = SLICE / ROWB;                      // output channels per full slice (multiple of 32)
    static_assert(SLN % 32 == 0 && SLN >= 32, "slice must hold whole 32-channel tiles");
    const int nslice = (p.N + SLN - 1) / SLN;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nblk = (p.M + 32 * NW * R - 1) / (32 * NW * R);   // row blocks of NW waves x 32 R rows
    const int myblocks = (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // b, b + grid, ...
    const int total = myblocks * nslice;                    // slices this workgroup walks (the ring never restarts)
    if (myblocks <= 0) return;
#ifdef GW_DELAY      // probe: odd workgroups start GW_DELAY x ~4 us late (are the phases of a round in lockstep?)
    if (blockIdx.x & 1) for (int i = 0; i < GW_DELAY; ++i) __builtin_amdgcn_s_sleep(127);
#endif

    const T* wg = reinterpret_cast<const T*>(p.w);
    char* scr = smem + 2 * SLICE + wave * GW_SCRATCH;       // this wave's scratch
    // bias in LDS: a global load in the epilogue would be consumed after earlier output stores, and vmcnt retires in
    // order and counts stores on this part -- every flush would wait for the previous flush's store acknowledgements
    float* bias_s = reinterpret_cast<float*>(smem + 2 * SLICE + NW * GW_SCRATCH);   // [N] (zeros without a bias)
    for (int n = threadIdx.x; n < p.N; n += 64 * NW) bias_s[n] = p.bias ? p.bias[n] : 0.f;

    // ---- weight staging through registers (issue early / write late): the next slice's coalesced 16-byte loads are
    // issued before the current slice's MFMAs and written to LDS (swizzled) after them, one barrier per slice.  (An
    // LDS-DMA version measured 1.3x slower: vmcnt is in-order and counts stores on this part, so the vmcnt(0) that
    // retires a DMA in front of the barrier also waits for every output store of the slice.)
    constexpr int NST = SLICE / 16 / (64 * NW);             // 16-byte chunks per thread and slice
    static_assert(NST >= 1, "slice smaller than one chunk per thread");
    u32x4 wst[NST];
    auto stage_load = [&](int gs) {                         // gs = running slice index of this workgroup
        const int n0 = (gs % nslice) * SLN;
        const int rows = min(SLN, p.N - n0);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int pos = (int)threadIdx.x + 64 * NW * i; // chunk index inside the slice (row-major)
            const int row = min(pos / CPR, rows - 1);       // (rows past the slice's end: harmless duplicates)
            wst[i] = *reinterpret_cast<const u32x4*>(wg + (int64_t)(n0 + row) * p.ldw + (pos % CPR) * EPC);
        }
    };
    auto stage_store = [&](int gs) {
        char* buf = smem + (gs & 1) * SLICE;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int pos = (int)threadIdx.x + 64 * NW * i;
            const int row = pos / CPR;
            *reinterpret_cast<u32x4*>(buf + row * ROWB + gw_swz(row, pos % CPR, CPR) * 16) = wst[i];
        }
    };

    // ---- a wave's activations: 32 rows x (XC 16-byte chunks) per phase, coalesced (XC lanes per row), through the
    // scratch, out as B-operand fragments (lane = row l31, elements [16 s + 8 hi, +8) of k-step s)
    constexpr int XC = CPR < 16 ? CPR : 16;                 // chunks per row and phase (<= 256 B of a row)
    constexpr int NPH = CPR / XC;                           // phases per row block
    constexpr int RPI = 64 / XC;                            // rows per load instruction
    constexpr int KSP = XC * EPC / 16;                      // k-steps per phase
    auto load_x = [&](Frag<T> (&xf)[R][KF], int blk) {
        const int m_wave = (blk * NW + wave) * 32 * R;
        const int xr = lane / XC, xc = lane % XC;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                u32x4 tmp[32 / RPI];
#pragma unroll
                for (int j = 0; j < 32 / RPI; ++j) {
                    const int row = j * RPI + xr;
                    const int64_t m = GW_ABL == 2 ? (int64_t)(row & 1) : min(m_wave + 32 * rr + row, p.M - 1);
                    const int e0 = (ph * XC + xc) * EPC;     // first element of this lane's chunk
                    const T* src = (TWO && e0 >= K / 2) ? reinterpret_cast<const T*>(p.x1) + m * p.ld1 + (e0 - K / 2)
                                                        : reinterpret_cast<const T*>(p.x0) + m * p.ld0 + e0;
                    tmp[j] = *reinterpret_cast<const u32x4*>(src);
                }
#pragma unroll
                for (int j = 0; j < 32 / RPI; ++j) {
                    const int row = j * RPI + xr;
                    *reinterpret_cast<u32x4*>(scr + row * (XC * 16) + gw_swz(row, xc, XC) * 16) = tmp[j];
                }
#pragma unroll
                for (int s = 0; s < KSP; ++s) {
                    const char* base = scr + l31 * (XC * 16);
                    if constexpr (sizeof(T) == 2) {
                        xf[rr][ph * KSP + s] = ld_frag8(reinterpret_cast<const T*>(base + gw_swz(l31, 2 * s + hi, XC) * 16));
                    } else {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(base + gw_swz(l31, 4 * s + 2 * hi, XC) * 16);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(base + gw_swz(l31, 4 * s + 2 * hi + 1, XC) * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { xf[rr][ph * KSP + s].v[e] = a[e]; xf[rr][ph * KSP + s].v[4 + e] = b[e]; }
                    }
                }
            }
        }
    };

    // ---- epilogue of up to two 32-channel tiles (channels [nt0, nt0 + 32 ntl)): accumulators -> scratch (fp32,
    // [32 rows][64 ch], 16-byte chunks swizzled by row) -> 8 consecutive channels per lane, 8 rows per instruction
    auto flush = [&](const f32x16 (&acc)[2], int ntl, int nt0, int m0) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
            if (tt < ntl) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[tt][4 * g], acc[tt][4 * g + 1], acc[tt][4 * g + 2], acc[tt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(scr + l31 * 256 + ((8 * tt + 2 * g + hi) ^ (l31 & 15)) * 16) = v;
                }
            }
        const int q = lane & 7;                              // channels [8 q, 8 q + 8) of the 64
        if (8 * q < 32 * ntl) {
            const int n = nt0 + 8 * q;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + n);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + n + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * j + (lane >> 3);
                const int64_t m = m0 + row;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + row * 256 + ((2 * q) ^ (row & 15)) * 16) + b0;
                f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + row * 256 + ((2 * q + 1) ^ (row & 15)) * 16) + b1;
                if (GW_ABL == 1 ? m < p.M - (int64_t)2000000000 : m < p.M) {
                    if (p.res) {
                        const T* rp = reinterpret_cast<const T*>(p.res) + m * p.ldr + n;
                        if constexpr (sizeof(T) == 2) {
                            const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(rp);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] += (float)r8[e]; v1[e] += (float)r8[4 + e]; }
                        } else {
                            v0 += *reinterpret_cast<const f32x4*>(rp);
                            v1 += *reinterpret_cast<const f32x4*>(rp + 4);
                        }
                    }
                    if (p.cs && n < p.rot_n) {
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.cs + m * 64 + (n & 63));
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.cs + m * 64 + (n & 63) + 4);
                        const f32x4 o0 = {v0[0] * c0[0] - v0[1] * c0[1], v0[1] * c0[0] + v0[0] * c0[1],
                                          v0[2] * c0[2] - v0[3] * c0[3], v0[3] * c0[2] + v0[2] * c0[3]};
                        const f32x4 o1 = {v1[0] * c1[0] - v1[1] * c1[1], v1[1] * c1[0] + v1[0] * c1[1],
                                          v1[2] * c1[2] - v1[3] * c1[3], v1[3] * c1[2] + v1[2] * c1[3]};
                        v0 = o0;
                        v1 = o1;
                    }
                    T* yp = reinterpret_cast<T*>(p.y) + m * p.ldy + n;
                    if constexpr (sizeof(T) == 2) {
                        const bf16x8 o = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                                          (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
                        *reinterpret_cast<bf16x8*>(yp) = o;
                    } else {
                        *reinterpret_cast<f32x4*>(yp) = v0;
                        *reinterpret_cast<f32x4*>(yp + 4) = v1;
                    }
                }
            }
        }
    };

    GW_STAMP(0);
    stage_load(0);
    Frag<T> xf[R][KF];
    load_x(xf, blockIdx.x);
    GW_STAMP(1);
    stage_store(0);
    __syncthreads();
    GW_STAMP(2);

    // per-lane swizzle key of the weight rows in byte units, with the lane's half (hi) folded in
    const int swz_key = (gw_swz(l31, 0, CPR) << 4) ^ ((sizeof(T) == 2 ? 16 : 32) * hi);
    int gs = 0;
    for (int bi = 0; bi < myblocks; ++bi) {
        const int blk = blockIdx.x + bi * gridDim.x;
        const int m_wave = (blk * NW + wave) * 32 * R;
        for (int s = 0; s < nslice; ++s, ++gs) {
            const char* cur = smem + (gs & 1) * SLICE;
            if (gs + 1 < total) stage_load(gs + 1);
            const int n0 = s * SLN;
            const int ntiles = min(SLN, p.N - n0) >> 5;
            for (int t = 0; t < ntiles; t += 2) {
                const int ntl = min(2, ntiles - t);
                f32x16 acc[R][2];
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[rr][tt][r] = 0.f;
                {
                    // byte offset of logical chunk ci of a weight row inside its LDS row = (16 ci) ^ (16 key(row));
                    // 32 t rows do not change the key (it depends on row bits < 5 only).  It is made opaque per
                    // tile pair: otherwise the compiler hoists all KF swizzled addresses out of the loops (KF or 2 KF
                    // address registers) instead of spending one v_xor per read.
                    int kb = swz_key;
                    asm volatile("" : "+v"(kb));
                    // the two tiles of the pair run as two INDEPENDENT accumulator chains, interleaved k-step by
                    // k-step (a single chain stalls on the MFMA's own result latency); an odd last tile reads tile t
                    // twice (its second accumulator is simply not flushed)
                    const char* wl0 = cur + (32 * t + l31) * ROWB;
                    const char* wl1 = cur + (32 * (t + ntl - 1) + l31) * ROWB;
                    auto rd = [&](const char* wl, int ks) {
                        Frag<T> wf;                           // logical elements [16 ks + 8 hi, +8) of the weight row
                        if constexpr (sizeof(T) == 2) {
                            wf = ld_frag8(reinterpret_cast<const T*>(wl + ((32 * ks) ^ kb)));
                        } else {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(wl + ((64 * ks) ^ kb));
                            const f32x4 b = *reinterpret_cast<const f32x4*>(wl + ((64 * ks + 16) ^ kb));
#pragma unroll
                            for (int e = 0; e < 4; ++e) { wf.v[e] = a[e]; wf.v[4 + e] = b[e]; }
                        }
                        return wf;
                    };
                    // k-steps in groups of G: the A fragments of group g+1 are requested before the MFMAs of group g
                    constexpr int G = KF < 2 ? KF : 2;
                    Frag<T> wfa[2][G], wfb[2][G];
#pragma unroll
                    for (int e = 0; e < G; ++e) { wfa[0][e] = rd(wl0, e); wfa[1][e] = rd(wl1, e); }
#pragma unroll
                    for (int g = 0; g < KF / G; ++g) {
                        Frag<T> (&cur_)[2][G] = (g & 1) ? wfb : wfa;
                        Frag<T> (&nxt_)[2][G] = (g & 1) ? wfa : wfb;
                        if (g + 1 < KF / G) {
#pragma unroll
                            for (int e = 0; e < G; ++e) {
                                nxt_[0][e] = rd(wl0, G * (g + 1) + e);
                                nxt_[1][e] = rd(wl1, G * (g + 1) + e);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < G; ++e)
#pragma unroll
                            for (int rr = 0; rr < R; ++rr) {
                                if (GW_ABL == 3) {
                                    asm volatile("" ::"v"(cur_[0][e].v), "v"(cur_[1][e].v));
                                } else {
                                    mma32(acc[rr][0], cur_[0][e], xf[rr][G * g + e]);
                                    mma32(acc[rr][1], cur_[1][e], xf[rr][G * g + e]);
                                }
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (s == 0 && t == 0) GW_STAMP(3);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) flush(acc[rr], ntl, n0 + 32 * t, m_wave + 32 * rr);
                if (s == 0 && t == 0) GW_STAMP(4);
            }
            if (s < 4) GW_STAMP(5 + 2 * s);
            if (gs + 1 < total) {                         // slice gs+1 -> the other buffer (free since the last barrier)
                stage_store(gs + 1);
                __syncthreads();
            }
            if (s < 4) GW_STAMP(6 + 2 * s);
        }
        if (bi + 1 < myblocks) load_x(xf, blk + gridDim.x);
    }
}

template <typename T, int KF, int R, bool TWO, int NW> int gw_launch(const GwParams& p, hipStream_t st) {
    constexpr int ROWB = 16 * KF * (int)sizeof(T);
    const size_t lds = 2 * (size_t)(ROWB * 32 > 16384 ? ROWB * 32 : 16384) + NW * GW_SCRATCH + (size_t)p.N * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<T, KF, R, TWO, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int rows_per_wg = NW * 32 * R;
    int grid = (p.M + rows_per_wg - 1) / rows_per_wg;
    if (GW_GRID > 0 && grid > GW_GRID) grid = GW_GRID;      // persistent: each workgroup walks row blocks b, b + grid, ...
    gemm_ws_kernel<T, KF, R, TWO, NW><<<dim3(grid), 64 * NW, lds, st>>>(p);
    return (int)hipGetLastError();
}

template <typename T, bool TWO> int gw_dispatch(const GwParams& p, int K, hipStream_t st) {
    // registers: x fragments = R * K/16 * (4 | 8) VGPRs, kept <= 128
    if constexpr (sizeof(T) == 2) {
        switch (K) {
            case 32: return gw_launch<T, 2, 2, TWO, 4>(p, st);
            case 64: return gw_launch<T, 4, 2, TWO, 4>(p, st);
            case 128: return gw_launch<T, 8, 2, TWO, 4>(p, st);
            case 256: return gw_launch<T, 16, GW_R256, TWO, GW_NW256>(p, st);
            case 512: return gw_launch<T, 32, 1, TWO, 8>(p, st);     // 32 KB slices: one 8-wave group per CU
        }
    } else {
        switch (K) {
            case 32: return gw_launch<T, 2, 2, TWO, 4>(p, st);
            case 64: return gw_launch<T, 4, 2, TWO, 4>(p, st);
            case 128: return gw_launch<T, 8, 1, TWO, 4>(p, st);
            case 256: return gw_launch<T, 16, 1, TWO, 8>(p, st);
        }
    }
    return GF_ERR_UNSUPPORTED;
}

}  // namespace

#if GW_TRACE
extern "C" int gf_gemm_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gw_trace_buf), sizeof(unsigned long long) * n);
}
#endif

// gemm_st.hip: the streamed-activation kernel for the large regular shapes (returns GF_ERR_UNSUPPORTED for the rest)
int gf_gemm_stream_try(const void* x0, const void* x1, const void* w, const float* bias, const void* res, void* y,
                       const float* cs, int rot_n, int M, int N, int K0, int K1, int64_t ld0, int64_t ld1, int64_t ldw,
                       int64_t ldr, int64_t ldy, hipStream_t st, const void* res2 = nullptr, int64_t ldr2 = 0);

// y = [x0 | x1] W^T + bias + res + res_b: gf_gemm with TWO residual inputs, streamed-activation kernel only (bf16,
// K0 + K1 in {256, 512}, N % 256 == 0, M % 64 == 0, no rotary epilogue; GF_ERR_UNSUPPORTED otherwise: add one of the
// residuals first and call gf_gemm)
extern "C" int gf_gemm_res2(const void* x0, const void* x1, const void* w, const float* bias, const void* res, const void* res_b,
                            void* y, int M, int N, int K0, int K1, int64_t ld0, int64_t ld1, int64_t ldw, int64_t ldr,
                            int64_t ldr_b, int64_t ldy, int dtype, void* stream) {
    if (M <= 0 || N <= 0 || K0 <= 0 || K1 < 0) return GF_ERR_SHAPE;
    if (dtype != GF_BF16 || !res || !res_b || (K1 && !x1)) return GF_ERR_UNSUPPORTED;
    if (ld0 % 8 || (K1 && ld1 % 8) || ldw % 8 || ldy % 8 || ldr % 8 || ldr_b % 8) return GF_ERR_ALIGN;
    return gf_gemm_stream_try(x0, x1, w, bias, res, y, nullptr, 0, M, N, K0, K1, ld0, ld1, ldw, ldr, ldy,
                              reinterpret_cast<hipStream_t>(stream), res_b, ldr_b);
}

extern "C" int gf_gemm(const void* x0, const void* x1, const void* w, const float* bias, const void* res, void* y,
                       const float* cs, int rot_n, int M, int N, int K0, int K1,
                       int64_t ld0, int64_t ld1, int64_t ldw, int64_t ldr, int64_t ldy, int dtype, void* stream) {
    if (M <= 0 || N <= 0 || K0 <= 0 || K1 < 0) return GF_ERR_SHAPE;
    if (dtype != GF_BF16 && dtype != GF_F32) return GF_ERR_DTYPE;
    const int K = K0 + K1;
    if (N % 32 || (K1 && K1 != K0) || (K1 && !x1)) return GF_ERR_UNSUPPORTED;   // K in {32..512} checked by the dispatch
    const int al = dtype == GF_BF16 ? 8 : 4;                // 16-byte aligned rows of x / W / y / res
    if (ld0 % al || (K1 && ld1 % al) || ldw % al || ldy % al || (res && ldr % al)) return GF_ERR_ALIGN;
    if (cs && (rot_n % 64 || rot_n > N)) return GF_ERR_SHAPE;
    GwParams p;
    p.x0 = x0; p.x1 = x1; p.w = w; p.bias = bias; p.res = res; p.y = y; p.cs = cs; p.rot_n = cs ? rot_n : 0;
    p.M = M; p.N = N; p.K0 = K0; p.K1 = K1; p.ld0 = ld0; p.ld1 = ld1; p.ldw = ldw; p.ldr = ldr; p.ldy = ldy;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifndef GW_NO_STREAM
    if (dtype == GF_BF16) {
        const int e = gf_gemm_stream_try(x0, x1, w, bias, res, y, cs, rot_n, M, N, K0, K1, ld0, ld1, ldw, ldr, ldy, st);
        if (e != GF_ERR_UNSUPPORTED) return e;
    }
#endif
    if (dtype == GF_BF16) return K1 ? gw_dispatch<bf16_t, true>(p, K, st) : gw_dispatch<bf16_t, false>(p, K, st);
    return K1 ? gw_dispatch<float, true>(p, K, st) : gw_dispatch<float, false>(p, K, st);
}
