// bf16 attention backward, part 1 (dQ and the per-row statistics), third generation.  Own translation unit, built with
// -fno-slp-vectorize like attention_fwd3.hip (SLP pairs independent f32 multiplies into half-rate v_pk_mul_f32).
//
// What bounds these kernels (tools/probe/ubench/mfma_valu.hip + SQ counters): a SIMD issues ONE
// vector-type instruction (VALU or MFMA) per 4-cycle slot, whichever wave it comes from, v_exp_f32 takes two slots, and
// the MFMA then runs 32 cycles in the background.  The second-generation dQ kernel spent 12.5 VALU instructions per MFMA
// (staging K^T through registers into LDS, fma + subtract + multiply per score): 50+ slot-cycles per 32 MFMA cycles.
// Here:
//   * K and V tiles arrive by `buffer_load ... lds` into the forward's 3-stage ring; S^T and dP^T read K / V rows with
//     ds_read_b128, the dQ product reads K^T straight from the same tile with ds_read_b64_tr_b16 (no staging VALU, no
//     transposed copy, no bank conflicts);
//   * the accumulators of S^T start at -lse (in the exponent's units) and those of dP^T at -delta: two 16-register
//     splats per wave that never change, passed as the MFMA's C operand -- so P = exp2(r x) and dS = P y take
//     mul + exp2 + mul per score (Q carries the exact power-of-two part of scale * log2(e), r the rest);
// i.e. 4.5 issue slots per score and 12 MFMAs per 16 scores: 7.5 slots (30 cycles) per MFMA.
// One wave = 32 query rows, 4 waves per workgroup, K/V streamed in 64-key tiles.  The kernel also writes the two
// per-row vectors the dK/dV kernel starts ITS accumulators from: stat[0] = -lse * log2(e) / r, stat[1] = -delta.
#include "gf_common.h"
#include "gf_amd.h"
#include "attn_common.h"

namespace gfattn {
namespace {

// Timing probes only (tools/probe/attn_stall_table.sh; shipped: 0): the dQ loop without ... 1 the exponentials, 2 the dQ
// product (4 of the 12 MFMAs of a half tile), 4 the DMA of the next tiles, 8 the hardware-transposed LDS reads (K^T), 16 the
// row-major LDS reads (K, V).  Results are wrong by construction.
#ifndef GF_DQ3_ABL
#define GF_DQ3_ABL 0
#endif
#if GF_DQ3_ABL & 2
#define GF_DQ3_OUT_MMA(acc, a, b) do { const auto a_ = (a); const auto b_ = (b); asm volatile("" ::"v"(a_), "v"(b_)); } while (0)
#else
#define GF_DQ3_OUT_MMA(acc, a, b) mma16(acc, a, b)
#endif
#ifndef DQ3_WPS
#define DQ3_WPS 2          // workgroups per CU = waves per SIMD
#endif

template <bool PRE, bool EVEN, bool SPLIT = false>   // PRE: rr == 1 (operands pre-multiplied by the caller): no multiply in front of exp2;
                                    // EVEN: Nk % 64 == 0 -- no tile-dependent branch in the loop (see attention_fwd3.hip);
                                    // SPLIT: dS = hi + lo bf16 pair in front of the dQ product (attention_fwd3.hip)
__global__ __launch_bounds__(256, DQ3_WPS) void attn_dq3_bf16_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nqb = (p.Nq + 127) / 128;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = qb * 128 + wave * 32 + l31;
    const int qld = min(qrow, p.Nq - 1);
#ifdef DQ3_PRIO
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(DQ3_PRIO);      // probe knob: priority asymmetry between co-resident waves
#endif

    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + h * p.sqh;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + h * p.skh;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.svb + h * p.svh;
    const bf16_t* op = reinterpret_cast<const bf16_t*>(p.o) + b * p.sob + h * p.soh;
    const bf16_t* dop = reinterpret_cast<const bf16_t*>(p.dout) + b * p.sdob + h * p.sdoh;

    KvDma dma;
    dma.init(kp, vp, p.skn, p.svn, p.Nk, wave, lane);
    const int nt = (p.Nk + 63) / 64;
    if (EVEN) {                       // no ragged tile; tiles past the end re-fetch the last one
        dma.issue_full(0, smem + wave * 1024);
        dma.issue_full(min(1, nt - 1), smem + FQ_STAGE + wave * 1024);
    } else {
        dma.issue(0, smem + wave * 1024);
        if (nt > 1) dma.issue(1, smem + FQ_STAGE + wave * 1024);
    }

    const float p2 = p.p2, rr = PRE ? 1.f : p.rr;
    bf16x8 qf[4], dof[4];
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 q8 = *reinterpret_cast<const bf16x8*>(qp + (int64_t)qld * p.sqn + 16 * s + 8 * hi);
        dof[s] = *reinterpret_cast<const bf16x8*>(dop + (int64_t)qld * p.sdon + 16 * s + 8 * hi);
        if (SPLIT) {          // `o` is the forward's fp32 copy (strides in fp32 elements): delta without the rounding of o
            const float* o32 = reinterpret_cast<const float*>(p.o) + b * p.sob + h * p.soh + (int64_t)qld * p.son + 16 * s + 8 * hi;
            const f32x4 oa = *reinterpret_cast<const f32x4*>(o32), ob = *reinterpret_cast<const f32x4*>(o32 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) delta += oa[e] * (float)dof[s][e] + ob[e] * (float)dof[s][4 + e];
        } else {
            const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(op + (int64_t)qld * p.son + 16 * s + 8 * hi);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += (float)o8[e] * (float)dof[s][e];
        }
        qf[s] = p2 != 1.f ? scale_frag(q8, p2) : q8;
    }
    delta += xhalf(delta);
    const int64_t stat = ((int64_t)b * p.H + h) * p.Nq + qld;
    const float nl = -p.lse[stat] * GF_LOG2E / rr;                  // S^T accumulators start here: P = exp2(rr * x)
    if (qrow < p.Nq && hi == 0) {
        p.delta[stat] = nl;
        p.delta[(int64_t)p.B * p.H * p.Nq + stat] = -delta;
    }
    f32x16 dq[2], nlb, ndb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; nlb[r] = nl; ndb[r] = -delta; }
    tie(nlb);                                                       // opaque: otherwise the splats are rebuilt every tile
    tie(ndb);

    unsigned aR[4], aT[4];
    {
        const FqAddr ad = fq_addresses(lds0, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] = ad.aR[i]; aT[i] = ad.aT[i]; }
    }
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        if (EVEN || t + 1 < nt) wait_vm<4>(); else wait_vm<0>();     // tile t landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                                // ... everyone's; the stage of tile t-1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (GF_DQ3_ABL & 4) {
        } else if (EVEN) dma.issue_full(min(t + 2, nt - 1), smem + (stage == 0 ? 2 : stage - 1) * FQ_STAGE + wave * 1024);
        else if (t + 2 < nt) dma.issue(t + 2, smem + (stage == 0 ? 2 : stage - 1) * FQ_STAGE + wave * 1024);
        const bool ragged = !EVEN && t * 64 + 64 > p.Nk;

        u32x4 ka[4], va[4];
        u32x2 kt[2][2][2];
        f32x16 s, dp;
#define GF_DQ3_HALF(KB)                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) ka[i] = (GF_DQ3_ABL & 16) ? u32x4{0u, 0u, 0u, 0u} : lds_rd128<KB * 4096>(aR[i]); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) va[i] = (GF_DQ3_ABL & 16) ? u32x4{0u, 0u, 0u, 0u} : lds_rd128<FT_TILE + KB * 4096>(aR[i]); \
        wait_lgkm<4>();                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) tie(ka[i]);                                                   \
        s = mma16c(as_frag(ka[0]), qf[0], nlb);                              /* S^T[key][q] - lse (exponent units) */ \
        _Pragma("unroll") for (int i = 1; i < 4; ++i) mma16(s, as_frag(ka[i]), qf[i]);                              \
        if (GF_DQ3_ABL & 8) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) kt[i_ >> 2][(i_ >> 1) & 1][i_ & 1] = u32x2{0u, 0u}; }   \
        else { GF_FQ_TR(kt, 0, KB, 0, 0) GF_FQ_TR(kt, 0, KB, 0, 1) GF_FQ_TR(kt, 0, KB, 1, 0) GF_FQ_TR(kt, 0, KB, 1, 1) }     \
        wait_lgkm<8>();                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) tie(va[i]);                                                   \
        dp = mma16c(as_frag(va[0]), dof[0], ndb);                            /* dP^T[key][q] - delta */             \
        _Pragma("unroll") for (int i = 1; i < 4; ++i) mma16(dp, as_frag(va[i]), dof[i]);                            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) s[r] = ((GF_DQ3_ABL & 1) ? s[r] : fast_exp2(PRE ? s[r] : s[r] * rr)) * dp[r];      /* dS */           \
        if (ragged) {                                                        /* keys past Nk contribute nothing */   \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                          \
                if (t * 64 + KB * 32 + crow(r, hi) >= p.Nk) s[r] = 0.f;                                             \
        }                                                                                                           \
        {                                                                                                           \
            const bf16x8 d0 = cvt_frag(s, 0), d1 = cvt_frag(s, 1);                                                  \
            wait_lgkm<0>();                                                                                         \
            _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                      \
                tie(kt[0][db][0]); tie(kt[0][db][1]); tie(kt[1][db][0]); tie(kt[1][db][1]);                         \
                GF_DQ3_OUT_MMA(dq[db], as_frag(kt[0][db][0], kt[0][db][1]), d0);      /* dQ^T[d][q] += K^T[d][key] dS */     \
                GF_DQ3_OUT_MMA(dq[db], as_frag(kt[1][db][0], kt[1][db][1]), d1);                                             \
            }                                                                                                       \
            if (SPLIT) {                                                                                            \
                const bf16x8 e0 = cvt_frag_lo(s, 0, d0), e1 = cvt_frag_lo(s, 1, d1);                                \
                _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                  \
                    GF_DQ3_OUT_MMA(dq[db], as_frag(kt[0][db][0], kt[0][db][1]), e0);                                         \
                    GF_DQ3_OUT_MMA(dq[db], as_frag(kt[1][db][0], kt[1][db][1]), e1);                                         \
                }                                                                                                   \
            }                                                                                                       \
        }
        GF_DQ3_HALF(0)
        GF_DQ3_HALF(1)
#undef GF_DQ3_HALF
        const int step = stage == 2 ? -2 * FQ_STAGE : FQ_STAGE;      // ring: per-lane read addresses follow the stage
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] += step; aT[i] += step; }
        stage = stage == 2 ? 0 : stage + 1;
    }
    if (EVEN) wait_vm<0>();                                          // the re-fetched tail tiles
    if (qrow < p.Nq) {
        bf16_t* dqp = reinterpret_cast<bf16_t*>(p.dq) + b * p.sdqb + h * p.sdqh + (int64_t)qrow * p.sdqn;
        if (p.flags & GF_ATTN_ACC_DQ) add_row<64>(dqp, dq, p.scale, hi); else store_row<bf16_t, 64>(dqp, dq, p.scale, hi);
    }
}

}  // namespace

int launch_dq3_bf16(const AttnParams& p, hipStream_t st) {
    const int total = ((p.Nq + 127) / 128) * p.H * p.B;
    const size_t lds = FQ_NSTAGE * FQ_STAGE;
    void (*const kern[8])(AttnParams) = {attn_dq3_bf16_kernel<false, false, false>, attn_dq3_bf16_kernel<false, true, false>,
                                         attn_dq3_bf16_kernel<true, false, false>, attn_dq3_bf16_kernel<true, true, false>,
                                         attn_dq3_bf16_kernel<false, false, true>, attn_dq3_bf16_kernel<false, true, true>,
                                         attn_dq3_bf16_kernel<true, false, true>, attn_dq3_bf16_kernel<true, true, true>};
    static unsigned long long attr_set = 0;          // function attributes are per device: one bit per device ordinal
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
        for (auto k : kern) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        if (dev < 64) attr_set |= 1ull << dev;
    }
    kern[((p.flags & GF_ATTN_SPLIT) ? 4 : 0) + (p.rr == 1.f ? 2 : 0) + (p.Nk % 64 == 0 ? 1 : 0)]<<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}

}  // namespace gfattn
