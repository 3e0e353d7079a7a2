// Third-generation bf16 attention forward (see the note below).  Own translation unit: built with -fno-slp-vectorize,
// because SLP pairs the row-sum adds of the two query blocks into v_pk_add_f32, which forces every exponential into
// a fresh register pair (+32 live VGPRs -> spills) and buys nothing (v_pk_add_f32 runs at half rate).
#include "gf_common.h"
#include "gf_amd.h"
#include "attn_common.h"

namespace gfattn {
namespace {

// ===========================================================================================
// bf16 forward, third generation: the softmax costs 2.5 VALU instructions per score instead of 5
// ===========================================================================================
// The second-generation kernel (attention.hip) is VALU-issue-bound (10 VALU per MFMA;
// `profiles/r02b_attention_sq_counters.txt`).  Per score it spent: max, fma (scale and subtract the running max), exp2,
// row-sum add, half a cvt_pk.  Here
//   * Q is pre-multiplied by the POWER-OF-TWO part of scale*log2(e) once per kernel (exact in bf16; pre-multiplying by
//     the whole factor would round Q a second time and costs up to 0.1 absolute in O at |logit| ~ 100), and
//   * the score accumulators START at -m (a 16-register splat passed as the MFMA's C operand with a different
//     destination, rewritten only when the reference m moves), so a score leaves the matrix pipe as x with
//     P = exp2(r x), r in [1, 2) the rest of the factor: one multiply instead of max + fma;
//   * no running max is tracked at all: m is a REFERENCE, not a maximum.  bf16 keeps fp32's exponent, so P may be
//     large; what has to be excluded is overflow, and the row sum that is computed anyway detects it after the fact
//     (ps <= 2^16 per lane and tile).  If the test fails -- always on the first tile (m = -1e30), on a ragged last
//     tile (keys past Nk need masking), and whenever a logit jumps more than ~11 nats above the reference -- the tile
//     is redone conventionally (S from zero, true max, O and l rescaled, m := running max).  Nothing of the fast
//     attempt has touched O or l at that point.
// Remaining per score: mul, exp2, add, half a cvt_pk.
// Geometry: one wave = ONE 32-row query block (o 32 + -m 16 + q 16 + scores 32 = 96 live registers), 4 waves = 128
// query rows per workgroup, THREE workgroups per CU (<= 168 VGPRs, 48 KiB LDS each): three unsynchronised waves per
// SIMD are in different phases, so one wave's exponentials run under another's MFMAs.  K/V tiles arrive through
// `buffer_load_dwordx4 ... lds` (one per-lane 32-bit offset per matrix, the tile advance in an SGPR) into the 3-stage
// ring.
constexpr float FW3_PS_LIMIT = 65536.f;
#ifndef FW3_ABL
#define FW3_ABL 0          // timing-only ablations (wrong results): 1 no barrier / DMA wait, 2 no K reads, 4 no V^T reads,
#endif                     // 8 no exp2, 16 no score MFMAs, 32 no PV MFMAs, 64 no DMA
#ifndef FW3_WPS
#define FW3_WPS 3          // workgroups per CU = waves per SIMD (probe knob)
#endif

// S^T[key][q] (+ init) of both 32-key blocks of the tile in stage ST; ZERO selects the conventional start
template <bool ZERO>
__device__ __forceinline__ void fw3_scores(f32x16 (&sc)[2], const unsigned (&aR)[4], const bf16x8 (&qf)[4], const f32x16& negm) {
    constexpr int KB = 0;
    u32x4 ka[2][4];
#if FW3_ABL & 2
#pragma unroll
    for (int s = 0; s < 4; ++s) { ka[0][s] = __builtin_bit_cast(u32x4, qf[s]); ka[1][s] = __builtin_bit_cast(u32x4, qf[3 - s]); }
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[0][s] = lds_rd128<KB>(aR[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[1][s] = lds_rd128<KB + 4096>(aR[s]);
#endif
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    wait_lgkm<4>();
#pragma unroll
    for (int s = 0; s < 4; ++s) tie(ka[0][s]);
#if FW3_ABL & 16
    wait_lgkm<0>();
#pragma unroll
    for (int s = 0; s < 4; ++s) tie(ka[1][s]);
    sc[0] = ZERO ? z : negm; sc[1] = ZERO ? z : negm;
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[0][r] += __builtin_bit_cast(float, ka[0][r][0]) * 1e-30f; sc[1][r] += __builtin_bit_cast(float, ka[1][r][0]) * 1e-30f; }
#else
    sc[0] = mma16c(as_frag(ka[0][0]), qf[0], ZERO ? z : negm);
    wait_lgkm<0>();
#pragma unroll
    for (int s = 0; s < 4; ++s) tie(ka[1][s]);
    sc[1] = mma16c(as_frag(ka[1][0]), qf[0], ZERO ? z : negm);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        mma16(sc[0], as_frag(ka[0][s]), qf[s]);
        mma16(sc[1], as_frag(ka[1][s]), qf[s]);
    }
#endif
}

// SPLIT ("fp32-equivalent second product", GF_ATTN_SPLIT): P = P_hi + P_lo with both halves bf16 (16 mantissa bits kept), and
// O accumulates V^T P_hi + V^T P_lo -- what remains of the difference to an fp32 softmax(QK^T) V on the same bf16 operands is
// the summation order.  Costs one more pass of the PV MFMAs (V^T read again from the same LDS tile) and 2.5 VALU per score.
template <bool PRE, bool SPLIT = false>      // PRE: rr == 1 (operands pre-multiplied by the caller): no multiply in front of exp2
__device__ __forceinline__ void fw3_tile(bool force_slow, int kv0, int Nk, const unsigned (&aR)[4], const unsigned (&aT)[4],
                                         const bf16x8 (&qf)[4], f32x16 (&o)[2], f32x16& negm, float& m, float& lsum, int hi, float rr) {
    constexpr int VB = FT_TILE;
    f32x16 sc[2];
    bool slow = force_slow;
    if (!slow) {
        fw3_scores<false>(sc, aR, qf, negm);
        float ps[2] = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if FW3_ABL & 8
                const float e = sc[kb][r] * rr;
#else
                const float e = fast_exp2(PRE ? sc[kb][r] : sc[kb][r] * rr);
#endif
                sc[kb][r] = e;
                ps[kb] += e;
            }
        const float pt = ps[0] + ps[1];
        if (__any(!(pt <= FW3_PS_LIMIT))) slow = true;
        else lsum += pt;
    }
    if (slow) {
        fw3_scores<true>(sc, aR, qf, negm);
        if (kv0 + 64 > Nk) {   // ragged last tile: keys past Nk never win the max and get P = 0
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + crow(r, hi) >= Nk) sc[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
        mx = fmaxf(mx, xhalf(mx));
        const float mnew = fmaxf(m, mx);
        const float alpha = fast_exp2((m - mnew) * rr);
        m = mnew;
        lsum *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mnew;
        tie(negm);                                       // opaque: otherwise the splat is rebuilt (16 moves) every tile
        const float mr = mnew * rr;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp2(fmaf(sc[kb][r], rr, -mr));
                sc[kb][r] = e;
                ps += e;
            }
        lsum += ps;
    }
    // ---- O^T[d][q] += V^T[d][key] P[key][q]
    u32x2 vt0[2][2][2], vt1[2][2][2];
    bf16x8 p00 = cvt_frag(sc[0], 0), p01 = cvt_frag(sc[0], 1), p10 = cvt_frag(sc[1], 0), p11 = cvt_frag(sc[1], 1);
    bf16x8 l00, l01, l10, l11;
    if (SPLIT) {
        l00 = cvt_frag_lo(sc[0], 0, p00); l01 = cvt_frag_lo(sc[0], 1, p01);
        l10 = cvt_frag_lo(sc[1], 0, p10); l11 = cvt_frag_lo(sc[1], 1, p11);
    }
#pragma unroll
    for (int pass = 0; pass < (SPLIT ? 2 : 1); ++pass) {
    if (SPLIT && pass == 1) { p00 = l00; p01 = l01; p10 = l10; p11 = l11; }
#if FW3_ABL & 4
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32x4 x = __builtin_bit_cast(u32x4, qf[i & 3]);
        vt0[i >> 2][(i >> 1) & 1][i & 1] = u32x2{x[0], x[1]}; vt1[i >> 2][(i >> 1) & 1][i & 1] = u32x2{x[2], x[3]};
    }
#else
    GF_FQ_TR(vt0, VB, 0, 0, 0) GF_FQ_TR(vt0, VB, 0, 0, 1) GF_FQ_TR(vt0, VB, 0, 1, 0) GF_FQ_TR(vt0, VB, 0, 1, 1)
    GF_FQ_TR(vt1, VB, 1, 0, 0) GF_FQ_TR(vt1, VB, 1, 0, 1)
    wait_lgkm<4>();
#endif
#if FW3_ABL & 32
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        o[db][0] += (float)p00[0] + (float)p01[0] + (float)p10[0] + (float)p11[0];
        o[db][1] += __builtin_bit_cast(float, vt0[0][db][0][0]) * 1e-30f + __builtin_bit_cast(float, vt1[0][db][0][0]) * 1e-30f;
    }
#else
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            tie(vt0[tt][db][0]); tie(vt0[tt][db][1]);
            mma16(o[db], as_frag(vt0[tt][db][0], vt0[tt][db][1]), tt ? p01 : p00);
        }
#if !(FW3_ABL & 4)
        if (tt == 0) { GF_FQ_TR(vt1, VB, 1, 1, 0) GF_FQ_TR(vt1, VB, 1, 1, 1) }
#endif
    }
    wait_lgkm<0>();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            tie(vt1[tt][db][0]); tie(vt1[tt][db][1]);
            mma16(o[db], as_frag(vt1[tt][db][0], vt1[tt][db][1]), tt ? p11 : p10);
        }
#endif
    }
}

// EVEN: Nk % 64 == 0 -- no ragged tile, so the loop carries no tile-dependent branches: tile 0 (always conventional) is
// peeled, tiles past the end re-fetch the last one (unconditional DMA, constant wait count).  Every instruction of the
// loop costs an issue slot: the scalar bookkeeping of the general loop is ~25 of them per tile.
template <bool PRE, bool EVEN, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 2 : FW3_WPS) void attn_fwd3_bf16_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nqb = (p.Nq + 127) / 128;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = qb * 128 + wave * 32 + l31;
#ifdef FW3_PRIO
    {                                                                     // probe knob: priority asymmetry between co-resident waves
        const int pr = (blockIdx.x >> 8) % 3;
        if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    }
#endif

    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + h * p.sqh;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + h * p.skh;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.svb + h * p.svh;
    KvDma dma;
    dma.init(kp, vp, p.skn, p.svn, p.Nk, wave, lane);
    const int nt = (p.Nk + 63) / 64;
    if (EVEN) {
        dma.issue_full(0, smem + wave * 1024);
        dma.issue_full(min(1, nt - 1), smem + FQ_STAGE + wave * 1024);
    } else {
        dma.issue(0, smem + wave * 1024);
        if (nt > 1) dma.issue(1, smem + FQ_STAGE + wave * 1024);
    }

    const float p2 = p.p2, rr = PRE ? 1.f : p.rr;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)min(qrow, p.Nq - 1) * p.sqn + 16 * s + 8 * hi);
        if (p2 != 1.f) qf[s] = scale_frag(qf[s], p2);
    }

    f32x16 o[2], negm;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -GF_NEG_BIG;
    tie(negm);
    float m = GF_NEG_BIG, lsum = 0.f;
    const FqAddr ad = fq_addresses(lds0, lane);

    unsigned aR[4], aT[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { aR[i] = ad.aR[i]; aT[i] = ad.aT[i]; }
    int stage = 0;
    if (EVEN) {
        char* const wbase = smem + wave * 1024;
        wait_vm<4>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dma.issue_full(min(2, nt - 1), wbase + 2 * FQ_STAGE);
        fw3_tile<PRE, SPLIT>(true, 0, p.Nk, aR, aT, qf, o, negm, m, lsum, hi, rr);
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] += FQ_STAGE; aT[i] += FQ_STAGE; }
        stage = 1;
        for (int t = 1; t < nt; ++t) {
            wait_vm<4>();                                                // tile t landed (this wave's pieces)
            __builtin_amdgcn_s_barrier();                                // ... everyone's; the stage of tile t-1 is free
            __builtin_amdgcn_sched_barrier(0);
            dma.issue_full(min(t + 2, nt - 1), wbase + (stage == 0 ? 2 : stage - 1) * FQ_STAGE);
            fw3_tile<PRE, SPLIT>(false, t * 64, p.Nk, aR, aT, qf, o, negm, m, lsum, hi, rr);
            const int step = stage == 2 ? -2 * FQ_STAGE : FQ_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) { aR[i] += step; aT[i] += step; }
            stage = stage == 2 ? 0 : stage + 1;
        }
        wait_vm<0>();                                                    // the re-fetched tail tiles
    } else {
    for (int t = 0; t < nt; ++t) {
#if !(FW3_ABL & 1)
        if (t + 1 >= nt) wait_vm<0>(); else wait_vm<4>();            // tile t landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                                // ... everyone's; the stage of tile t-1 is free
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(FW3_ABL & 64)
        if (t + 2 < nt) dma.issue(t + 2, smem + (stage == 0 ? 2 : stage - 1) * FQ_STAGE + wave * 1024);
#endif
        fw3_tile<PRE, SPLIT>(t == 0 || t * 64 + 64 > p.Nk, t * 64, p.Nk, aR, aT, qf, o, negm, m, lsum, hi, rr);
        const int step = stage == 2 ? -2 * FQ_STAGE : FQ_STAGE;      // ring: per-lane read addresses follow the stage
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] += step; aT[i] += step; }
        stage = stage == 2 ? 0 : stage + 1;
    }
    }
    const float l = lsum + xhalf(lsum);
    if (qrow < p.Nq) {
        bf16_t* op = reinterpret_cast<bf16_t*>(p.o) + b * p.sob + h * p.soh + (int64_t)qrow * p.son;
        store_row<bf16_t, 64>(op, o, 1.f / l, hi);
        if (SPLIT && p.o32 != nullptr)
            store_row<float, 64>(p.o32 + (((int64_t)b * p.Nq + qrow) * p.H + h) * 64, o, 1.f / l, hi);
        if (hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + qrow] = (m * rr + fast_log2(l)) * GF_LN2;
    }
}


}  // namespace

int launch_fwd3_bf16(const AttnParams& p, hipStream_t st) {
    const int total = ((p.Nq + 127) / 128) * p.H * p.B;
    const size_t lds = FQ_NSTAGE * FQ_STAGE;
    void (*const kern[8])(AttnParams) = {
        attn_fwd3_bf16_kernel<false, false, false>, attn_fwd3_bf16_kernel<false, true, false>,
        attn_fwd3_bf16_kernel<true, false, false>, attn_fwd3_bf16_kernel<true, true, false>,
        attn_fwd3_bf16_kernel<false, false, true>, attn_fwd3_bf16_kernel<false, true, true>,
        attn_fwd3_bf16_kernel<true, false, true>, attn_fwd3_bf16_kernel<true, true, true>};
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
#ifdef FW3_NO_EVEN
    const bool pre = p.rr == 1.f, even = false;
#else
    const bool pre = p.rr == 1.f, even = p.Nk % 64 == 0;
#endif
    const bool split = (p.flags & GF_ATTN_SPLIT) != 0;
    kern[(split ? 4 : 0) + (pre ? 2 : 0) + (even ? 1 : 0)]<<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}

}  // namespace gfattn
