// Flash-style multi-head attention over N keypoints, forward and backward, for gfx950.
//
// Replaces (reference) gluefactory/models/matchers/lightglue.py:97-128 (Attention / SDPA),
// :161 (self attention context) and :203-216 (cross attention, both directions as two calls
// with (q,k,v) = (qk0,qk1,v1) and (qk1,qk0,v0)), and gluefactory_nonfree/superglue.py:112-116.
//
// Layout: q,k,v,o are [B, N, H, hd] views with arbitrary element strides for (b, n, h) and
// hd contiguous (so the fused Wqkv output is consumed in place).  lse/delta are [B,H,N] fp32.
// One workgroup = 4 waves = 128 query rows (forward, dQ) or 128 keys (dK/dV); every wave
// owns 32 rows; K/V (resp. Q/dO) stream through LDS in 64-row tiles.  The score tile is
// produced transposed (keys on the MFMA i axis, the owning row on j = lane&31), so the
// softmax statistics, the rescale of O and the lse/delta factors are all lane-local; P is
// fed back to the second MFMA straight from the accumulator registers with a matching
// key-order on the V^T fragments (no LDS round trip, no permutes).
#include "gf_common.h"
#include "gf_amd.h"
#include "attn_common.h"

using namespace gfattn;

namespace {


// Position of tile row r inside a transposed LDS row: bits 2 and 3 of r are swapped so that the 8
// rows a lane needs for one k-step of the second MFMA — {16t + 4hi + e, 16t + 8 + 4hi + e}, e<4, the
// C-layout rows of accumulator registers 8t..8t+7 — are 8 CONSECUTIVE elements (one 16-byte read).
__device__ __forceinline__ int tpos(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
// Transposed tiles are additionally XOR-swizzled in 8-element (16-byte) blocks by the low bits of
// (row d >> 3): with the coalesced staging order (consecutive lanes = consecutive 16-byte chunks of one
// source row) the 8 lanes of a chunk group would otherwise hit one LDS bank; reads stay 16-byte.
__device__ __forceinline__ int tswz(int d, int pos) { return pos ^ (((d >> 3) & 7) << 3); }

// ---- global -> LDS staging of a 64-row tile (rows clamped to the last valid row) ----------
template <typename T, int HD>
__device__ __forceinline__ void stage_rowmajor(T* lds, const T* g, int64_t ld, int row0, int nmax) {
    using L = Lay<T, HD>;
    for (int c = threadIdx.x; c < 64 * L::CPR; c += 256) {
        int r = c / L::CPR, cc = c % L::CPR;
        int gr = min(row0 + r, nmax - 1);
        u32x4 v = *reinterpret_cast<const u32x4*>(g + (int64_t)gr * ld + cc * L::VEC);
        *reinterpret_cast<u32x4*>(lds + r * L::LDR + cc * L::VEC) = v;
    }
}

template <typename T> struct Pair;
template <> struct Pair<bf16_t> { typedef bf16x2 type; };
template <> struct Pair<float> { typedef f32x2 type; };

// B-operand style fragments of one row held in registers: row[16 s + 8 hi + e], s = 0..HD/16-1
template <typename T, int HD>
__device__ __forceinline__ void load_row_frags(Frag<T> (&f)[HD / 16], const T* rowptr, int hi) {
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) f[s] = ld_frag8(rowptr + 16 * s + 8 * hi);
}

// C[i][j] (+)= sum_d A_lds[i0 + i][d] * Bfrag_j[d]   for one 32-row block of a row-major LDS tile
template <typename T, int HD>
__device__ __forceinline__ void mma_rows(f32x16& acc, const T* ldsR, int i0, const Frag<T> (&b)[HD / 16],
                                         int l31, int hi) {
    using L = Lay<T, HD>;
    const T* base = ldsR + (i0 + l31) * L::LDR + 8 * hi;
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) mma32(acc, ld_frag8(base + 16 * s), b[s]);
}

// acc_d[db][d][j] += sum_i X^T[d][i0 + i] * P[i][j]  where P = regs of a C tile (rows i, cols j) and
// X^T comes from the transposed LDS tile; row order of i matches the C-layout (see gf_common.h).
template <typename T, int HD>
__device__ __forceinline__ void mma_transposed(f32x16 (&acc)[HD / 32], const T* ldsT, int i0,
                                               const f32x16& p, int l31, int hi) {
    using L = Lay<T, HD>;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        Frag<T> pf = acc_to_frag<T>(p, t);
#pragma unroll
        for (int db = 0; db < HD / 32; ++db) {
            const int d = db * 32 + l31;
            mma32(acc[db], ld_frag8(ldsT + d * L::LDT + tswz(d, i0 + 16 * t + 8 * hi)), pf);
        }
    }
}

// ===========================================================================================
// forward
// ===========================================================================================
// One wave owns 64 query rows (two 32-row blocks): every K / V^T fragment read from LDS feeds two
// MFMAs.  K/V tiles are double-buffered: the next tile's global loads are issued before the
// compute of the current one and land in LDS after it (one barrier per tile).  The running max is
// only raised (and O / l rescaled) when it grows by more than RESCALE_THR (0 in fp32 mode).
template <typename T> struct RescaleThr { static constexpr float value = 0.f; };
template <> struct RescaleThr<bf16_t> { static constexpr float value = 4.f; };   // P <= 2^4, log2 units

template <typename T, int HD> struct StageRegs {
    static constexpr int NKI = 64 * Lay<T, HD>::CPR, NVI = 32 * Lay<T, HD>::CPR;   // work items of a tile
    static constexpr int NK = (NKI + 255) / 256;            // row-major chunks per thread
    static constexpr int NV = (NVI + 255) / 256;            // row pairs x chunks per thread (bf16 at head_dim 32: half a round)
    u32x4 k[NK];
    u32x4 v0[NV], v1[NV];
};

template <typename T, int HD>
__device__ __forceinline__ void stage_load(StageRegs<T, HD>& rg, const T* kp, int64_t kld, const T* vp,
                                           int64_t vld, int row0, int nmax) {
    using L = Lay<T, HD>;
#pragma unroll
    for (int i = 0; i < StageRegs<T, HD>::NK; ++i) {
        int c = threadIdx.x + 256 * i;
        if (StageRegs<T, HD>::NKI % 256 && c >= StageRegs<T, HD>::NKI) continue;
        int r = c / L::CPR, cc = c % L::CPR;
        int gr = min(row0 + r, nmax - 1);
        rg.k[i] = *reinterpret_cast<const u32x4*>(kp + (int64_t)gr * kld + cc * L::VEC);
    }
#pragma unroll
    for (int i = 0; i < StageRegs<T, HD>::NV; ++i) {
        int it = threadIdx.x + 256 * i;
        if (StageRegs<T, HD>::NVI % 256 && it >= StageRegs<T, HD>::NVI) continue;
        int cc = it % L::CPR, p = it / L::CPR;
        int r0 = min(row0 + 2 * p, nmax - 1), r1 = min(row0 + 2 * p + 1, nmax - 1);
        rg.v0[i] = *reinterpret_cast<const u32x4*>(vp + (int64_t)r0 * vld + cc * L::VEC);
        rg.v1[i] = *reinterpret_cast<const u32x4*>(vp + (int64_t)r1 * vld + cc * L::VEC);
    }
}

template <typename T, int HD>
__device__ __forceinline__ void stage_store(const StageRegs<T, HD>& rg, T* Ks, T* Vt) {
    using L = Lay<T, HD>;
    typedef typename Pair<T>::type pair_t;
#pragma unroll
    for (int i = 0; i < StageRegs<T, HD>::NK; ++i) {
        int c = threadIdx.x + 256 * i;
        if (StageRegs<T, HD>::NKI % 256 && c >= StageRegs<T, HD>::NKI) continue;
        int r = c / L::CPR, cc = c % L::CPR;
        *reinterpret_cast<u32x4*>(Ks + r * L::LDR + cc * L::VEC) = rg.k[i];
    }
#pragma unroll
    for (int i = 0; i < StageRegs<T, HD>::NV; ++i) {
        int it = threadIdx.x + 256 * i;
        if (StageRegs<T, HD>::NVI % 256 && it >= StageRegs<T, HD>::NVI) continue;
        int cc = it % L::CPR, p = it / L::CPR;
        union { u32x4 u; T e[L::VEC]; } a, b;
        a.u = rg.v0[i];
        b.u = rg.v1[i];
#pragma unroll
        for (int e = 0; e < L::VEC; ++e) {
            pair_t pr = {a.e[e], b.e[e]};
            const int d = cc * L::VEC + e;
            *reinterpret_cast<pair_t*>(Vt + d * L::LDT + tswz(d, tpos(2 * p))) = pr;
        }
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_fwd_kernel(AttnParams p) {
    using L = Lay<T, HD>;
    constexpr int BUF = L::ROWMAJOR + L::TRANSP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lds = reinterpret_cast<T*>(smem);

    const int nqb = (p.Nq + 255) / 256;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow0 = qb * 256 + wave * 64 + l31;

    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.sqb + h * p.sqh;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.skb + h * p.skh;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.svb + h * p.svh;

    Frag<T> qf[2][HD / 16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        load_row_frags<T, HD>(qf[j], qp + (int64_t)min(qrow0 + 32 * j, p.Nq - 1) * p.sqn, hi);

    f32x16 o[2][HD / 32];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < HD / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][db][r] = 0.f;
    float m[2] = {GF_NEG_BIG, GF_NEG_BIG}, lsum[2] = {0.f, 0.f};
    const float c = p.scale * GF_LOG2E;

    StageRegs<T, HD> rg;
    stage_load<T, HD>(rg, kp, p.skn, vp, p.svn, 0, p.Nk);
    stage_store<T, HD>(rg, lds, lds + L::ROWMAJOR);
    __syncthreads();

    const int nt = (p.Nk + 63) / 64;
    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * 64;
        const T* Ks = lds + (t & 1) * BUF;
        const T* Vt = Ks + L::ROWMAJOR;
        if (t + 1 < nt) stage_load<T, HD>(rg, kp, p.skn, vp, p.svn, kv0 + 64, p.Nk);

        f32x16 s[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[j][kb][r] = 0.f;
            const T* base = Ks + (kb * 32 + l31) * L::LDR + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                Frag<T> kf = ld_frag8(base + 16 * ks);
                mma32(s[0][kb], kf, qf[0][ks]);
                mma32(s[1][kb], kf, qf[1][ks]);
            }
        }
        if (kv0 + 64 > p.Nk) {   // ragged last tile: keys past Nk never win the max and get P = 0
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + kb * 32 + crow(r, hi) >= p.Nk) s[j][kb][r] = -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][kb][r]);
            mx = fmaxf(mx, xhalf(mx)) * c;
            if (__any(mx > m[j] + RescaleThr<T>::value)) {
                const float mnew = fmaxf(m[j], mx);
                const float alpha = fast_exp2(m[j] - mnew);
                m[j] = mnew;
                lsum[j] *= alpha;
#pragma unroll
                for (int db = 0; db < HD / 32; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[j][db][r] *= alpha;
            }
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float e = fast_exp2(fmaf(s[j][kb][r], c, -m[j]));
                    s[j][kb][r] = e;
                    ps += e;
                }
            lsum[j] += ps;
        }
        // O^T[d][q] += V^T[d][key] P[key][q]; each V^T fragment feeds both query blocks
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                Frag<T> p0 = acc_to_frag<T>(s[0][kb], tt), p1 = acc_to_frag<T>(s[1][kb], tt);
#pragma unroll
                for (int db = 0; db < HD / 32; ++db) {
                    const int d = db * 32 + l31;
                    Frag<T> vf = ld_frag8(Vt + d * L::LDT + tswz(d, kb * 32 + 16 * tt + 8 * hi));
                    mma32(o[0][db], vf, p0);
                    mma32(o[1][db], vf, p1);
                }
            }
        if (t + 1 < nt) {
            T* nb = lds + ((t + 1) & 1) * BUF;
            stage_store<T, HD>(rg, nb, nb + L::ROWMAJOR);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qrow = qrow0 + 32 * j;
        const float l = lsum[j] + xhalf(lsum[j]);
        if (qrow < p.Nq) {
            T* op = reinterpret_cast<T*>(p.o) + b * p.sob + h * p.soh + (int64_t)qrow * p.son;
            store_row<T, HD>(op, o[j], 1.f / l, hi);
            if (hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + qrow] = (m[j] + fast_log2(l)) * GF_LN2;
        }
    }
}

// ===========================================================================================
// backward, part 1: dQ (and delta = rowsum(dO * O)); wave = 64 query rows, K/V tiles double-buffered
// ===========================================================================================
template <typename T, int HD> struct PairRegs {
    static constexpr int NI = 32 * Lay<T, HD>::CPR;         // (row pair, chunk) items of a tile
    static constexpr int N = (NI + 255) / 256;              // ... per thread
    u32x4 a[N], b[N];
};
template <typename T, int HD>
__device__ __forceinline__ void pair_load(PairRegs<T, HD>& rg, const T* g, int64_t ld, int row0, int nmax) {
    using L = Lay<T, HD>;
#pragma unroll
    for (int i = 0; i < PairRegs<T, HD>::N; ++i) {
        int it = threadIdx.x + 256 * i;
        if (PairRegs<T, HD>::NI % 256 && it >= PairRegs<T, HD>::NI) continue;
        int cc = it % L::CPR, p = it / L::CPR;
        int r0 = min(row0 + 2 * p, nmax - 1), r1 = min(row0 + 2 * p + 1, nmax - 1);
        rg.a[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)r0 * ld + cc * L::VEC);
        rg.b[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)r1 * ld + cc * L::VEC);
    }
}
template <typename T, int HD, bool ROWM, bool TRAN>
__device__ __forceinline__ void pair_store(const PairRegs<T, HD>& rg, T* ldsR, T* ldsT) {
    using L = Lay<T, HD>;
    typedef typename Pair<T>::type pair_t;
#pragma unroll
    for (int i = 0; i < PairRegs<T, HD>::N; ++i) {
        int it = threadIdx.x + 256 * i;
        if (PairRegs<T, HD>::NI % 256 && it >= PairRegs<T, HD>::NI) continue;
        int cc = it % L::CPR, p = it / L::CPR;
        if (ROWM) {
            *reinterpret_cast<u32x4*>(ldsR + (2 * p) * L::LDR + cc * L::VEC) = rg.a[i];
            *reinterpret_cast<u32x4*>(ldsR + (2 * p + 1) * L::LDR + cc * L::VEC) = rg.b[i];
        }
        if (TRAN) {
            union { u32x4 u; T e[L::VEC]; } x, y;
            x.u = rg.a[i];
            y.u = rg.b[i];
#pragma unroll
            for (int e = 0; e < L::VEC; ++e) {
                pair_t pr = {x.e[e], y.e[e]};
                const int d = cc * L::VEC + e;
                *reinterpret_cast<pair_t*>(ldsT + d * L::LDT + tswz(d, tpos(2 * p))) = pr;
            }
        }
    }
}

constexpr int ATTN_PLAIN_STATS = 0x100;      // internal flag (launch_bwd_generic): bf16 dQ kernel paired with the GENERIC dK/dV kernel
template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dq_kernel(AttnParams p) {
    using L = Lay<T, HD>;
    constexpr int BUF = 2 * L::ROWMAJOR + L::TRANSP;   // K row-major | V row-major | K^T
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lds = reinterpret_cast<T*>(smem);

    const int nqb = (p.Nq + 255) / 256;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow0 = qb * 256 + wave * 64 + l31;

    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.sqb + h * p.sqh;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.skb + h * p.skh;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.svb + h * p.svh;
    const T* op = reinterpret_cast<const T*>(p.o) + b * p.sob + h * p.soh;
    const T* dop = reinterpret_cast<const T*>(p.dout) + b * p.sdob + h * p.sdoh;

    Frag<T> qf[2][HD / 16], dof[2][HD / 16];
    float delta[2], lse2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qld = min(qrow0 + 32 * j, p.Nq - 1);
        load_row_frags<T, HD>(qf[j], qp + (int64_t)qld * p.sqn, hi);
        load_row_frags<T, HD>(dof[j], dop + (int64_t)qld * p.sdon, hi);
        Frag<T> of[HD / 16];
        load_row_frags<T, HD>(of, op + (int64_t)qld * p.son, hi);
        float d = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) d += to_f32(of[s].v[e]) * to_f32(dof[j][s].v[e]);
        d += xhalf(d);
        delta[j] = d;
        const int64_t stat = ((int64_t)b * p.H + h) * p.Nq + qld;
        lse2[j] = p.lse[stat] * GF_LOG2E;
        if (qrow0 + 32 * j < p.Nq && hi == 0) {
            if (sizeof(T) == 2 && !(p.flags & ATTN_PLAIN_STATS)) {   // what attn_bwd_dkv_bf16_kernel starts its accumulators from (attention_bwd3.hip)
                p.delta[stat] = -lse2[j] / p.rr;
                p.delta[(int64_t)p.B * p.H * p.Nq + stat] = -d;
            } else {                             // the generic dK/dV kernel reads lse and delta as they are
                p.delta[stat] = d;
            }
        }
    }
    const float c = p.scale * GF_LOG2E;

    f32x16 dq[2][HD / 32];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < HD / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[j][db][r] = 0.f;

    PairRegs<T, HD> kr, vr;
    pair_load<T, HD>(kr, kp, p.skn, 0, p.Nk);
    pair_load<T, HD>(vr, vp, p.svn, 0, p.Nk);
    pair_store<T, HD, true, true>(kr, lds, lds + 2 * L::ROWMAJOR);
    pair_store<T, HD, true, false>(vr, lds + L::ROWMAJOR, nullptr);
    __syncthreads();

    const int nt = (p.Nk + 63) / 64;
    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * 64;
        const T* Ks = lds + (t & 1) * BUF;
        const T* Vs = Ks + L::ROWMAJOR;
        const T* Kt = Vs + L::ROWMAJOR;
        if (t + 1 < nt) {
            pair_load<T, HD>(kr, kp, p.skn, kv0 + 64, p.Nk);
            pair_load<T, HD>(vr, vp, p.svn, kv0 + 64, p.Nk);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s[2], dp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[j][r] = 0.f; dp[j][r] = 0.f; }
            const T* kbase = Ks + (kb * 32 + l31) * L::LDR + 8 * hi;
            const T* vbase = Vs + (kb * 32 + l31) * L::LDR + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                Frag<T> kf = ld_frag8(kbase + 16 * ks);
                mma32(s[0], kf, qf[0][ks]);
                mma32(s[1], kf, qf[1][ks]);
                Frag<T> vf = ld_frag8(vbase + 16 * ks);
                mma32(dp[0], vf, dof[0][ks]);
                mma32(dp[1], vf, dof[1][ks]);
            }
            const bool ragged = kv0 + 64 > p.Nk;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pr = fast_exp2(fmaf(s[j][r], c, -lse2[j]));
                    if (ragged && kv0 + kb * 32 + crow(r, hi) >= p.Nk) pr = 0.f;
                    s[j][r] = pr * (dp[j][r] - delta[j]);
                }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                Frag<T> d0 = acc_to_frag<T>(s[0], tt), d1 = acc_to_frag<T>(s[1], tt);
#pragma unroll
                for (int db = 0; db < HD / 32; ++db) {
                    const int d = db * 32 + l31;
                    Frag<T> kt = ld_frag8(Kt + d * L::LDT + tswz(d, kb * 32 + 16 * tt + 8 * hi));
                    mma32(dq[0][db], kt, d0);
                    mma32(dq[1][db], kt, d1);
                }
            }
        }
        if (t + 1 < nt) {
            T* nb = lds + ((t + 1) & 1) * BUF;
            pair_store<T, HD, true, true>(kr, nb, nb + 2 * L::ROWMAJOR);
            pair_store<T, HD, true, false>(vr, nb + L::ROWMAJOR, nullptr);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qrow = qrow0 + 32 * j;
        if (qrow < p.Nq) {
            T* dqp = reinterpret_cast<T*>(p.dq) + b * p.sdqb + h * p.sdqh + (int64_t)qrow * p.sdqn;
            if (p.flags & GF_ATTN_ACC_DQ) add_row<HD>(dqp, dq[j], p.scale, hi); else store_row<T, HD>(dqp, dq[j], p.scale, hi);
        }
    }
}

// ===========================================================================================
// backward, part 2: dK, dV (one workgroup per 128 keys, Q / dO tiles double-buffered)
// ===========================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dkv_kernel(AttnParams p) {
    using L = Lay<T, HD>;
    constexpr int BUF = 2 * L::ROWMAJOR + 2 * L::TRANSP + 128 * (int)(sizeof(float) / sizeof(T));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lds = reinterpret_cast<T*>(smem);

    const int nkb = (p.Nk + 127) / 128;
    const int total = nkb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int kb_ = lb % nkb, h = (lb / nkb) % p.H, b = lb / (nkb * p.H);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int krow = kb_ * 128 + wave * 32 + l31;
    const int kld = min(krow, p.Nk - 1);

    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.sqb + h * p.sqh;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.skb + h * p.skh;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.svb + h * p.svh;
    const T* dop = reinterpret_cast<const T*>(p.dout) + b * p.sdob + h * p.sdoh;
    const float* lsep = p.lse + ((int64_t)b * p.H + h) * p.Nq;
    const float* delp = p.delta + ((int64_t)b * p.H + h) * p.Nq;

    Frag<T> kf[HD / 16], vf[HD / 16];
    load_row_frags<T, HD>(kf, kp + (int64_t)kld * p.skn, hi);
    load_row_frags<T, HD>(vf, vp + (int64_t)kld * p.svn, hi);
    const float c = p.scale * GF_LOG2E;

    f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

    auto stats_of = [&](T* buf) { return reinterpret_cast<float*>(buf + 2 * L::ROWMAJOR + 2 * L::TRANSP); };
    // raw prefetch only: consuming the values here (scale / select) would force the wave to wait for the
    // whole prefetch batch at the top of the iteration; they are finished in store_stats, after the MFMAs.
    auto load_stats = [&](int q0, float& l, float& d) {
        if (threadIdx.x < 64) {
            int qi = min(q0 + (int)threadIdx.x, p.Nq - 1);
            l = lsep[qi];
            d = delp[qi];
        }
    };
    auto store_stats = [&](T* buf, int q0, float l, float d) {
        if (threadIdx.x < 64) {
            float* st = stats_of(buf);
            const bool ok = q0 + (int)threadIdx.x < p.Nq;     // rows past Nq: lse = +inf makes P exactly 0
            st[threadIdx.x] = ok ? l * GF_LOG2E : INFINITY;
            st[64 + threadIdx.x] = ok ? d : 0.f;
        }
    };

    PairRegs<T, HD> qr, dor;
    float ls = 0.f, dl = 0.f;
    pair_load<T, HD>(qr, qp, p.sqn, 0, p.Nq);
    pair_load<T, HD>(dor, dop, p.sdon, 0, p.Nq);
    load_stats(0, ls, dl);
    pair_store<T, HD, true, true>(qr, lds, lds + 2 * L::ROWMAJOR);
    pair_store<T, HD, true, true>(dor, lds + L::ROWMAJOR, lds + 2 * L::ROWMAJOR + L::TRANSP);
    store_stats(lds, 0, ls, dl);
    __syncthreads();

    const int nt = (p.Nq + 63) / 64;
    for (int t = 0; t < nt; ++t) {
        const int q0 = t * 64;
        T* cur = lds + (t & 1) * BUF;
        const T* Qs = cur;
        const T* dOs = Qs + L::ROWMAJOR;
        const T* Qt = dOs + L::ROWMAJOR;
        const T* dOt = Qt + L::TRANSP;
        const float* lse_s = stats_of(cur);
        const float* del_s = lse_s + 64;
        if (t + 1 < nt) {
            pair_load<T, HD>(qr, qp, p.sqn, q0 + 64, p.Nq);
            pair_load<T, HD>(dor, dop, p.sdon, q0 + 64, p.Nq);
            load_stats(q0 + 64, ls, dl);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // every LDS fragment of a phase is requested before the phase's first MFMA, so the reads
            // overlap instead of forming a read -> wait -> MFMA chain
            Frag<T> qa[HD / 16], da[HD / 16];
            {
                const T* qb_ = Qs + (qb * 32 + l31) * L::LDR + 8 * hi;
                const T* db_ = dOs + (qb * 32 + l31) * L::LDR + 8 * hi;
#pragma unroll
                for (int s_ = 0; s_ < HD / 16; ++s_) qa[s_] = ld_frag8(qb_ + 16 * s_);
#pragma unroll
                for (int s_ = 0; s_ < HD / 16; ++s_) da[s_] = ld_frag8(db_ + 16 * s_);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s_ = 0; s_ < HD / 16; ++s_) mma32(s, qa[s_], kf[s_]);      // S[q][key]
#pragma unroll
            for (int s_ = 0; s_ < HD / 16; ++s_) mma32(dp, da[s_], vf[s_]);     // dP[q][key]
            f32x4 l4[4], d4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                l4[g] = *reinterpret_cast<const f32x4*>(lse_s + qb * 32 + 8 * g + 4 * hi);
                d4[g] = *reinterpret_cast<const f32x4*>(del_s + qb * 32 + 8 * g + 4 * hi);
            }
            Frag<T> dot[2][HD / 32], qt[2][HD / 32];
#pragma unroll
            for (int t_ = 0; t_ < 2; ++t_)
#pragma unroll
                for (int db = 0; db < HD / 32; ++db) {
                    const int d = db * 32 + l31;
                    const int off = d * L::LDT + tswz(d, qb * 32 + 16 * t_ + 8 * hi);
                    dot[t_][db] = ld_frag8(dOt + off);
                    qt[t_][db] = ld_frag8(Qt + off);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int r = 4 * g + e;
                    float pr = fast_exp2(fmaf(s[r], c, -l4[g][e]));
                    s[r] = pr;
                    dp[r] = pr * (dp[r] - d4[g][e]);             // dS overwrites dP
                }
#pragma unroll
            for (int t_ = 0; t_ < 2; ++t_) {
                Frag<T> pf = acc_to_frag<T>(s, t_);
#pragma unroll
                for (int db = 0; db < HD / 32; ++db) mma32(dv[db], dot[t_][db], pf);
            }
#pragma unroll
            for (int t_ = 0; t_ < 2; ++t_) {
                Frag<T> pf = acc_to_frag<T>(dp, t_);
#pragma unroll
                for (int db = 0; db < HD / 32; ++db) mma32(dk[db], qt[t_][db], pf);
            }
        }
        if (t + 1 < nt) {
            T* nb = lds + ((t + 1) & 1) * BUF;
            pair_store<T, HD, true, true>(qr, nb, nb + 2 * L::ROWMAJOR);
            pair_store<T, HD, true, true>(dor, nb + L::ROWMAJOR, nb + 2 * L::ROWMAJOR + L::TRANSP);
            store_stats(nb, q0 + 64, ls, dl);
        }
        __syncthreads();
    }
    if (krow < p.Nk) {
        T* dkp = reinterpret_cast<T*>(p.dk) + b * p.sdkb + h * p.sdkh + (int64_t)krow * p.sdkn;
        T* dvp = reinterpret_cast<T*>(p.dv) + b * p.sdvb + h * p.sdvh + (int64_t)krow * p.sdvn;
        if (p.flags & GF_ATTN_ACC_DK) add_row<HD>(dkp, dk, p.scale, hi); else store_row<T, HD>(dkp, dk, p.scale, hi);
        store_row<T, HD>(dvp, dv, 1.f, hi);
    }
}

// dK / dV: one workgroup per 128 keys (32 per wave, K and V fragments in registers; K carries the exact power-of-two
// part p2 of scale * log2(e) = p2 * rr), Q / dO tiles of 64 query rows and the two per-row vectors the dQ kernel wrote
// (stat[0] = -lse * log2(e) / rr, stat[1] = -delta) streamed through the ring.  The vectors are the INITIAL VALUES of
// the S and dP accumulators -- they land there straight from LDS -- so P = exp2(rr * acc) and dS = P * acc.
constexpr int DKV_STATS = 2 * FT_TILE;                 // per wave: 16 lse | 16 delta | duplicates (256 B)
constexpr int DKV_STAGE = 2 * FT_TILE + 1024;
constexpr int DKV_NSTAGE = 3;

// Timing probes only (tools/probe/attn_stall_table.sh; the shipped library builds with 0): what does the dK/dV loop cost
// without ... 1 the exponentials, 2 the two output products (dV += P^T dO, dK += dS^T Q: 8 of the 16 MFMAs of a half tile),
// 4 the DMA of the next tiles, 8 the hardware-transposed LDS reads of the output products' operands, 16 the row-major LDS
// reads + statistics.  Results are wrong by construction.
#ifndef GF_DKV_ABL
#define GF_DKV_ABL 0
#endif
#if GF_DKV_ABL & 2
#define GF_DKV_OUT_MMA(acc, a, b) do { const auto a_ = (a); const auto b_ = (b); asm volatile("" ::"v"(a_), "v"(b_)); } while (0)
#else
#define GF_DKV_OUT_MMA(acc, a, b) mma16(acc, a, b)
#endif
template <int QB, bool PRE, bool SPLIT, typename Mid>      // SPLIT: P and dS as hi + lo bf16 pairs (attention_fwd3.hip)
__device__ __forceinline__ void dkv_half_tile(f32x16 (&dk)[2], f32x16 (&dv)[2], const bf16x8 (&kf)[4],
                                              const bf16x8 (&vf)[4], const unsigned (&aR)[4],
                                              const unsigned (&aT)[4], unsigned aS, float c,
                                              int hi, int nvalid, Mid&& mid) {
    f32x4 l4[4], d4[4];
#define GF_ST(g) l4[g] = __builtin_bit_cast(f32x4, lds_rd128<(2 * QB + (g >> 1)) * 256 + 32 * (g & 1)>(aS)); \
                 d4[g] = __builtin_bit_cast(f32x4, lds_rd128<(2 * QB + (g >> 1)) * 256 + 32 * (g & 1) + 64>(aS));
    if (!(GF_DKV_ABL & 16)) { GF_ST(0) GF_ST(1) GF_ST(2) GF_ST(3) }
    else {
#pragma unroll
        for (int g = 0; g < 4; ++g) { l4[g] = f32x4{0.f, 0.f, 0.f, 0.f}; d4[g] = l4[g]; }
    }
#undef GF_ST
    u32x4 qa[4], da[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qa[s] = (GF_DKV_ABL & 16) ? u32x4{0u, 0u, 0u, 0u} : lds_rd128<QB * 4096>(aR[s]);
    wait_lgkm<4>();
    f32x16 sa, dp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                                   // the per-row vectors ARE the initial values (stored
        tie(l4[g]);                                                 // negated and in the exponent's units by the dQ kernel)
        tie(d4[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sa[4 * g + e] = l4[g][e];
            dp[4 * g + e] = d4[g][e];
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) da[s] = (GF_DKV_ABL & 16) ? u32x4{0u, 0u, 0u, 0u} : lds_rd128<FT_TILE + QB * 4096>(aR[s]);
    wait_lgkm<4>();
#pragma unroll
    for (int s = 0; s < 4; ++s) {                               // k-steps chained on ONE accumulator: switching
        tie(qa[s]);                                             // accumulators between MFMAs measured 9 % slower
        mma16(sa, as_frag(qa[s]), kf[s]);                       // S[q][key] - lse/scale
    }
    // transposed operands: [t][db] -> rows 16t + 4hi + {0..3} (lo) and + 8 (hi half), columns db*32 + l31
    u32x2 dot[2][2][2], qt[2][2][2];
#define GF_TR(dst, base, t, db) dst[t][db][0] = (GF_DKV_ABL & 8) ? u32x2{0u, 0u} : lds_rdtr<base + QB * 4096 + t * 2048>(aT[db]); \
                                dst[t][db][1] = (GF_DKV_ABL & 8) ? u32x2{0u, 0u} : lds_rdtr<base + QB * 4096 + t * 2048 + 1024>(aT[2 + db]);
    GF_TR(dot, FT_TILE, 0, 0) GF_TR(dot, FT_TILE, 0, 1) GF_TR(dot, FT_TILE, 1, 0) GF_TR(dot, FT_TILE, 1, 1)
    wait_lgkm<8>();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tie(da[s]);
        mma16(dp, as_frag(da[s]), vf[s]);                       // dP[q][key] - delta
    }
    GF_TR(qt, 0, 0, 0) GF_TR(qt, 0, 0, 1) GF_TR(qt, 0, 1, 0) GF_TR(qt, 0, 1, 1)
#undef GF_TR
    if (!(GF_DKV_ABL & 4)) mid();                               // DMA issue rides in the VALU gap
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pr = (GF_DKV_ABL & 1) ? sa[r] : fast_exp2(PRE ? sa[r] : sa[r] * c);
        sa[r] = pr;
        dp[r] = pr * dp[r];                                     // dS overwrites dP
    }
    if (nvalid < 64) {                                          // ragged last tile: rows past Nq contribute 0
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (QB * 32 + crow(r, hi) >= nvalid) { sa[r] = 0.f; dp[r] = 0.f; }
    }
    wait_lgkm<8>();
    {
        const bf16x8 pf0 = cvt_frag(sa, 0), pf1 = cvt_frag(sa, 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            tie(dot[0][db][0]); tie(dot[0][db][1]); tie(dot[1][db][0]); tie(dot[1][db][1]);
            GF_DKV_OUT_MMA(dv[db], as_frag(dot[0][db][0], dot[0][db][1]), pf0);
            GF_DKV_OUT_MMA(dv[db], as_frag(dot[1][db][0], dot[1][db][1]), pf1);
        }
        if (SPLIT) {
            const bf16x8 pl0 = cvt_frag_lo(sa, 0, pf0), pl1 = cvt_frag_lo(sa, 1, pf1);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                GF_DKV_OUT_MMA(dv[db], as_frag(dot[0][db][0], dot[0][db][1]), pl0);
                GF_DKV_OUT_MMA(dv[db], as_frag(dot[1][db][0], dot[1][db][1]), pl1);
            }
        }
    }
    wait_lgkm<0>();
    {
        const bf16x8 pf0 = cvt_frag(dp, 0), pf1 = cvt_frag(dp, 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            tie(qt[0][db][0]); tie(qt[0][db][1]); tie(qt[1][db][0]); tie(qt[1][db][1]);
            GF_DKV_OUT_MMA(dk[db], as_frag(qt[0][db][0], qt[0][db][1]), pf0);
            GF_DKV_OUT_MMA(dk[db], as_frag(qt[1][db][0], qt[1][db][1]), pf1);
        }
        if (SPLIT) {
            const bf16x8 pl0 = cvt_frag_lo(dp, 0, pf0), pl1 = cvt_frag_lo(dp, 1, pf1);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                GF_DKV_OUT_MMA(dk[db], as_frag(qt[0][db][0], qt[0][db][1]), pl0);
                GF_DKV_OUT_MMA(dk[db], as_frag(qt[1][db][0], qt[1][db][1]), pl1);
            }
        }
    }
}

// NW waves per workgroup (32 keys each) share the Q/dO stream; PRE: rr == 1, no multiply per score; EVEN: Nq % 64 == 0 -- no
// ragged tile: unconditional re-fetching DMA (tiles past the end fetch the last one again), constant wait counts, no
// row masking: no tile-dependent branch in the loop (attention_fwd3.hip)
template <int NW, bool PRE, bool EVEN, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, 8 / NW) void attn_bwd_dkv_bf16_kernel(AttnParams p) {
    constexpr int KPB = 32 * NW, PPW = 8 / NW;    // keys per block, 1-KiB DMA pieces per wave and matrix
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int nkb = (p.Nk + KPB - 1) / KPB;
    const int total = nkb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int kb_ = lb % nkb, h = (lb / nkb) % p.H, b = lb / (nkb * p.H);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5, s16 = lane & 15, half = (lane >> 4) & 1;
#ifdef GF_DKV_PRIO
    // static priority asymmetry between the waves that share a SIMD (probe knob; see DESIGN.md "convoy")
    if (NW == 8) { if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(GF_DKV_PRIO); }
    else if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(GF_DKV_PRIO);
#endif
    const int krow = kb_ * KPB + wave * 32 + l31;
    const int kld = min(krow, p.Nk - 1);

    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + h * p.sqh;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + h * p.skh;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.svb + h * p.svh;
    const bf16_t* dop = reinterpret_cast<const bf16_t*>(p.dout) + b * p.sdob + h * p.sdoh;
    const float* lsep = p.delta + ((int64_t)b * p.H + h) * p.Nq;                       // stat[0]
    const float* delp = lsep + (int64_t)p.B * p.H * p.Nq;                               // stat[1]

    // ---- DMA descriptors: chunk (2 wave + i) * 64 + lane of a tile -> row, swizzled source column
    int drow[PPW], dcol[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        drow[i] = (PPW * wave + i) * 8 + (lane >> 3);
        dcol[i] = ((lane & 7) ^ fswz(drow[i])) * 8;
    }
    const float* statp = (lane & 16) ? delp : lsep;
    const int srow = 16 * wave + s16;
    const bool stat_wave = __builtin_amdgcn_readfirstlane(wave) < 4;
    // part 0: Q pieces + stats, part 1: dO pieces (issued in the VALU gaps of the two half tiles)
    const int64_t qstep = 64 * p.sqn, dostep = 64 * p.sdon;
    const bf16_t* gq[PPW];
    const bf16_t* gdo[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        gq[i] = qp + (int64_t)drow[i] * p.sqn + dcol[i];
        gdo[i] = dop + (int64_t)drow[i] * p.sdon + dcol[i];
    }
    auto issue_part = [&](int part, int t, int stage) {
        char* sb = smem + stage * DKV_STAGE;
        const int q0 = t * 64;
        if (EVEN || q0 + 64 <= p.Nq) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                if (part == 0) dma16(gq[i] + t * qstep, sb + (PPW * wave + i) * 1024);
                else dma16(gdo[i] + t * dostep, sb + FT_TILE + (PPW * wave + i) * 1024);
            }
        } else {                                                // ragged last tile: rows clamped to Nq - 1
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int64_t r = min(q0 + drow[i], p.Nq - 1);
                if (part == 0) dma16(qp + r * p.sqn + dcol[i], sb + (PPW * wave + i) * 1024);
                else dma16(dop + r * p.sdon + dcol[i], sb + FT_TILE + (PPW * wave + i) * 1024);
            }
        }
        if (part == 0 && stat_wave) dma4(statp + (EVEN ? q0 + srow : min(q0 + srow, p.Nq - 1)), sb + DKV_STATS + wave * 256);
    };
    auto issue_tile = [&](int t, int stage) { issue_part(0, t, stage); issue_part(1, t, stage); };

    const int nt = (p.Nq + 63) / 64;
    issue_tile(0, 0);
    if (EVEN) issue_tile(min(1, nt - 1), 1);
    else if (nt > 1) issue_tile(1, 1);

    const float p2 = p.p2, c = PRE ? 1.f : p.rr;                   // c: the non-power-of-two rest rr of scale * log2(e)
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const bf16x8*>(kp + (int64_t)kld * p.skn + 16 * s + 8 * hi);
        if (p2 != 1.f) kf[s] = scale_frag(kf[s], p2);
        vf[s] = *reinterpret_cast<const bf16x8*>(vp + (int64_t)kld * p.svn + 16 * s + 8 * hi);
    }

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

    // ---- per-lane LDS read addresses (stage 0); see the layout note above
    unsigned bR[4], bT[4];
    {
        const unsigned rb = l31 * 128 + 16 * (hi ^ fswz(l31));
#pragma unroll
        for (int s = 0; s < 4; ++s) bR[s] = lds0 + (rb ^ (32 * s));
        const int bq = s16 >> 3;
        const unsigned tb = (4 * hi + (s16 >> 2)) * 128 + 8 * (s16 & 1) +
                            16 * ((2 * half + ((s16 & 3) >> 1)) ^ (4 * bq + hi));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int db = 0; db < 2; ++db) bT[2 * u + db] = lds0 + (tb ^ (32 * u) ^ (64 * db));
    }
    const unsigned bS = lds0 + DKV_STATS + 16 * hi;

    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        if (!EVEN && t + 1 >= nt) wait_vm<0>();                   // tile t landed (this wave's share)
        else if (stat_wave) wait_vm<2 * PPW + 1>();
        else wait_vm<2 * PPW>();
        __builtin_amdgcn_s_barrier();                             // ... everyone's; stage of tile t-1 is free
        __builtin_amdgcn_sched_barrier(0);
        const unsigned so = stage * DKV_STAGE;
        unsigned aR[4], aT[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] = bR[i] + so; aT[i] = bT[i] + so; }
        const int nvalid = EVEN ? 64 : p.Nq - t * 64;
        const int nstage = stage == 0 ? 2 : stage - 1;
        const bool more = EVEN || t + 2 < nt;
        const int tn = EVEN ? min(t + 2, nt - 1) : t + 2;
        dkv_half_tile<0, PRE, SPLIT>(dk, dv, kf, vf, aR, aT, bS + so, c, hi, nvalid,
                         [&] { if (more) issue_part(0, tn, nstage); });
        dkv_half_tile<1, PRE, SPLIT>(dk, dv, kf, vf, aR, aT, bS + so, c, hi, nvalid,
                         [&] { if (more) issue_part(1, tn, nstage); });
        stage = stage == 2 ? 0 : stage + 1;
    }
    if (EVEN) wait_vm<0>();                                       // the re-fetched tail tiles
    if (krow < p.Nk) {
        bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dk) + b * p.sdkb + h * p.sdkh + (int64_t)krow * p.sdkn;
        bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dv) + b * p.sdvb + h * p.sdvh + (int64_t)krow * p.sdvn;
        if (p.flags & GF_ATTN_ACC_DK) add_row<64>(dkp, dk, p.scale, hi); else store_row<bf16_t, 64>(dkp, dk, p.scale, hi);
        store_row<bf16_t, 64>(dvp, dv, 1.f, hi);
    }
}


__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nqb = (p.Nq + 255) / 256;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow0 = qb * 256 + wave * 64 + l31;

    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + h * p.sqh;
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + h * p.skh;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + b * p.svb + h * p.svh;

    const int nt = (p.Nk + 63) / 64;
    auto issue_tile = [&](int t, int stage) {
        char* sb = smem + stage * FQ_STAGE;
        fq_issue(kp, p.skn, t * 64, p.Nk, sb, wave, lane);
        fq_issue(vp, p.svn, t * 64, p.Nk, sb + FT_TILE, wave, lane);
    };
    issue_tile(0, 0);
    if (nt > 1) issue_tile(1, 1);

    bf16x8 qf[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[j][s] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)min(qrow0 + 32 * j, p.Nq - 1) * p.sqn + 16 * s + 8 * hi);

    f32x16 o[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][db][r] = 0.f;
    float m[2] = {GF_NEG_BIG, GF_NEG_BIG}, lsum[2] = {0.f, 0.f};
    const float c = p.scale * GF_LOG2E;
    const FqAddr ad = fq_addresses(lds0, lane);

    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 >= nt) wait_vm<0>();                            // tile t landed (this wave's pieces)
        else wait_vm<4>();
        __builtin_amdgcn_s_barrier();                             // ... everyone's; the stage of tile t-1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) issue_tile(t + 2, stage == 0 ? 2 : stage - 1);
        const unsigned so = stage * FQ_STAGE;
        unsigned aR[4], aT[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] = ad.aR[i] + so; aT[i] = ad.aT[i] + so; }
        const int kv0 = t * 64;

        // ---- S^T[key][q] for both 32-key blocks: all eight K fragments requested up front
        u32x4 ka[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ka[0][s] = lds_rd128<0>(aR[s]);
#pragma unroll
        for (int s = 0; s < 4; ++s) ka[1][s] = lds_rd128<4096>(aR[s]);
        f32x16 sc[2][2];                                          // [q block j][key block kb]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[j][kb][r] = 0.f;
        wait_lgkm<4>();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            tie(ka[0][s]);
            mma16(sc[0][0], as_frag(ka[0][s]), qf[0][s]);
            mma16(sc[1][0], as_frag(ka[0][s]), qf[1][s]);
        }
        // V^T fragments of this tile: requested now, consumed after the softmax
        u32x2 vt0[2][2][2], vt1[2][2][2];
        GF_FQ_TR(vt0, FT_TILE, 0, 0, 0) GF_FQ_TR(vt0, FT_TILE, 0, 0, 1) GF_FQ_TR(vt0, FT_TILE, 0, 1, 0) GF_FQ_TR(vt0, FT_TILE, 0, 1, 1)
        wait_lgkm<8>();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            tie(ka[1][s]);
            mma16(sc[0][1], as_frag(ka[1][s]), qf[0][s]);
            mma16(sc[1][1], as_frag(ka[1][s]), qf[1][s]);
        }

        if (kv0 + 64 > p.Nk) {   // ragged last tile: keys past Nk never win the max and get P = 0
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + kb * 32 + crow(r, hi) >= p.Nk) sc[j][kb][r] = -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[j][kb][r]);
            mx = fmaxf(mx, xhalf(mx)) * c;
            if (__any(mx > m[j] + RescaleThr<bf16_t>::value)) {
                const float mnew = fmaxf(m[j], mx);
                const float alpha = fast_exp2(m[j] - mnew);
                m[j] = mnew;
                lsum[j] *= alpha;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[j][db][r] *= alpha;
            }
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = fast_exp2(fmaf(sc[j][kb][r], c, -m[j]));
                    sc[j][kb][r] = e;
                    ps += e;
                }
            lsum[j] += ps;
        }
        // ---- O^T[d][q] += V^T[d][key] P[key][q]; each V^T fragment feeds both query blocks.  The second key block's
        // V^T fragments are requested while the first block's MFMAs run (lgkmcnt holds at most 15 requests)
        wait_lgkm<0>();
        GF_FQ_TR(vt1, FT_TILE, 1, 0, 0) GF_FQ_TR(vt1, FT_TILE, 1, 0, 1) GF_FQ_TR(vt1, FT_TILE, 1, 1, 0) GF_FQ_TR(vt1, FT_TILE, 1, 1, 1)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const bf16x8 p0 = cvt_frag(sc[0][0], tt), p1 = cvt_frag(sc[1][0], tt);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                tie(vt0[tt][db][0]); tie(vt0[tt][db][1]);
                const bf16x8 vf = as_frag(vt0[tt][db][0], vt0[tt][db][1]);
                mma16(o[0][db], vf, p0);
                mma16(o[1][db], vf, p1);
            }
        }
        wait_lgkm<0>();
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const bf16x8 p0 = cvt_frag(sc[0][1], tt), p1 = cvt_frag(sc[1][1], tt);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                tie(vt1[tt][db][0]); tie(vt1[tt][db][1]);
                const bf16x8 vf = as_frag(vt1[tt][db][0], vt1[tt][db][1]);
                mma16(o[0][db], vf, p0);
                mma16(o[1][db], vf, p1);
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qrow = qrow0 + 32 * j;
        const float l = lsum[j] + xhalf(lsum[j]);
        if (qrow < p.Nq) {
            bf16_t* op = reinterpret_cast<bf16_t*>(p.o) + b * p.sob + h * p.soh + (int64_t)qrow * p.son;
            store_row<bf16_t, 64>(op, o[j], 1.f / l, hi);
            if (hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + qrow] = (m[j] + fast_log2(l)) * GF_LN2;
        }
    }
}


// ===========================================================================================
// Assignment-head backward, one side (lightglue.py:256-290 autograd; the "dual softmax" part):
//   S_so = oth_s . own_o,   dS_so = exp(S_so - ns_s) gs_s + exp(S_so - no_o) go_o,   dOwn_o = sum_s dS_so oth_s
// with (ns, gs) / (no, go) the log-sum-exp normaliser and incoming coefficient of the streamed row / of the owner.
// Called twice (owner = md1 rows -> d md1, owner = md0 rows -> d md0): no [B,N,N] dS tensor is written and no
// library GEMM follows.  It lives in this file because it IS the attention forward's machinery with D = 256: the
// streamed [64 x 256] tile is four 64 x 64 sub-tiles in the forward's LDS-DMA ring layout, S^T comes from row
// fragments (ds_read_b128), dS goes from the accumulator registers straight into the second product, whose other
// operand oth^T is read with ds_read_b64_tr_b16 from the SAME tile.  One wave owns 32 owner rows and the whole
// 256-wide output row (8 accumulator tiles): one wave per SIMD, 512-register budget.
// ===========================================================================================
constexpr int HB_TILE = 4 * FT_TILE;                 // 64 rows x 256 channels
constexpr int HB_STAGE = HB_TILE + 1024;             // + ns | gs (64 floats each) | spare copies
constexpr int HB_NSTAGE = 3;

struct HeadBwdParams {
    const bf16_t* own; const bf16_t* oth;            // [B, No, 256], [B, Ns, 256]
    const float* no; const float* go;                // [B, No]
    const float* ns; const float* gs;                // [B, Ns]
    bf16_t* down;                                    // [B, No, 256]
    int B, No, Ns;
};

__global__ __launch_bounds__(256, 1) void head_bwd_bf16_kernel(HeadBwdParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nob = (p.No + 127) / 128;
    const int lb = xcd_remap(blockIdx.x, nob * p.B);
    const int ob = lb % nob, b = lb / nob;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int orow = ob * 128 + wave * 32 + l31;
    const int old_ = min(orow, p.No - 1);
    const bf16_t* ownp = p.own + ((int64_t)b * p.No + old_) * 256;
    const bf16_t* othp = p.oth + (int64_t)b * p.Ns * 256;
    const float* nsp = p.ns + (int64_t)b * p.Ns;
    const float* gsp = p.gs + (int64_t)b * p.Ns;

    const int nt = (p.Ns + 63) / 64;
    // part i of tile t's DMA: i < 4 the two pieces of sub-tile i, i == 4 the per-row vectors (waves 0 / 1 bring ns / gs,
    // waves 2 / 3 the same into spare slots: equal vmcnt in every wave).  Tiles past the end re-fetch the last one.
    auto issue_part = [&](int t, int stage, int i) {
        char* sb = smem + stage * HB_STAGE;
        const int tc = min(t, nt - 1);
        if (i < 4) {
            fq_issue(othp + 64 * i, 256, tc * 64, p.Ns, sb + i * FT_TILE, wave, lane);
        } else {
            const float* src = (wave & 1) ? gsp : nsp;
            dma4(src + min(tc * 64 + lane, p.Ns - 1), sb + HB_TILE + (wave & 1) * 256 + (wave >> 1) * 512);
        }
    };
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_part(0, 0, i);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_part(1, 1, i);

    bf16x8 of[16];                                     // owner row: B operand of S^T, k-step 4c + s
#pragma unroll
    for (int k = 0; k < 16; ++k) of[k] = *reinterpret_cast<const bf16x8*>(ownp + 16 * k + 8 * hi);
    const float no2 = p.no[(int64_t)b * p.No + old_] * GF_LOG2E;
    const float go = p.go[(int64_t)b * p.No + old_];

    f32x16 acc[8];                                     // dOwn^T[d][o]: d-tile 2c + db
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const FqAddr ad = fq_addresses(lds0, lane);

    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        wait_vm<9>();                                             // tile t landed (this wave's pieces; t + 1 in flight)
        __builtin_amdgcn_s_barrier();                             // ... everyone's; the stage of tile t-1 is free
        __builtin_amdgcn_sched_barrier(0);
        const int nstage = stage == 0 ? 2 : stage - 1;            // tile t + 2 goes there, piece by piece between MFMAs
        const unsigned so = stage * HB_STAGE;
        unsigned aR[4], aT[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { aR[i] = ad.aR[i] + so; aT[i] = ad.aT[i] + so; }
        const unsigned aV = lds0 + so + HB_TILE + 16 * hi;        // ns of rows 8 g + 4 hi .. + 3 (gs: + 256 bytes)
        const int s0 = t * 64;
        if (s0 + 64 > p.Ns) {        // ragged last tile: the clamped duplicate rows become zero rows of oth (no contribution)
            const int lim = p.Ns - s0;
            for (int i = threadIdx.x; i < 4 * 64 * 8; i += 256) {
                const int row = (i >> 3) & 63;
                if (row >= lim) *reinterpret_cast<u32x4*>(smem + so + (i >> 9) * FT_TILE + row * 128 + (i & 7) * 16) = u32x4{0, 0, 0, 0};
            }
            __syncthreads();
        }

        // One wave per SIMD: the exponentials only overlap the matrix pipe if they sit BETWEEN MFMAs in program order.
        // Schedule per tile: S(rows 0-31) | S(rows 32-63) with dS(rows 0-31) one element per MFMA gap |
        // second product (rows 0-31) with dS(rows 32-63) in its gaps | second product (rows 32-63).
        u32x4 vn[2][4], vg[2][4];
        f32x16 sc0, sc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc0[r] = 0.f; sc1[r] = 0.f; }
        auto elem = [&](f32x16& sc, const u32x4 (&n_)[4], const u32x4 (&g_)[4], int i) {
            const f32x4 n4 = __builtin_bit_cast(f32x4, n_[i >> 2]), g4 = __builtin_bit_cast(f32x4, g_[i >> 2]);
            const float x = sc[i];
            sc[i] = fast_exp2((x - n4[i & 3]) * GF_LOG2E) * g4[i & 3] + fast_exp2(fmaf(x, GF_LOG2E, -no2)) * go;
        };
        u32x4 ka[4], kc[4];
#define GF_HB_RD(dst, c, KB) _Pragma("unroll") for (int s = 0; s < 4; ++s) dst[s] = lds_rd128<(c) * FT_TILE + (KB) * 4096>(aR[s]);
#define GF_HB_S(sc, src, c, SIDE) _Pragma("unroll") for (int s = 0; s < 4; ++s) { tie(src[s]); mma16(sc, as_frag(src[s]), of[4 * (c) + s]); SIDE(4 * (c) + s) }
#define GF_HB_NONE(i)
#define GF_HB_DMA(i) if ((i) % 3 == 0 && (i) / 3 < 5) issue_part(t + 2, nstage, (i) / 3);
#define GF_HB_E0(i) elem(sc0, vn[0], vg[0], i);
#define GF_HB_E1(i) elem(sc1, vn[1], vg[1], i);
        // (never more than 12 LDS requests in flight: the counter holds 15)
#pragma unroll
        for (int g = 0; g < 4; ++g) { vn[0][g] = lds_rd128<0>(aV + 32 * g); vg[0][g] = lds_rd128<256>(aV + 32 * g); }
        GF_HB_RD(ka, 0, 0)
        wait_lgkm<4>();                                       // the row vectors of rows 0-31
        GF_HB_RD(kc, 1, 0)
        wait_lgkm<4>();
        GF_HB_S(sc0, ka, 0, GF_HB_DMA)
        GF_HB_RD(ka, 2, 0)
        wait_lgkm<4>();
        GF_HB_S(sc0, kc, 1, GF_HB_DMA)
        GF_HB_RD(kc, 3, 0)
        wait_lgkm<4>();
        GF_HB_S(sc0, ka, 2, GF_HB_DMA)
        GF_HB_RD(ka, 0, 1)
        wait_lgkm<4>();
        GF_HB_S(sc0, kc, 3, GF_HB_DMA)
#pragma unroll
        for (int g = 0; g < 4; ++g) { vn[1][g] = lds_rd128<128>(aV + 32 * g); vg[1][g] = lds_rd128<256 + 128>(aV + 32 * g); }
#pragma unroll
        for (int g = 0; g < 4; ++g) { tie(vn[0][g]); tie(vg[0][g]); }
        wait_lgkm<8>();                                       // ka (rows 32-63, sub-tile 0); the vectors still in flight
        GF_HB_RD(kc, 1, 1)
        GF_HB_S(sc1, ka, 0, GF_HB_E0)
        wait_lgkm<4>();                                       // the row vectors of rows 32-63
        GF_HB_RD(ka, 2, 1)
        wait_lgkm<4>();
        GF_HB_S(sc1, kc, 1, GF_HB_E0)
        GF_HB_RD(kc, 3, 1)
        wait_lgkm<4>();
        GF_HB_S(sc1, ka, 2, GF_HB_E0)
        // oth^T fragments of sub-tile 0, rows 0-31, requested under the last S block
        u32x2 va[2][2][2], vb[2][2][2];
#define GF_HB_TR(dst, c, KB) GF_FQ_TR(dst, (c) * FT_TILE, KB, 0, 0) GF_FQ_TR(dst, (c) * FT_TILE, KB, 0, 1) \
                             GF_FQ_TR(dst, (c) * FT_TILE, KB, 1, 0) GF_FQ_TR(dst, (c) * FT_TILE, KB, 1, 1)
        GF_HB_TR(va, 0, 0)
        wait_lgkm<8>();
        GF_HB_S(sc1, kc, 3, GF_HB_E0)
#pragma unroll
        for (int g = 0; g < 4; ++g) { tie(vn[1][g]); tie(vg[1][g]); }
        bf16x8 p0 = cvt_frag(sc0, 0), p1 = cvt_frag(sc0, 1);

        // ---- dOwn^T[d][o] += oth^T[d][s] dS[s][o]: sub-tile c feeds d-tiles 2c, 2c + 1 for both 16-row k-steps
#define GF_HB_MMA(src, c, SIDE)                                                                            \
        _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                 \
            tie(src[0][db][0]); tie(src[0][db][1]); tie(src[1][db][0]); tie(src[1][db][1]);                \
            mma16(acc[2 * (c) + db], as_frag(src[0][db][0], src[0][db][1]), p0); SIDE(4 * (c) + 2 * db)    \
            mma16(acc[2 * (c) + db], as_frag(src[1][db][0], src[1][db][1]), p1); SIDE(4 * (c) + 2 * db + 1) \
        }
        wait_lgkm<0>();
        GF_HB_TR(vb, 1, 0)
        GF_HB_MMA(va, 0, GF_HB_E1)
        wait_lgkm<0>();
        GF_HB_TR(va, 2, 0)
        GF_HB_MMA(vb, 1, GF_HB_E1)
        wait_lgkm<0>();
        GF_HB_TR(vb, 3, 0)
        GF_HB_MMA(va, 2, GF_HB_E1)
        wait_lgkm<0>();
        GF_HB_TR(va, 0, 1)
        GF_HB_MMA(vb, 3, GF_HB_E1)
        p0 = cvt_frag(sc1, 0); p1 = cvt_frag(sc1, 1);
        wait_lgkm<0>();
        GF_HB_TR(vb, 1, 1)
        GF_HB_MMA(va, 0, GF_HB_NONE)
        wait_lgkm<0>();
        GF_HB_TR(va, 2, 1)
        GF_HB_MMA(vb, 1, GF_HB_NONE)
        wait_lgkm<0>();
        GF_HB_TR(vb, 3, 1)
        GF_HB_MMA(va, 2, GF_HB_NONE)
        wait_lgkm<0>();
        GF_HB_MMA(vb, 3, GF_HB_NONE)
#undef GF_HB_MMA
#undef GF_HB_TR
#undef GF_HB_RD
#undef GF_HB_S
#undef GF_HB_NONE
#undef GF_HB_DMA
#undef GF_HB_E0
#undef GF_HB_E1
        stage = stage == 2 ? 0 : stage + 1;
    }
    wait_vm<0>();                                                 // the re-fetched tail tiles
    if (orow < p.No) {
        bf16_t* dst = p.down + ((int64_t)b * p.No + orow) * 256;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x16 pair[2] = {acc[2 * c], acc[2 * c + 1]};
            store_row<bf16_t, 64>(dst + 64 * c, pair, 1.f, hi);
        }
    }
}

template <typename T, int HD> size_t fwd_lds() { return 2 * (Lay<T, HD>::ROWMAJOR + Lay<T, HD>::TRANSP) * sizeof(T); }
template <typename T, int HD> size_t dq_lds() { return 2 * (2 * Lay<T, HD>::ROWMAJOR + Lay<T, HD>::TRANSP) * sizeof(T); }
template <typename T, int HD> size_t dkv_lds() {
    return 2 * ((2 * Lay<T, HD>::ROWMAJOR + 2 * Lay<T, HD>::TRANSP) * sizeof(T) + 128 * sizeof(float));
}

template <typename K> int set_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

template <typename T> int launch_fwd(const AttnParams& p, hipStream_t st) {
    int total = ((p.Nq + 255) / 256) * p.H * p.B;
#if !defined(GF_ATTN_FWD_V1) && !defined(GF_ATTN_FWD_V2)
    if constexpr (sizeof(T) == 2) {
        // buffer descriptors address rows with 32-bit byte offsets
        if (kvdma_ok(p.Nk, p.skn) && kvdma_ok(p.Nk, p.svn)) {
            return launch_fwd3_bf16(p, st);
        }
    }
#endif
#ifndef GF_ATTN_FWD_V1
    if constexpr (sizeof(T) == 2) {
        const size_t l2 = FQ_NSTAGE * FQ_STAGE;
        if (int e = set_lds(attn_fwd_bf16_kernel, l2)) return e;
        attn_fwd_bf16_kernel<<<dim3(total), dim3(256), l2, st>>>(p);
        return (int)hipGetLastError();
    }
#endif
    size_t lds = fwd_lds<T, 64>();
    if (int e = set_lds(attn_fwd_kernel<T, 64>, lds)) return e;
    attn_fwd_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}
template <typename T> int launch_bwd(const AttnParams& p, hipStream_t st) {
    int total = ((p.Nq + 255) / 256) * p.H * p.B;
    size_t lds = dq_lds<T, 64>();
    bool dq_done = false;
#if !defined(GF_ATTN_DQ_V2)
    if constexpr (sizeof(T) == 2) {
        if (kvdma_ok(p.Nk, p.skn) && kvdma_ok(p.Nk, p.svn)) {
            if (int e = launch_dq3_bf16(p, st)) return e;
            dq_done = true;
        }
    }
#endif
    if (!dq_done) {
        if (int e = set_lds(attn_bwd_dq_kernel<T, 64>, lds)) return e;
        attn_bwd_dq_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
        if (int e = (int)hipGetLastError()) return e;
    }
    total = ((p.Nk + 127) / 128) * p.H * p.B;
    if constexpr (sizeof(T) == 2) {
#ifdef GF_DKV_NW
        constexpr int NW = GF_DKV_NW;
#else
        constexpr int NW = 4;                       // 8 waves sharing one Q/dO stream measured 7 % slower
#endif
        total = ((p.Nk + 32 * NW - 1) / (32 * NW)) * p.H * p.B;
        lds = DKV_NSTAGE * DKV_STAGE;
        void (*const kern[8])(AttnParams) = {attn_bwd_dkv_bf16_kernel<NW, false, false>, attn_bwd_dkv_bf16_kernel<NW, false, true>,
                                             attn_bwd_dkv_bf16_kernel<NW, true, false>, attn_bwd_dkv_bf16_kernel<NW, true, true>,
                                             attn_bwd_dkv_bf16_kernel<NW, false, false, true>, attn_bwd_dkv_bf16_kernel<NW, false, true, true>,
                                             attn_bwd_dkv_bf16_kernel<NW, true, false, true>, attn_bwd_dkv_bf16_kernel<NW, true, true, true>};
        static unsigned long long attr_set = 0;     // function attributes are per DEVICE: one bit per device ordinal
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
            for (auto k : kern)
                if (int e = set_lds(k, lds)) return e;
            if (dev < 64) attr_set |= 1ull << dev;
        }
        kern[((p.flags & GF_ATTN_SPLIT) ? 4 : 0) + (p.rr == 1.f ? 2 : 0) + (p.Nq % 64 == 0 ? 1 : 0)]<<<dim3(total), dim3(64 * NW), lds, st>>>(p);
        return (int)hipGetLastError();
    }
    lds = dkv_lds<T, 64>();
    if (int e = set_lds(attn_bwd_dkv_kernel<T, 64>, lds)) return e;
    attn_bwd_dkv_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}

// head_dim 32 / 128: the generic register-staged kernels (the LDS-DMA kernels above are specialised on 64-wide heads).  A capability
// path, not a tuned one (at 128 the fragments of a row exceed the register budget and spill).
constexpr size_t ATTN_LDS_MAX = 160 * 1024;
template <typename T, int HD> int launch_fwd_generic(const AttnParams& p, hipStream_t st) {
    const size_t lds = fwd_lds<T, HD>();
    if (lds > ATTN_LDS_MAX) return GF_ERR_UNSUPPORTED;
    if (int e = set_lds(attn_fwd_kernel<T, HD>, lds)) return e;
    attn_fwd_kernel<T, HD><<<dim3(((p.Nq + 255) / 256) * p.H * p.B), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}
template <typename T, int HD> int launch_bwd_generic(const AttnParams& p_, hipStream_t st) {
    AttnParams p = p_;
    p.flags |= ATTN_PLAIN_STATS;
    const size_t l1 = dq_lds<T, HD>(), l2 = dkv_lds<T, HD>();
    if (l1 > ATTN_LDS_MAX || l2 > ATTN_LDS_MAX) return GF_ERR_UNSUPPORTED;
    if (int e = set_lds(attn_bwd_dq_kernel<T, HD>, l1)) return e;
    attn_bwd_dq_kernel<T, HD><<<dim3(((p.Nq + 255) / 256) * p.H * p.B), dim3(256), l1, st>>>(p);
    if (int e = (int)hipGetLastError()) return e;
    if (int e = set_lds(attn_bwd_dkv_kernel<T, HD>, l2)) return e;
    attn_bwd_dkv_kernel<T, HD><<<dim3(((p.Nk + 127) / 128) * p.H * p.B), dim3(256), l2, st>>>(p);
    return (int)hipGetLastError();
}
template <typename T> int launch_fwd_any(const AttnParams& p, hipStream_t st, int D) {
    if (D == 64) return launch_fwd<T>(p, st);
    if (D == 32) return launch_fwd_generic<T, 32>(p, st);
    if (D == 128) return launch_fwd_generic<T, 128>(p, st);
    return GF_ERR_UNSUPPORTED;
}
template <typename T> int launch_bwd_any(const AttnParams& p, hipStream_t st, int D) {
    if (D == 64) return launch_bwd<T>(p, st);
    if (D == 32) return launch_bwd_generic<T, 32>(p, st);
    if (D == 128) return launch_bwd_generic<T, 128>(p, st);
    return GF_ERR_UNSUPPORTED;
}

bool bad_stride(const int64_t* s, int n, int align) {
    for (int i = 0; i < n; ++i)
        if (s[i] % align) return true;
    return false;
}

}  // namespace

extern "C" int gf_attn_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                              int B, int H, int Nq, int Nk, int D,
                              const int64_t* q_strides, const int64_t* k_strides,
                              const int64_t* v_strides, const int64_t* o_strides,
                              float scale, int dtype, int flags, float* o32, void* stream);
extern "C" int gf_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                           int B, int H, int Nq, int Nk, int D,
                           const int64_t* q_strides, const int64_t* k_strides,
                           const int64_t* v_strides, const int64_t* o_strides,
                           float scale, int dtype, void* stream) {
    return gf_attn_fwd_ex(q, k, v, o, lse, B, H, Nq, Nk, D, q_strides, k_strides, v_strides, o_strides, scale, dtype, 0, nullptr, stream);
}
extern "C" int gf_attn_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                              int B, int H, int Nq, int Nk, int D,
                              const int64_t* q_strides, const int64_t* k_strides,
                              const int64_t* v_strides, const int64_t* o_strides,
                              float scale, int dtype, int flags, float* o32, void* stream) {
    if (D != 64 && D != 32 && D != 128) return GF_ERR_UNSUPPORTED;
    if (flags & ~GF_ATTN_SPLIT) return GF_ERR_UNSUPPORTED;
    if (D != 64 && (flags & GF_ATTN_SPLIT) && dtype != GF_F32) return GF_ERR_UNSUPPORTED;   // split products: 64-wide heads only
    if (dtype == GF_F32) flags = 0;                                  // fp32 operands: nothing to split
    // the split products exist in the LDS-DMA bf16 kernels only (rows addressed through 32-bit buffer offsets)
    if ((flags & GF_ATTN_SPLIT) && !(kvdma_ok(Nk, k_strides[1]) && kvdma_ok(Nk, v_strides[1]))) return GF_ERR_UNSUPPORTED;
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return GF_ERR_SHAPE;
    const int align = dtype == GF_BF16 ? 8 : 4;
    if (bad_stride(q_strides, 3, align) || bad_stride(k_strides, 3, align) ||
        bad_stride(v_strides, 3, align) || bad_stride(o_strides, 3, align))
        return GF_ERR_ALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    host_split_scale(scale, p.p2, p.rr);
    p.flags = flags;
    p.o32 = (flags & GF_ATTN_SPLIT) ? o32 : nullptr;
    p.sqb = q_strides[0]; p.sqn = q_strides[1]; p.sqh = q_strides[2];
    p.skb = k_strides[0]; p.skn = k_strides[1]; p.skh = k_strides[2];
    p.svb = v_strides[0]; p.svn = v_strides[1]; p.svh = v_strides[2];
    p.sob = o_strides[0]; p.son = o_strides[1]; p.soh = o_strides[2];
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_fwd_any<float>(p, st, D);
    if (dtype == GF_BF16) return launch_fwd_any<bf16_t>(p, st, D);
    return GF_ERR_DTYPE;
}

extern "C" int gf_attn_bwd(const void* q, const void* k, const void* v, const void* o,
                           const void* dout, const float* lse, float* delta,
                           void* dq, void* dk, void* dv,
                           int B, int H, int Nq, int Nk, int D,
                           const int64_t* q_strides, const int64_t* k_strides,
                           const int64_t* v_strides, const int64_t* o_strides,
                           const int64_t* do_strides, const int64_t* dq_strides,
                           const int64_t* dk_strides, const int64_t* dv_strides,
                           float scale, int dtype, void* stream) {
    return gf_attn_bwd_acc(q, k, v, o, dout, lse, delta, dq, dk, dv, B, H, Nq, Nk, D, q_strides, k_strides, v_strides, o_strides,
                           do_strides, dq_strides, dk_strides, dv_strides, scale, dtype, 0, stream);
}

extern "C" int gf_attn_bwd_acc(const void* q, const void* k, const void* v, const void* o,
                               const void* dout, const float* lse, float* delta,
                               void* dq, void* dk, void* dv,
                               int B, int H, int Nq, int Nk, int D,
                               const int64_t* q_strides, const int64_t* k_strides,
                               const int64_t* v_strides, const int64_t* o_strides,
                               const int64_t* do_strides, const int64_t* dq_strides,
                               const int64_t* dk_strides, const int64_t* dv_strides,
                               float scale, int dtype, int flags, void* stream) {
    if (D != 64 && D != 32 && D != 128) return GF_ERR_UNSUPPORTED;
    if (flags & ~7) return GF_ERR_UNSUPPORTED;
    if (D != 64 && (flags & GF_ATTN_SPLIT) && dtype != GF_F32) return GF_ERR_UNSUPPORTED;
    if (dtype == GF_F32) flags &= ~GF_ATTN_SPLIT;                    // fp32 operands: nothing to split
    if ((flags & GF_ATTN_SPLIT) && !(kvdma_ok(Nk, k_strides[1]) && kvdma_ok(Nk, v_strides[1]))) return GF_ERR_UNSUPPORTED;
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return GF_ERR_SHAPE;
    const int align = dtype == GF_BF16 ? 8 : 4;
    const int64_t* all[8] = {q_strides, k_strides, v_strides, o_strides,
                             do_strides, dq_strides, dk_strides, dv_strides};
    for (int i = 0; i < 8; ++i)          // (GF_ATTN_SPLIT: `o` is the forward's fp32 copy, strides in fp32 elements)
        if (bad_stride(all[i], 3, (i == 3 && (flags & GF_ATTN_SPLIT)) ? 4 : align)) return GF_ERR_ALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.o = const_cast<void*>(o); p.dout = dout;
    p.lse = const_cast<float*>(lse); p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    host_split_scale(scale, p.p2, p.rr);
    p.flags = flags;
    p.sqb = q_strides[0]; p.sqn = q_strides[1]; p.sqh = q_strides[2];
    p.skb = k_strides[0]; p.skn = k_strides[1]; p.skh = k_strides[2];
    p.svb = v_strides[0]; p.svn = v_strides[1]; p.svh = v_strides[2];
    p.sob = o_strides[0]; p.son = o_strides[1]; p.soh = o_strides[2];
    p.sdob = do_strides[0]; p.sdon = do_strides[1]; p.sdoh = do_strides[2];
    p.sdqb = dq_strides[0]; p.sdqn = dq_strides[1]; p.sdqh = dq_strides[2];
    p.sdkb = dk_strides[0]; p.sdkn = dk_strides[1]; p.sdkh = dk_strides[2];
    p.sdvb = dv_strides[0]; p.sdvn = dv_strides[1]; p.sdvh = dv_strides[2];
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_bwd_any<float>(p, st, D);
    if (dtype == GF_BF16) return launch_bwd_any<bf16_t>(p, st, D);
    return GF_ERR_DTYPE;
}

extern "C" int gf_head_bwd(const void* a, const void* b, const float* r, const float* c, const float* gr, const float* gc,
                           void* da, void* db, int B, int M, int N, int D, int dtype, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0) return GF_ERR_SHAPE;
    if (dtype != GF_BF16 || D != 256) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)HB_NSTAGE * HB_STAGE;
    if (int e = set_lds(head_bwd_bf16_kernel, lds)) return e;
    HeadBwdParams p;
    // d md1: owner = md1 rows (columns of S), streamed = md0 rows
    p.own = static_cast<const bf16_t*>(b); p.oth = static_cast<const bf16_t*>(a); p.no = c; p.go = gc; p.ns = r; p.gs = gr;
    p.down = static_cast<bf16_t*>(db); p.B = B; p.No = N; p.Ns = M;
    head_bwd_bf16_kernel<<<dim3(((N + 127) / 128) * B), dim3(256), lds, st>>>(p);
    if (int e = (int)hipGetLastError()) return e;
    // d md0: owner = md0 rows, streamed = md1 rows
    p.own = static_cast<const bf16_t*>(a); p.oth = static_cast<const bf16_t*>(b); p.no = r; p.go = gr; p.ns = c; p.gs = gc;
    p.down = static_cast<bf16_t*>(da); p.No = M; p.Ns = N;
    head_bwd_bf16_kernel<<<dim3(((M + 127) / 128) * B), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}
