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

namespace {

struct AttnParams {
    const void* q; const void* k; const void* v; void* o;
    const void* dout; void* dq; void* dk; void* dv;
    float* lse; float* delta;
    int B, H, Nq, Nk;
    int64_t sqb, sqn, sqh, skb, skn, skh, svb, svn, svh, sob, son, soh;
    // gradients: dq/dout use the o-like strides given below
    int64_t sdob, sdon, sdoh, sdqb, sdqn, sdqh, sdkb, sdkn, sdkh, sdvb, sdvn, sdvh;
    float scale;
};

template <typename T, int HD> struct Lay {
    static constexpr int VEC = 16 / sizeof(T);   // elements per 16-byte chunk
    static constexpr int CPR = HD / VEC;         // chunks per row
    static constexpr int LDR = HD + VEC;         // row-major LDS stride (+16 B: conflict-free b128)
    static constexpr int LDT = 64 + 8;           // transposed LDS stride (64 rows of the tile + 16 B pad)
    static constexpr int ROWMAJOR = 64 * LDR;    // elements
    static constexpr int TRANSP = HD * LDT;      // elements
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// write acc^T: lane owns row (rowptr), acc[db][r] is column db*32 + crow(r,hi)
template <typename T, int HD>
__device__ __forceinline__ void store_row(T* rowptr, const f32x16 (&acc)[HD / 32], float mul, int hi) {
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            st4(rowptr + db * 32 + 8 * g + 4 * hi, acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul,
                acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
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
    static constexpr int NK = 64 * Lay<T, HD>::CPR / 256;   // row-major chunks per thread
    static constexpr int NV = 32 * Lay<T, HD>::CPR / 256;   // row pairs x chunks per thread
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
        int r = c / L::CPR, cc = c % L::CPR;
        int gr = min(row0 + r, nmax - 1);
        rg.k[i] = *reinterpret_cast<const u32x4*>(kp + (int64_t)gr * kld + cc * L::VEC);
    }
#pragma unroll
    for (int i = 0; i < StageRegs<T, HD>::NV; ++i) {
        int it = threadIdx.x + 256 * i;
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
        int r = c / L::CPR, cc = c % L::CPR;
        *reinterpret_cast<u32x4*>(Ks + r * L::LDR + cc * L::VEC) = rg.k[i];
    }
#pragma unroll
    for (int i = 0; i < StageRegs<T, HD>::NV; ++i) {
        int it = threadIdx.x + 256 * i;
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
    static constexpr int N = 32 * Lay<T, HD>::CPR / 256;   // (row pair, chunk) items per thread
    u32x4 a[N], b[N];
};
template <typename T, int HD>
__device__ __forceinline__ void pair_load(PairRegs<T, HD>& rg, const T* g, int64_t ld, int row0, int nmax) {
    using L = Lay<T, HD>;
#pragma unroll
    for (int i = 0; i < PairRegs<T, HD>::N; ++i) {
        int it = threadIdx.x + 256 * i;
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
        if (qrow0 + 32 * j < p.Nq && hi == 0) p.delta[stat] = d;
        lse2[j] = p.lse[stat] * GF_LOG2E;
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
            store_row<T, HD>(dqp, dq[j], p.scale, hi);
        }
    }
}

// ===========================================================================================
// backward, part 2: dK, dV (one workgroup per 128 keys, Q / dO tiles double-buffered)
// ===========================================================================================
// Single LDS buffer, no register prefetch, <= 168 registers: three workgroups (12 waves) per CU hide
// the LDS / HBM latency that two double-buffered workgroups could not (this kernel reads four staged
// tiles per step and sat 53 % of its wave cycles in s_waitcnt at two waves per SIMD).
template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 1) void attn_bwd_dkv_kernel(AttnParams p) {
    using L = Lay<T, HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = reinterpret_cast<T*>(smem);
    T* dOs = Qs + L::ROWMAJOR;
    T* Qt = dOs + L::ROWMAJOR;
    T* dOt = Qt + L::TRANSP;
    float* lse_s = reinterpret_cast<float*>(dOt + L::TRANSP);
    float* del_s = lse_s + 64;

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

    for (int q0 = 0; q0 < p.Nq; q0 += 64) {
        __syncthreads();
        {
            PairRegs<T, HD> rg;
            pair_load<T, HD>(rg, qp, p.sqn, q0, p.Nq);
            pair_store<T, HD, true, true>(rg, Qs, Qt);
            pair_load<T, HD>(rg, dop, p.sdon, q0, p.Nq);
            pair_store<T, HD, true, true>(rg, dOs, dOt);
        }
        if (threadIdx.x < 64) {
            const int qi = q0 + threadIdx.x;
            const bool ok = qi < p.Nq;                       // rows past Nq: lse = +inf makes P exactly 0
            lse_s[threadIdx.x] = ok ? lsep[qi] * GF_LOG2E : INFINITY;
            del_s[threadIdx.x] = ok ? delp[qi] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            mma_rows<T, HD>(s, Qs, qb * 32, kf, l31, hi);     // S[q][key]
            mma_rows<T, HD>(dp, dOs, qb * 32, vf, l31, hi);   // dP[q][key]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb * 32 + 8 * g + 4 * hi);
                f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int r = 4 * g + e;
                    float pr = fast_exp2(fmaf(s[r], c, -l4[e]));
                    s[r] = pr;
                    dp[r] = pr * (dp[r] - d4[e]);             // dS overwrites dP
                }
            }
            mma_transposed<T, HD>(dv, dOt, qb * 32, s, l31, hi);
            mma_transposed<T, HD>(dk, Qt, qb * 32, dp, l31, hi);
        }
    }
    if (krow < p.Nk) {
        T* dkp = reinterpret_cast<T*>(p.dk) + b * p.sdkb + h * p.sdkh + (int64_t)krow * p.sdkn;
        T* dvp = reinterpret_cast<T*>(p.dv) + b * p.sdvb + h * p.sdvh + (int64_t)krow * p.sdvn;
        store_row<T, HD>(dkp, dk, p.scale, hi);
        store_row<T, HD>(dvp, dv, 1.f, hi);
    }
}

template <typename T, int HD> size_t fwd_lds() { return 2 * (Lay<T, HD>::ROWMAJOR + Lay<T, HD>::TRANSP) * sizeof(T); }
template <typename T, int HD> size_t dq_lds() { return 2 * (2 * Lay<T, HD>::ROWMAJOR + Lay<T, HD>::TRANSP) * sizeof(T); }
template <typename T, int HD> size_t dkv_lds() {
    return (2 * Lay<T, HD>::ROWMAJOR + 2 * Lay<T, HD>::TRANSP) * sizeof(T) + 128 * sizeof(float);
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
    size_t lds = fwd_lds<T, 64>();
    if (int e = set_lds(attn_fwd_kernel<T, 64>, lds)) return e;
    attn_fwd_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}
template <typename T> int launch_bwd(const AttnParams& p, hipStream_t st) {
    int total = ((p.Nq + 255) / 256) * p.H * p.B;
    size_t lds = dq_lds<T, 64>();
    if (int e = set_lds(attn_bwd_dq_kernel<T, 64>, lds)) return e;
    attn_bwd_dq_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
    if (int e = (int)hipGetLastError()) return e;
    total = ((p.Nk + 127) / 128) * p.H * p.B;
    lds = dkv_lds<T, 64>();
    if (int e = set_lds(attn_bwd_dkv_kernel<T, 64>, lds)) return e;
    attn_bwd_dkv_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}

bool bad_stride(const int64_t* s, int n, int align) {
    for (int i = 0; i < n; ++i)
        if (s[i] % align) return true;
    return false;
}

}  // namespace

extern "C" int gf_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                           int B, int H, int Nq, int Nk, int D,
                           const int64_t* q_strides, const int64_t* k_strides,
                           const int64_t* v_strides, const int64_t* o_strides,
                           float scale, int dtype, void* stream) {
    if (D != 64) return GF_ERR_UNSUPPORTED;
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return GF_ERR_SHAPE;
    const int align = dtype == GF_BF16 ? 8 : 4;
    if (bad_stride(q_strides, 3, align) || bad_stride(k_strides, 3, align) ||
        bad_stride(v_strides, 3, align) || bad_stride(o_strides, 3, align))
        return GF_ERR_ALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    p.sqb = q_strides[0]; p.sqn = q_strides[1]; p.sqh = q_strides[2];
    p.skb = k_strides[0]; p.skn = k_strides[1]; p.skh = k_strides[2];
    p.svb = v_strides[0]; p.svn = v_strides[1]; p.svh = v_strides[2];
    p.sob = o_strides[0]; p.son = o_strides[1]; p.soh = o_strides[2];
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_fwd<float>(p, st);
    if (dtype == GF_BF16) return launch_fwd<bf16_t>(p, st);
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
    if (D != 64) return GF_ERR_UNSUPPORTED;
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return GF_ERR_SHAPE;
    const int align = dtype == GF_BF16 ? 8 : 4;
    const int64_t* all[8] = {q_strides, k_strides, v_strides, o_strides,
                             do_strides, dq_strides, dk_strides, dv_strides};
    for (int i = 0; i < 8; ++i)
        if (bad_stride(all[i], 3, align)) return GF_ERR_ALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.o = const_cast<void*>(o); p.dout = dout;
    p.lse = const_cast<float*>(lse); p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    p.sqb = q_strides[0]; p.sqn = q_strides[1]; p.sqh = q_strides[2];
    p.skb = k_strides[0]; p.skn = k_strides[1]; p.skh = k_strides[2];
    p.svb = v_strides[0]; p.svn = v_strides[1]; p.svh = v_strides[2];
    p.sob = o_strides[0]; p.son = o_strides[1]; p.soh = o_strides[2];
    p.sdob = do_strides[0]; p.sdon = do_strides[1]; p.sdoh = do_strides[2];
    p.sdqb = dq_strides[0]; p.sdqn = dq_strides[1]; p.sdqh = dq_strides[2];
    p.sdkb = dk_strides[0]; p.sdkn = dk_strides[1]; p.sdkh = dk_strides[2];
    p.sdvb = dv_strides[0]; p.sdvn = dv_strides[1]; p.sdvh = dv_strides[2];
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GF_F32) return launch_bwd<float>(p, st);
    if (dtype == GF_BF16) return launch_bwd<bf16_t>(p, st);
    return GF_ERR_DTYPE;
}
