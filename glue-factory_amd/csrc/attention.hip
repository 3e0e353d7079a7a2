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
    static constexpr int LDT = 64 + 4;           // transposed LDS stride (64 rows of the tile + pad)
    static constexpr int ROWMAJOR = 64 * LDR;    // elements
    static constexpr int TRANSP = HD * LDT;      // elements
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// rows (2p, 2p+1) x chunk cc: optional row-major copy + transposed copy ldsT[d][row]
template <typename T, int HD, bool ROWM, bool TRAN>
__device__ __forceinline__ void stage_tile(T* ldsR, T* ldsT, const T* g, int64_t ld, int row0, int nmax) {
    using L = Lay<T, HD>;
    typedef typename Pair<T>::type pair_t;
    for (int it = threadIdx.x; it < 32 * L::CPR; it += 256) {
        int p = it & 31, cc = it >> 5;
        int r0 = min(row0 + 2 * p, nmax - 1), r1 = min(row0 + 2 * p + 1, nmax - 1);
        union { u32x4 u; T e[L::VEC]; } v0, v1;
        v0.u = *reinterpret_cast<const u32x4*>(g + (int64_t)r0 * ld + cc * L::VEC);
        v1.u = *reinterpret_cast<const u32x4*>(g + (int64_t)r1 * ld + cc * L::VEC);
        if (ROWM) {
            *reinterpret_cast<u32x4*>(ldsR + (2 * p) * L::LDR + cc * L::VEC) = v0.u;
            *reinterpret_cast<u32x4*>(ldsR + (2 * p + 1) * L::LDR + cc * L::VEC) = v1.u;
        }
        if (TRAN) {
#pragma unroll
            for (int e = 0; e < L::VEC; ++e) {
                pair_t pr = {v0.e[e], v1.e[e]};
                *reinterpret_cast<pair_t*>(ldsT + (cc * L::VEC + e) * L::LDT + 2 * p) = pr;
            }
        }
    }
}

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
            const T* base = ldsT + (db * 32 + l31) * L::LDT + i0 + 16 * t + 4 * hi;
            mma32(acc[db], ld_frag4x2(base, base + 8), pf);
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
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    using L = Lay<T, HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vt = Ks + L::ROWMAJOR;

    const int nqb = (p.Nq + 127) / 128;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = qb * 128 + wave * 32 + l31;
    const int qld = min(qrow, p.Nq - 1);

    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.sqb + h * p.sqh;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.skb + h * p.skh;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.svb + h * p.svh;

    Frag<T> qf[HD / 16];
    load_row_frags<T, HD>(qf, qp + (int64_t)qld * p.sqn, hi);

    f32x16 o[HD / 32];
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = GF_NEG_BIG, lsum = 0.f;
    const float c = p.scale * GF_LOG2E;

    for (int kv0 = 0; kv0 < p.Nk; kv0 += 64) {
        __syncthreads();
        stage_rowmajor<T, HD>(Ks, kp, p.skn, kv0, p.Nk);
        stage_tile<T, HD, false, true>(nullptr, Vt, vp, p.svn, kv0, p.Nk);
        __syncthreads();

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            mma_rows<T, HD>(s[kb], Ks, kb * 32, qf, l31, hi);
        }
        // scale into log2 units, mask keys beyond Nk, tile max
        float mx = GF_NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int key = kv0 + kb * 32 + crow(r, hi);
                float x = (key < p.Nk) ? s[kb][r] * c : -INFINITY;
                s[kb][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, xhalf(mx));
        const float mnew = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mnew);
        m = mnew;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = fast_exp2(s[kb][r] - mnew);
                s[kb][r] = e;
                ps += e;
            }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int db = 0; db < HD / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mma_transposed<T, HD>(o, Vt, kb * 32, s[kb], l31, hi);
    }
    lsum += xhalf(lsum);
    if (qrow < p.Nq) {
        T* op = reinterpret_cast<T*>(p.o) + b * p.sob + h * p.soh + (int64_t)qrow * p.son;
        store_row<T, HD>(op, o, 1.f / lsum, hi);
        if (hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + qrow] = (m + fast_log2(lsum)) * GF_LN2;
    }
}

// ===========================================================================================
// backward, part 1: dQ (and delta = rowsum(dO * O))
// ===========================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
    using L = Lay<T, HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + L::ROWMAJOR;
    T* Kt = Vs + L::ROWMAJOR;

    const int nqb = (p.Nq + 127) / 128;
    const int total = nqb * p.H * p.B;
    int lb = xcd_remap(blockIdx.x, total);
    const int qb = lb % nqb, h = (lb / nqb) % p.H, b = lb / (nqb * p.H);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = qb * 128 + wave * 32 + l31;
    const int qld = min(qrow, p.Nq - 1);

    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.sqb + h * p.sqh;
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.skb + h * p.skh;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.svb + h * p.svh;
    const T* op = reinterpret_cast<const T*>(p.o) + b * p.sob + h * p.soh;
    const T* dop = reinterpret_cast<const T*>(p.dout) + b * p.sdob + h * p.sdoh;

    Frag<T> qf[HD / 16], dof[HD / 16];
    load_row_frags<T, HD>(qf, qp + (int64_t)qld * p.sqn, hi);
    load_row_frags<T, HD>(dof, dop + (int64_t)qld * p.sdon, hi);
    float delta = 0.f;
    {
        Frag<T> of[HD / 16];
        load_row_frags<T, HD>(of, op + (int64_t)qld * p.son, hi);
#pragma unroll
        for (int s = 0; s < HD / 16; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += to_f32(of[s].v[e]) * to_f32(dof[s].v[e]);
        delta += xhalf(delta);
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.Nq + qld;
    if (qrow < p.Nq && hi == 0) p.delta[stat] = delta;
    const float lse2 = p.lse[stat] * GF_LOG2E;
    const float c = p.scale * GF_LOG2E;

    f32x16 dq[HD / 32];
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

    for (int kv0 = 0; kv0 < p.Nk; kv0 += 64) {
        __syncthreads();
        stage_tile<T, HD, true, true>(Ks, Kt, kp, p.skn, kv0, p.Nk);
        stage_rowmajor<T, HD>(Vs, vp, p.svn, kv0, p.Nk);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            mma_rows<T, HD>(s, Ks, kb * 32, qf, l31, hi);
            mma_rows<T, HD>(dp, Vs, kb * 32, dof, l31, hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int key = kv0 + kb * 32 + crow(r, hi);
                float pr = (key < p.Nk) ? fast_exp2(s[r] * c - lse2) : 0.f;
                s[r] = pr * (dp[r] - delta);
            }
            mma_transposed<T, HD>(dq, Kt, kb * 32, s, l31, hi);
        }
    }
    if (qrow < p.Nq) {
        T* dqp = reinterpret_cast<T*>(p.dq) + b * p.sdqb + h * p.sdqh + (int64_t)qrow * p.sdqn;
        store_row<T, HD>(dqp, dq, p.scale, hi);
    }
}

// ===========================================================================================
// backward, part 2: dK, dV (one workgroup per 128 keys, streaming Q / dO tiles)
// ===========================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
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
        stage_tile<T, HD, true, true>(Qs, Qt, qp, p.sqn, q0, p.Nq);
        stage_tile<T, HD, true, true>(dOs, dOt, dop, p.sdon, q0, p.Nq);
        if (threadIdx.x < 64) {
            int qi = q0 + threadIdx.x;
            // rows past Nq: lse = +inf makes P exactly 0
            lse_s[threadIdx.x] = (qi < p.Nq) ? lsep[qi] * GF_LOG2E : INFINITY;
            del_s[threadIdx.x] = (qi < p.Nq) ? delp[qi] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            mma_rows<T, HD>(s, Qs, qb * 32, kf, l31, hi);     // S[q][key]
            mma_rows<T, HD>(dp, dOs, qb * 32, vf, l31, hi);   // dP[q][key]
            f32x16 ds;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb * 32 + 8 * g + 4 * hi);
                f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int r = 4 * g + e;
                    float pr = fast_exp2(s[r] * c - l4[e]);
                    s[r] = pr;
                    ds[r] = pr * (dp[r] - d4[e]);
                }
            }
            mma_transposed<T, HD>(dv, dOt, qb * 32, s, l31, hi);
            mma_transposed<T, HD>(dk, Qt, qb * 32, ds, l31, hi);
        }
    }
    if (krow < p.Nk) {
        T* dkp = reinterpret_cast<T*>(p.dk) + b * p.sdkb + h * p.sdkh + (int64_t)krow * p.sdkn;
        T* dvp = reinterpret_cast<T*>(p.dv) + b * p.sdvb + h * p.sdvh + (int64_t)krow * p.sdvn;
        store_row<T, HD>(dkp, dk, p.scale, hi);
        store_row<T, HD>(dvp, dv, 1.f, hi);
    }
}

template <typename T, int HD> size_t fwd_lds() { return (Lay<T, HD>::ROWMAJOR + Lay<T, HD>::TRANSP) * sizeof(T); }
template <typename T, int HD> size_t dq_lds() { return (2 * Lay<T, HD>::ROWMAJOR + Lay<T, HD>::TRANSP) * sizeof(T); }
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
    int total = ((p.Nq + 127) / 128) * p.H * p.B;
    size_t lds = fwd_lds<T, 64>();
    if (int e = set_lds(attn_fwd_kernel<T, 64>, lds)) return e;
    attn_fwd_kernel<T, 64><<<dim3(total), dim3(256), lds, st>>>(p);
    return (int)hipGetLastError();
}
template <typename T> int launch_bwd(const AttnParams& p, hipStream_t st) {
    int total = ((p.Nq + 127) / 128) * p.H * p.B;
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
