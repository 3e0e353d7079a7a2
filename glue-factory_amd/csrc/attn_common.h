// Shared pieces of the attention kernels (attention.hip, attention_fwd3.hip): launch parameters, the row store of a
// transposed accumulator and the bf16 LDS-DMA / transposing-read helpers.  Everything is static / in a named namespace
// so that both translation units can include it.
#pragma once
#include "gf_common.h"
#include <cmath>

namespace gfattn {

enum { GF_ATTN_ACC_DQ = 1, GF_ATTN_ACC_DK = 2 };      // (+ GF_ATTN_SPLIT = 4, include/gf_amd.h)

struct AttnParams {
    const void* q; const void* k; const void* v; void* o;
    const void* dout; void* dq; void* dk; void* dv;
    float* lse; float* delta;
    float* o32;       // GF_ATTN_SPLIT forward: fp32 copy of the output, [B, Nq, H, 64] contiguous (the backward's delta = sum o dO
                      // then carries no bf16 rounding of o: the reference's fp32 attention keeps its output in fp32 as well)
    int B, H, Nq, Nk;
    int64_t sqb, sqn, sqh, skb, skn, skh, svb, svn, svh, sob, son, soh;
    // gradients: dq/dout use the o-like strides given below
    int64_t sdob, sdon, sdoh, sdqb, sdqn, sdqh, sdkb, sdkn, sdkh, sdvb, sdvn, sdvh;
    float scale;
    int flags;        // GF_ATTN_ACC_DQ / GF_ATTN_ACC_DK: add to what dq / dk hold instead of overwriting (gf_attn_bwd_acc);
                      // GF_ATTN_SPLIT: P and dS enter the second products as hi + lo bf16 pairs (fp32-equivalent)
    float p2, rr;     // scale * log2(e) = p2 * rr, p2 a power of two, rr in [1, 2) (host_split_scale); rr == 1 exactly when the
                      // caller pre-multiplied its operands (scale = ln 2): the kernels then skip the multiply per score
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

// acc^T added to what the row already holds (fp32 sum, one rounding)
template <int HD>
__device__ __forceinline__ void add_row(float* rowptr, const f32x16 (&acc)[HD / 32], float mul, int hi) {
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float* p = rowptr + db * 32 + 8 * g + 4 * hi;
            const f32x4 old = *reinterpret_cast<const f32x4*>(p);
            st4(p, old[0] + acc[db][4 * g] * mul, old[1] + acc[db][4 * g + 1] * mul, old[2] + acc[db][4 * g + 2] * mul,
                old[3] + acc[db][4 * g + 3] * mul);
        }
}
template <int HD>
__device__ __forceinline__ void add_row(bf16_t* rowptr, const f32x16 (&acc)[HD / 32], float mul, int hi) {
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16_t* p = rowptr + db * 32 + 8 * g + 4 * hi;
            const bf16x4 old = *reinterpret_cast<const bf16x4*>(p);
            st4(p, (float)old[0] + acc[db][4 * g] * mul, (float)old[1] + acc[db][4 * g + 1] * mul,
                (float)old[2] + acc[db][4 * g + 2] * mul, (float)old[3] + acc[db][4 * g + 3] * mul);
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
// bf16 fast path: LDS-DMA staging + hardware-transposed LDS reads
// ===========================================================================================
// Tiles are 64 rows x 128 B, row-major and UNPADDED in LDS (a `global_load_lds_dwordx4` writes
// 64 lanes x 16 B contiguously, so there is no room for padding): instead the 16-byte chunk index of
// row r is XORed with fswz(r).  Conflict-free for both access patterns used below:
//   * ds_read_b128 of one chunk per row, rows = the lanes of a 16-lane service group;
//   * ds_read_b64_tr_b16 of a 4-row x 32-column block per 32 lanes (the operand of the second MFMA,
//     read TRANSPOSED straight from the row-major tile: no transposed copy, no VALU transposition).
// The DMA writes LDS asynchronously (tracked by vmcnt): a 3-stage ring keeps two tiles in flight and
// needs ONE raw s_barrier per tile.  Every LDS read is inline asm with hand-placed s_waitcnt, since
// the compiler would otherwise drain vmcnt (= the prefetch) in front of each read.
constexpr int FT_TILE = 8192;              // bytes of one 64 x 64 bf16 tile
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int fswz(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}
__device__ __forceinline__ void dma16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 4, 0, 0);
}
template <int OFF> __device__ __forceinline__ u32x4 lds_rd128(unsigned a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ u32x2 lds_rdtr(unsigned a) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the asm reads are invisible to the compiler's waitcnt bookkeeping: a tie after the wait orders every use
template <typename V> __device__ __forceinline__ void tie(V& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 as_frag(u32x2 lo, u32x2 hi) {
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 cvt_frag(const f32x16& c, int t) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (bf16_t)c[8 * t + e];
    return f;
}
// the low half of a split operand: bf16(c - float(hi)) for the 8 values hi = cvt_frag(c, t) was rounded from
__device__ __forceinline__ bf16x8 cvt_frag_lo(const f32x16& c, int t, bf16x8 hi) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (bf16_t)(c[8 * t + e] - (float)hi[e]);
    return f;
}
__device__ __forceinline__ void mma16(f32x16& acc, bf16x8 a, bf16x8 b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// ===========================================================================================
// bf16 forward and dQ on the same staging scheme as the dK/dV kernel above
// ===========================================================================================
// K and V tiles (64 keys x 128 B, row-major, chunk-swizzled by fswz) arrive by LDS-DMA into a 3-stage ring; the
// score MFMAs read K rows with ds_read_b128, the second product reads V^T (forward) or K^T (dQ) straight from the
// row-major tile with ds_read_b64_tr_b16 -- no transposed copy, no staging registers, no bank conflicts (the
// register-staged kernels spent 36-43 % of their LDS cycles on conflicts of the transposed-tile stores).
// One wave owns 64 query rows (two 32-row blocks: every K / V fragment feeds two MFMAs), 4 waves per workgroup.
constexpr int FQ_STAGE = 2 * FT_TILE;          // K tile | V tile
constexpr int FQ_NSTAGE = 3;

struct FqAddr { unsigned aR[4], aT[4]; };      // per-lane LDS read addresses of stage 0 (see attn_bwd_dkv_bf16_kernel)

__device__ __forceinline__ FqAddr fq_addresses(unsigned lds0, int lane) {
    const int l31 = lane & 31, hi = lane >> 5, s16 = lane & 15, half = (lane >> 4) & 1;
    FqAddr a;
    const unsigned rb = l31 * 128 + 16 * (hi ^ fswz(l31));
#pragma unroll
    for (int s = 0; s < 4; ++s) a.aR[s] = lds0 + (rb ^ (32 * s));
    const int bq = s16 >> 3;
    const unsigned tb = (4 * hi + (s16 >> 2)) * 128 + 8 * (s16 & 1) + 16 * ((2 * half + ((s16 & 3) >> 1)) ^ (4 * bq + hi));
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int db = 0; db < 2; ++db) a.aT[2 * u + db] = lds0 + (tb ^ (32 * u) ^ (64 * db));
    return a;
}

// DMA of one 64-row tile of a [rows, 64] bf16 matrix (row stride ld): wave w moves pieces 2w, 2w+1 (8 rows each)
__device__ __forceinline__ void fq_issue(const bf16_t* base, int64_t ld, int row0, int nmax, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = 2 * wave + i;
        const int r = piece * 8 + (lane >> 3);
        const int col = ((lane & 7) ^ fswz(r)) * 8;
        dma16(base + (int64_t)min(row0 + r, nmax - 1) * ld + col, dst + piece * 1024);
    }
}

// transposed operand [t][db] of the 32-row block KB of the tile at byte offset BASE (rows 16t + 4hi + {0..3} and + 8)
#define GF_FQ_TR(dst, BASE, KB, t, db) dst[t][db][0] = lds_rdtr<BASE + KB * 4096 + t * 2048>(aT[db]); \
                                       dst[t][db][1] = lds_rdtr<BASE + KB * 4096 + t * 2048 + 1024>(aT[2 + db]);

__device__ __forceinline__ f32x16 mma16c(bf16x8 a, bf16x8 b, const f32x16& c) {          // D = A B + C with D != C
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// K | V tiles of 64 keys (FQ_STAGE bytes per ring stage) fetched with `buffer_load_dwordx4 ... lds` by a 4-wave
// workgroup: wave w moves rows 8w .. 8w+7 and 32+8w .. 32+8w+7 of each matrix (the same chunk swizzle: fswz has period
// 16 rows), i.e. ONE per-lane 32-bit byte offset per matrix; the tile advance and the +32 rows ride in the scalar offset.
struct KvDma {
    __amdgpu_buffer_rsrc_t rk, rv;
    int vk, vv, r8, col, skn2, svn2, Nk;
    __device__ __forceinline__ void init(const bf16_t* kp, const bf16_t* vp, int64_t skn, int64_t svn, int Nk_, int wave, int lane) {
        skn2 = (int)skn * 2; svn2 = (int)svn * 2; Nk = Nk_;                              // row strides in bytes
        rk = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (Nk - 1) * skn2 + 128, 0x00020000);
        rv = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (Nk - 1) * svn2 + 128, 0x00020000);
        r8 = wave * 8 + (lane >> 3);
        col = ((lane & 7) ^ fswz(r8)) * 16;
        vk = r8 * skn2 + col; vv = r8 * svn2 + col;
    }
    // a FULL tile `t` (t * 64 + 64 <= Nk) into the stage whose first byte (for this wave: + wave * 1024) is `dst`
    __device__ __forceinline__ void issue_full(int t, char* dst) const {
        const int sk = t * 64 * skn2, sv = t * 64 * svn2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)dst, 16, vk, sk, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(dst + 4096), 16, vk, sk + 32 * skn2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + FT_TILE), 16, vv, sv, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + FT_TILE + 4096), 16, vv, sv + 32 * svn2, 0, 0);
    }
    // any tile (the last one may be ragged)
    __device__ __forceinline__ void issue(int t, char* dst) const {
        const int sk = t * 64 * skn2, sv = t * 64 * svn2;
        if (t * 64 + 64 <= Nk) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)dst, 16, vk, sk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(dst + 4096), 16, vk, sk + 32 * skn2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + FT_TILE), 16, vv, sv, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + FT_TILE + 4096), 16, vv, sv + 32 * svn2, 0, 0);
        } else {
            // ragged last tile: rows past Nk - 1 are clamped to it.  Offsets computed here, once per kernel; the opaque zero
            // keeps the compiler from hoisting them out of the tile loop and carrying four more registers through it.
            int zero;
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
            const int rl = Nk - 1 - t * 64, ra = min(r8 + zero, rl), rb = min(r8 + 32 + zero, rl);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)dst, 16, ra * skn2 + col, sk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(dst + 4096), 16, rb * skn2 + col, sk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + FT_TILE), 16, ra * svn2 + col, sv, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + FT_TILE + 4096), 16, rb * svn2 + col, sv, 0, 0);
        }
    }
};
// buffer descriptors address rows with 32-bit byte offsets
inline bool kvdma_ok(int64_t rows, int64_t ld) { return rows * ld < (1 << 29); }

// scale * log2(e) = p2 * rr with p2 a power of two and rr in [1, 2): an operand takes p2 (exact in bf16), the exponent
// argument the rest (multiplying by the whole factor would round the operand a second time).  Computed on the host in
// double precision and snapped: a caller that pre-multiplies its operands by scale * log2(e) (in the fp32 epilogue of the
// producing GEMM: ONE rounding) passes scale = ln 2, for which rr must come out as exactly 1.
inline void host_split_scale(float scale, float& p2, float& rr) {
    const double c = (double)scale * 1.4426950408889634074;
    int e;
    double m = std::frexp(c, &e);                      // c = m * 2^e, m in [0.5, 1)
    m *= 2.0; e -= 1;                                  // m in [1, 2)
    if (m - 1.0 < 1e-6) m = 1.0;
    if (2.0 - m < 2e-6) { m = 1.0; e += 1; }
    p2 = (float)std::ldexp(1.0, e);
    rr = (float)m;
}
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 v, float p2) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)((float)v[e] * p2);
    return r;
}

// attention_fwd3.hip / attention_bwd3.hip
int launch_fwd3_bf16(const AttnParams& p, hipStream_t st);
int launch_dq3_bf16(const AttnParams& p, hipStream_t st);

}  // namespace gfattn
