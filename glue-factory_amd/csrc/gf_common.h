// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of glue_factory_amd.
//
// MFMA convention used by every kernel in this directory ("swapped product"):
//   C[i][j] = sum_k A[i][k] * B[k][j]   with the 32x32 MFMA family, where
//   * a lane (l) supplies A[i = l&31][8 k-elements] and B[8 k-elements][j = l&31]; the
//     8 elements belong to k-group (l>>5).  Both operands are always fetched with the
//     SAME (k-group, element) -> memory-index rule, so the contraction is correct for any
//     hardware ordering of k inside the instruction;
//   * C/D: lane l holds column j = l&31 and rows i = (r&3) + 8*(r>>2) + 4*(l>>5), r=0..15.
//   Reductions over the i axis are therefore in-lane (+ one exchange with lane^32), which is
//   why score tiles are always produced with the softmax axis on i (keys) and the owning
//   row (query / keypoint) on j.
// T = float uses v_mfma_f32_32x32x2_f32 eight times per 16-deep k-step (exact fp32, parity
// mode); T = bf16 uses one v_mfma_f32_32x32x16_bf16 (perf mode).  Accumulation is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GF_LOG2E 1.4426950408889634f
#define GF_LN2 0.6931471805599453f
#define GF_NEG_BIG (-1.0e30f)

enum { GF_F32 = 0, GF_BF16 = 1 };

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

// ---- 16-deep k-step on a 32x32 accumulator ---------------------------------------------
__device__ __forceinline__ void mma32(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[e], b.v[e], acc, 0, 0, 0);
}

// ---- fragment loads: 8 consecutive elements starting at p (16-byte aligned) --------------
__device__ __forceinline__ Frag<bf16_t> ld_frag8(const bf16_t* p) {
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const bf16x8*>(p);
    return f;
}
__device__ __forceinline__ Frag<float> ld_frag8(const float* p) {
    Frag<float> f;
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
    return f;
}
// two groups of 4 consecutive elements (p0, p1), each 8-byte (bf16) / 16-byte (f32) aligned
__device__ __forceinline__ Frag<bf16_t> ld_frag4x2(const bf16_t* p0, const bf16_t* p1) {
    Frag<bf16_t> f;
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
    bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
    return f;
}
__device__ __forceinline__ Frag<float> ld_frag4x2(const float* p0, const float* p1) {
    Frag<float> f;
    f32x4 a = *reinterpret_cast<const f32x4*>(p0);
    f32x4 b = *reinterpret_cast<const f32x4*>(p1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
    return f;
}

// ---- accumulator registers [8t .. 8t+7] of a C tile -> operand fragment ------------------
template <typename T> __device__ __forceinline__ Frag<T> acc_to_frag(const f32x16& c, int t);
template <> __device__ __forceinline__ Frag<float> acc_to_frag<float>(const f32x16& c, int t) {
    Frag<float> f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = c[8 * t + e];
    return f;
}
template <> __device__ __forceinline__ Frag<bf16_t> acc_to_frag<bf16_t>(const f32x16& c, int t) {
    Frag<bf16_t> f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = (bf16_t)c[8 * t + e];
    return f;
}
// C-tile register r of lane-half hi <-> row index inside the 32-row tile
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- scalar conversions -------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// store 4 consecutive values (p 8-byte aligned for bf16 / 16-byte for f32)
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf16x4*>(p) = v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float xhalf(float x) { return __shfl_xor(x, 32); }

// Sum over the 64 lanes, result in every lane, without touching the LDS pipe: a __shfl_xor butterfly
// lowers to six dependent ds_bpermute round trips; DPP moves run at VALU rate.  Steps: quad_perm
// [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror leave the 16-lane row sum in every lane of
// the row; the four row sums meet through v_readlane.
__device__ __forceinline__ float row16_allsum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ float wave_allsum(float v) {
    v = row16_allsum(v);
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}


// Max over the 64 lanes, result in every lane (same DPP ladder as row16_allsum).
__device__ __forceinline__ float row16_allmax(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float wave_allmax(float v) {
    v = row16_allmax(v);
    const int b = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}

// Logical block index such that consecutive logical blocks run on the same XCD (block b is
// dispatched to XCD b % 8): each XCD walks one contiguous chunk of the logical range, so
// blocks sharing K/V (or md1) panels hit the same private L2.  Bijective for any total.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int nx = 8;
    int q = total / nx, r = total % nx;
    int xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Small fp32 fills / copies as KERNELS.  Not hipMemsetAsync / hipMemcpyAsync: inside a captured hipGraph those become
// memset / memcpy nodes, and on ROCm 7.2 such nodes were observed to be dropped from some replays when eager work
// (the extractor) is queued on the same stream between replays (the loss accumulators then held stale pool memory
// while the kernels around them ran correctly).  Kernel nodes are not affected.
namespace {
__global__ void gf_fill_f32_kernel(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void gf_copy_f32_kernel(float* __restrict__ d, const float* __restrict__ s, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}
}  // namespace
static inline hipError_t gf_zero_f32(float* p, size_t n, hipStream_t st) {
    if (n) gf_fill_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(p, 0.f, n);
    return hipGetLastError();
}
static inline hipError_t gf_copy_f32(float* d, const float* s, size_t n, hipStream_t st) {
    if (n) gf_copy_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d, s, n);
    return hipGetLastError();
}

// chunk sizes for 16-byte vector staging
template <typename T> struct Vec16 { static constexpr int N = 16 / sizeof(T); };
