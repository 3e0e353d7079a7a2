// Micro-benchmarks behind the attention kernels' structure (gfx950): how fast do v_mfma_f32_32x32x16_bf16 issue
//   (1) as a dependent chain on one accumulator vs round-robin over 2 / 4 accumulators,
//   (2) with k independent VALU instructions (v_fma / v_exp) placed after every MFMA of the same wave,
//   (3) next to a VALU-only wave on the same SIMD (8-wave workgroup: waves w and w+4 share a SIMD),
// at 1 / 2 / 3 waves per SIMD, on random operands.  Prints shader cycles per MFMA per SIMD (s_memtime) and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_valu.hip -o mfma_valu.bin && ./mfma_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: all waves run MFMA (+ NV VALU after each).  MODE 1: waves with (wave & 4) run only the VALU stream (NV per "slot"),
// the others only MFMAs.  VK: 0 = v_fma_f32, 1 = v_exp_f32
template <int NACC, int NV, int VK, int MODE>
__global__ __launch_bounds__(512) void bench(const bf16x8* __restrict__ src, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a = src[threadIdx.x], b = src[512 + threadIdx.x];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i] * 0.01f;
    const float mulc = 0.999f, addc = 1e-3f;
    const bool valu_only = MODE == 1 && (wave & 4);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#define VALU_SLOT()                                                                              \
    _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                              \
        if (VK == 0) v[k & 7] = __builtin_fmaf(v[k & 7], mulc, addc);                            \
        else { v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]); v[k & 7] = __builtin_fmaf(v[k & 7], mulc, addc); } \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 && valu_only) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) { VALU_SLOT() }
        }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                VALU_SLOT()
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][lane & 15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int NACC, int NV, int VK, int MODE>
void run(const char* name, int threads, int blocks_per_cu, const bf16x8* src, float* out, long long* cyc, int iters) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench<NACC, NV, VK, MODE><<<grid, threads>>>(src, out, cyc, iters / 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    bench<NACC, NV, VK, MODE><<<grid, threads>>>(src, out, cyc, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = grid * threads / 64;
    std::vector<long long> h(waves);
    CHECK(hipMemcpy(h.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost));
    double mean = 0, mean_v = 0; int nm = 0, nv = 0;
    const int wpb = threads / 64;
    for (int i = 0; i < waves; ++i) {
        if (MODE == 1 && ((i % wpb) & 4)) { mean_v += (double)h[i]; ++nv; } else { mean += (double)h[i]; ++nm; }
    }
    mean /= nm; if (nv) mean_v /= nv;
    const int wps = threads / 64 * blocks_per_cu / 4;                     // waves per SIMD
    const double mfma_waves_per_simd = MODE == 1 ? wps / 2.0 : wps;
    const double n_mfma_per_simd = (double)iters * NACC * mfma_waves_per_simd;
    const double flops = 2.0 * 32 * 32 * 16 * n_mfma_per_simd * 1024;
    // per MFMA wave: ticks between its first and last instruction / its own MFMA count (x waves sharing the SIMD's pipe)
    const double own = (double)iters * NACC;
    printf("%-44s w/SIMD %d %8.3f ms | MFMA waves: %7.2f ticks per own MFMA = %6.2f per MFMA of the SIMD | %6.2f ns/MFMA/SIMD %7.1f TF",
           name, wps, ms, mean / own, mean / own / mfma_waves_per_simd, ms * 1e6 / n_mfma_per_simd, flops / (ms * 1e-3) * 1e-12);
    if (nv) printf(" | VALU waves: %7.2f ticks per slot of %d", mean_v / own, NV * (VK ? 2 : 1));
    printf("\n");
}

int main() {
    bf16x8* src; float* out; long long* cyc;
    std::vector<unsigned short> h(1024 * 8);
    srand(1);
    for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    CHECK(hipMalloc(&src, h.size() * 2)); CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 256 * 4 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 4 * 8 * 8));
    const int IT = 20000;
    printf("== (1) MFMA only: accumulators in rotation x waves per SIMD\n");
    run<1, 0, 0, 0>("1 acc (dependent chain)", 256, 1, src, out, cyc, IT);
    run<2, 0, 0, 0>("2 acc", 256, 1, src, out, cyc, IT);
    run<4, 0, 0, 0>("4 acc", 256, 1, src, out, cyc, IT);
    run<1, 0, 0, 0>("1 acc (dependent chain)", 256, 2, src, out, cyc, IT);
    run<2, 0, 0, 0>("2 acc", 256, 2, src, out, cyc, IT);
    run<4, 0, 0, 0>("4 acc", 256, 2, src, out, cyc, IT);
    run<1, 0, 0, 0>("1 acc (dependent chain)", 256, 3, src, out, cyc, IT);
    run<2, 0, 0, 0>("2 acc", 256, 3, src, out, cyc, IT);
    printf("== (2) MFMA + k VALU after each, same wave (4 accumulators)\n");
    run<4, 2, 0, 0>("k=2 v_fma", 256, 1, src, out, cyc, IT);
    run<4, 4, 0, 0>("k=4 v_fma", 256, 1, src, out, cyc, IT);
    run<4, 6, 0, 0>("k=6 v_fma", 256, 1, src, out, cyc, IT);
    run<4, 8, 0, 0>("k=8 v_fma", 256, 1, src, out, cyc, IT);
    run<4, 12, 0, 0>("k=12 v_fma", 256, 1, src, out, cyc, IT);
    run<4, 4, 0, 0>("k=4 v_fma", 256, 2, src, out, cyc, IT);
    run<4, 6, 0, 0>("k=6 v_fma", 256, 2, src, out, cyc, IT);
    run<4, 8, 0, 0>("k=8 v_fma", 256, 2, src, out, cyc, IT);
    run<4, 12, 0, 0>("k=12 v_fma", 256, 2, src, out, cyc, IT);
    run<4, 6, 0, 0>("k=6 v_fma", 256, 3, src, out, cyc, IT);
    run<4, 2, 1, 0>("k=2 (v_exp + v_fma pairs: 4 instr)", 256, 1, src, out, cyc, IT);
    run<4, 4, 1, 0>("k=4 (v_exp + v_fma pairs: 8 instr)", 256, 1, src, out, cyc, IT);
    run<4, 2, 1, 0>("k=2 (v_exp + v_fma pairs: 4 instr)", 256, 2, src, out, cyc, IT);
    run<4, 4, 1, 0>("k=4 (v_exp + v_fma pairs: 8 instr)", 256, 2, src, out, cyc, IT);
    printf("== (3) MFMA-only waves next to VALU-only waves on the same SIMD (512-thread workgroups, waves w / w+4)\n");
    run<4, 0, 0, 1>("partner idle-ish (k=0)", 512, 1, src, out, cyc, IT);
    run<4, 4, 0, 1>("partner: 4 v_fma per MFMA slot", 512, 1, src, out, cyc, IT);
    run<4, 8, 0, 1>("partner: 8 v_fma per MFMA slot", 512, 1, src, out, cyc, IT);
    run<4, 16, 0, 1>("partner: 16 v_fma per MFMA slot", 512, 1, src, out, cyc, IT);
    run<4, 4, 1, 1>("partner: 4 (v_exp+v_fma) per MFMA slot", 512, 1, src, out, cyc, IT);
    return 0;
}
