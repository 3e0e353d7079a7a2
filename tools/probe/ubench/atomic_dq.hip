// What would a single-pass self-attention backward pay to sum its partial dQ across workgroups?  (round-4 review item 3)
// A dK/dV-owner workgroup (128 keys) of the 64 img x 4 heads x 2048^2 launch produces a partial dQ for ALL 2048 queries;
// 16 owners per (image, head) have to be summed.  This program issues exactly that traffic and nothing else (no MFMA, no
// softmax): 4096 workgroups x 32 query tiles of 64 x 64 values (already pre-reduced over the 4 waves of the workgroup), as
//   store32   plain fp32 dwordx4 stores into ONE slab                  (the bandwidth floor of moving the bytes once)
//   slab32    plain fp32 dwordx4 stores into 16 slabs + a reduce pass  (candidate iii of DESIGN.md section 4)
//   atom32    global_atomic_add_f32, device scope                      (candidate i)
//   atombf    global_atomic_pk_add_bf16, device scope                  (candidate ii)
// each with the launch order's natural placement (the 16 owners of an (image, head) spread over the 8 XCDs) and with an
// XCD-aware placement (all 16 owners of an (image, head) on one XCD: their atomics meet in one L2).
// Owners start at staggered query tiles so that the 16 never hit the same rows at the same time.
// Compare the printed times with the dQ kernel this would replace: 0.37 ms per launch in the step (attn_dq3_bf16_kernel).
//   hipcc --offload-arch=gfx950 -O3 atomic_dq.hip -o atomic_dq.bin && ./atomic_dq.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int IH = 256, OWN = 16, NQ = 2048, D = 64, TQ = 64, NT = NQ / TQ;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void decode(int bid, int xcd_aware, int& ih, int& owner) {
    if (xcd_aware) {                       // workgroup ids go round-robin over the 8 XCDs: id % 8 is the XCD
        const int xcd = bid & 7, slot = bid >> 3;
        ih = (slot / OWN) * 8 + xcd;
        owner = slot % OWN;
    } else {
        ih = bid / OWN;
        owner = bid % OWN;
    }
}

// MODE 0 store32, 1 slab32, 2 atom32, 3 atombf
template <int MODE>
__global__ __launch_bounds__(256) void partial_dq(float* __restrict__ dq32, unsigned* __restrict__ dqbf, float* __restrict__ slabs,
                                                   int xcd_aware, float val) {
    int ih, owner;
    decode(blockIdx.x, xcd_aware, ih, owner);
    const int l = threadIdx.x;
    const int c4 = (l & 15) * 4, r0 = l >> 4;                 // 16 lanes cover a 64-float row; 16 rows per pass, 4 passes
    for (int t = 0; t < NT; ++t) {
        const int tile = (t + owner * 2) % NT;
        const size_t base = ((size_t)ih * NQ + (size_t)tile * TQ) * D;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const size_t off = base + (size_t)(r0 + 16 * p) * D + c4;
            const float v = val * (float)(1 + ((l + p) & 3));
            if (MODE == 0) {
                *(f32x4*)(dq32 + off) = f32x4{v, v, v, v};
            } else if (MODE == 1) {
                *(f32x4*)(slabs + (size_t)owner * IH * NQ * D + off) = f32x4{v, v, v, v};
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __hip_atomic_fetch_add(dq32 + off + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __bf16 b = (__bf16)v;
                unsigned short h = *(unsigned short*)&b;
                const unsigned pk = (unsigned)h | ((unsigned)h << 16);
                unsigned* a = dqbf + (off >> 1);
                asm volatile("global_atomic_pk_add_bf16 %0, %1, off\n\tglobal_atomic_pk_add_bf16 %0, %1, off offset:4"
                             :: "v"(a), "v"(pk) : "memory");
            }
        }
    }
}

__global__ __launch_bounds__(256) void slab_reduce(const float* __restrict__ slabs, float* __restrict__ dq32) {
    const size_t n4 = (size_t)IH * NQ * D / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 s = ((const f32x4*)slabs)[i];
#pragma unroll
        for (int o = 1; o < OWN; ++o) s += ((const f32x4*)slabs)[(size_t)o * n4 + i];
        ((f32x4*)dq32)[i] = s;
    }
}

template <class F>
static float time_ms(F&& f, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t n = (size_t)IH * NQ * D;
    float *dq32, *slabs;
    unsigned* dqbf;
    CHECK(hipMalloc(&dq32, n * 4));
    CHECK(hipMalloc(&dqbf, n * 2));
    CHECK(hipMalloc(&slabs, n * 4 * OWN));
    CHECK(hipMemset(dq32, 0, n * 4));
    CHECK(hipMemset(dqbf, 0, n * 2));
    const int grid = IH * OWN, reps = 10;
    printf("partial-dQ traffic of a single-pass attention backward: %d workgroups x %d tiles of %dx%d; dQ = %.0f MB fp32\n", grid, NT, TQ, D, n * 4 / 1e6);
    for (int aware = 0; aware < 2; ++aware) {
        const char* tag = aware ? "one XCD per (image, head)" : "launch-order placement  ";
        float t0 = time_ms([&] { partial_dq<0><<<grid, 256>>>(dq32, dqbf, slabs, aware, 0.f); }, reps);
        float t1 = time_ms([&] { partial_dq<1><<<grid, 256>>>(dq32, dqbf, slabs, aware, 0.f); slab_reduce<<<2048, 256>>>(slabs, dq32); }, reps);
        float t1a = time_ms([&] { partial_dq<1><<<grid, 256>>>(dq32, dqbf, slabs, aware, 0.f); }, reps);
        CHECK(hipMemset(dq32, 0, n * 4));
        float t2 = time_ms([&] { partial_dq<2><<<grid, 256>>>(dq32, dqbf, slabs, aware, 1.f / 1024); }, reps);
        float t3 = time_ms([&] { partial_dq<3><<<grid, 256>>>(dq32, dqbf, slabs, aware, 1.f / 1024); }, reps);
        printf("%s  store32 %.3f ms | slab32 %.3f ms (writes %.3f + reduce) | atom32 %.3f ms (%.1f G atomics/s) | atombf %.3f ms (%.1f G/s)\n",
               tag, t0, t1, t1a, t2, n * OWN / t2 / 1e6, t3, n / 2 * OWN / t3 / 1e6);
    }
    // the sums are what they should be (16 owners x 11 launches x val x (1..4), last placement): the atomics did meet across XCDs
    std::vector<float> h(4096);
    CHECK(hipMemcpy(h.data(), dq32 + 12352 * 64, 4096 * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < 4096; ++i) {
        const int r = i / 64, l = (r % 16) * 16 + (i % 64) / 4, p = r / 16;         // the lane / pass that wrote element i of its tile
        const double want = 11.0 * OWN / 1024.0 * (1 + ((l + p) & 3));
        worst = fmax(worst, fabs(h[i] - want) / want);
    }
    printf("fp32 atomic sums: worst relative deviation from 11 launches x 16 owners: %.2e\n", worst);
    return 0;
}
