// TEST-ONLY shared object (tests/libgf_test_probe.so; built by __graft_entry__.build() and tests/conftest.py, never linked into or
// loaded by the product library): a kernel that occupies compute units for a given time -- the stand-in for "another
// stream's kernel holds part of the chip" (an RCCL reduction, a second process) under which the chip-resident Sinkhorn's
// bounded waits are tested (tests/test_gpu_sinkhorn_safety.py).
#include <hip/hip_runtime.h>

namespace {
// A workgroup that claims (nearly) all of a CU's LDS -- so no other LDS-heavy workgroup shares the CU -- and spins on the
// constant-rate wall clock until `ticks` have passed.
__global__ __launch_bounds__(256) void hold_cus_kernel(long long ticks, unsigned* sink) {
    extern __shared__ unsigned held[];
    const long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < 16000000u) {      // (the count bounds the spin should the clock ever stand still)
        __builtin_amdgcn_s_sleep(64);
        ++n;
    }
    held[threadIdx.x] = n;
    if (sink != nullptr && threadIdx.x == 0) sink[blockIdx.x] = held[0];
}
}  // namespace

// occupies `n_cus` compute units (one 150 KB-LDS workgroup each) for `milliseconds` (<= 5000) on `stream`; 0 or a hipError_t
extern "C" int gf_test_hold_cus(int n_cus, int milliseconds, void* stream) {
    if (n_cus < 1 || n_cus > 1024 || milliseconds < 0 || milliseconds > 5000) return -1;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz < 1) khz = 100000;
    const int lds = 150 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hold_cus_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hold_cus_kernel<<<dim3((unsigned)n_cus), 256, lds, reinterpret_cast<hipStream_t>(stream)>>>(
        (long long)milliseconds * khz, nullptr);
    return (int)hipGetLastError();
}


namespace {
// fills a CU's whole LDS allocation (one 159 KB workgroup per CU) with a bit pattern and leaves: the next kernel on that CU finds it
// there -- LDS is not cleared between kernels, so a kernel that reads shared memory it never wrote picks the pattern up
__global__ __launch_bounds__(256) void dirty_lds_kernel(unsigned pattern, int words, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
    __syncthreads();
    if (sink != nullptr && threadIdx.x == 0) sink[blockIdx.x] = lds[words - 1];
}
}  // namespace

// every CU's LDS <- pattern (n_wg workgroups of 159 KB: one per CU when n_wg = the CU count, idle device)
extern "C" int gf_test_dirty_lds(int n_wg, unsigned pattern, void* stream) {
    if (n_wg < 1 || n_wg > 4096) return -1;
    const int lds = 159 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dirty_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    dirty_lds_kernel<<<dim3((unsigned)n_wg), 256, lds, reinterpret_cast<hipStream_t>(stream)>>>(pattern, lds / 4, nullptr);
    return (int)hipGetLastError();
}
