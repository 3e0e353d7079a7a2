#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned short* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int lane = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = lane * 8;                                  // linear: lane -> 4 consecutive elems
    else if (pattern == 1) addr = (lane & 15) * 128 + (lane >> 4) * 8;  // 16 rows (stride 64 elems), 4 col groups
    else addr = (lane & 15) * 32 + (lane >> 4) * 512;                   // 16 rows stride 16 elems, blocks of 256 elems
    unsigned base = (unsigned)(size_t)lds;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    out[lane * 4 + 0] = v[0] & 0xffff; out[lane * 4 + 1] = v[0] >> 16;
    out[lane * 4 + 2] = v[1] & 0xffff; out[lane * 4 + 3] = v[1] >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    std::vector<unsigned short> h(256);
    for (int p = 0; p < 3; ++p) {
        probe<<<1, 64>>>(d, p); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
