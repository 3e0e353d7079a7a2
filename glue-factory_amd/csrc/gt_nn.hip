// Ground-truth nearest-neighbour search under a homography, fused (no B x M x N tensors).
//
// Replaces the dense part of gluefactory/geometry/gt_generation.py:120-150
// (gt_matches_from_homography): with a = kp0 warped into image 1, b = kp1 warped into image 0,
//     dist0(i,j) = |a_i - kp1_j|^2,  dist1(i,j) = |kp0_i - b_j|^2,  dist = max(dist0, dist1),
// the reference materialises three fp32 [B,M,N] tensors (+ bool masks) to take row / column
// arg-mins.  Here one thread owns one point, streams the other image's points through LDS and
// keeps (min dist, arg-min, min of its one-sided distance) in registers; called once per
// direction.  Ties resolve to the lowest index (what torch's CPU min returns).
#include "gf_common.h"
#include "gf_amd.h"

namespace {

// own point i: (p_i = own coords, q_i = own warped coords); other point j: (r_j = other coords,
// s_j = other warped coords).  d_own(i,j) = |q_i - r_j|^2 (own warped vs other), d_oth = |p_i - s_j|^2.
__global__ __launch_bounds__(256) void gt_nn_kernel(const float* __restrict__ own, const float* __restrict__ own_w,
                                                    const float* __restrict__ oth, const float* __restrict__ oth_w,
                                                    int64_t* __restrict__ arg, float* __restrict__ dmin,
                                                    float* __restrict__ own_min, int No, int Ns) {
    __shared__ float sr[256 * 2], ss[256 * 2];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = min(i, No - 1);
    const float px = own[((int64_t)b * No + ic) * 2], py = own[((int64_t)b * No + ic) * 2 + 1];
    const float qx = own_w[((int64_t)b * No + ic) * 2], qy = own_w[((int64_t)b * No + ic) * 2 + 1];
    float best = INFINITY, bown = INFINITY;
    int barg = 0;
    for (int j0 = 0; j0 < Ns; j0 += 256) {
        __syncthreads();
        const int j = min(j0 + (int)threadIdx.x, Ns - 1);
        sr[threadIdx.x * 2] = oth[((int64_t)b * Ns + j) * 2];
        sr[threadIdx.x * 2 + 1] = oth[((int64_t)b * Ns + j) * 2 + 1];
        ss[threadIdx.x * 2] = oth_w[((int64_t)b * Ns + j) * 2];
        ss[threadIdx.x * 2 + 1] = oth_w[((int64_t)b * Ns + j) * 2 + 1];
        __syncthreads();
        const int nj = min(256, Ns - j0);
        for (int t = 0; t < nj; ++t) {
            const float dx0 = qx - sr[2 * t], dy0 = qy - sr[2 * t + 1];
            const float dx1 = px - ss[2 * t], dy1 = py - ss[2 * t + 1];
            const float d_own = dx0 * dx0 + dy0 * dy0;
            const float d_oth = dx1 * dx1 + dy1 * dy1;
            const float d = fmaxf(d_own, d_oth);
            if (d < best) { best = d; barg = j0 + t; }
            bown = fminf(bown, d_own);
        }
    }
    if (i < No) {
        arg[(int64_t)b * No + i] = barg;
        dmin[(int64_t)b * No + i] = best;
        own_min[(int64_t)b * No + i] = bown;
    }
}

}  // namespace

extern "C" int gf_gt_nn(const float* own, const float* own_warped, const float* oth, const float* oth_warped,
                        int64_t* arg, float* dmin, float* own_min, int B, int No, int Ns, void* stream) {
    if (B <= 0 || No <= 0 || Ns <= 0) return GF_ERR_SHAPE;
    gt_nn_kernel<<<dim3((No + 255) / 256, B), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        own, own_warped, oth, oth_warped, arg, dmin, own_min, No, Ns);
    return (int)hipGetLastError();
}
