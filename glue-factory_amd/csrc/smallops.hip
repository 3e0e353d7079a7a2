// Small per-step utility kernels that replace swarms of tiny stock-torch launches in the train step
// (tools/probe/op_profile.py: 486 aten launches, 4.6 ms of the 55 ms LightGlue step before this file):
//   gf_multi_cast_transpose  ONE launch per step converts every fp32 master parameter into the compute dtype AND writes
//                            the transposed copy W^T of every matrix (the "weight" of the input-gradient GEMM
//                            dx = dy W) -- instead of one multi-tensor cast plus ~107 transposing copies per step;
//                            derived weights (a projection's rows gathered / scaled / stacked from several parameters,
//                            e.g. LightGlue's Wqkv in kernel order with the softmax scale folded into its q rows) are
//                            entries of the same table; gf_weight_grad_map sends their gradient back to the sources;
//   gf_colsum_f32            deterministic column sums of an [R, C] fp32 matrix of per-block partials (LayerNorm
//                            gamma / beta gradients: 36 reductions of [2048, 512] per step, 16 us each in torch);
//   gf_small_dw              dW[o, k] = sum_m dy[m, o] x[m, k] for a tall dy [M, O] and a FEW input columns (K <= 8):
//                            the gradient of the Fourier positional encoding's Wr (lightglue.py:52-65; M = 131072,
//                            O = 32, K = 2), which the library ran as a 0.4 ms skinny GEMM.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

struct CastEntry {            // one parameter (or one row block of a DERIVED weight): src fp32 [src rows, cols] row-major
    const float* src;
    void* dst;                // compute-dtype copy [rows, cols] (may be NULL when only the transpose is wanted)
    void* dst_t;              // transposed copy: element (r, c) at dst_t[c * ldt + r] (NULL for vectors)
    int rows, cols;
    int tile0;                // first 32 x 32 tile of this tensor in the launch
    int tiles_x;              // tiles per row of tiles (cols direction)
    // derived weights (the matcher's prepared projections: row gather, per-row / scalar scale, row blocks of one output)
    const int* perm;          // dst row r reads src row perm[r] (NULL: r)
    const float* rscale;      // dst row r is scaled by rscale[r] (NULL: 1)
    float scale;              // ... and by this scalar, all in fp32 before the single rounding
    int ldt;                  // leading dimension of dst_t (rows of the WHOLE output when this is one block of it)
    int flags;                // bit 0: dst is fp32 whatever the launch's dtype (biases stay fp32 for the GEMM epilogues)
    int ldd;                  // leading dimension of dst (0: cols) -- a COLUMN block of a wider output (the folded FFN weight
                              // [W0a | W0b Wo], csrc/fold.hip) has ldd = the whole output's width
    const int* cperm;         // dst column c reads src column cperm[c] (NULL: c)
    int lds;                  // leading dimension of src (0: cols) -- the left columns of a wider parameter
    int pad_;
};
static_assert(sizeof(CastEntry) == 88, "host table layout (ops.precast)");

template <typename T>
__global__ __launch_bounds__(256) void multi_cast_transpose_kernel(const CastEntry* __restrict__ tab, int n) {
    __shared__ float tile[32][33];
    // binary search: last entry with tile0 <= blockIdx.x
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const CastEntry e = tab[lo];
    const int t = blockIdx.x - e.tile0;
    const int r0 = (t / e.tiles_x) * 32, c0 = (t % e.tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8 threads
    T* dst = static_cast<T*>(e.dst);
    float* dst32 = static_cast<float*>(e.dst);
    const int lds = e.lds ? e.lds : e.cols, ldd = e.ldd ? e.ldd : e.cols;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        float v = 0.f;
        if (r < e.rows && c < e.cols) {
            const int sr = e.perm ? e.perm[r] : r, sc = e.cperm ? e.cperm[c] : c;
            v = e.src[(size_t)sr * lds + sc] * e.scale;
            if (e.rscale) v *= e.rscale[r];
            if (e.dst) {
                if (e.flags & 1) dst32[(size_t)r * ldd + c] = v;
                else dst[(size_t)r * ldd + c] = from_f32<T>(v);
            }
        }
        tile[ty + 8 * i][tx] = v;
    }
    if (!e.dst_t) return;
    __syncthreads();
    T* dt = static_cast<T*>(e.dst_t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < e.rows && c < e.cols) dt[(size_t)c * e.ldt + r] = from_f32<T>(tile[tx][ty + 8 * i]);
    }
}

// gradient of a derived-weight row block back to its source parameter: out[perm[r]][cperm[c]] = g[r][c] * rscale[r] * scale
// (perm / cperm bijections onto the source rows / columns: a scatter without accumulation)
__global__ __launch_bounds__(256) void weight_grad_map_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                              const int* __restrict__ perm, const int* __restrict__ cperm,
                                                              const float* __restrict__ rscale, float scale, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
        float v = g[i] * scale;
        if (rscale) v *= rscale[r];
        out[(size_t)(perm ? perm[r] : r) * cols + (cperm ? cperm[c] : c)] = v;
    }
}

constexpr int CS_CHUNKS = 32;
// stage 1: block (column group of 64, row chunk) -> part2[chunk][c]; stage 2: sum of the CS_CHUNKS rows
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ x, float* __restrict__ part2, int R, int C) {
    __shared__ float sm[4][64];
    x += (size_t)blockIdx.z * R * C;
    part2 += (size_t)blockIdx.z * CS_CHUNKS * C;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int per = (R + CS_CHUNKS - 1) / CS_CHUNKS, r0 = blockIdx.y * per, r1 = min(R, r0 + per);
    float s = 0.f;
    if (c < C) {
        int r = r0 + w;
        for (; r + 28 < r1; r += 32) {            // eight rows in flight (a loop of single loads is a chain of round trips)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(r + 4 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < r1; r += 4) s += x[(size_t)r * C + c];
    }
    sm[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) part2[(size_t)blockIdx.y * C + c] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
__global__ void colsum_stage2(const float* __restrict__ part2, float* __restrict__ out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    part2 += (size_t)blockIdx.y * CS_CHUNKS * C;
    out += (size_t)blockIdx.y * C;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CS_CHUNKS; ++k) s += part2[(size_t)k * C + c];
    out[c] = s;
}

constexpr int SDW_BLOCKS = 256;
// thread = (row slot g, output o); K <= 8 columns of x per row; partial sums per block, then colsum_stage-like finish
template <int K>
__global__ __launch_bounds__(256) void small_dw_stage1(const float* __restrict__ dy, const float* __restrict__ x,
                                                       float* __restrict__ part, int M, int O) {
    extern __shared__ float sm[];                       // [G][O * K]
    const int G = 256 / O, g = threadIdx.x / O, o = threadIdx.x % O;
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    if (g < G) {
        for (int m = blockIdx.x * G + g; m < M; m += SDW_BLOCKS * G) {
            const float d = dy[(size_t)m * O + o];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += d * x[(size_t)m * K + k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) sm[(g * O + o) * K + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < O * K) {
        float s = 0.f;
        for (int gg = 0; gg < G; ++gg) s += sm[gg * O * K + threadIdx.x];
        part[(size_t)blockIdx.x * O * K + threadIdx.x] = s;
    }
}
// y[m, o] = sum_k x[m, k] w[o, k]  (K <= 8): the forward of the same input linears -- one thread per output, w in registers
template <int K>
__global__ __launch_bounds__(256) void small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, size_t total, int O) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / O;
        const int o = (int)(i - m * O);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(x[m * K + k], w[o * K + k], acc);
        y[i] = acc;
    }
}
__global__ void small_dw_stage2(const float* __restrict__ part, float* __restrict__ dw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < SDW_BLOCKS; ++b) s += part[(size_t)b * n + i];
    dw[i] = s;
}

}  // namespace

extern "C" int gf_multi_cast_transpose(const void* table, int n_entries, int total_tiles, int dtype, void* stream) {
    if (n_entries <= 0 || total_tiles <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const CastEntry* tab = static_cast<const CastEntry*>(table);
    if (dtype == GF_BF16) multi_cast_transpose_kernel<bf16_t><<<dim3(total_tiles), dim3(256), 0, st>>>(tab, n_entries);
    else if (dtype == GF_F32) multi_cast_transpose_kernel<float><<<dim3(total_tiles), dim3(256), 0, st>>>(tab, n_entries);
    else return GF_ERR_DTYPE;
    return (int)hipGetLastError();
}
extern "C" int gf_cast_entry_bytes(void) { return (int)sizeof(CastEntry); }

extern "C" int gf_weight_grad_map(const float* g, float* out, const int* perm, const int* cperm, const float* rscale,
                                  float scale, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return GF_ERR_SHAPE;
    const size_t total = (size_t)rows * cols;
    const int nb = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    weight_grad_map_kernel<<<dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(g, out, perm, cperm, rscale, scale, rows, cols);
    return (int)hipGetLastError();
}

extern "C" int gf_colsum_f32(const float* x, float* ws, float* out, int G, int R, int C, void* stream) {
    if (G <= 0 || R <= 0 || C <= 0) return GF_ERR_SHAPE;
    if (G > 65535) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    colsum_stage1<<<dim3((C + 63) / 64, CS_CHUNKS, G), dim3(256), 0, st>>>(x, ws, R, C);
    colsum_stage2<<<dim3((C + 255) / 256, G), dim3(256), 0, st>>>(ws, out, C);
    return (int)hipGetLastError();
}
extern "C" int gf_colsum_ws_floats(int G, int C) { return G * CS_CHUNKS * C; }

extern "C" int gf_small_dw(const float* dy, const float* x, float* ws, float* dw, int M, int O, int K, void* stream) {
    if (M <= 0 || O <= 0 || K <= 0) return GF_ERR_SHAPE;
    if (O > 256 || K > 8 || O * K > 256) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)(256 / O) * O * K * sizeof(float);
#define GF_SDW(K_) case K_: small_dw_stage1<K_><<<dim3(SDW_BLOCKS), dim3(256), lds, st>>>(dy, x, ws, M, O); break;
    switch (K) {
        GF_SDW(1) GF_SDW(2) GF_SDW(3) GF_SDW(4) GF_SDW(5) GF_SDW(6) GF_SDW(7)
        default: small_dw_stage1<8><<<dim3(SDW_BLOCKS), dim3(256), lds, st>>>(dy, x, ws, M, O); break;
    }
#undef GF_SDW
    small_dw_stage2<<<dim3((O * K + 255) / 256), dim3(256), 0, st>>>(ws, dw, O * K);
    return (int)hipGetLastError();
}
extern "C" int gf_small_dw_ws_floats(int O, int K) { return SDW_BLOCKS * O * K; }

extern "C" int gf_small_fwd(const float* x, const float* w, float* y, int M, int O, int K, void* stream) {
    if (M <= 0 || O <= 0 || K <= 0) return GF_ERR_SHAPE;
    if (K > 8) return GF_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)M * O;
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
#define GF_SFW(K_) case K_: small_fwd_kernel<K_><<<dim3(nb), dim3(256), 0, st>>>(x, w, y, total, O); break;
    switch (K) {
        GF_SFW(1) GF_SFW(2) GF_SFW(3) GF_SFW(4) GF_SFW(5) GF_SFW(6) GF_SFW(7)
        default: small_fwd_kernel<8><<<dim3(nb), dim3(256), 0, st>>>(x, w, y, total, O); break;
    }
#undef GF_SFW
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// gf_multi_adam: the Adam update of EVERY parameter tensor of a model in a handful of launches (train.py:513
// optimizer.step()).  torch's fused Adam walks its tensor lists in 7 launches of <= 320 blocks for the 12 M parameters of
// LightGlue: 0.5 ms for 336 MB of traffic; here the tensor table travels BY VALUE in the kernel arguments (<= 80 tensors
// per launch: no device table to refresh when the gradients move, nothing but kernel nodes in a captured step) and every
// 4096-element chunk is a block: the update runs at the HBM rate.
// Semantics = torch.optim.Adam (amsgrad = maximize = False), entry by entry:
//     g' = g / grad_scale + wd p;  m = m + (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;
//     p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps),    t = *step + 1
// `found_inf` (device float, may be NULL) > 0 skips the whole update (the GradScaler protocol train_step.py uses for its
// non-finite-loss / gradient skip); adam_step_kernel then leaves *step alone.  lr / step are DEVICE scalars, so a
// captured graph follows a scheduler and its own step count.
namespace {

struct AdamEntry {
    float* p; const float* g; float* m; float* v;
    long long n;
    int block0, pad_;
};
static_assert(sizeof(AdamEntry) == 48, "host table layout (optim.FusedAdam)");
constexpr int ADAM_CHUNK = 4096, ADAM_MAXT = 80;                // 80 x 48 B = 3840 B of kernel arguments
struct AdamTable { AdamEntry e[ADAM_MAXT]; };

__global__ __launch_bounds__(256) void multi_adam_kernel(const AdamTable tab, int n_entries,
                                                         const float* __restrict__ lr_p, const float* __restrict__ step_p,
                                                         const float* __restrict__ found_inf, const float* __restrict__ grad_scale,
                                                         double b1d, double b2d, float eps, float wd) {
    if (found_inf != nullptr && *found_inf > 0.f) return;
    // beta, 1 - beta: rounded to fp32 from the DOUBLE values, each on its own (1.f - 0.999f is 1.3e-5 away from 0.001f)
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab.e[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamEntry e = tab.e[lo];
    const float t = *step_p + 1.f;
    // bias corrections in DOUBLE from the double betas (torch.optim.Adam keeps them as python floats): 1 - 0.999^t in fp32
    // loses 5 digits to cancellation for small t -- and 0.999f itself is 1.3e-5 off in (1 - beta2) -- which showed as a
    // 1e-5 relative error of the first updates after a resume
    const float bc1 = (float)(-expm1((double)t * log(b1d))), bc2 = (float)(-expm1((double)t * log(b2d)));
    const float step_size = *lr_p / bc1, rsq_bc2 = 1.f / sqrtf(bc2);
    const float inv_scale = grad_scale != nullptr ? 1.f / *grad_scale : 1.f;
    const long long base = (long long)(blockIdx.x - e.block0) * ADAM_CHUNK;
    const bool vec = (((size_t)e.p | (size_t)e.g | (size_t)e.m | (size_t)e.v) & 15) == 0;
    auto upd = [&](float& p, float g, float& m, float& v) {
        g = g * inv_scale + wd * p;
        m = m + omb1 * (g - m);
        v = b2 * v + omb2 * g * g;
        p -= step_size * m / (sqrtf(v) * rsq_bc2 + eps);
    };
#pragma unroll
    for (int it = 0; it < ADAM_CHUNK / 1024; ++it) {
        const long long i = base + it * 1024 + threadIdx.x * 4;
        if (i >= e.n) break;
        if (vec && i + 4 <= e.n) {
            f32x4 p = *reinterpret_cast<const f32x4*>(e.p + i), g = *reinterpret_cast<const f32x4*>(e.g + i);
            f32x4 m = *reinterpret_cast<const f32x4*>(e.m + i), v = *reinterpret_cast<const f32x4*>(e.v + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                upd(pk, g[k], mk, vk);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
            *reinterpret_cast<f32x4*>(e.p + i) = p;
            *reinterpret_cast<f32x4*>(e.m + i) = m;
            *reinterpret_cast<f32x4*>(e.v + i) = v;
        } else {
            for (long long j = i; j < min(i + 4, e.n); ++j) {
                float p = e.p[j], m = e.m[j], v = e.v[j];
                upd(p, e.g[j], m, v);
                e.p[j] = p; e.m[j] = m; e.v[j] = v;
            }
        }
    }
}
__global__ void adam_step_kernel(float* step, const float* found_inf) {
    if (found_inf == nullptr || !(*found_inf > 0.f)) *step += 1.f;
}

}  // namespace

extern "C" int gf_adam_entry_bytes(void) { return (int)sizeof(AdamEntry); }
// `table`: HOST array of n_entries records {p, g, m, v, n, (block0, pad: ignored on entry)}
extern "C" int gf_multi_adam(const void* table, int n_entries, const float* lr, float* step, const float* found_inf,
                             const float* grad_scale, double beta1, double beta2, float eps, float weight_decay, void* stream) {
    if (n_entries <= 0) return GF_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const AdamEntry* src = static_cast<const AdamEntry*>(table);
    for (int i0 = 0; i0 < n_entries; i0 += ADAM_MAXT) {
        AdamTable tab;
        const int n = n_entries - i0 < ADAM_MAXT ? n_entries - i0 : ADAM_MAXT;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            tab.e[i] = src[i0 + i];
            if (tab.e[i].n <= 0) return GF_ERR_SHAPE;
            tab.e[i].block0 = blocks;
            blocks += (int)((tab.e[i].n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        multi_adam_kernel<<<dim3(blocks), dim3(256), 0, st>>>(tab, n, lr, step, found_inf, grad_scale, beta1, beta2, eps, weight_decay);
    }
    adam_step_kernel<<<dim3(1), dim3(1), 0, st>>>(step, found_inf);          // (after every block read the old count: stream order)
    return (int)hipGetLastError();
}
