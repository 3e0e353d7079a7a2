// gf_topk_candidates: the K best entries of each image's NMS candidate list, sorted by score (the `torch.topk(..., sorted=
// True)` + index gather of superpoint_open.py:165-170 on the lists gf_nms_candidates wrote).
//
// Why an own kernel for a library-shaped op: a captured graph of the extractor tail that contains torch.topk (ROCm 7.2,
// torch 2.10) faults on its SECOND replay -- bisected with tools/probe/capture_scope_p.py: `tail_nms` replays cleanly,
// `tail_topk` ("Memory access fault by GPU node") does not; still reproducible with the shipped library.  torch.topk is the
// only op of the step that adds non-kernel nodes (four hipMemsetAsync on temporaries), but memset nodes alone do not
// reproduce it (tools/probe/repro/): the defect sits below this code and is layout dependent (DESIGN.md section 4,
// "hipGraph").  This kernel uses no memset, no global scratch and no atomics on global memory: ONE workgroup per image,
//   1. radix select of the K-th largest key (4 passes of 8 bits, LDS histograms) over the non-negative scores -- negative
//      scores are the lists' "unfilled slot" marker and rank below everything, in list order;
//   2. ordered compaction (block scans in list order, so ties at the threshold are taken lowest position first:
//      deterministic) of the selected entries into LDS;
//   3. bitonic sort of the <= 4096 survivors by (score descending, list position ascending);
//   4. scores and the int64 payload (the flat pixel index of the candidate) written in that order.
#include "gf_common.h"
#include "gf_amd.h"

namespace {

constexpr int TK_THREADS = 1024, TK_WAVES = TK_THREADS / 64, TK_MAXK = 4096;

__device__ __forceinline__ unsigned tk_key(float s) {          // monotone map of the NON-NEGATIVE floats; negatives -> 0
    return s >= 0.f ? (__float_as_uint(s) | 0x80000000u) : 0u;
}

// exclusive scan of one flag per thread over the block (wave ballots + per-wave totals in LDS); returns (offset, total)
__device__ __forceinline__ int block_scan_flag(bool flag, int* wsum, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TK_WAVES; ++w) {
        const int c = wsum[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return base + within;
}

__global__ __launch_bounds__(TK_THREADS) void topk_candidates_kernel(const float* __restrict__ scores, const int* __restrict__ payload,
                                                                     float* __restrict__ out_s, int64_t* __restrict__ out_p,
                                                                     int n, int K, int K2) {
    extern __shared__ unsigned long long items[];              // [K2] (key << 32 | ~position)
    __shared__ int hist[256];
    __shared__ int wsum[TK_WAVES];
    __shared__ unsigned sel_prefix;
    __shared__ int sel_need;
    const float* s = scores + (size_t)blockIdx.x * n;
    const int* pl = payload + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;

    // ---- 1. the K-th largest key among the non-negative scores (need = how many of the threshold's ties are taken)
    unsigned prefix = 0, mask = 0;
    int need = K;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += TK_THREADS) {
            const unsigned k = tk_key(s[i]);
            if (k != 0u && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int cum = 0, b = 255;
            for (; b >= 0; --b) {
                if (cum + hist[b] >= need) break;
                cum += hist[b];
            }
            if (b < 0) {                       // fewer than `need` non-negative entries carry this prefix: take them all
                sel_prefix = 0xffffffffu;      // (only possible in the first pass: later passes refine a bin that holds >= need)
                sel_need = need - cum;
            } else {
                sel_prefix = prefix | ((unsigned)b << shift);
                sel_need = need - cum;
            }
        }
        __syncthreads();
        if (sel_prefix == 0xffffffffu) break;
        prefix = sel_prefix;
        need = sel_need;
        mask |= 255u << shift;
        __syncthreads();
    }
    const bool short_list = sel_prefix == 0xffffffffu;          // fewer than K non-negative entries in the list
    const unsigned T = short_list ? 1u : prefix;                // keys >= T are selected (all positives in the short case)
    const int ties_wanted = short_list ? 0 : need;              // entries == T taken (lowest position first)
    const int fills_wanted = short_list ? sel_need : 0;         // unfilled slots appended (list order) to reach K
    __syncthreads();

    // ---- 2. ordered compaction into LDS: [ keys > T | first `ties_wanted` keys == T | first `fills_wanted` negatives ]
    int n_gt = 0, n_eq = 0, n_fill = 0;
    for (int i0 = 0; i0 < n; i0 += TK_THREADS) {
        const int i = i0 + tid;
        const unsigned k = i < n ? tk_key(s[i]) : 0u;
        const bool valid = i < n;
        const bool gt = valid && (short_list ? k != 0u : k > T);
        const bool eq = valid && !short_list && k == T;
        const bool fl = valid && short_list && k == 0u;
        int tot;
        int o = block_scan_flag(gt, wsum, tot);
        const unsigned long long it = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        if (gt) items[n_gt + o] = it;
        n_gt += tot;
        if (!short_list) {
            o = block_scan_flag(eq, wsum, tot);
            if (eq && n_eq + o < ties_wanted) items[K - ties_wanted + n_eq + o] = it;     // (n_gt ends at K - ties_wanted)
            n_eq += tot;
        } else {
            o = block_scan_flag(fl, wsum, tot);
            if (fl && n_fill + o < fills_wanted) items[K - fills_wanted + n_fill + o] = it;
            n_fill += tot;
        }
    }
    for (int i = K + tid; i < K2; i += TK_THREADS) items[i] = 0ull;          // padding sorts last
    __syncthreads();

    // ---- 3. bitonic sort, descending
    for (int size = 2; size <= K2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < K2 / 2; t += TK_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = items[lo], b = items[hi];
                if ((a < b) == desc) { items[lo] = b; items[hi] = a; }
            }
            __syncthreads();
        }
    }

    // ---- 4. outputs
    for (int j = tid; j < K; j += TK_THREADS) {
        const unsigned long long it = items[j];
        const int pos = (int)(0xffffffffu - (unsigned)(it & 0xffffffffu));
        out_s[(size_t)blockIdx.x * K + j] = s[pos];
        out_p[(size_t)blockIdx.x * K + j] = (int64_t)pl[pos];
    }
}

}  // namespace

extern "C" int gf_topk_candidates(const float* scores, const int* payload, float* out_scores, int64_t* out_payload, int B,
                                  int n, int K, void* stream) {
    if (B <= 0 || n <= 0 || K <= 0) return GF_ERR_SHAPE;
    if (K > n || K > TK_MAXK) return GF_ERR_UNSUPPORTED;
    int K2 = 1;
    while (K2 < K) K2 <<= 1;
    topk_candidates_kernel<<<dim3(B), dim3(TK_THREADS), (size_t)K2 * sizeof(unsigned long long),
                             reinterpret_cast<hipStream_t>(stream)>>>(scores, payload, out_scores, out_payload, n, K, K2);
    return (int)hipGetLastError();
}
