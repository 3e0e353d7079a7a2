// Chip-resident Sinkhorn sweeps (included into sinkhorn.hip's anonymous namespace).
//
// The streaming fast path (skf_*) reads the whole coupling matrix from HBM once per iteration: 100 iterations
// of superglue.py:186-191 = 100 sweeps of 537 MB at B = 32, N = 2048, and every sweep is two dependent launches.
// Here a chunk of pairs is loaded ONCE and stays on the chip for all T iterations: 256 CUs x (512 KB of vector
// registers + 160 KB of LDS) hold 8 pairs of 2049 x 2049 fp32 (134 MB).  A pair is spread over `wpp` workgroups
// (one per CU, 4 waves, one wave per SIMD with the full 512-register budget); a wave owns `nrows` consecutive whole
// rows: the first SKR_RR of them in registers (statically indexed), the rest in LDS.  Columns follow the fast
// path's layout: lane l holds float4 columns 4 (l + 64 k), k < NSM (N = 256 NSM), plus the dustbin column N as a
// per-row scalar (lane r keeps row r's).  One iteration =
//     row pass over the resident rows (the same shared-exponential update as skf_fwd_iter / skf_bwd_iter)
//     -> every wave writes ONE partial row of column sums (8 KB; 128 per pair instead of the matrix)
//     -> pair barrier -> the pair's workgroups each finish a slice of the columns (fixed summation order)
//     -> pair barrier -> the new column vector(s) are pulled into LDS.
// The only HBM/L2 traffic inside the loop is the partial rows and the column vectors (~1 MB per pair and
// iteration instead of 16.8 MB), so the sweep is bounded by the exponentials, not by memory.
//
// Barriers are per PAIR (its `wpp` workgroups; with 8 pairs per launch and the observed round-robin dispatch a pair's
// workgroups share an XCD -- a speed bonus, never relied upon).  Hand-off protocol (placement independent; per-XCD L2s
// are not coherent and a CU's L1 is never refreshed by another CU's stores): everything one workgroup publishes for
// another -- the partial rows and the finished column vectors -- is written with 16-byte write-through (`sc0 sc1`)
// stores and read with `sc0 sc1` loads; a wave drains its stores (`s_waitcnt vmcnt(0)`) before the workgroup arrives
// on a monotonic agent-scope counter, the pollers use relaxed agent-scope loads.  No cache-wide fence is needed: the
// first version used agent-scope release / acquire fences in every wave (`buffer_wbl2` + `buffer_inv`), 25 us per
// barrier = 3x the row pass; this form costs 2-3 us.  (Also measured: the finished column vectors as self-validating
// tagged granules polled by every reader instead of the second barrier -- slower, 8.35 -> 9.84 ms forward at B = 32:
// 8192 polling lanes per pair cost more than one counter.)
//
// Co-residency and what happens without it.  The grid is <= the CU count and a workgroup fills a CU (LDS + registers), so
// on an idle device all workgroups of a pair run at once.  They need NOT: a workgroup that finds its CU held by another
// stream's kernel (an RCCL reduction, a second process) is simply dispatched later, and the resident ones poll until it
// arrives -- no workgroup waits for one that can never be scheduled, because the kernels that hold CUs do not depend on this
// one (no circular wait; grid <= CU count rules out waiting for a sibling that needs a poller's own CU).  Contention
// therefore costs time, not correctness (tests/test_gpu_sinkhorn_safety.py holds 32 CUs for 30 ms under a launch and
// compares bit for bit).  Every wait is still BOUNDED, in wall-clock time (the 100 MHz s_memrealtime counter; the bound is
// an argument of the call, default 10 s): when it expires the pair is marked failed, its later waits are released, and at
// the end of the kernel every wave of the pair overwrites its rows of the last iterate (forward: u^T, backward: ubar^1)
// with NaN -- the pair's output / gradient is NaN in every row, the loss is NaN, and TrainStep's device-side skip flag
// (train.py:477-480) drops the update.  Never a silently wrong number, never a hung device.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "sinkhorn_resident.h: the same-XCD hand-off is written against gfx950's cache hierarchy (see skr_same_xcd_allowed)"
#endif
constexpr int SKR_RR = 12;                 // rows of a wave that live in registers
constexpr int SKR_MAX_BC = 16;             // pairs per launch (counter slots; the failure flags follow them)
constexpr unsigned SKR_DEFAULT_WAIT_MS = 10000;

struct SkrPlan {
    int bc, wpp, nw, base, extra, cs, nsm;
    size_t lds;
};

struct SkrArgs {
    const float* Zraw;          // forward, round 6: the couplings themselves [bc, R, C] -- scaled by log2(e) while they are loaded,
                                // rows only 4-byte aligned (C = N + 1): no pre-scaled padded copy is made for the resident forward
    float* out;                 // forward, round 6: out = Z + u + v - norm written by the kernel's last iteration (null: not fused)
    const float* Zp;            // backward: [bc, R, Cp] prescaled padded copy
    float* part;                // [bc, nw, Cp] per-wave column partials
    unsigned* ctr;              // [4 SKR_MAX_BC]: barrier counters, failure flags, XCD masks, same-XCD counters; zeroed before the launch
    int safe_only;              // 1: placement-independent (write-through) hand-offs even when a pair sits on one XCD
    long long wait_ticks;       // bound of every wait in wall_clock64() ticks
    float* colA;                // [bc, Cp]  forward: running v (log2 units); backward: a2p
    float* colB;                // [bc, Cp]  backward: vbp
    float* u_hist;              // forward: written; backward: read           (chunk-offset, iteration stride ustride)
    float* v_hist;
    const float* base_row;      // backward: rowsum(G) of the chunk (k == T)
    float* ubar_hist;           // backward: written (index k - 1)
    float* vbar_hist;           // backward: written (index k - 1)
    size_t ustride, vstride;
    int iters;
    SkrPlan d;
    Geo g;
};

__global__ void skr_reset(unsigned* ctr) {
    if (threadIdx.x < 4 * SKR_MAX_BC) ctr[threadIdx.x] = 0u;
}

#ifndef SKR_ABL
#define SKR_ABL 0          // timing probes only (tools/probe/build_sk_variants.sh): 4 no barrier, 8 no row pass,
#endif                     // 16 no column phase
typedef unsigned skr_u32x4 __attribute__((ext_vector_type(4)));
// write-through 16-byte accesses (aux 17 = sc0 sc1): coherent at every placement without cache-wide fences
__device__ __forceinline__ f32x4 skr_ld(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 17));
}
__device__ __forceinline__ void skr_st(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(skr_u32x4, v), r, byte_off, 0, 17);
}
// Same-XCD hand-off.  When EVERY workgroup of a pair runs on one XCD -- established at run time from the hardware's XCC id,
// never assumed (see skr_kernel) -- the pair's readers and writers share one L2: a PLAIN store keeps the published line in
// that L2 (a write-through `sc0 sc1` store drops it: the reader then fetches it from the memory side at the cross-XCD
// rate), the readers' `sc0 sc1` loads bypass their own L1 and are served from it, and the producer's `s_waitcnt vmcnt(0)`
// before it arrives on the counter is the L2's acknowledgement.  Coherent by construction of the part (one L2 per XCD),
// and roughly half the latency of the memory-side round trip in every hop of the hand-off chain.
__device__ __forceinline__ void skr_pub(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v, bool same_xcd) {
    if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(skr_u32x4, v), r, byte_off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(skr_u32x4, v), r, byte_off, 0, 17);
}

// 16 bytes at a 4-byte aligned address (rows of the [R, N + 1] couplings): one global_load / global_store_dwordx4
struct __attribute__((packed, aligned(4))) SkrF4u { float v[4]; };
__device__ __forceinline__ f32x4 skr_ldu(const float* p) {
    const SkrF4u u = *reinterpret_cast<const SkrF4u*>(p);
    return f32x4{u.v[0], u.v[1], u.v[2], u.v[3]};
}
__device__ __forceinline__ void skr_stu(float* p, f32x4 x) {
    SkrF4u u;
    u.v[0] = x[0]; u.v[1] = x[1]; u.v[2] = x[2]; u.v[3] = x[3];
    *reinterpret_cast<SkrF4u*>(p) = u;
}

// L1-bypassing load of one word (the poll of the same-XCD counter: a plain load could be served by this CU's L1 for ever)
__device__ __forceinline__ unsigned skr_peek(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// same_xcd: the pair's own L2 is the meeting point -- the arrival is an atomic on a counter word that only ever sees this
// protocol (ctr + 3 SKR_MAX_BC), polled past the L1
__device__ __forceinline__ void skr_barrier(unsigned* ctr, unsigned target, long long wait_ticks, bool same_xcd = false) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's published stores have left the CU
    __syncthreads();
    if ((SKR_ABL & 4) == 0 && threadIdx.x == 0) {
        unsigned* c = same_xcd ? ctr + 3 * SKR_MAX_BC : ctr;
        // (round 6: the arrival is an AGENT-scope atomic in both protocols -- legal between workgroups under the HSA memory
        // model; it executes in the XCD's L2 like the workgroup-scope form it replaces and measures the same: 5.97 / 7.95 ms
        // against 5.94-5.97 / 7.95-7.96, bit-identical, tools/probe/time_sinkhorn.py.  SKR_SAME_XCD_SCOPE is the A/B knob.)
#ifndef SKR_SAME_XCD_SCOPE
#define SKR_SAME_XCD_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
        if (same_xcd) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, SKR_SAME_XCD_SCOPE);
        else __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        long long t0 = 0;
        while ((same_xcd ? skr_peek(c) : __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++polls & 1023) == 0) {                  // (the clock is read every ~1 ms of polling only)
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > wait_ticks) {
                    // expired: mark the pair failed FIRST (the returned value is waited for), then release this and
                    // every later wait of the pair; whoever leaves a barrier after this sees the flag at the kernel's end
                    const unsigned was = __hip_atomic_fetch_or(ctr + SKR_MAX_BC, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::"v"(was) : "memory");
                    // BOTH counters: should a pair's workgroups ever disagree about the protocol (they cannot after the guard
                    // behind the first barrier, but a release must not depend on that), nobody sits out a second full bound
                    __hip_atomic_fetch_add(ctr, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(ctr + 3 * SKR_MAX_BC, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float skr_lane(float x, int r) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), r));
}

// One resident row, forward:  e = exp2(z + v + ref), rs = sum e, u' = lmu - log2 rs + ref, S += e * 2^SHIFT mu / rs
template <int NSM, bool FIRST>
__device__ __forceinline__ float skr_row_fwd(const f32x4 (&z)[NSM], float zt, const f32x4* __restrict__ vA, float vt,
                                             float uprev, float lmu2, f32x4 (&S)[NSM], float& st, int lane) {
    f32x4 e[NSM];
    float ref;
    if (FIRST) {
        f32x4 m = z[0];
#pragma unroll
        for (int k = 1; k < NSM; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], z[k][c]);
        ref = -fmaxf(wave_allmax(fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]))), zt);
    } else {
        ref = uprev + SKF_SHIFT;
    }
    f32x4 rs4 = splat4(0.f);
#pragma unroll
    for (int k = 0; k < NSM; ++k) {
        f32x4 t = z[k] + splat4(ref);
        if (!FIRST) t += vA[lane + 64 * k];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = fast_exp2(t[c]);
        e[k] = t;
        rs4 += t;
    }
    const float xt = fast_exp2(zt + vt + ref);
    const float rs = fmaxf(wave_allsum((rs4[0] + rs4[1]) + (rs4[2] + rs4[3])) + xt, 1.17549435e-38f);
    const float l2 = fast_log2(rs);
    const float f = fast_exp2(lmu2 - l2 + SKF_SHIFT);
    const f32x4 f4 = splat4(f);
#pragma unroll
    for (int k = 0; k < NSM; ++k) S[k] = __builtin_elementwise_fma(e[k], f4, S[k]);
    st = fmaf(xt, f, st);
    return lmu2 - l2 + ref;
}

// One resident row, backward:  e = exp2(z + a2 + u), ubar = base - sum e vbar, S += e * ubar / mu
template <int NSM>
__device__ __forceinline__ float skr_row_bwd(const f32x4 (&z)[NSM], float zt, const f32x4* __restrict__ vA,
                                             const f32x4* __restrict__ vB, float a2t, float vbt, float u2, float base,
                                             float imu, f32x4 (&S)[NSM], float& st, int lane) {
    f32x4 e[NSM];
    f32x4 acc4 = splat4(0.f);
#pragma unroll
    for (int k = 0; k < NSM; ++k) {
        f32x4 t = z[k] + splat4(u2) + vA[lane + 64 * k];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = fast_exp2(t[c]);
        e[k] = t;
        acc4 = __builtin_elementwise_fma(t, vB[lane + 64 * k], acc4);
    }
    const float xt = fast_exp2(zt + a2t + u2);
    const float acc = wave_allsum((acc4[0] + acc4[1]) + (acc4[2] + acc4[3])) + xt * vbt;
    const float ub = base - acc;
    const float w = ub * imu;
    const f32x4 w4 = splat4(w);
#pragma unroll
    for (int k = 0; k < NSM; ++k) S[k] = __builtin_elementwise_fma(e[k], w4, S[k]);
    st = fmaf(xt, w, st);
    return ub;
}

// grid = wpp * bc workgroups of 256 threads; blockIdx.x % bc = pair, blockIdx.x / bc = workgroup of the pair
template <int NSM, bool BWD>
__global__ __launch_bounds__(256, 1) void skr_kernel(const SkrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int N4 = NSM * 64;                          // float4 per row of the main block (N / 4)
    const Geo& g = a.g;
    f32x4* vA = reinterpret_cast<f32x4*>(smem);           // forward: v; backward: a2
    f32x4* vB = vA + N4;                                  // backward: vbar
    f32x4* red = vA + (BWD ? 2 : 1) * N4;                 // [8][32] column-phase scratch
    float* misc = reinterpret_cast<float*>(red + 256);    // [8]: tail (dustbin column) of vA, vB
    f32x4* zl = reinterpret_cast<f32x4*>(misc + 8);       // LDS-resident rows of the four waves

    const int bc = a.d.bc;
    const int pair = blockIdx.x % bc, wg = blockIdx.x / bc;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = wg * 4 + wave;
    const int nrows = a.d.base + (gw < a.d.extra ? 1 : 0);
    const int row0 = gw * a.d.base + min(gw, a.d.extra);
    int lrow0 = 0;
    for (int w = 0; w < wave; ++w) lrow0 += max(0, a.d.base + (wg * 4 + w < a.d.extra ? 1 : 0) - SKR_RR);
    const int nreg = min(nrows, SKR_RR), nlds = nrows - nreg;
    f32x4* zw = zl + (size_t)lrow0 * N4;
    unsigned* ctr = a.ctr + pair;
    unsigned nbar = 0, nbar2 = 0;                          // barriers passed on the agent-scope / the same-XCD counter
    // which XCD am I on?  every workgroup ORs its XCC id into the pair's mask BEFORE it arrives at the first barrier (which,
    // like everything of the first iteration, uses the placement-independent protocol); whoever leaves that barrier reads the
    // COMPLETE mask, so all workgroups of the pair take the same decision: one bit set = one shared L2 = `same_xcd` hand-offs
    bool same_xcd = false;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(ctr + 2 * SKR_MAX_BC, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- one-time load of the wave's rows.  Forward: straight from the couplings (scaled by log2 e here: no pre-scaled
    // copy, 2 x 537 MB of traffic and a launch per chunk less); backward: from the padded pre-scaled copy.
    // (forward only: measured in the backward too -- 7.91 -> 8.02 ms, the 4-byte aligned 16-byte loads cost more than the copy
    // they replace when there is no fused final pass to pay for them)
    const bool raw = !BWD && a.Zraw != nullptr;
    const size_t zld = raw ? (size_t)g.C : (size_t)g.Cp;
    const float* zb = (raw ? a.Zraw : a.Zp) + ((size_t)pair * g.R + row0) * zld;
    const f32x4 zsc = splat4(raw ? GF_LOG2E : 1.f);
    f32x4 zr[SKR_RR][NSM];
#pragma unroll
    for (int r = 0; r < SKR_RR; ++r)
#pragma unroll
        for (int k = 0; k < NSM; ++k)
            zr[r][k] = r < nreg ? (raw ? skr_ldu(zb + (size_t)r * zld + 4 * (lane + 64 * k))
                                       : *reinterpret_cast<const f32x4*>(zb + (size_t)r * zld + 4 * (lane + 64 * k))) * zsc
                                : splat4(0.f);
    for (int r = 0; r < nlds; ++r)
#pragma unroll
        for (int k = 0; k < NSM; ++k)
            zw[r * N4 + lane + 64 * k] = (raw ? skr_ldu(zb + (size_t)(SKR_RR + r) * zld + 4 * (lane + 64 * k))
                                              : *reinterpret_cast<const f32x4*>(zb + (size_t)(SKR_RR + r) * zld + 4 * (lane + 64 * k))) * zsc;
    const float ztl = lane < nrows ? zb[(size_t)lane * zld + 4 * N4] * (raw ? GF_LOG2E : 1.f) : 0.f;      // dustbin column of row `lane`
    const int gil = row0 + lane;                                                  // the row this lane keeps scalars of
    const float lmu2l = lmu(g, gil) * GF_LOG2E;

    // column phase: thread (grp, ql) sums every 8th partial row of float4 column q; thread ql of group 0 finishes it
    const int ql = tid & 31, grp = tid >> 5;
    const int nvec = N4 + 1;
    const int q = wg * a.d.cs + ql;
    const bool colact = ql < a.d.cs && q < nvec;
    const unsigned rowb = (unsigned)g.Cp * 4u;             // bytes of one partial row / column vector
    const __amdgpu_buffer_rsrc_t rpart = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.part + (size_t)pair * a.d.nw * g.Cp), 0, (int)(a.d.nw * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rcA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.colA + (size_t)pair * g.Cp), 0, (int)rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.colB + (size_t)pair * g.Cp), 0, (int)rowb, 0x00020000);
    const unsigned prow = (unsigned)gw * rowb;

    // the column vector(s) of the next row pass: LDS copy of the finished float4 columns + their dustbin entry
    auto pull_columns = [&]() {
        // (all loads of a thread first, then the LDS writes: one round trip, not one per 256 columns)
        constexpr int PC = (N4 + 255) / 256;
        f32x4 pa[PC], pb[PC];
#pragma unroll
        for (int u = 0; u < PC; ++u) {
            const int i = tid + 256 * u;
            if (i < N4) {
                pa[u] = skr_ld(rcA, 16u * i);
                if (BWD) pb[u] = skr_ld(rcB, 16u * i);
            }
        }
#pragma unroll
        for (int u = 0; u < PC; ++u) {
            const int i = tid + 256 * u;
            if (i < N4) {
                vA[i] = pa[u];
                if (BWD) vB[i] = pb[u];
            }
        }
        if (tid == 0) {
            misc[0] = skr_ld(rcA, 16u * N4)[0];
            if (BWD) misc[1] = skr_ld(rcB, 16u * N4)[0];
        }
        __syncthreads();
    };
    if (BWD) pull_columns();                               // a2p / vbp of k = T come from skf_bwd_prep
    else __syncthreads();

    float ul = 0.f;                                        // forward: u of row `lane` (log2 units)
    f32x4 vkeep = splat4(0.f);                             // forward: this thread's finished columns of v
    // backward: the iterates of the history that an iteration needs -- u^k of the row pass, v^k / v^{k-1} of the column
    // phase -- are plain HBM loads; they are requested one phase ahead (u^{k-1} during iteration k, the v's in front of the
    // row pass), not where they are used: with one wave per SIMD nothing else would hide their latency.
    float u_next = 0.f;
    if (BWD && lane < nrows && a.iters > 0)
        u_next = a.u_hist[(size_t)(a.iters - 1) * a.ustride + (size_t)pair * g.R + gil];
    for (int it = 0; it < a.iters; ++it) {
        const int k = a.iters - it;                        // backward: reverse iteration index T .. 1
        f32x4 S[NSM];
#pragma unroll
        for (int s_ = 0; s_ < NSM; ++s_) S[s_] = splat4(0.f);
        float st = 0.f, outl = 0.f;
        float u2l = 0.f, basel = 0.f;
        f32x4 vp4 = splat4(0.f), vk4 = splat4(0.f);
        if (BWD) {
            u2l = u_next * GF_LOG2E;
            if (lane < nrows) {
                if (k >= 2) u_next = a.u_hist[(size_t)(k - 2) * a.ustride + (size_t)pair * g.R + gil];
                if (it == 0 && a.base_row) basel = a.base_row[(size_t)pair * g.R + gil];
            }
            if (grp == 0 && colact) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * q + c;
                    if (j < g.C) {
                        vk4[c] = a.v_hist[(size_t)(k - 1) * a.vstride + (size_t)pair * g.C + j];
                        if (k >= 2) vp4[c] = a.v_hist[(size_t)(k - 2) * a.vstride + (size_t)pair * g.C + j];
                    }
                }
            }
        }
        const float t0 = misc[0], t1 = BWD ? misc[1] : 0.f;
        const float imul = fast_exp2(-lmu2l);

#define SKR_ROW(Z_, R_, FIRST_)                                                                                       \
        {                                                                                                             \
            const int r_ = (R_);                                                                                      \
            float o_;                                                                                                 \
            if (BWD) o_ = skr_row_bwd<NSM>(Z_, skr_lane(ztl, r_), vA, vB, t0, t1, skr_lane(u2l, r_),                  \
                                           skr_lane(basel, r_), skr_lane(imul, r_), S, st, lane);                     \
            else if (FIRST_) o_ = skr_row_fwd<NSM, true>(Z_, skr_lane(ztl, r_), vA, 0.f, 0.f, skr_lane(lmu2l, r_),    \
                                                         S, st, lane);                                                \
            else o_ = skr_row_fwd<NSM, false>(Z_, skr_lane(ztl, r_), vA, t0, skr_lane(ul, r_), skr_lane(lmu2l, r_),   \
                                              S, st, lane);                                                           \
            if (lane == r_) outl = o_;                                                                                \
        }
#define SKR_ROWS(FIRST_)                                                                                              \
        if (nreg == SKR_RR) {               /* the usual case as ONE basic block: rows interleave freely */          \
            _Pragma("unroll") for (int r = 0; r < SKR_RR; ++r) SKR_ROW(zr[r], r, FIRST_)                              \
        } else {                                                                                                      \
            _Pragma("unroll") for (int r = 0; r < SKR_RR; ++r)                                                        \
                if (r < nreg) SKR_ROW(zr[r], r, FIRST_)                                                               \
        }                                                                                                             \
        for (int r = 0; r < nlds; ++r) {                                                                              \
            f32x4 zz[NSM];                                                                                            \
            _Pragma("unroll") for (int s_ = 0; s_ < NSM; ++s_) zz[s_] = zw[r * N4 + lane + 64 * s_];                  \
            SKR_ROW(zz, SKR_RR + r, FIRST_)                                                                           \
        }
        if (SKR_ABL & 8) {
        } else if (!BWD && it == 0) {
            SKR_ROWS(true)
        } else {
            SKR_ROWS(false)
        }
#undef SKR_ROWS
#undef SKR_ROW

        if (lane < nrows) {
            if (BWD) a.ubar_hist[(size_t)(k - 1) * a.ustride + (size_t)pair * g.R + gil] = outl;
            else a.u_hist[(size_t)it * a.ustride + (size_t)pair * g.R + gil] = outl * GF_LN2;
        }
        ul = outl;
#pragma unroll
        for (int s_ = 0; s_ < NSM; ++s_) skr_pub(rpart, prow + 16u * (lane + 64 * s_), S[s_], same_xcd);
        if (lane == 0) {
            f32x4 tl = {st, 0.f, 0.f, 0.f};
            skr_pub(rpart, prow + 16u * N4, tl, same_xcd);
        }
        if (same_xcd) skr_barrier(ctr, ++nbar2 * (unsigned)a.d.wpp, a.wait_ticks, true);
        else skr_barrier(ctr, ++nbar * (unsigned)a.d.wpp, a.wait_ticks);
        if (it == 0 && !a.safe_only) {
            // (a pair whose FIRST barrier expired left it with an incomplete mask: it is poisoned anyway and stays on the
            // placement-independent protocol, so its workgroups cannot split over the two counters)
            const unsigned mask = __hip_atomic_load(ctr + 2 * SKR_MAX_BC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned failed = __hip_atomic_load(ctr + SKR_MAX_BC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            same_xcd = __builtin_amdgcn_readfirstlane((int)(__builtin_popcount(mask) == 1 && failed == 0u)) != 0;
        }

        // ---- column phase: this workgroup finishes float4 columns [wg cs, wg cs + cs)
        f32x4 acc = splat4(0.f);
        if (colact && !(SKR_ABL & 16)) {
            // eight partial rows in flight per thread (the loop of single loads waited for every one of them: a chain of
            // nw / 8 dependent round trips); summed in the same fixed order
            int w = grp;
            for (; w + 56 < a.d.nw; w += 64) {
                f32x4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = skr_ld(rpart, (unsigned)(w + 8 * u) * rowb + 16u * q);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += x[u];
            }
            for (; w < a.d.nw; w += 8) acc += skr_ld(rpart, (unsigned)w * rowb + 16u * q);
        }
        red[grp * 32 + ql] = acc;
        __syncthreads();
        if (grp == 0 && colact) {
            const f32x4 tot = ((red[ql] + red[32 + ql]) + (red[64 + ql] + red[96 + ql])) +
                              ((red[128 + ql] + red[160 + ql]) + (red[192 + ql] + red[224 + ql]));
            f32x4 oA = splat4(0.f), oB = splat4(0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 4 * q + c;
                if (j < g.C) {
                    if (BWD) {
                        const float vp = vp4[c], vk = vk4[c];
                        const float vbn = -__expf(vp - vk + lnu(g, j)) * tot[c];
                        a.vbar_hist[(size_t)(k - 1) * a.vstride + (size_t)pair * g.C + j] = vbn;
                        oA[c] = (vp - lnu(g, j)) * GF_LOG2E;
                        oB[c] = vbn;
                    } else {
                        const float vn = vkeep[c] + lnu(g, j) * GF_LOG2E - fast_log2(fmaxf(tot[c], 1.17549435e-38f)) + SKF_SHIFT;
                        a.v_hist[(size_t)it * a.vstride + (size_t)pair * g.C + j] = vn * GF_LN2;
                        oA[c] = vn;
                    }
                }
            }
            if (!BWD) vkeep = oA;
            skr_pub(rcA, 16u * q, oA, same_xcd);
            if (BWD) skr_pub(rcB, 16u * q, oB, same_xcd);
        }
        if (same_xcd) skr_barrier(ctr, ++nbar2 * (unsigned)a.d.wpp, a.wait_ticks, true);
        else skr_barrier(ctr, ++nbar * (unsigned)a.d.wpp, a.wait_ticks);

        if (it + 1 < a.iters || (!BWD && a.out != nullptr)) pull_columns();
    }
    // ---- forward, fused final pass: out = Z + u^T + v^T - norm from the resident rows, the wave's own u and the final column
    // vector just pulled (all in log2 units until the last multiply) -- the couplings are not read a second time
    if (!BWD && a.out != nullptr && a.iters > 0) {
        const bool failed = __hip_atomic_load(ctr + SKR_MAX_BC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        const float qnan = __builtin_nanf("");
        float* ob = a.out + ((size_t)pair * g.R + row0) * g.C;
        const float vt = misc[0];
        auto put_row = [&](const f32x4 (&z)[NSM], int r) {
            const float ur = skr_lane(ul, r);
            const f32x4 us = splat4(ur), ln2 = splat4(GF_LN2), nm = splat4(g.norm);
#pragma unroll
            for (int k = 0; k < NSM; ++k) {
                f32x4 o = (z[k] + us + vA[lane + 64 * k]) * ln2 - nm;
                if (failed) o = splat4(qnan);
                skr_stu(ob + (size_t)r * g.C + 4 * (lane + 64 * k), o);
            }
        };
#pragma unroll
        for (int r = 0; r < SKR_RR; ++r)
            if (r < nreg) put_row(zr[r], r);
        for (int r = 0; r < nlds; ++r) {
            f32x4 zz[NSM];
#pragma unroll
            for (int s_ = 0; s_ < NSM; ++s_) zz[s_] = zw[r * N4 + lane + 64 * s_];
            put_row(zz, SKR_RR + r);
        }
        if (lane < nrows) ob[(size_t)lane * g.C + 4 * N4] = failed ? qnan : (ztl + ul + vt) * GF_LN2 - g.norm;
    }
    // a wait of this pair expired (see the header): poison the iterate the final passes read -- forward: u^T of this wave's
    // rows (out = Z + u + v - norm), backward: ubar^1 (a column of the rank-2T factor P of dZ = G - E o (P Q^T))
    if (a.iters > 0 && __hip_atomic_load(ctr + SKR_MAX_BC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u && lane < nrows) {
        const float qnan = __builtin_nanf("");
        if (BWD) a.ubar_hist[(size_t)pair * g.R + gil] = qnan;
        else a.u_hist[(size_t)(a.iters - 1) * a.ustride + (size_t)pair * g.R + gil] = qnan;
    }
}

// `schedule` argument of gf_sinkhorn_fwd / _bwd / _plan, bits 0-1: 0 = streaming kernels only, 1 = resident from SKR_MIN_BC
// pairs per launch, 2 = resident whenever the problem fits (tests: small batches too); bit 2: placement-independent
// (write-through) hand-offs only, i.e. no same-XCD fast path (A/B tests); bits 8-31: bound of every inter-
// workgroup wait in milliseconds (0 = SKR_DEFAULT_WAIT_MS).  Measured on MI355X (tools/probe/time_sinkhorn.py, N = 2048,
// T = 100, forward / backward ms, streaming -> resident): B = 32 10.81 / 12.18 -> 8.35 / 10.66, B = 8 3.05 / 3.45 -> 2.10 / 2.73,
// B = 4 2.25 / 2.51 -> 2.64 / 3.12, B = 1 1.46 / 1.56 -> 5.92 / 6.11 (a pair spread over the whole chip pays a 256-workgroup
// barrier and 1024 partial rows per iteration): few pairs stay on the streaming path.
constexpr int SKR_MIN_BC = 5;

// Distribution of B pairs over the chip, or false when the problem does not fit the resident layout
bool skr_plan(const Geo& g, int B, int ncu, bool bwd, int mode, SkrPlan& d) {
    if (mode == 0 || !g.fast || g.N < 256 || g.N % 256 || g.N / 256 > 8 || g.C != g.N + 1) return false;
    const int n4 = g.N / 4, nvec = n4 + 1;
    const int cap = batch_chunk(g);
    for (int top = B < SKR_MAX_BC ? B : SKR_MAX_BC; top >= 1; --top) {
        const int nch = (B + top - 1) / top, bc = (B + nch - 1) / nch;
        if (bc > cap) continue;
        if (bc < SKR_MIN_BC && mode != 2) continue;
        const int wpp = ncu / bc;
        if (wpp < 1) continue;
        const int nw = 4 * wpp, base = g.R / nw, extra = g.R % nw;
        if (base + (extra ? 1 : 0) > 64) continue;
        const int cs = (nvec + wpp - 1) / wpp;
        if (cs > 32) continue;
        int lrows = 0;
        for (int w = 0; w < 4; ++w) {
            const int n = base + (w < extra ? 1 : 0) - SKR_RR;
            lrows += n > 0 ? n : 0;
        }
        const size_t lds = ((size_t)(bwd ? 2 : 1) * n4 + 256) * 16 + 32 + (size_t)lrows * n4 * 16;
        if (lds > LDS_BUDGET) continue;
        d.bc = bc; d.wpp = wpp; d.nw = nw; d.base = base; d.extra = extra; d.cs = cs; d.nsm = g.N / 256; d.lds = lds;
        return true;
    }
    return false;
}

int skr_cus() {                            // CU count of the CURRENT device (asked per call: one process may drive several)
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) return 0;
    return v;
}

// The same-XCD hand-off leans on how THIS part is built, not on the HSA memory model (workgroups of different CUs meeting
// through plain stores and an L2-resident counter): one L2 per XCD that every CU of the XCD shares, atomics executed in
// that L2, HW_REG_XCC_ID naming the XCD.  It is therefore enabled per architecture, at run time, and nowhere else: gfx950
// (MI350X / MI355X; measured bit-identical to the write-through protocol, tests/test_gpu_sinkhorn_safety.py).  Any other
// device -- and any failure to ask -- takes the placement-independent protocol.
bool skr_same_xcd_allowed() {
    // (an immutable fact about a device, remembered per device ordinal: 0 unknown, 1 yes, 2 no -- not a setting)
    static std::atomic<int> known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int k = known[dev].load(std::memory_order_relaxed);
    if (k == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        const char* a = prop.gcnArchName;
        const bool ok = a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0' &&
                        (a[6] == 0 || a[6] == ':');
        k = ok ? 1 : 2;
        known[dev].store(k, std::memory_order_relaxed);
    }
    return k == 1;
}

long long skr_wait_ticks(int schedule) {   // the call's wait bound in wall_clock64() ticks of the CURRENT device
    const unsigned ms = ((unsigned)schedule >> 8) ? ((unsigned)schedule >> 8) : SKR_DEFAULT_WAIT_MS;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz < 1) khz = 100000;
    return (long long)ms * khz;
}

template <bool BWD> int skr_launch(const SkrArgs& a, hipStream_t st) {
    skr_reset<<<1, 64, 0, st>>>(a.ctr);
    const dim3 grid((unsigned)(a.d.wpp * a.d.bc));
#define SKR_CASE(NSM_)                                                                                                 \
    case NSM_: {                                                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(skr_kernel<NSM_, BWD>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.d.lds);                  \
        if (e != hipSuccess) return (int)e;                                                                            \
        skr_kernel<NSM_, BWD><<<grid, 256, a.d.lds, st>>>(a);                                                          \
        break;                                                                                                         \
    }
    switch (a.d.nsm) {
        SKR_CASE(1) SKR_CASE(2) SKR_CASE(3) SKR_CASE(4) SKR_CASE(5) SKR_CASE(6) SKR_CASE(7) SKR_CASE(8)
        default: return GF_ERR_UNSUPPORTED;
    }
#undef SKR_CASE
    return (int)hipGetLastError();
}
