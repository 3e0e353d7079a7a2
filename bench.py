"""Benchmark of the SuperPoint + LightGlue train-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no torchrun environment self-spawns N ranks (one process per GPU, RCCL) through
``python -m torch.distributed.run`` on 127.0.0.1, the way gluefactory/train.py:727-734 spawns its workers;
under torchrun (RANK / WORLD_SIZE set) it just joins.

One "step" (the headline, BASELINE.json configs[1], scope P of SURVEY.md §8d) = frozen SuperPoint forward on a
batch of 2x32 synthetic 1024x1024 images resident in HBM (first block and the three 64-channel blocks as fused HIP
kernels; library convolutions + fused HIP tails for the 128/256-channel blocks) -> homography ground truth
(gf_gt_nn) -> one full LightGlue train step
(forward, loss, backward, fused Adam) at B=32 pairs per GPU, N=2048 keypoints, d=256, L=9, bf16 compute.
Data parallel over N GPUs is weak scaling (32 pairs per GPU) with DDP/RCCL gradient all-reduce.
Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "matcher_step":  the matcher-only scope (M): the same train step on keypoint pairs already resident in HBM
                   (the reference's cached-feature training mode) -- the scope the roofline fractions refer to;
  "roofline":      the dominant MFMA kernel (attention backward) timed live with HIP events on its stream vs
                   the bf16 MFMA peak, "traffic" = HBM bytes per launch from the committed PMC passes;
  "roofline_hbm":  the dominant HBM-bound kernels (assignment write, Sinkhorn iteration) vs the 8 TB/s peak;
  "other_configs": BASELINE.json configs[3] (SuperGlue, 100 Sinkhorn iterations) and configs[4] (GlueStick);
  "cpu_baseline":  the REFERENCE's own LightGlue (oracle/_ref, byte-compiled by oracle/build_ref.py) -- or,
                   when that is absent, the oracle port -- timed on the host cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Library GEMM selection for the remaining library calls (SuperPoint is MIOpen; a few small GEMMs): replay
# the hipBLASLt/rocBLAS solutions tuned once on gfx950 (PyTorch TunableOp, tuning itself disabled -> no timing
# side effects).  TunableOp reads "<name><device ordinal>.csv", so each rank gets a private copy of the table.
_TUNED = os.path.join(ROOT, "glue-factory_amd", "tunableop_gfx950.csv")
if os.path.exists(_TUNED) and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ:
    import shutil
    import tempfile
    _ord = int(os.environ.get("LOCAL_RANK", "0"))
    _dir = os.path.join(tempfile.gettempdir(), f"gf_amd_tunableop_{os.getuid()}")
    os.makedirs(_dir, exist_ok=True)
    shutil.copyfile(_TUNED, os.path.join(_dir, f"table{_ord}.csv"))
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(_dir, "table.csv")

import glue_factory_amd  # noqa: E402,F401  (sets the MIOpen find-db / find mode of the stock convolutions before torch runs one)

import torch  # noqa: E402  (after the library-selection environment is set)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E spec (6.3 TB/s achievable by a float4 copy), same guide
EXP_PEAK_T = 19.7                # transcendental (v_exp_f32) issue rate, T/s: 256 CUs x 4 SIMDs x 8 lanes/clk x 2.4 GHz
N_KPTS, DIM, HEADS, LAYERS, BATCH = 2048, 256, 4, 9, 32
IMG = 1024
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "roofline_traffic.json")


def flops_per_pair_train(n=N_KPTS, d=DIM, L=LAYERS):
    """Algorithmic FLOPs of one matcher train step per pair: 3 x forward."""
    fwd = L * (76 * n * d * d + 14 * n * n * d) + (L + 1) * (4 * n * d * d + 2 * n * n * d)
    return 3 * fwd


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="pairs per GPU")
    ap.add_argument("--kpts", type=int, default=N_KPTS)
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive companion measurement of scope P")
    ap.add_argument("--roofline-only", action="store_true", help="only run the roofline kernels (rocprofv3 target)")
    ap.add_argument("--micro", action="store_true", help="print per-kernel micro timings and exit")
    ap.add_argument("--model", default="lightglue", choices=["lightglue", "superglue", "gluestick"],
                    help="lightglue = the headline; superglue / gluestick: time only that matcher step")
    ap.add_argument("--lines", type=int, default=512, help="gluestick: line segments per image")
    ap.add_argument("--sinkhorn-iters", type=int, default=100)
    ap.add_argument("--no-graph", action="store_true", help="launch the step kernel by kernel instead of replaying a hipGraph")
    ap.add_argument("--matcher-only", action="store_true", help="skip the extractor: scope M becomes the only line")
    ap.add_argument("--dp-graph", action="store_true", help="(accepted for compatibility; N > 1 always measures kernel-by-kernel first and then tries the captured step)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- kernel timing
def time_kernel(fn, iters=10, warm=2):
    """Average seconds per call, HIP events recorded on the stream the launchers use (torch's current stream)."""
    for _ in range(warm):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters * 1e-3


def _traffic_table():
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _traffic(key):
    """Measured HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 (gfx950 half count) + WRITE_SIZE; collected by
    tools/collect_pmc.sh over `bench.py --roofline-only`, summarised into profiles/roofline_traffic.json).  NOT
    measured in this run: `traffic_source` in the JSON line names the collection it comes from."""
    t = _traffic_table()
    if t.get("lib_sha", _sha_of_source(t)) != _lib_sha():        # counters of ANOTHER build say nothing about this one's kernels
        return None
    return t.get(key, {}).get("hbm_bytes_per_launch")


def _sha_of_source(t):
    """Tables written before `lib_sha` existed name the build inside `source` ("... sha1 <12 hex> ...")."""
    src = t.get("source", "")
    i = src.find("sha1 ")
    return src[i + 5:i + 17] if i >= 0 else None


def _lib_sha():
    import hashlib
    try:
        with open(os.path.join(ROOT, "glue-factory_amd", "libgf_amd.so"), "rb") as f:
            return hashlib.sha1(f.read()).hexdigest()[:12]
    except OSError:
        return "missing"


def _traffic_source():
    t = _traffic_table()
    src = t.get("source", "none") if t else "none (profiles/roofline_traffic.json missing)"
    if t and t.get("lib_sha", _sha_of_source(t)) != _lib_sha():
        src = "STALE, `traffic` withheld (null): " + src
    return f"{src}; this run: libgf_amd.so sha1 {_lib_sha()}"


def roofline_attention(batch, n, dtype):
    """Dominant kernel: the self-attention backward (gf_attn_bwd) at the step's own shape (2*batch images, H
    heads, N tokens, hd=64).  Algorithmic FLOPs per launch = 2.5 x forward = 10*N*N*hd per (image, head)."""
    from glue_factory_amd import ops
    B2, H, D = 2 * batch, HEADS, DIM // HEADS
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B2, n, 3, H, D, device="cuda", dtype=dtype, generator=g)
    do = torch.randn(B2, n, H, D, device="cuda", dtype=dtype, generator=g)
    qkv[:, :, 0] *= D ** -0.5 / ops.LN2            # the step's own mode: head_dim^-1/2 * log2(e) folded into the q rows of
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]     # Wqkv (matchers/lightglue.py), the kernels run at scale = ln 2
    o, lse = ops.attn_fwd_raw(q, k, v, ops.LN2)
    dqkv = torch.empty_like(qkv)
    t_fwd = time_kernel(lambda: ops.attn_fwd_raw(q, k, v, ops.LN2, out=o, lse=lse))
    t_bwd = time_kernel(lambda: ops.attn_bwd_raw(q, k, v, o, do, lse, dqkv[:, :, 0], dqkv[:, :, 1],
                                                 dqkv[:, :, 2], ops.LN2))
    f_fwd = 4.0 * n * n * D * B2 * H
    f_bwd = 2.5 * f_fwd
    ach = f_bwd / t_bwd / 1e12
    # the cross layers' backward (both directions of one layer = the same algorithmic FLOPs as one self-attention launch):
    # ONE gf_attn_cross_bwd launch (csrc/attention_xbwd.hip, 10 MFMA products per tile pair) against the two gf_attn_bwd_acc
    # launches (14 products) it replaced, same process, same operands
    pc = (torch.randn(B2, n, 2, H, D, device="cuda", generator=g) * 0.6).to(dtype).requires_grad_(True)
    dm = torch.randn(B2, n, H, D, device="cuda", dtype=dtype, generator=g)
    mc = ops.cross_attention_stacked(pc, scale=ops.LN2)

    def cross_bwd(fused):
        ops.XBWD_ENABLED = fused
        try:
            pc.grad = None
            mc.backward(dm, retain_graph=True)
        finally:
            ops.XBWD_ENABLED = True
    t_x = time_kernel(lambda: cross_bwd(True)) if dtype == torch.bfloat16 else None
    t_x2 = time_kernel(lambda: cross_bwd(False))
    cross = {"kernel": "gf_attn_cross_bwd (attn_stats_kernel + attn_xbwd_bf16_kernel): backward of one cross layer, both directions",
             "two_launch_ms": round(t_x2 * 1e3, 4), "two_launch_frac": round(f_bwd / t_x2 / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
    if t_x is not None:
        cross.update({"launch_ms": round(t_x * 1e3, 4), "achieved": round(f_bwd / t_x / 1e12, 2),
                      "frac": round(f_bwd / t_x / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": _traffic("gf_attn_cross_bwd")})
    if t_x is not None:     # the step's attention backward as a whole: L self layers on gf_attn_bwd + L cross layers on gf_attn_cross_bwd
        cross["step_attention_backward"] = {
            "what": "one self layer + one cross layer (the step runs L of each): algorithmic FLOPs over the sum of the two launches",
            "achieved": round(2 * f_bwd / (t_bwd + t_x) / 1e12, 2), "frac": round(2 * f_bwd / (t_bwd + t_x) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
    return {
        "bound": "mfma", "kernel": "gf_attn_bwd (attn_dq3_bf16_kernel + attn_bwd_dkv_bf16_kernel of one launch)",
        "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": _traffic("gf_attn_bwd"),
        "launch_ms": round(t_bwd * 1e3, 4), "algorithmic_flop_per_launch": f_bwd,
        "traffic_source": _traffic_source(),
        "cross_bwd": cross,
        "fwd_kernel": {"kernel": "attn_fwd3_bf16_kernel", "launch_ms": round(t_fwd * 1e3, 4),
                       "achieved": round(f_fwd / t_fwd / 1e12, 2),
                       "frac": round(f_fwd / t_fwd / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                       "traffic": _traffic("attn_fwd_kernel")},
    }


def roofline_hbm(batch, n, dtype, sinkhorn_iters=100):
    """HBM regime: (1) the materialised log-assignment, one fp32 write of B*(N+1)^2*4 bytes
    (gf_assign_write); (2) one Sinkhorn iteration = a row sweep + a column sweep over the couplings, algorithmic
    2 * B*(N+1)^2*4 bytes per iteration and direction (SuperGlue, config 4); forward timed over all iterations."""
    from glue_factory_amd import lib as L_
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(batch, n, 256, device="cuda", dtype=dtype, generator=g) * 0.3
    b = torch.randn(batch, n, 256, device="cuda", dtype=dtype, generator=g) * 0.3
    rb = torch.randn(batch, n, device="cuda", generator=g)
    out = {}
    t = time_kernel(lambda: ops.assign_write(a, b, rb, rb, rb, rb))
    byt = batch * (n + 1) * (n + 1) * 4.0
    out["assign_write"] = {"bound": "hbm", "kernel": "assign_write_kernel", "achieved": round(byt / t / 1e9, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byt / t / 1e9 / HBM_PEAK_GBS, 4),
                           "traffic": _traffic("assign_write_kernel"), "launch_ms": round(t * 1e3, 4),
                           "algorithmic_bytes_per_launch": byt}
    Z = torch.randn(batch, n + 1, n + 1, device="cuda", generator=g)
    lib = L_.load()
    sched = ops.sinkhorn_schedule()              # GF_SINKHORN_RESIDENT / GF_SINKHORN_WAIT_MS, as ops.sinkhorn applies them
    ws = torch.empty(int(lib.gf_sinkhorn_ws_bytes(batch, n, n, sinkhorn_iters)), dtype=torch.uint8, device="cuda")
    o = torch.empty_like(Z)
    uh = torch.empty((sinkhorn_iters, batch, n + 1), device="cuda")
    vh = torch.empty((sinkhorn_iters, batch, n + 1), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t = time_kernel(lambda: lib.gf_sinkhorn_fwd(Z.data_ptr(), o.data_ptr(), uh.data_ptr(), vh.data_ptr(), ws.data_ptr(),
                                                batch, n, n, sinkhorn_iters, sched, st), iters=10, warm=2)
    byt = batch * (n + 1) * (n + 1) * 4.0 * sinkhorn_iters          # ONE sweep of the couplings per iteration
    nexp = batch * (n + 1) * (n + 1) * float(sinkhorn_iters)

    plan = (ctypes.c_int64 * 8)()
    resident = lib.gf_sinkhorn_plan(batch, n, n, torch.cuda.get_device_properties(0).multi_processor_count, 0, sched, plan) == 1

    def sk_entry(name, t):
        # the couplings stay on the chip for all iterations (csrc/sinkhorn_resident.h): no HBM roofline applies; what bounds
        # the sweep is the exponential -- ONE exp2 per element and iteration serves the row sums and the next half-step's
        # column sums -- priced against the chip's transcendental issue rate, 256 CUs x 4 SIMDs x 8 lanes/clk x 2.4 GHz
        e = {"bound": "valu", "kernel": f"{name} ({sinkhorn_iters} iterations, B={batch})",
             "achieved": round(nexp / t / 1e12, 3), "peak": EXP_PEAK_T, "unit": "Texp/s",
             "frac": round(nexp / t / 1e12 / EXP_PEAK_T, 4),
             "traffic": _traffic(name), "launch_ms": round(t * 1e3, 3),
             "algorithmic_exp_per_launch": nexp,
             "hbm_accounting": {"one_sweep_per_iteration_bytes": byt, "GBps": round(byt / t / 1e9, 1),
                                "frac_of_hbm_peak": round(byt / t / 1e9 / HBM_PEAK_GBS, 4),
                                "frac_two_sweeps": round(2 * byt / t / 1e9 / HBM_PEAK_GBS, 4)},
             "path": "resident" if resident else "streaming", "note": note}
        if not resident:        # the streaming kernels DO sweep the couplings once per iteration: the HBM line is their bound
            e.update({"bound": "hbm", "achieved": round(byt / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(byt / t / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": byt})
        return e
    note = ("resident path: `frac` = exp2 evaluations per second / the transcendental peak (19.7 T/s); `traffic` (PMC) is per CALL, not per "
            "iteration: forward = one read of the couplings + one write of the output (the resident kernel loads Z itself and writes `out` "
            "from its last iteration) plus the published partial rows; backward = the pre-scaled copy, the rank-2T product's reads and the "
            "partial rows.  `hbm_accounting` keeps the figures of "
            "a kernel that streams the couplings (one fp32 sweep per iteration; SURVEY 8(d) counts two) for comparison with earlier "
            "rounds -- an accounting figure, not a bound")
    out["sinkhorn_fwd"] = sk_entry("gf_sinkhorn_fwd", t)
    G = torch.randn_like(Z)
    gZ = torch.empty_like(Z)
    gr, gc = G.sum(2).contiguous(), G.sum(1).contiguous()
    t = time_kernel(lambda: lib.gf_sinkhorn_bwd(Z.data_ptr(), G.data_ptr(), gr.data_ptr(), gc.data_ptr(), uh.data_ptr(),
                                                vh.data_ptr(), gZ.data_ptr(), ws.data_ptr(), batch, n, n,
                                                sinkhorn_iters, sched, st), iters=10, warm=2)
    out["sinkhorn_bwd"] = sk_entry("gf_sinkhorn_bwd", t)
    return out


def roofline_extra(batch, n, dtype):
    """The two next-largest own kernels of the headline step: the streamed-activation GEMM (HBM-bound: x read + y
    write, weights stationary in registers) at the step's most frequent shape, and the fused 64-channel 3x3
    convolution block of the extractor (MFMA-bound implicit GEMM)."""
    from glue_factory_amd import lib as L_
    from glue_factory_amd import ops
    lib = L_.load()
    out = {}
    M = 2 * batch * n
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, 256, device="cuda", dtype=dtype, generator=g)
    w = torch.randn(256, 256, device="cuda", dtype=dtype, generator=g) * 0.05
    b = torch.zeros(256, device="cuda")
    y = torch.empty(M, 256, device="cuda", dtype=dtype)
    t = time_kernel(lambda: ops.gemm(x, w, b, None, y), iters=20)
    byt = 2.0 * M * (256 + 256) + 2.0 * 256 * 256
    out["gemm_st"] = {"bound": "hbm", "kernel": "gemm_st_kernel (gf_gemm, M=131072, 256 <- 256, bf16)",
                      "achieved": round(byt / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(byt / t / 1e9 / HBM_PEAK_GBS, 4), "traffic": _traffic("gemm_st_kernel"),
                      "launch_ms": round(t * 1e3, 4), "algorithmic_bytes_per_launch": byt}
    # assignment-head similarity passes (csrc/assignment.hip; lightglue.py:256-290, 68-94): every pass recomputes the N x N
    # similarity of a pair on MFMA (2 N^2 d FLOP) and reduces it on the fly -- no N x N tensor exists.  A LightGlue step runs
    # 28 such forward passes (3 per layer: column LSE, row LSE + row arg-max, column arg-max; + the assignment write) and 18
    # backward ones; DESIGN.md section 4 says why the three dependent forward passes are not two.
    md = (torch.randn(2 * batch, n, 256, device="cuda", generator=g) * 0.5).to(dtype)
    a_, b_ = md[:batch], md[batch:]
    zb = torch.randn(batch, n, device="cuda", generator=g)
    fl = 2.0 * batch * n * n * 256
    t1 = time_kernel(lambda: ops.rows_lse(b_, a_), iters=20)
    cl = ops.rows_lse(b_, a_)
    v0 = torch.empty(batch, n, device="cuda")
    a0 = torch.empty(batch, n, dtype=torch.int64, device="cuda")
    r0 = torch.empty(batch, n, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    dt_code = 1 if dtype == torch.bfloat16 else 0
    t2 = time_kernel(lambda: lib.gf_rows_lse_argmax(a_.data_ptr(), b_.data_ptr(), zb.data_ptr(), cl.data_ptr(), 2.0, r0.data_ptr(),
                                                    v0.data_ptr(), a0.data_ptr(), batch, n, n, 256, dt_code, st), iters=20)
    out["head_pass"] = {"bound": "mfma", "kernel": f"rows_lse_kernel / rows_lse_argmax_kernel (B={batch} pairs, {n} x {n} x 256)",
                        "achieved": round(fl / t1 / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / t1 / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "launch_ms": round(t1 * 1e3, 4),
                        "rows_lse_argmax": {"achieved": round(fl / t2 / 1e12, 1), "frac": round(fl / t2 / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                            "launch_ms": round(t2 * 1e3, 4)},
                        "traffic": _traffic("rows_lse_kernel"), "algorithmic_flop_per_launch": fl,
                        "passes_per_lightglue_step": {"forward": 28, "backward": 18}}
    if dtype == torch.bfloat16:
        # ---- the next three by per-step time (profiles/r05c_lightglue_step_kernel_stats.csv): head backward (MFMA), the
        # weight gradient of ffn.0(cat[x, msg]) (HBM) and LayerNorm + GELU forward (HBM)
        gr = torch.randn(batch, n, device="cuda", generator=g) * 1e-3
        gc = torch.randn(batch, n, device="cuda", generator=g) * 1e-3
        da, db_ = torch.empty_like(a_), torch.empty_like(b_)
        t3 = time_kernel(lambda: ops._head_bwd(a_, b_, r0, cl, gr, gc, da, db_), iters=20)
        fl3 = 6.0 * batch * n * n * 256             # algorithmic: S once, dS md1, dS^T md0 (the kernel recomputes S: 8 N^2 d executed)
        out["head_bwd"] = {"bound": "mfma", "kernel": f"head_bwd_bf16_kernel (B={batch} pairs, {n} x {n} x 256)",
                           "achieved": round(fl3 / t3 / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(fl3 / t3 / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "launch_ms": round(t3 * 1e3, 4),
                           "executed_over_algorithmic": round(8.0 / 6.0, 3), "traffic": _traffic("head_bwd_bf16_kernel"),
                           "algorithmic_flop_per_launch": fl3}
        x1 = torch.randn(M, 256, device="cuda", dtype=dtype, generator=g)
        x2 = torch.randn(M, 256, device="cuda", dtype=dtype, generator=g)
        dyw = torch.randn(M, 512, device="cuda", dtype=dtype, generator=g)
        wsd = torch.empty(int(lib.gf_linear_dw_ws_bytes(M, 512, 512)), dtype=torch.uint8, device="cuda")
        dww = torch.empty(512, 512, device="cuda")
        dbw = torch.empty(512, device="cuda")
        t4 = time_kernel(lambda: lib.gf_linear_dw2(dyw.data_ptr(), x1.data_ptr(), x2.data_ptr(), 256, dww.data_ptr(), dbw.data_ptr(),
                                                   wsd.data_ptr(), M, 512, 512, 1, st), iters=20)
        by4 = 2.0 * M * (512 + 512)
        out["linear_dw"] = {"bound": "hbm", "kernel": "linear_dw_dma_kernel + linear_dw_reduce (gf_linear_dw2, M=131072, 512 <- 256 + 256, bf16)",
                            "achieved": round(by4 / t4 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(by4 / t4 / 1e9 / HBM_PEAK_GBS, 4), "launch_ms": round(t4 * 1e3, 4),
                            "traffic": _traffic("linear_dw_dma_kernel"), "algorithmic_bytes_per_launch": by4,
                            "mfma_frac": round(2.0 * M * 512 * 512 / t4 / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                            "note": "256 FLOP per HBM byte: the shape sits AT the ridge of the (power-limited) roofline, so neither line alone "
                                    "bounds it -- `mfma_frac` is the same launch against the MFMA peak (PMC: MFMA pipe 33 % busy); the 16 output "
                                    "tiles of a slice also re-read the dY / X panels through L2 (1.1 GB L2 -> LDS per launch, 4x the HBM bytes; "
                                    "DESIGN.md section 7)"}
        xl = torch.randn(M, 512, device="cuda", dtype=dtype, generator=g)
        gam, bet = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
        with torch.no_grad():
            t5 = time_kernel(lambda: ops.ln_gelu(xl, gam, bet), iters=20)
        by5 = 2.0 * M * 512 * 2
        out["ln_gelu_fwd"] = {"bound": "hbm", "kernel": "ln_gelu_fwd_kernel (M=131072, C=512, bf16)",
                              "achieved": round(by5 / t5 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(by5 / t5 / 1e9 / HBM_PEAK_GBS, 4), "launch_ms": round(t5 * 1e3, 4),
                              "traffic": _traffic("ln_gelu_fwd_kernel"), "algorithmic_bytes_per_launch": by5}
        del x1, x2, dyw, wsd, xl, da, db_
    del md, a_, b_
    if dtype == torch.bfloat16:
        # calibration, not a product kernel: what the vendor library's plain bf16 GEMM reaches on THIS box in THIS process
        # (the part clocks down under sustained MFMA load; DESIGN.md section 5 reads the attention fractions against it)
        ga = torch.randn(8192, 8192, device="cuda", dtype=dtype, generator=g)
        gb = torch.randn(8192, 8192, device="cuda", dtype=dtype, generator=g)
        gc = torch.empty(8192, 8192, device="cuda", dtype=dtype)
        t = time_kernel(lambda: torch.matmul(ga, gb, out=gc), iters=10, warm=3)
        out["library_gemm_calibration"] = {"kernel": "hipBLASLt bf16 GEMM 8192^3 via torch.matmul (not on the product path)",
                                           "achieved": round(2.0 * 8192 ** 3 / t / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                           "unit": "TFLOP/s", "frac": round(2.0 * 8192 ** 3 / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
        del ga, gb, gc
        nimg, hh, ww = 2 * batch, IMG, IMG
        xi = torch.randn(nimg, hh, ww, 64, device="cuda", dtype=dtype, generator=g)
        wc = (torch.randn(9, 64, 64, device="cuda", generator=g) * 0.05).to(dtype)       # [tap][c_out][c_in]
        v = torch.zeros(64, device="cuda")
        one = torch.ones(64, device="cuda")
        yo = torch.empty(nimg, hh, ww, 64, device="cuda", dtype=dtype)
        st = torch.cuda.current_stream().cuda_stream

        def conv():
            rc = lib.gf_conv3x3_c64(xi.data_ptr(), wc.data_ptr(), v.data_ptr(), one.data_ptr(), v.data_ptr(), yo.data_ptr(),
                                    nimg, hh, ww, 1, 0, 1, st)          # relu, no pool, GF_BF16
            if rc != 0:
                raise RuntimeError(f"gf_conv3x3_c64 rc={rc}")
        try:
            t = time_kernel(conv, iters=5)
            fl = 2.0 * 64 * 576 * nimg * hh * ww
            out["conv3x3_c64"] = {"bound": "mfma", "kernel": f"conv3x3_c64_kernel ({nimg} x {hh} x {ww} x 64, conv + bias + ReLU + BN)",
                                  "achieved": round(fl / t / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": _traffic("conv3x3_c64_kernel"),
                                  "launch_ms": round(t * 1e3, 3), "algorithmic_flop_per_launch": fl}
        except Exception as e:      # noqa: BLE001  (an extra entry must never cost the headline line)
            out["conv3x3_c64"] = {"error": str(e)[:200]}
    return out


def micro_bench(batch, n, dtype):
    """Per-kernel timings at the step's own shapes (tuning aid; not part of the JSON contract)."""
    from glue_factory_amd import ops
    from glue_factory_amd import lib as L_
    M = 2 * batch * n
    dev = "cuda"
    out = {}
    g = torch.Generator(device=dev).manual_seed(0)
    lib = L_.load()
    for nout, k in ((768, 256), (256, 256), (512, 512), (256, 512), (512, 256)):
        dy = torch.randn(M, nout, device=dev, dtype=dtype, generator=g)
        x = torch.randn(M, k, device=dev, dtype=dtype, generator=g)
        ws = torch.empty(int(lib.gf_linear_dw_ws_bytes(M, nout, k)), dtype=torch.uint8, device=dev)
        dw = torch.empty(nout, k, device=dev)
        db = torch.empty(nout, device=dev)
        t = time_kernel(lambda: lib.gf_linear_dw(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                 ws.data_ptr(), M, nout, k, 1 if dtype == torch.bfloat16 else 0,
                                                 torch.cuda.current_stream().cuda_stream))
        out[f"linear_dw_{nout}x{k}_us"] = round(t * 1e6, 1)
        out[f"linear_dw_{nout}x{k}_TF"] = round(2.0 * M * nout * k / t / 1e12, 1)
    x = torch.randn(M, 512, device=dev, dtype=dtype, generator=g)
    gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    xr = x.clone().requires_grad_(True)
    y = ops.ln_gelu(xr, gam, bet)
    out["ln_gelu_fwd_us"] = round(time_kernel(lambda: ops.ln_gelu(x, gam, bet)) * 1e6, 1)
    dy = torch.randn_like(y)
    out["ln_gelu_fwd+bwd_us"] = round(time_kernel(lambda: ops.ln_gelu(xr, gam, bet).backward(dy)) * 1e6, 1)
    a = torch.randn(batch, n, 256, device=dev, dtype=dtype, generator=g) * 0.5
    b = torch.randn(batch, n, 256, device=dev, dtype=dtype, generator=g) * 0.5
    out["rows_lse_us"] = round(time_kernel(lambda: ops.rows_lse(a, b)) * 1e6, 1)
    cb = torch.zeros(batch, n, device=dev)
    out["rows_argmax_us"] = round(time_kernel(lambda: ops.rows_argmax(a, b, cb, 2.0)) * 1e6, 1)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    r, c = ops.dual_lse(ar, br)
    gr, gc = torch.randn_like(r), torch.randn_like(c)
    out["dual_lse_fwd+bwd_us"] = round(time_kernel(
        lambda: torch.autograd.backward(ops.dual_lse(ar, br), (gr, gc))) * 1e6, 1)
    return out


# ------------------------------------------------------------------------------------------- CPU baseline
def _calibrate_threads(step_small):
    """torch's CPU backend collapses when given every hardware thread of a large host (measured: 541 s/step with
    256 threads vs ~7 s with 8), so the thread count is calibrated on a small problem; `cores` reports it."""
    avail = os.cpu_count() or 1
    best, cores = None, 1
    for nt in (8, 16, 32, 64):
        if nt > avail:
            break
        torch.set_num_threads(nt)
        step_small()
        t0 = time.time()
        step_small()
        dt = time.time() - t0
        if best is None or dt < best:
            best, cores = dt, nt
    torch.set_num_threads(cores)
    return cores, avail


def cpu_baseline(n, layers):
    """The reference's own CPU path beside the GPU number: full LightGlue train step (forward + loss + backward + Adam,
    fp32, `flash: false` as in the training yamls) at B=1 pair, same N and L; pairs/s = 1 / step.
    kind "reference": gluefactory's LightGlue module itself, from oracle/_ref (byte-compiled from /root/reference by
    oracle/build_ref.py in the build container; the GPU box only has the .pyc files).  kind "port": the oracle
    restatement (oracle/lightglue_oracle.py), used only when oracle/_ref is absent."""
    from glue_factory_amd.synthetic import make_pairs
    from oracle import build_ref
    from oracle import lightglue_oracle as lgo

    if build_ref.import_reference():
        from gluefactory.models.matchers.lightglue import LightGlue as RefLightGlue
        kind = "reference"

        def make_step(nn_, ll):
            params = lgo.init_params(ll, DIM, HEADS, seed=0)
            model = RefLightGlue({"n_layers": ll, "descriptor_dim": DIM, "input_dim": DIM, "num_heads": HEADS,
                                  "weights": None, "flash": False, "checkpointed": False}).train()
            model.load_state_dict(params, strict=True)
            data = make_pairs(1, nn_, dim=DIM, seed=1)
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)      # the optimiser of the reference's training configs

            def step():
                opt.zero_grad(set_to_none=True)
                pred = model(data)
                losses, _ = model.loss(pred, {**pred, **data})
                losses["total"].mean().backward()
                opt.step()
            return step
    else:
        kind = "port"

        def make_step(nn_, ll):
            params = lgo.init_params(ll, DIM, HEADS, seed=0)
            data = make_pairs(1, nn_, dim=DIM, seed=1)
            data = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
            leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
            opt = torch.optim.Adam(list(leaves.values()), lr=1e-4)

            def step():
                _, _, grads = lgo.train_step_grads(params, data, ll, HEADS)
                for k, v in leaves.items():
                    v.grad = grads[k]
                opt.step()
            return step

    cores, avail = _calibrate_threads(make_step(512, 1))
    step = make_step(n, layers)
    t0 = time.time()
    step()                                       # warm-up (also the cold cost)
    cold = time.time() - t0
    reps = 1 if cold > 12 else (2 if cold > 5 else 4)
    t0 = time.time()
    for _ in range(reps):
        step()
    dt = (time.time() - t0) / reps
    what = ("gluefactory.models.matchers.lightglue.LightGlue (the reference module, oracle/_ref)" if kind == "reference"
            else "the torch-CPU oracle port")
    return {"value": round(1.0 / dt, 4), "unit": "image-pairs/s", "cores": cores, "kind": kind,
            "sample": f"B=1 pair, N={n}, L={layers}, fp32, 1 warm + {reps} timed train steps = forward + loss + backward + "
                      f"torch.optim.Adam step (`checkpointed: false`, no extractor) ({dt:.2f} s/step) of {what} on {cores} of "
                      f"{avail} host threads (thread count calibrated on a small problem: torch's CPU backend collapses with "
                      f"all {avail})"}


# ------------------------------------------------------------------------------------------- model setup
def build_matcher(args, rank, name, conf=None):
    from glue_factory_amd.synthetic import make_pairs
    torch.manual_seed(0)
    if name == "lightglue":
        from glue_factory_amd.matchers.lightglue import LightGlue
        model = LightGlue({"n_layers": args.layers, "filter_threshold": 0.1}).cuda().train()
        cpu_data = make_pairs(args.batch, args.kpts, dim=DIM, seed=100 + rank)
    elif name == "superglue":
        from glue_factory_amd.matchers.superglue import SuperGlue
        model = SuperGlue({"num_sinkhorn_iterations": args.sinkhorn_iters}).cuda().train()
        cpu_data = make_pairs(args.batch, args.kpts, dim=DIM, seed=100 + rank)
    else:
        from glue_factory_amd.matchers.gluestick import GlueStick
        from glue_factory_amd.synthetic import make_point_line_pairs
        model = GlueStick(conf or {}).cuda().train()
        cpu_data = make_point_line_pairs(args.batch, args.kpts, args.lines, dim=DIM, seed=100 + rank)
    return model, cpu_data


# GF_FORCE_DIST=1 python bench.py --gpus 1: the N > 1 code path (process group, gradient buckets, SyncBatchNorm exchange,
# eager-then-captured measurement, data_parallel report) in a group of ONE rank over RCCL -- the smoke run of that path on a
# single-GPU box (every collective is the identity; the line must equal the plain N = 1 one up to the launch mode).
FORCE_DIST = os.environ.get("GF_FORCE_DIST") == "1"


def make_stepper(args, model, local, allow_graph=True):
    """TrainStep around `model`.  One process: the whole step is captured once and replayed as a hipGraph.  Several ranks:
    main() measures the step launched kernel by kernel FIRST (bucketed all-reduces overlapped from autograd hooks: plain
    torch.distributed calls, the safe path) and then tries the CAPTURED bucket-reducer step (collectives inside the
    hipGraph, RCCL only) under a watchdog -- `allow_graph` selects which of the two this stepper is."""
    from glue_factory_amd.train_step import TrainStep
    graph = allow_graph and not args.no_graph
    # Adam as ONE table-driven launch per 80 tensors (glue_factory_amd.optim.FusedAdam = torch.optim.Adam's numbers;
    # torch's own fused kernel needs 7 launches and 0.5 ms for these 12 M parameters)
    from glue_factory_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-4)
    return TrainStep(model, opt, amp_dtype=torch.bfloat16 if args.dtype == "bf16" else None, device_ids=[local],
                     graph=graph, force_distributed=FORCE_DIST)


def scope_p_inputs(batch, rank):
    """a batch of synthetic images ~U(0,1) [B,1,IMG,IMG], a homography per pair SAMPLED like the reference's
    homography dataset does (corner perturbation of the full frame: datasets/homographies.py:37-44 ->
    geometry/homography.py:40-67 sample_homography_corners; here
    every corner moves by up to 12 % of the image side, seeded), and the second view = the first WARPED by it (bilinear,
    zeros outside) -- generated here, outside every timed region, resident in HBM.  H_0to1 maps view0 pixels to view1."""
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    img0 = torch.rand(batch, 1, IMG, IMG, device="cuda", generator=g)
    side = float(IMG)
    src = torch.tensor([[0.0, 0.0], [side, 0.0], [side, side], [0.0, side]], device="cuda")[None].repeat(batch, 1, 1)
    dst = src + (torch.rand(batch, 4, 2, device="cuda", generator=g) - 0.5) * 0.24 * side
    # DLT: the 8 x 8 system of the four corner correspondences (fp64)
    x, y, u, v = src[..., 0].double(), src[..., 1].double(), dst[..., 0].double(), dst[..., 1].double()
    z, o = torch.zeros_like(x), torch.ones_like(x)
    A = torch.cat([torch.stack([x, y, o, z, z, z, -u * x, -u * y], -1), torch.stack([z, z, z, x, y, o, -v * x, -v * y], -1)], 1)
    h = torch.linalg.solve(A, torch.cat([u, v], 1)[..., None])[..., 0]
    H = torch.cat([h, torch.ones(batch, 1, device="cuda", dtype=torch.float64)], 1).reshape(batch, 3, 3)
    # view1(p1) = view0(H^-1 p1): sample view0 at the back-projected pixel centres
    ys, xs = torch.meshgrid(torch.arange(IMG, device="cuda", dtype=torch.float64) + 0.5,
                            torch.arange(IMG, device="cuda", dtype=torch.float64) + 0.5, indexing="ij")
    p1 = torch.stack([xs, ys, torch.ones_like(xs)], -1).reshape(1, -1, 3)
    p0 = p1 @ torch.linalg.inv(H).transpose(1, 2)
    p0 = p0[..., :2] / p0[..., 2:]
    grid = (p0 / side * 2 - 1).reshape(batch, IMG, IMG, 2).float()
    img1 = torch.nn.functional.grid_sample(img0, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    size = torch.tensor([[side, side]], device="cuda").repeat(batch, 1)
    return {"view0": {"image": img0, "image_size": size}, "view1": {"image": img1.contiguous(), "image_size": size.clone()},
            "H_0to1": H.float()}


def make_pipeline_step(args, rank, local, graph=True):
    """Scope P through the product API: ``glue_factory_amd.pipeline.TwoViewPipeline`` (frozen SuperPoint-open extractor ->
    homography ground truth -> LightGlue) inside ``TrainStep`` -- forward, ground truth, loss, backward and the fused Adam
    update of one step, captured as ONE hipGraph (the extractor's top-k is csrc/topk.hip: a captured extractor tail
    with torch.topk in it faults on its second replay).  Inputs: 2 x batch synthetic IMG x IMG images resident in HBM."""
    from glue_factory_amd.pipeline import TwoViewPipeline
    torch.manual_seed(0)
    pipe = TwoViewPipeline({
        "extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": args.kpts, "force_num_keypoints": True,
                      "detection_threshold": 0.0, "nms_radius": 3, "trainable": False, "freeze_batch_normalization": True},
        "ground_truth": {"name": "matchers.homography_matcher", "th_positive": 3.0, "th_negative": 3.0, "with_reward": False},
        "matcher": {"name": "matchers.lightglue", "n_layers": args.layers, "filter_threshold": 0.1},
    }).cuda()
    stepper = make_stepper(args, pipe, local, allow_graph=graph)
    data = scope_p_inputs(args.batch, rank)
    images = torch.cat([data["view0"]["image"], data["view1"]["image"]], 0)
    state = {"data": data}

    def extract():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            return pipe.extractor({"image": images})

    def pipeline_step():
        out = stepper(state["data"])["total"].mean()
        static = stepper.static_inputs()
        if static is not None and state["data"] is not static:
            state["data"] = static        # the graph's own input buffers (same values): no per-step copy of the images
        return out

    return pipeline_step, extract, stepper


def timed_steps(step, warmup, steps, barrier, dist):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # a matching NLL on random-init weights sits around log(N) ~ 8-10 and can only go down from there
    if not torch.isfinite(loss.detach()).item() or not 0.0 <= float(loss.item()) < 100.0:
        raise RuntimeError(f"implausible loss in the benchmark step: {float(loss.item())}")
    # every product of a timed configuration must have run on the hand-written kernels: a shape that fell through to the
    # vendor library (ops.gemm counts them) makes the number a measurement of something else -- fail, do not warn
    from glue_factory_amd import ops
    if ops.LIBRARY_GEMMS:
        raise SystemExit(f"bench.py: products left the HIP path for the vendor library during a timed configuration: {ops.LIBRARY_GEMMS}")
    return dt, float(loss.item())


def pcie_inclusive(args, p_stepper, data):
    """The same scope-P step with the batch arriving from HOST memory, the way the reference's loop receives it
    (train.py:462-469: `batch_to_device(data, device, non_blocking=True)` on what the DataLoader's pinned-memory workers
    produced): 2 x B float32 images + image sizes + H_0to1 per step over PCIe.  `serial`: the host-to-device copies are
    enqueued on the step's stream in front of the replay (what the reference loop does).  `prefetched`: batch k + 1 is copied
    by a second stream into one of two device staging sets while step k replays; the step then starts with a device-side
    copy into the graph's input buffers.  A REPORTED companion of `value`, never `value` itself."""
    def tree(fn, d):
        return {k: tree(fn, v) if isinstance(v, dict) else (fn(v) if torch.is_tensor(v) else v) for k, v in d.items()}

    def leaves(d):
        for v in d.values():
            if isinstance(v, dict):
                yield from leaves(v)
            elif torch.is_tensor(v):
                yield v

    def copy_tree(dst, src):
        for k, v in src.items():
            if isinstance(v, dict):
                copy_tree(dst[k], v)
            elif torch.is_tensor(v):
                dst[k].copy_(v, non_blocking=True)

    host = tree(lambda t: t.detach().cpu().pin_memory(), data)
    nbytes = sum(t.numel() * t.element_size() for t in leaves(host))
    steps = min(args.steps, 10)
    cur = torch.cuda.current_stream()

    def timed(step_k):
        for k in range(2):
            step_k(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 2 + steps):
            loss = step_k(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if not torch.isfinite(loss["total"]).all().item():
            raise RuntimeError("non-finite loss in the PCIe-inclusive step")
        return dt

    # the copies alone (the box's pinned host-to-device rate)
    stage = [tree(torch.empty_like, data) for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        copy_tree(stage[0], host)
    torch.cuda.synchronize()
    h2d = (time.perf_counter() - t0) / 3
    serial = timed(lambda k: p_stepper(host))
    side = torch.cuda.Stream()
    landed = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    for e in freed:
        e.record(cur)

    def prefetch(i):
        with torch.cuda.stream(side):
            side.wait_event(freed[i])                # the step that read staging set i has copied it out
            copy_tree(stage[i], host)
            landed[i].record(side)

    prefetch(0)

    def step_prefetched(k):
        i = k & 1
        prefetch(1 - i)
        cur.wait_event(landed[i])
        out = p_stepper(stage[i])
        freed[i].record(cur)
        return out

    pref = timed(step_prefetched)
    torch.cuda.synchronize()

    def entry(dt):
        return {"ms_per_step": round(dt * 1e3, 3), "value": round(args.batch / dt, 2)}

    return {"unit": "image-pairs/s", "host_bytes_per_step": nbytes, "h2d_ms": round(h2d * 1e3, 3),
            "h2d_GBps": round(nbytes / h2d / 1e9, 1), "steps": steps, "serial": entry(serial), "prefetched": entry(pref),
            "note": "scope P with the batch in pinned HOST memory (2 x B float32 images, sizes, H_0to1): copies on the step's "
                    "stream / on a second stream under the previous step; never the headline value"}


def other_config(args, name, local, conf=None):
    """BASELINE.json configs[3] / configs[4]: matcher train step of SuperGlue / GlueStick, inputs resident in HBM."""
    from glue_factory_amd.synthetic import to_device
    model, cpu_data = build_matcher(args, 0, name, conf)
    # captured like the LightGlue step: their losses gather the positives through the fixed-length col0 vectors
    stepper = make_stepper(args, model, local, allow_graph=True)
    data = to_device(cpu_data, "cuda")
    steps = min(args.steps, 10)
    for _ in range(PRIME_STEPS):
        stepper(data)
    dt, loss = timed_steps(lambda: stepper(data)["total"].mean(), min(args.warmup, 3), steps,
                           torch.cuda.synchronize, None)
    desc = (f"SuperGlue (18 GNN layers, {args.sinkhorn_iters} Sinkhorn iterations)" if name == "superglue"
            else f"GlueStick ({args.kpts} keypoints + {args.lines} lines = {args.kpts + 2 * args.lines} tokens per image)")
    out = {"value": round(args.batch * steps / dt, 2), "unit": "image-pairs/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "workload": f"{desc} matcher train step, B={args.batch} pairs, N={args.kpts}, bf16, "
                                       "inputs resident in HBM", "final_loss": round(loss, 4)}
    del stepper, model, data
    torch.cuda.empty_cache()
    return out


def data_parallel_report(dist, rank, world, stepper):
    """What a reader needs to trust an N > 1 line: which ranks actually took part (all-gathered RANK / device), how the
    gradient buckets were cut, and the measured bus bandwidth of an all-reduce of one bucket over RCCL (ring: 2 (N-1)/N x
    bytes / time) -- xGMI is point to point, so this is the number the bucket size was chosen against."""
    ids = torch.tensor([rank, torch.cuda.current_device()], device="cuda", dtype=torch.int64)
    seen = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(seen, ids)
    out = {"ranks_seen": [int(t[0]) for t in seen], "devices": [int(t[1]) for t in seen], "backend": dist.get_backend(),
           "reducer": "buckets" if stepper.buckets is not None else "ddp"}
    if getattr(stepper, "last_collectives", None):
        # per step and rank: gradient-bucket all-reduces + SyncBatchNorm exchanges (one per BatchNorm call-site and direction for
        # BOTH views: ops._BatchNormActSetsSync); LightGlue has no BatchNorm
        out["collectives_per_step"] = stepper.last_collectives
    if stepper.buckets is not None:
        out["buckets"] = len(stepper.buckets.buckets)
        out["bucket_mbytes"] = [round((hi - lo) * 4 / 2 ** 20, 2) for lo, hi, _ in stepper.buckets.buckets]
    buf = torch.zeros(4 * 2 ** 20, device="cuda")                       # 16 MB of fp32: the bucket cap
    for _ in range(3):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["allreduce_16MB_ms"] = round(dt * 1e3, 4)
    out["allreduce_busbw_GBps"] = round(2.0 * (world - 1) / world * buf.numel() * 4 / dt / 1e9, 2)
    return out


DP_GRAPH_TIMEOUT_S = 240.0
PRIME_STEPS = 3          # TrainStep(graph=True) runs two eager steps and captures on the third: all three belong to the set-up, so that
                         # `--warmup W` with W < 3 cannot push the capture into the timed region


class _Watchdog:
    """N > 1 only: bounds the attempt to run the multi-rank step as a captured hipGraph.  On expiry rank 0 prints the line
    it already has (the kernel-by-kernel numbers) and every rank leaves with exit code 0 -- a hung collective must not cost
    the measurement that was already taken."""

    def __init__(self, seconds, line):
        import threading
        self._line = line
        self._t = threading.Timer(seconds, self._expire)
        self._t.daemon = True
        self._t.start()

    def _expire(self):
        if self._line is not None:
            print(json.dumps(self._line), flush=True)
        os._exit(0)

    def cancel(self):
        self._t.cancel()


def build_line(args, world, matcher, pipe_res, dp_info, dp_modes, pairs):
    """The ONE JSON line of the contract from what has been measured so far."""
    extra = {}
    if dp_info is not None:
        extra["data_parallel"] = dict(dp_info, **({"step_launch_modes": dp_modes} if dp_modes else {}))
    if pipe_res is None:
        return {"metric": f"image-pairs/sec (train step) {args.model} matcher only", "n_gpus": world,
                "warmup": args.warmup, "dtype": args.dtype, "data": "synthetic",
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "config": {"workload": f"{args.model} matcher train step, {args.batch} pairs/GPU, N={args.kpts}"},
                **extra, **matcher}
    p_dt, p_loss, p_stepper, _ = pipe_res
    return {
        "metric": "image-pairs/sec (train step) SP+LightGlue N=2048 d=256 L=9",
        "value": round(pairs / p_dt, 2), "unit": "image-pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(p_dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: SuperPoint + LightGlue train step -- frozen SuperPoint-open forward on 2x32 "
                               f"synthetic {IMG}x{IMG} images resident in HBM (view1 = view0 warped by a sampled homography, SURVEY 8(d)(P); "
                               "first block and the 64-channel 3x3 blocks as fused HIP "
                               "kernels, library convolutions + fused HIP tails for the 128/256-channel blocks), homography "
                               "ground truth (gf_gt_nn), LightGlue fwd + loss + bwd + fused Adam; glue_factory_amd.pipeline."
                               "TwoViewPipeline inside TrainStep, the whole step "
                               + ("replayed as ONE hipGraph" if p_stepper.graph else "launched kernel by kernel"),
                   "pairs_per_gpu": args.batch, "global_batch": args.batch * world, "keypoints": args.kpts,
                   "descriptor_dim": DIM, "layers": args.layers, "image_size": [IMG, IMG], "parallelism": f"dp{world}"},
        "final_loss": round(p_loss, 4), "matcher_step": matcher, **extra}


def self_spawn(args):
    """`python bench.py --gpus N` without a torchrun environment: launch N ranks of this file on one node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_spawn(args))
    if args.micro:
        torch.cuda.set_device(0)
        print(json.dumps(micro_bench(args.batch, args.kpts, dtype)))
        return
    if args.roofline_only:
        torch.cuda.set_device(0)
        out = {"roofline": roofline_attention(args.batch, args.kpts, dtype),
               "roofline_hbm": roofline_hbm(args.batch, args.kpts, dtype, args.sinkhorn_iters),
               "roofline_extra": roofline_extra(args.batch, args.kpts, dtype)}
        print(json.dumps(out))
        return
    from glue_factory_amd import lib
    from glue_factory_amd.synthetic import to_device
    from glue_factory_amd.train_step import init_distributed
    import torch.distributed as dist_mod

    rank, world, local = init_distributed()          # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % torch.cuda.device_count()        # (GF_DIST_BACKEND=gloo smoke runs put several ranks on one GPU)
    torch.cuda.set_device(local)
    dist = dist_mod if (world > 1 or FORCE_DIST) else None
    lib.load()

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    model, cpu_data = build_matcher(args, rank, args.model)
    data = to_device(cpu_data, "cuda")
    headline_is_pipeline = args.model == "lightglue" and not args.matcher_only
    pairs = args.batch * world * args.steps

    def matcher_entry(m_dt, m_loss, graphed):
        return {"value": round(pairs / m_dt, 2), "unit": "image-pairs/s", "ms_per_step": round(m_dt / args.steps * 1e3, 3),
                "steps": args.steps,
                "scope": "M: matcher train step (fwd + loss + bwd + Adam) on SuperPoint-shaped keypoint pairs resident in "
                         "HBM (the reference's cached-feature training mode)",
                "launch": "one hipGraph replay per step" if graphed else "kernel by kernel",
                "mfma_frac_step": round(pairs / m_dt * flops_per_pair_train(args.kpts, DIM, args.layers)
                                        / world / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
                "final_loss": round(m_loss, 4)}

    def measure(graph):
        """(matcher entry, pipeline (dt, loss, stepper, extract) or None, data-parallel report or None) with every step
        either launched kernel by kernel or replayed as one hipGraph."""
        stepper = make_stepper(args, model, local, allow_graph=graph)
        for _ in range(PRIME_STEPS):                  # set-up, not warm-up: two eager steps, then the capture of the hipGraph
            stepper(data)
        m_dt, m_loss = timed_steps(lambda: stepper(data)["total"].mean(), args.warmup, args.steps, barrier, dist)
        m = matcher_entry(m_dt, m_loss, stepper.graph)
        dp = data_parallel_report(dist, rank, world, stepper) if dist is not None else None
        if not headline_is_pipeline:
            stepper.close()
            return m, None, dp
        pipeline_step, extract, p_stepper = make_pipeline_step(args, rank, local, graph=graph)
        for _ in range(PRIME_STEPS):                  # MIOpen's convolution search and the graph capture happen here, outside any timing
            pipeline_step()
        p_dt, p_loss = timed_steps(pipeline_step, args.warmup, args.steps, barrier, dist)
        stepper.close()
        return m, (p_dt, p_loss, p_stepper, extract), dp

    # ---- N = 1: the step is ONE hipGraph.  N > 1: kernel by kernel first (plain torch.distributed calls from autograd
    # hooks: the path the CPU / single-GPU multi-process tests cover), then -- RCCL only -- the CAPTURED bucket-reducer step
    # (collectives inside the hipGraph) under a watchdog: if the capture raises, the run falls back to the eager numbers; if
    # it hangs (a collective inside a capture that one rank abandoned), every rank prints / exits on the eager numbers after
    # DP_GRAPH_TIMEOUT_S instead of hanging the job.  When both work the line carries both and `value` is the better one.
    dp_modes = None
    if dist is None:
        matcher, pipe_res, dp_info = measure(graph=True)
    else:
        matcher, pipe_res, dp_info = measure(graph=False)
        dp_modes = {"eager": {"matcher_ms_per_step": matcher["ms_per_step"],
                              **({"pipeline_ms_per_step": round(pipe_res[0] / args.steps * 1e3, 3)} if pipe_res else {})}}
        try_graph = dist.get_backend() == "nccl" and not args.no_graph
        if try_graph:
            eager_line = build_line(args, world, matcher, pipe_res, dp_info, dict(dp_modes, graph={"status": "timed out"}), pairs)
            dog = _Watchdog(DP_GRAPH_TIMEOUT_S, eager_line if rank == 0 else None)
            try:
                g_matcher, g_pipe, _ = measure(graph=True)
                ok = torch.ones((), device="cuda")
            except Exception as e:          # capture / replay error on this rank
                g_matcher = g_pipe = None
                ok = torch.zeros((), device="cuda")
                dp_modes["graph"] = {"status": f"failed: {type(e).__name__}: {str(e)[:200]}"}
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank must have replayed its graph
            torch.cuda.synchronize()
            dog.cancel()
            if float(ok.item()) > 0:
                dp_modes["graph"] = {"status": "ok", "matcher_ms_per_step": g_matcher["ms_per_step"],
                                     **({"pipeline_ms_per_step": round(g_pipe[0] / args.steps * 1e3, 3)} if g_pipe else {})}
                better = (g_pipe[0] < pipe_res[0]) if pipe_res else (g_matcher["ms_per_step"] < matcher["ms_per_step"])
                if better:
                    matcher, pipe_res = g_matcher, g_pipe
            else:
                dp_modes.setdefault("graph", {"status": "failed on another rank"})
        else:
            dp_modes["graph"] = {"status": "not attempted (" + ("--no-graph" if args.no_graph else
                                                               f"{dist.get_backend()}: collectives are capturable on RCCL only") + ")"}
    out = build_line(args, world, matcher, pipe_res, dp_info, dp_modes, pairs)
    if not headline_is_pipeline:
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    p_stepper, extract = pipe_res[2], pipe_res[3]
    out["extractor_ms"] = round(time_kernel(extract, iters=5, warm=1) * 1e3, 2)
    if rank == 0 and world == 1:
        if not args.no_roofline:
            try:
                out["roofline"] = roofline_attention(args.batch, args.kpts, dtype)
                out["roofline_hbm"] = roofline_hbm(args.batch, args.kpts, dtype, args.sinkhorn_iters)
                out["roofline_extra"] = roofline_extra(args.batch, args.kpts, dtype)
            except Exception as e:   # a secondary measurement must never cost the headline line
                out.setdefault("roofline", {"error": f"{type(e).__name__}: {e}"})
        if not args.no_pcie:
            try:
                out["pcie_inclusive"] = pcie_inclusive(args, p_stepper, p_stepper.static_inputs())
            except Exception as e:   # a secondary measurement must never cost the headline line
                out["pcie_inclusive"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_other_configs:
            del model, data, p_stepper, extract, pipe_res
            torch.cuda.empty_cache()
            oc = {}
            for name in ("superglue", "gluestick"):
                try:
                    oc[name] = other_config(args, name, local)
                    if name == "gluestick":
                        # the line above runs the module's default, the reference's own AMP arithmetic (its attention pinned
                        # to fp32-equivalent products, gluestick.py:524-529); the all-bf16 kernels beside it
                        oc[name]["attention_precision"] = "reference"
                        alt = other_config(args, name, local, {"attention_precision": "bf16"})
                        oc[name]["attention_precision_bf16"] = {k: alt[k] for k in ("value", "ms_per_step", "final_loss")}
                except Exception as e:
                    oc[name] = {"error": f"{type(e).__name__}: {e}"}
            out["other_configs"] = oc
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kpts, args.layers)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
