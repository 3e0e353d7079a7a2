"""Benchmark of the matcher train-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full LightGlue train step (forward, loss, backward, Adam update) on one batch of
synthetic SuperPoint-shaped keypoint pairs (BASELINE.json configs[1]: B=32 pairs per GPU,
N=2048 keypoints, d=256, L=9, bf16 compute under autocast, inputs resident in HBM).  Data
parallel over N GPUs is weak scaling (32 pairs per GPU) with DDP/RCCL gradient all-reduce.
Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     live HIP-event timing of the dominant kernel vs the bf16 MFMA peak,
  "cpu_baseline": the CPU oracle (port of the reference algorithm) timed on the host cores on a
                  bounded sample (B=1 pair, same N/L) — reported, never the thing shipped.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Library GEMM selection: replay the hipBLASLt/rocBLAS solutions tuned once on gfx950 for this
# workload's GEMM shapes (PyTorch TunableOp, tuning itself disabled -> no timing side effects).
# TunableOp reads "<name><device ordinal>.csv", so each rank gets a private copy of the table.
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

import torch  # noqa: E402  (after the TunableOp environment is set)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
N_KPTS, DIM, HEADS, LAYERS, BATCH = 2048, 256, 4, 9, 32


def flops_per_pair_train(n=N_KPTS, d=DIM, L=LAYERS):
    """Algorithmic FLOPs of one train step per pair: 3 x forward."""
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
    ap.add_argument("--roofline-only", action="store_true", help="only time the attention kernels")
    ap.add_argument("--micro", action="store_true", help="print per-kernel micro timings and exit")
    ap.add_argument("--model", default="lightglue", choices=["lightglue", "superglue", "gluestick"],
                    help="lightglue = the headline configs[1]; superglue / gluestick = configs[3] / [4] (extra lines)")
    ap.add_argument("--lines", type=int, default=512, help="gluestick: line segments per image")
    ap.add_argument("--sinkhorn-iters", type=int, default=100)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="skip the secondary pipeline-scope measurement (SuperPoint forward + GT + matcher step)")
    return ap.parse_args()


def time_kernel(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters * 1e-3   # seconds per launch


def roofline_attention(batch, n, dtype):
    """Dominant kernel: the self-attention backward (dK/dV kernel + dQ kernel of gf_attn_bwd) at
    the step's own shape (2*batch images, H heads, N tokens, hd=64).  Algorithmic FLOPs per launch
    = 2.5 x forward = 10*N*N*hd per (image, head); forward kernel reported beside it."""
    from glue_factory_amd import ops
    B2, H, D = 2 * batch, HEADS, DIM // HEADS
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B2, n, 3, H, D, device="cuda", dtype=dtype, generator=g)
    do = torch.randn(B2, n, H, D, device="cuda", dtype=dtype, generator=g)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o, lse = ops.attn_fwd_raw(q, k, v, D ** -0.5)
    dqkv = torch.empty_like(qkv)
    t_fwd = time_kernel(lambda: ops.attn_fwd_raw(q, k, v, D ** -0.5, out=o, lse=lse))
    t_bwd = time_kernel(lambda: ops.attn_bwd_raw(q, k, v, o, do, lse, dqkv[:, :, 0], dqkv[:, :, 1],
                                                 dqkv[:, :, 2], D ** -0.5))
    f_fwd = 4.0 * n * n * D * B2 * H
    f_bwd = 2.5 * f_fwd
    ach = f_bwd / t_bwd / 1e12
    return {
        "bound": "mfma", "kernel": "gf_attn_bwd (attn_bwd_dq_kernel + attn_bwd_dkv_bf16_kernel)",
        "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
        "launch_ms": round(t_bwd * 1e3, 4),
        "fwd_kernel": {"kernel": "attn_fwd_kernel", "launch_ms": round(t_fwd * 1e3, 4),
                       "achieved": round(f_fwd / t_fwd / 1e12, 2)},
    }


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
    mean = torch.zeros(M, device=dev)
    rstd = torch.ones(M, device=dev)
    nblk = lib.gf_ln_gelu_nblk(M)
    dxb, dgp, dbp = torch.empty_like(x), torch.empty(nblk, 512, device=dev), torch.empty(nblk, 512, device=dev)
    out["ln_gelu_bwd_kernel_us"] = round(time_kernel(lambda: lib.gf_ln_gelu_bwd(
        x.data_ptr(), gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dxb.data_ptr(),
        dgp.data_ptr(), dbp.data_ptr(), M, 512, 1 if dtype == torch.bfloat16 else 0,
        torch.cuda.current_stream().cuda_stream)) * 1e6, 1)
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


def cpu_baseline(n, layers):
    """CPU oracle (port of the reference algorithm, oracle/lightglue_oracle.py) on the host cores:
    full train step (forward + loss + backward) at B=1 pair, same N and L; pairs/s = 1/step.
    torch's CPU backend collapses when given every hardware thread of a large host (measured:
    541 s/step with 256 threads vs ~7 s with 8), so the thread count is calibrated on a small
    problem first and the count actually used is what `cores` reports."""
    from glue_factory_amd.synthetic import make_pairs
    from oracle import lightglue_oracle as lgo
    avail = os.cpu_count() or 1

    def one_step(nn, ll, params, data):
        t0 = time.time()
        lgo.train_step_grads(params, data, ll, HEADS)
        return time.time() - t0

    def setup(nn, ll):
        params = lgo.init_params(ll, DIM, HEADS, seed=0)
        data = make_pairs(1, nn, dim=DIM, seed=1)
        return params, dict(data, image_size0=data["view0"]["image_size"],
                            image_size1=data["view1"]["image_size"])

    small = setup(512, 1)
    best, cores = None, 1
    for nt in (8, 16, 32, 64):
        if nt > avail:
            break
        torch.set_num_threads(nt)
        one_step(512, 1, *small)
        dt = one_step(512, 1, *small)
        if best is None or dt < best:
            best, cores = dt, nt
    torch.set_num_threads(cores)
    params, data = setup(n, layers)
    cold = one_step(n, layers, params, data)       # warm-up (also the cold cost)
    reps = 1 if cold > 12 else 2
    dt = sum(one_step(n, layers, params, data) for _ in range(reps)) / reps
    return {"value": round(1.0 / dt, 4), "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "sample": f"B=1 pair, N={n}, L={layers}, fp32, 1 warm + {reps} timed full train steps "
                      f"({dt:.2f} s/step) of the torch-CPU oracle on {cores} of {avail} host threads"}


def pipeline_scope(args, stepper):
    """Secondary scope (P): frozen SuperPoint forward on synthetic 1024x1024 images (stock library
    convolutions + the fused HIP tails) + homography ground truth + the same matcher train step."""
    import torch
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    from glue_factory_amd.gt import gt_matches_from_homography_fused as gt_matches_from_homography
    sp = SuperPoint({"max_num_keypoints": args.kpts, "force_num_keypoints": True, "detection_threshold": 0.0,
                     "nms_radius": 3}).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(7)
    img0 = torch.rand(args.batch, 1, 1024, 1024, device="cuda", generator=g)
    img1 = img0.roll(8, -1)
    Hm = torch.tensor([[1.0, 0, 8], [0, 1, 0], [0, 0, 1]], device="cuda")[None].repeat(args.batch, 1, 1)
    size = torch.tensor([[1024.0, 1024.0]], device="cuda").repeat(args.batch, 1)

    def extract():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            return sp({"image": torch.cat([img0, img1], 0)})

    def pipeline_step():
        f = extract()
        b = args.batch
        d = {"keypoints0": f["keypoints"][:b], "keypoints1": f["keypoints"][b:],
             "descriptors0": f["descriptors"][:b], "descriptors1": f["descriptors"][b:],
             "view0": {"image_size": size}, "view1": {"image_size": size}}
        gt = gt_matches_from_homography(d["keypoints0"], d["keypoints1"], Hm, 3.0, 3.0)
        d.update({"gt_assignment": gt["assignment"], "gt_assignment_col0": gt["assignment_col0"],
                  "gt_matches0": gt["matches0"], "gt_matches1": gt["matches1"]})
        return stepper(d)["total"].mean()

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    pipeline_step()                      # library autotuning of the convolutions happens here
    te = timed(extract)
    tp = timed(pipeline_step)
    return {"value": round(args.batch / tp, 2), "unit": "image-pairs/s", "ms_per_step": round(tp * 1e3, 2),
            "extractor_ms": round(te * 1e3, 2),
            "scope": "frozen SuperPoint-open forward on 2x32 synthetic 1024x1024 images (library convolutions + fused HIP "
                     "bias/ReLU/BN/pool and NMS kernels) + homography GT (gf_gt_nn) + LightGlue train step"}


def main():
    args = parse()
    if args.micro:
        torch.cuda.set_device(0)
        print(json.dumps(micro_bench(args.batch, args.kpts, torch.bfloat16 if args.dtype == "bf16" else torch.float32)))
        return
    if args.roofline_only:
        torch.cuda.set_device(0)
        print(json.dumps(roofline_attention(args.batch, args.kpts,
                                            torch.bfloat16 if args.dtype == "bf16" else torch.float32)))
        return
    from glue_factory_amd import lib
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep, init_distributed
    import torch.distributed as dist_mod

    rank, world, local = init_distributed()          # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dist = dist_mod if world > 1 else None
    lib.load()

    torch.manual_seed(0)
    if args.model == "lightglue":
        model = LightGlue({"n_layers": args.layers, "filter_threshold": 0.1}).cuda().train()
        cpu_data = make_pairs(args.batch, args.kpts, dim=DIM, seed=100 + rank)
    elif args.model == "superglue":
        from glue_factory_amd.matchers.superglue import SuperGlue
        model = SuperGlue({"num_sinkhorn_iterations": args.sinkhorn_iters}).cuda().train()
        cpu_data = make_pairs(args.batch, args.kpts, dim=DIM, seed=100 + rank)
    else:
        from glue_factory_amd.matchers.gluestick import GlueStick
        from glue_factory_amd.synthetic import make_point_line_pairs
        model = GlueStick({}).cuda().train()
        cpu_data = make_point_line_pairs(args.batch, args.kpts, args.lines, dim=DIM, seed=100 + rank)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    stepper = TrainStep(model, opt, amp_dtype=torch.bfloat16 if args.dtype == "bf16" else None,
                        device_ids=[local])
    data = to_device(cpu_data, "cuda")

    def step():
        return stepper(data)["total"].mean()

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not torch.isfinite(loss.detach()).item():
        raise RuntimeError("non-finite loss in the benchmark step")

    pairs = args.batch * world * args.steps
    value = pairs / dt
    if args.model != "lightglue":      # extra (non-headline) configurations: short report
        if rank == 0:
            print(json.dumps({"metric": f"image-pairs/sec (train step) {args.model}", "value": round(value, 2),
                              "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": args.dtype, "data": "synthetic",
                              "config": {"workload": f"{args.model} matcher train step, {args.batch} pairs/GPU, "
                                                     f"N={args.kpts}" + (f" + {args.lines} lines" if args.model == "gluestick" else "")
                                                     + (f", {args.sinkhorn_iters} Sinkhorn iterations" if args.model == "superglue" else "")},
                              "final_loss": round(float(loss.item()), 4)}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    out = {
        "metric": "image-pairs/sec (train step) SP+LightGlue N=2048 d=256 L=9",
        "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: LightGlue matcher train step (fwd+loss+bwd+Adam) on synthetic "
                               "SuperPoint-shaped keypoint pairs resident in HBM (the reference's cached-feature training mode, "
                               "two_view_pipeline allow_no_extract / README feature export); on-the-fly extraction is "
                               "reported under 'pipeline'",
                   "pairs_per_gpu": args.batch, "global_batch": args.batch * world,
                   "keypoints": args.kpts, "descriptor_dim": DIM, "layers": args.layers,
                   "parallelism": f"dp{world}"},
        "mfma_frac_step": round(value * flops_per_pair_train(args.kpts, DIM, args.layers)
                                / world / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
        "final_loss": round(float(loss.item()), 4),
    }
    if rank == 0 and not args.no_pipeline and world == 1 and args.model == "lightglue":
        # secondary scope (P): frozen SuperPoint forward on synthetic 1024x1024 images (stock
        # PyTorch-ROCm conv, by design) + homography ground truth + the same matcher train step.
        try:
            out["pipeline"] = pipeline_scope(args, stepper)
        except Exception as e:  # the secondary scope must never cost the headline line
            out["pipeline"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not args.no_roofline:
            out["roofline"] = roofline_attention(args.batch, args.kpts,
                                                 torch.bfloat16 if args.dtype == "bf16" else torch.float32)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kpts, args.layers)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
