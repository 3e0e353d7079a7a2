"""The bench line the driver parses (README of the task: one JSON line with metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline):
checked on the committed output of the last GPU run, and on bench.py's own helpers where no GPU is needed."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return files[-1]


def test_committed_bench_line_has_the_contract_keys():
    line = open(_latest()).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) < 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    assert 0.0 < d["final_loss"] < 100.0           # bench.py refuses anything else (a dropped graph node showed up as 1e22)


def test_flop_model_matches_survey():
    """761.8 GFLOP per pair for the N=2048, d=256, L=9 train step (3 x forward)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    try:
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    except Exception as e:            # bench.py imports torch + the package at module level
        pytest.skip(f"bench.py not importable here: {e}")
    assert abs(mod.flops_per_pair_train(2048, 256, 9) / 1e9 - 761.8) < 0.5


def test_grad_chain_bookkeeping():
    """ops.GradChain: contributions are parked until the designated last consumer takes them; a missing one is an error."""
    from glue_factory_amd import ops
    ch = ops.GradChain(3)
    ch.park("a")
    with pytest.raises(RuntimeError):
        ch.take()
    ch.got = 1
    ch.park("a+b")
    assert ch.take() == "a+b" and ch.acc is None and ch.got == 0


def test_multi_rank_watchdog_prints_the_eager_line_and_exits_cleanly():
    """bench.py --gpus N: the attempt to run the multi-rank step as a captured hipGraph is bounded by a watchdog -- when it
    expires (a hung collective), rank 0 prints the line it already has (the kernel-by-kernel numbers) and the process leaves
    with exit code 0.  Simulated here: a subprocess arms the watchdog with a line and then blocks."""
    import subprocess
    import sys
    code = (
        "import importlib.util, json, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(ROOT, 'bench.py')!r})\n"
        "mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)\n"
        "mod._Watchdog(0.5, {'metric': 'm', 'value': 1.0, 'data_parallel': {'step_launch_modes': {'graph': {'status': 'timed out'}}}})\n"
        "time.sleep(60)\n"
        "print('NOT REACHED')\n")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-500:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "NOT REACHED" not in res.stdout
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["data_parallel"]["step_launch_modes"]["graph"]["status"] == "timed out"
    # a cancelled watchdog never fires
    code2 = code.replace("time.sleep(60)", "").replace("mod._Watchdog(0.5,", "w = mod._Watchdog(0.5,").replace("print('NOT REACHED')", "w.cancel(); time.sleep(1.0); print('done')")
    res = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and res.stdout.strip().splitlines()[-1] == "done"


def test_gradchain_second_residual_slot():
    """GradChain.extra: a parked tensor that has not been added yet rides to the next link (gf_gemm_res2) or is folded in by take()."""
    from glue_factory_amd import ops
    ch = ops.GradChain(2)
    ch.extra = 3.0
    ch.park(4.0)
    assert ch.take() == 7.0 and ch.extra is None
    ch = ops.GradChain(2)
    ch.extra = 3.0
    assert ch.pop_extra() == 3.0 and ch.pop_extra() is None
    s = ops.SharedGradSum(3)
    assert s.add("g1") is None and s.acc == "g1" and s.add("g2") is None and s.add("g3") == "g3" and s.acc is None and s.got == 0


def test_replay_node_leaves_the_buffers_a_checkpointed_reference_block_leaves():
    """ops._ReplayRunningStats (the generic form of the BatchNorm replay: SyncBatchNorm / single-set / fallback paths) against
    the real thing on the CPU: a BatchNorm1d called for image 0 and image 1 inside torch.utils.checkpoint, as
    gluefactory_nonfree/superglue.py:160-169 does -- after the backward its running statistics hold the sequence
    s0, s1, s0, s1 and num_batches_tracked == 4; a forward without a backward holds s0, s1."""
    import copy
    import torch
    import torch.utils.checkpoint
    from glue_factory_amd import ops
    torch.manual_seed(0)
    ref_bn = torch.nn.BatchNorm1d(6)
    with torch.no_grad():
        ref_bn.running_mean.normal_()
        ref_bn.running_var.uniform_(0.5, 2.0)
    our_bn = copy.deepcopy(ref_bn).train()
    x0 = torch.randn(4, 6, 50, requires_grad=True)
    x1 = (torch.randn(4, 6, 50) * 2 + 1).requires_grad_(True)
    ref_bn.train()
    y0, y1 = torch.utils.checkpoint.checkpoint(lambda a, b: (ref_bn(a), ref_bn(b)), x0, x1, preserve_rng_state=False,
                                               use_reentrant=False)
    assert int(ref_bn.num_batches_tracked) == 2
    (y0.sum() + (y1 * y1).sum()).backward()
    assert int(ref_bn.num_batches_tracked) == 4                 # the backward re-ran the block in training mode

    def stats(x):
        m = x.detach().mean((0, 2))
        return m, x.detach().var((0, 2), unbiased=True)

    def forward_update(x):      # what ops.batch_norm_act does around the HIP kernels: batch statistics, then the buffers by hand
        y = torch.nn.functional.batch_norm(x, None, None, our_bn.weight, our_bn.bias, training=True, eps=our_bn.eps)
        with torch.no_grad():
            m, v = stats(x)
            our_bn.num_batches_tracked += 1
            our_bn.running_mean.mul_(1 - our_bn.momentum).add_(m, alpha=our_bn.momentum)
            our_bn.running_var.mul_(1 - our_bn.momentum).add_(v, alpha=our_bn.momentum)
        return y

    z0, z1 = forward_update(x0), forward_update(x1)              # the two forward updates
    z0 = ops.replay_running_stats(z0, [(our_bn, [stats(x0), stats(x1)])])
    assert int(our_bn.num_batches_tracked) == 2
    (z0.sum() + (z1 * z1).sum()).backward()
    assert int(our_bn.num_batches_tracked) == 4
    torch.testing.assert_close(our_bn.running_mean, ref_bn.running_mean, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(our_bn.running_var, ref_bn.running_var, rtol=1e-6, atol=1e-7)
    assert ops.replay_running_stats(z1, []) is z1                # nothing collected: no node
