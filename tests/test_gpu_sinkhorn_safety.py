"""The chip-resident Sinkhorn (csrc/sinkhorn_resident.h; superglue.py:186-214) when it does NOT have the chip to itself.

Its workgroups hand partial column sums to each other through per-pair counters, so every workgroup of a pair has to
run before any of them can finish an iteration.  On an idle device they all start at once; here another stream's kernel
(`gf_test_hold_cus` of the test-only tests/libgf_test_probe.so: one 150 KB-LDS workgroup per compute unit, the stand-in for an RCCL reduction or a second process)
holds 32 of the 256 CUs while the sweep is launched:
  * the 32 workgroups that find no CU are dispatched when the holder leaves; the result is BIT-IDENTICAL to the idle run
    (contention costs time, never correctness);
  * when the holder outlasts the call's wait bound, every row of the affected pairs' output (forward) / gradient
    (backward) is NaN -- what TrainStep's device-side skip flag and the reference's NaN check (train.py:477-480) act on --
    and the device is healthy afterwards.
"""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

B, N, T = 8, 2048, 20        # one launch of 8 pairs x 32 workgroups = all 256 CUs


def _inputs():
    g = torch.Generator(device="cuda").manual_seed(5)
    Z = torch.randn(B, N + 1, N + 1, device="cuda", generator=g) * 2
    G = torch.randn(B, N + 1, N + 1, device="cuda", generator=g)
    return Z, G


def _run(Z, G, schedule):
    from glue_factory_amd import ops
    z = Z.clone().requires_grad_(True)
    out = ops.sinkhorn(z, T, schedule=schedule)
    (out * G).sum().backward()
    return out.detach(), z.grad


def _hold(n_cus, ms, stream):
    from conftest import test_probe
    rc = test_probe().gf_test_hold_cus(n_cus, ms, stream.cuda_stream)
    assert rc == 0, f"gf_test_hold_cus -> {rc}"


def _resident_here():
    import ctypes
    from glue_factory_amd import lib, ops
    out = (ctypes.c_int64 * 8)()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    ok = lib.load().gf_sinkhorn_plan(B, N, N, ncu, 0, ops.sinkhorn_schedule(2), out)
    return ok == 1 and int(out[0]) * int(out[1]) > ncu - 32      # the launch needs CUs the holder takes


def test_resident_sweeps_under_cu_contention_equal_the_idle_run():
    from glue_factory_amd import ops
    assert _resident_here()
    Z, G = _inputs()
    sched = ops.sinkhorn_schedule(2)
    ref_out, ref_g = _run(Z, G, sched)
    stream_out, stream_g = _run(Z, G, ops.sinkhorn_schedule(0))
    assert float((ref_out - stream_out).abs().max()) < 4e-5          # (two summation orders of the column partials)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _run(Z, G, sched)
    torch.cuda.synchronize()
    idle_s = time.perf_counter() - t0
    side = torch.cuda.Stream()
    for hold_ms in (30, 60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _hold(32, hold_ms, side)                                     # 32 CUs are gone before the sweep is launched
        out, g = _run(Z, G, sched)
        torch.cuda.synchronize()
        busy_s = time.perf_counter() - t0
        print(f"holder {hold_ms} ms on 32 CUs: forward + backward {busy_s * 1e3:.1f} ms (idle {idle_s * 1e3:.1f} ms)")
        assert busy_s > hold_ms * 1e-3                               # the sweep did wait for the holder ...
        assert torch.equal(out, ref_out) and torch.equal(g, ref_g)   # ... and computed exactly what it computes alone


def test_an_expired_wait_poisons_the_pairs_instead_of_returning_numbers():
    from glue_factory_amd import ops
    assert _resident_here()
    Z, G = _inputs()
    good_out, good_g = _run(Z, G, ops.sinkhorn_schedule(2))
    side = torch.cuda.Stream()
    short = ops.sinkhorn_schedule(2, wait_ms=20)
    # ---- forward expires
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _hold(32, 600, side)
    z = Z.clone().requires_grad_(True)
    out = ops.sinkhorn(z, T, schedule=short)
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    ev.synchronize()
    fwd_s = time.perf_counter() - t0
    assert bool(torch.isnan(out).all()), "an expired wait must poison every row of every pair of the launch"
    print(f"forward with a 20 ms bound under a 600 ms holder: returned NaN after {fwd_s * 1e3:.0f} ms")
    torch.cuda.synchronize()
    # ---- backward expires (histories of a healthy forward)
    z = Z.clone().requires_grad_(True)
    out = ops.sinkhorn(z, T, schedule=short)
    torch.cuda.synchronize()
    assert torch.equal(out.detach(), good_out)
    _hold(32, 600, side)
    (out * G).sum().backward()
    torch.cuda.synchronize()
    assert bool(torch.isnan(z.grad).all())
    # ---- and the device is healthy: the same call, idle, is exact again
    out2, g2 = _run(Z, G, ops.sinkhorn_schedule(2))
    assert torch.equal(out2, good_out) and torch.equal(g2, good_g)


def test_train_step_skips_the_update_when_the_sweep_was_poisoned(monkeypatch):
    """End to end: a SuperGlue TrainStep whose Sinkhorn wait expires sees a NaN loss, its device-side flag skips the fused
    optimiser update (parameters untouched, `skipped` counts it), and the next -- uncontended -- step trains."""
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    monkeypatch.setenv("GF_SINKHORN_RESIDENT", "2")
    monkeypatch.setenv("GF_SINKHORN_WAIT_MS", "20")
    torch.manual_seed(0)
    model = SuperGlue({"num_sinkhorn_iterations": T, "GNN_layers": ["self", "cross"]}).cuda()
    step = TrainStep(model, FusedAdam(model.parameters(), lr=1e-3), amp_dtype=torch.bfloat16)
    data = to_device(make_pairs(B, N, dim=256, size=(1024, 1024), seed=6), "cuda")
    first = step(data)["total"]
    assert bool(torch.isfinite(first).all()) and step.skipped == 0
    before = [p.detach().clone() for p in model.parameters()]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    _hold(32, 800, side)
    losses = step(data)["total"]
    torch.cuda.synchronize()
    assert bool(torch.isnan(losses).all())
    assert step.skipped == 1
    assert all(torch.equal(p, q) for p, q in zip(model.parameters(), before))
    again = step(data)["total"]
    assert bool(torch.isfinite(again).all()) and step.skipped == 1
    assert any(not torch.equal(p, q) for p, q in zip(model.parameters(), before))


def test_same_xcd_handoffs_equal_the_write_through_protocol_bit_for_bit():
    """When the resident kernel FINDS every workgroup of a pair on one XCD (hardware XCC id, checked at run time) it
    publishes partial rows / column vectors with plain stores through the shared L2 instead of writing them through to the
    memory side -- same arithmetic, same order, so the result must be bit-identical to the placement-independent protocol
    (schedule bit 2), run after run (a stale hand-off would show up as a run that differs), forward and backward, at the
    benchmarked geometry (B = 32: four launches of 8 pairs, T = 100) and at B = 9 (launches of 5 + 4 pairs: mixed placement)."""
    from glue_factory_amd import ops
    for Bn, Tn, reps in ((32, 100, 6), (9, 30, 10)):
        g = torch.Generator(device="cuda").manual_seed(Bn)
        Z = torch.randn(Bn, N + 1, N + 1, device="cuda", generator=g) * 2
        G = torch.randn(Bn, N + 1, N + 1, device="cuda", generator=g)

        def run(schedule):
            z = Z.clone().requires_grad_(True)
            out = ops.sinkhorn(z, Tn, schedule=schedule)
            (out * G).sum().backward()
            return out.detach(), z.grad

        ref_out, ref_g = run(ops.sinkhorn_schedule(1, safe_handoff=True))
        assert bool(torch.isfinite(ref_out).all()) and bool(torch.isfinite(ref_g).all())
        for _ in range(reps):
            out, gz = run(ops.sinkhorn_schedule(1, safe_handoff=False))
            assert torch.equal(out, ref_out) and torch.equal(gz, ref_g)
        # ... and under an uneven load: a streaming kernel on another stream while the sweeps run
        side = torch.cuda.Stream()
        junk = torch.randn(64 * 1024 * 1024, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(20):
                junk = junk * 1.0001 + 0.5
        out, gz = run(ops.sinkhorn_schedule(1, safe_handoff=False))
        torch.cuda.synchronize()
        assert torch.equal(out, ref_out) and torch.equal(gz, ref_g)
