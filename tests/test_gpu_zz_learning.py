"""End to end: do the HIP matchers LEARN like the reference?  Parity tests pin every tensor of single steps to the
reference; this one runs the thing the path exists for -- 300 optimiser steps (TrainStep: fused Adam, one hipGraph, bf16
autocast or fp32) on FRESH synthetic pairs every step (matched descriptors correlate at cos ~ 0.2, positions follow a
similarity warp; GlueStick: + line segments) -- and checks that the run follows the learning curve of the UNMODIFIED
reference module trained on the CPU in fp32 on the very same batches from the very same initial parameters
(tools/probe/ref_learning_curve.py, tests/learning_cases.py, profiles/r05e_learning_curve_reference_cpu.txt) and ends where
the reference ends on held-out pairs (loss, match precision, match recall).  Wrong-but-finite gradients anywhere on the
path -- attention, Sinkhorn, train-mode BatchNorm, loss heads, the fused optimiser -- fail this.
(The file name sorts it last: three 300-step runs, ~2.3 min of the suite.)"""
import pytest
import torch

import learning_cases as lc

pytestmark = pytest.mark.gpu

# the reference's runs: train loss at steps 50, 100, ..., 300; held-out (loss, precision, recall[, line precision, line recall])
REF = {
    "lightglue": {"trace": [5.812, 5.398, 4.639, 4.127, 3.868, 3.742], "before": [6.4633, 0.0, 0.0],
                  "after": [1.9514, 0.7943, 0.4959]},
    # SuperGlue (4 GNN layers, 20 Sinkhorn iterations, lr 2e-4).  What is asserted after training, and why (round 6,
    # profiles/r06_learning_anchor.txt -- the same run in SIX arithmetics, every end state evaluated by the reference module):
    #   * the EVAL-mode held-out loss is NOT a reproducible quantity of this model family after 300 steps: the unmodified
    #     reference lands at 2.5 ... 5.1 in fp32 depending on the thread count, at 7.78 in fp64 (two thread counts, and
    #     bit-for-bit the same state from stock torch fp64 on the MI355X), stock PyTorch-ROCm fp32 on the MI355X at 21.3,
    #     the HIP path at 17.3 (fp32) / 10.8 (bf16) -- printed, never asserted;
    #   * in TRAINING mode (per-image batch statistics: what the loss is trained under) all six agree: 1.071 ... 1.080 held-out,
    #     precision 0.746 ... 0.763, recall 0.653 ... 0.722 -- asserted against the fp64 anchor's values;
    #   * the distance of the trained weights to the reference's own fp32 run (the `superglue_trained_ref` fixture), in units of
    #     that run's drift from the initial state: HIP fp32 0.195-0.197, stock ROCm fp32 0.197, reference fp32 vs fp64 0.157, HIP
    #     bf16 0.288-0.315 (the HIP runs are not bit-reproducible from box to box) -- asserted (0.25 / 0.40).
    "superglue": {"trace": [1.735, 1.431, 1.292, 1.15, 1.15, 1.058], "before": [3.4872, 0.0, 0.0],
                  "after": [3.3805, 0.4306, 0.0158], "assert_after": False,
                  "train_mode_after": [1.074, 0.76, 0.708], "trained_ref": "superglue_trained_ref",
                  "max_distance": {False: 0.25, True: 0.40}},
    # GlueStick (4 GNN layers + line layers, 192 keypoints + 32 lines, lr 2e-4): the optimisation is chaotic from step ~150 on
    # -- the reference's own runs: 16 threads 6.593 6.296 5.523 4.617 3.516 3.075, 3 threads 6.593 6.298 5.523 3.951 4.391 4.617,
    # fp64 6.592 6.299 5.645 4.281 3.633 3.996, any two end states ~0.8 of the drift apart (HIP fp32 vs fp64: 0.75) -- so only the
    # first 100 steps are compared point by point; afterwards: it must have learnt what the reference's runs learn (held-out,
    # reference runs: eval loss 3.04 / 3.99 / 3.95, precision 0.32 / 0.40 / 0.40, line precision 0.62 / 0.54 / 0.62; HIP fp32
    # 3.12 / 0.37 / 0.59, bf16 7.55 / 0.39 / 0.49; training-mode held-out loss 3.34 (fp64), 3.46, HIP 2.83).
    "gluestick": {"trace": [6.593, 6.296, 5.523, 4.617, 3.516, 3.075], "before": [8.3008, 0.0, 0.0, 0.0, 0.0],
                  "after": [3.0362, 0.3219, 0.102, 0.6212, 0.4261], "assert_after": False, "chaotic_after": 2,
                  "learnt": {"train_mode_loss_below": 4.6, "precision_above": 0.25, "line_precision_above": 0.35}},
}


@pytest.fixture(autouse=True, scope="module")
def _few_cpu_threads():
    """The batches are generated on the CPU (seeded, the same tensors the reference's run saw); torch's CPU backend crawls on
    small tensors when it spreads them over the GPU box's 256 hardware threads."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 8))
    yield
    torch.set_num_threads(n)


def _model(kind):
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.matchers.superglue import SuperGlue
    torch.manual_seed(0)
    model = {"lightglue": LightGlue, "superglue": SuperGlue, "gluestick": GlueStick}[kind](lc.conf(kind))
    params = lc.initial_params(kind)
    if params is not None:
        res = model.load_state_dict(params, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    return model.cuda()


def _batch(kind, seed):
    from glue_factory_amd.synthetic import to_device
    return to_device(lc.batch(kind, seed), "cuda")      # (pageable host batches: copied synchronously, see synthetic.to_device)


def _evaluate(kind, model, bf16, mode="eval"):
    """Held-out loss / precision / recall [/ line precision / line recall]; mode "train": through per-image batch statistics
    (the BatchNorm models' training arithmetic; no_grad, but the running statistics move -- call it last)."""
    from glue_factory_amd.metrics import matcher_metrics
    model.train() if mode == "train" else model.eval()
    rows = []
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        for s in lc.HELD_OUT:
            data = _batch(kind, s)
            pred = model(data)
            losses, _ = model.loss(pred, {**pred, **data})
            m = matcher_metrics(pred, {**pred, **data})
            row = [float(losses["total"].mean()), float(m["match_precision"].mean()), float(m["match_recall"].mean())]
            if kind == "gluestick":
                ml = matcher_metrics(pred, {**pred, **data}, prefix="line_", prefix_gt="line_")
                row += [float(ml["line_match_precision"].mean()), float(ml["line_match_recall"].mean())]
            rows.append(row)
    return [sum(v) / len(v) for v in zip(*rows)]


def _distance_to(model, kind, fixture):
    """|theta - theta_ref| / |theta_ref - theta_0| over the weight matrices, theta_ref = the reference's own trained state
    (a `*_trained_ref` fixture), theta_0 the shared initial state."""
    from conftest import load_golden
    init = lc.initial_params(kind)
    ref = lc.trained_state_from_delta(init, load_golden(fixture))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    keys = [k for k in init if k.endswith(".weight") and init[k].ndim >= 2]
    num = sum(float((sd[k] - ref[k]).double().pow(2).sum()) for k in keys) ** 0.5
    den = sum(float((ref[k] - init[k]).double().pow(2).sum()) for k in keys) ** 0.5
    return num / den


@pytest.mark.parametrize("kind,bf16", [("lightglue", True), ("lightglue", False), ("superglue", True), ("superglue", False),
                                       ("gluestick", True), ("gluestick", False)])
def test_matcher_learns_like_the_reference_on_fresh_synthetic_pairs(kind, bf16):
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.train_step import TrainStep
    ref = REF[kind]
    model = _model(kind)
    before = _evaluate(kind, model, bf16)
    assert abs(before[0] - ref["before"][0]) < 3e-2 * max(1.0, ref["before"][0])       # same starting point as the reference's run
    # deterministic_replay: this loop builds every batch on the CPU and uploads it while the previous replay may still be in
    # flight; the host waits for every replay instead (belt and braces: on the final tree the run is bit-reproducible either
    # way, profiles/r06_graph_replay_determinism.txt -- the noise this test once showed was an asynchronous copy from pageable
    # memory, fixed in train_step._async_ok / synthetic.to_device).
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16 if bf16 else None,
                     graph=True, graph_warmup=2, deterministic_replay=True)
    trace = []
    for i in range(lc.STEPS):
        out = step(_batch(kind, 1000 + i))
        if i % 50 == 49:
            trace.append(round(float(out["total"].mean()), 3))
    assert step.skipped == 0
    after = _evaluate(kind, model, bf16)
    tag = f"{kind} {'bf16' if bf16 else 'fp32'}"
    print(f"{tag} train loss every 50 steps: {trace}   (reference, CPU fp32: {ref['trace']})")
    print(f"{tag} held-out before {[round(v, 3) for v in before]} -> after {[round(v, 3) for v in after]}   (reference: {ref['after']})")
    # the same trajectory while rounding differences have not been amplified by the optimiser yet, the same place afterwards
    # (measured on MI355X, LightGlue: bf16 1.817 / 0.791 / 0.558, fp32 2.064 / 0.771 / 0.442 -- 250 Adam steps amplify rounding-level
    # differences into a few hundredths; the margins are several times that)
    scale = max(1.0, ref["trace"][0])
    assert abs(trace[0] - ref["trace"][0]) < 0.005 * scale and abs(trace[1] - ref["trace"][1]) < 0.01 * scale
    n_cmp = ref.get("chaotic_after", len(trace))
    assert all(abs(a - b) < 0.06 * scale for a, b in zip(trace[:n_cmp], ref["trace"][:n_cmp]))
    assert all(t == t and t < ref["trace"][0] + 0.5 for t in trace) and trace[-1] < 0.8 * trace[0]
    if ref.get("assert_after", True):
        assert after[0] < ref["after"][0] + 0.15 * scale
        assert all(a > r - 0.2 for a, r in zip(after[1:], ref["after"][1:]))
    if "trained_ref" in ref:
        dist = _distance_to(model, kind, ref["trained_ref"])
        print(f"{tag} distance of the trained weights to the reference's own trained state, in units of its drift: {dist:.3f}")
        assert dist < ref["max_distance"][bf16], dist
    if "train_mode_after" in ref or "learnt" in ref:
        tm = _evaluate(kind, model, bf16, mode="train")
        print(f"{tag} held-out in TRAINING mode (per-image batch statistics): {[round(v, 3) for v in tm]}   "
              f"(fp64 anchor: {ref.get('train_mode_after')})")
        if "train_mode_after" in ref:
            r = ref["train_mode_after"]
            assert abs(tm[0] - r[0]) < 0.06 * max(1.0, r[0]) and tm[1] > r[1] - 0.06 and tm[2] > r[2] - 0.12, tm
        if "learnt" in ref:
            b = ref["learnt"]
            assert tm[0] < b["train_mode_loss_below"] and after[1] > b["precision_above"] and after[3] > b["line_precision_above"], (tm, after)
    step.close()


@pytest.mark.parametrize("kind", ["superglue", "gluestick"])
def test_replayed_run_with_uploads_in_the_loop_is_bit_reproducible(kind):
    """The loop every training script runs -- build the next batch on the CPU and upload it while the previous hipGraph replay may
    still be in flight, no host wait anywhere -- must give the same numbers twice, bit for bit: nothing on the replayed path may
    depend on timing (atomics in gradient paths, workgroup placement of the resident Sinkhorn, what the host does meanwhile:
    profiles/r06_graph_replay_determinism.txt).  (The input race round 6 fixed -- asynchronous copies from PAGEABLE host memory
    reading a batch the loop had already refilled, train_step._async_ok / synthetic.to_device -- is timing dependent and does
    not show in 40 steps of this loop; tools/probe/graph_input_race.py and op_h2d_race.py provoke it directly.)"""
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.train_step import TrainStep

    def run():
        model = _model(kind)
        step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
        losses = []
        for i in range(40):
            out = step(_batch(kind, 2000 + i))             # (the device batch is dropped right away: its memory is reused)
            losses.append(out["total"].clone())             # device-side: no host synchronisation in the loop
        torch.cuda.synchronize()
        assert step.skipped == 0
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        step.close()
        return torch.stack([v.float().mean() for v in losses]), state

    (la, sa), (lb, sb) = run(), run()
    assert torch.equal(la, lb), (la - lb).abs().max()
    assert all(torch.equal(sa[k], sb[k]) for k in sa), [k for k in sa if not torch.equal(sa[k], sb[k])][:5]
