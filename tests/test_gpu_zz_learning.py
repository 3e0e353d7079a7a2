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
    # SuperGlue (4 GNN layers, 20 Sinkhorn iterations, lr 2e-4): the training curve is compared; the held-out numbers AFTER
    # training are printed, not asserted -- they are taken in eval mode, i.e. through BatchNorm running statistics that lag
    # 300 steps of moving activations: the reference's own eval loss (3.38) sits far from its training loss (1.06), and
    # percent-level differences between two runs' parameters move it by factors (profiles/r05e_learning_curve_reference_cpu.txt).
    # What CAN be pinned about those statistics is pinned exactly: the buffers after one step against the reference's
    # (tests/test_gpu_matcher_options.py, incl. the double update under the reference's activation checkpointing).
    "superglue": {"trace": [1.735, 1.431, 1.292, 1.15, 1.15, 1.058], "before": [3.4872, 0.0, 0.0],
                  "after": [3.3805, 0.4306, 0.0158], "assert_after": False},
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
    return to_device(lc.batch(kind, seed), "cuda")


def _evaluate(kind, model, bf16):
    from glue_factory_amd.metrics import matcher_metrics
    model.eval()
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


# (SuperGlue in the benchmarked precision only: its fp32 run follows the reference the same way -- 1.735 1.431 1.29 1.151 1.138 1.034,
# profiles/r05e_learning_curve_reference_cpu.txt -- and costs another minute of the suite)
@pytest.mark.parametrize("kind,bf16", [("lightglue", True), ("lightglue", False), ("superglue", True)])
def test_matcher_learns_like_the_reference_on_fresh_synthetic_pairs(kind, bf16):
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.train_step import TrainStep
    ref = REF[kind]
    model = _model(kind)
    before = _evaluate(kind, model, bf16)
    assert abs(before[0] - ref["before"][0]) < 3e-2 * max(1.0, ref["before"][0])       # same starting point as the reference's run
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16 if bf16 else None,
                     graph=True, graph_warmup=2)
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
    assert all(abs(a - b) < 0.06 * scale for a, b in zip(trace, ref["trace"]))
    if ref.get("assert_after", True):
        assert after[0] < ref["after"][0] + 0.15 * scale
        assert all(a > r - 0.2 for a, r in zip(after[1:], ref["after"][1:]))
    step.close()
