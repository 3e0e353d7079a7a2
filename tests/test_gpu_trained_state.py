"""Parity AT TRAINED STATES (round-5 review, item 1a).  Every other golden pins the kernels at random-init (or hand-sharpened)
weights; these take the states that 300-step runs of tests/learning_cases.py END in -- the reference's own CPU run
(`superglue_trained_ref`) and the HIP path's runs on the MI355X (`superglue_trained_hip`, `gluestick_trained_hip`): attention
and assignment rows far from uniform, BatchNorm statistics far from their initial values, a grown bin_score -- and hold the
HIP modules to what the UNMODIFIED reference module computes there (oracle/gen_golden.py::gen_trained_state): eval forward,
train step (log-assignments, every loss entry, every parameter gradient) and the BatchNorm buffers after the step, fp32 at
north_star's 1e-4; the bf16 mode with stated bounds.  `superglue_trained_hip` is the state whose EVAL-mode loss (18.4 on this
batch) is far from the reference run's (3.8): if a kernel lost accuracy in that regime -- Sinkhorn / softmax / BatchNorm
cancellation -- this is where it would show."""
import numpy as np
import pytest
import torch

import learning_cases as lc
from config_golden import (assert_disagreements_are_ties, check_la_digest, grad_digest_errors, la_digest_error,
                           significant_grads)
from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = [("superglue_trained_ref", "superglue"), ("superglue_trained_hip", "superglue"), ("gluestick_trained_hip", "gluestick")]


def _setup(name, kind):
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import to_device
    z = load_golden(name)
    state = lc.trained_state_from_delta(lc.initial_params(kind), z)
    data = lc.batch(kind, int(z["meta"][0]))
    dchk = float(sum(v.double().abs().sum() for v in data.values() if torch.is_tensor(v) and v.is_floating_point()))
    assert abs(dchk - float(z["data_checksum"][0])) < 1e-9 * dchk
    model = {"superglue": SuperGlue, "gluestick": GlueStick}[kind](lc.conf(kind))
    res = model.load_state_dict(state, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return z, model.cuda(), to_device(data, "cuda"), int(z["meta"][1])


def _heads(kind):
    return [("log_assignment", "matches0", "", 0)] + ([("line_log_assignment", "line_matches0", "line_", 7)] if kind == "gluestick" else [])


@pytest.mark.parametrize("name,kind", CASES)
def test_trained_state_eval_forward_vs_reference(name, kind):
    z, model, data, stride = _setup(name, kind)
    model.eval()
    with torch.no_grad():
        pred = model(data)
        losses, _ = model.loss(pred, {**pred, **data})
    for la, mk, pre, st in _heads(kind):
        check_la_digest(z, pred[la], st or stride, prefix="eval." + pre, tol=1e-4)
        n = assert_disagreements_are_ties(pred[la], pred[mk], z["eval." + mk], 0.2)
        print(f"{name} eval {mk}: rows that differ from the reference's (all near-ties): {n}")
        assert n <= 2
    np.testing.assert_allclose(losses["total"].cpu().numpy(), z["eval.loss_total"], rtol=1e-4, atol=1e-4)
    print(f"{name}: eval-mode loss on the held-out batch {float(losses['total'].mean()):.4f} (reference {float(z['eval.loss_total'].mean()):.4f})")


@pytest.mark.parametrize("name,kind", CASES)
def test_trained_state_fp32_train_step_vs_reference(name, kind):
    z, model, data, stride = _setup(name, kind)
    model.train()
    pred = model(data)
    losses, _ = model.loss(pred, {**pred, **data})
    losses["total"].mean().backward()
    for la, mk, pre, st in _heads(kind):
        check_la_digest(z, pred[la], st or stride, prefix="train." + pre, tol=1e-4)
        assert assert_disagreements_are_ties(pred[la], pred[mk], z["train." + mk], 0.2) <= 2
    for k in [k[5:] for k in z if k.startswith("loss.")]:
        np.testing.assert_allclose(losses[k].detach().float().cpu().numpy(), z["loss." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    errs = significant_grads(grad_digest_errors(z, grads))
    worst_n = max((e[0], k) for k, e in errs.items())
    worst_s = max((e[1], k) for k, e in errs.items())
    print(f"{name} fp32 train step: worst gradient-norm error {worst_n}, worst gradient-sample error {worst_s}, tensors {len(errs)}")
    assert worst_n[0] <= 1e-3, worst_n
    assert worst_s[0] <= 4e-3, worst_s
    # BatchNorm buffers after the step (the GNN layers' second update under the reference's activation checkpointing included)
    post = model.state_dict()
    worst_b = 0.0
    for k in [k[5:] for k in z if k.startswith("post.")]:
        r = z["post." + k]
        d = float(np.abs(post[k].float().cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-6))
        worst_b = max(worst_b, d)
        assert d <= 1e-4, (k, d)
    print(f"{name}: BatchNorm buffers after the step, worst relative difference {worst_b:.2e}")


# bf16 bounds: 1.5 x the values measured on MI355X in round 6 -- log_assignment max / p99 / mean |d|, total loss (relative), worst
# per-tensor gradient-sample error: superglue_trained_ref 0.133 / 0.049 / 0.012, 9.5e-5, 5.1 %; superglue_trained_hip 0.115 / 0.051 /
# 0.013, 1.3e-4, 7.7 %; gluestick_trained_hip 0.064 / 0.033 / 0.0087, 4.6e-4, 4.3 % -- i.e. SMALLER than at random weights (the
# random-init bounds of tests/test_gpu_configs45.py: SuperGlue 0.48 / 0.21 / 0.042, 14.5 %; GlueStick 0.62 / 0.33 / 0.083, 30 %)
BF16 = {"superglue_trained_ref": (0.20, 0.076, 0.019, 5e-4, 0.12), "superglue_trained_hip": (0.20, 0.076, 0.019, 5e-4, 0.12),
        "gluestick_trained_hip": (0.10, 0.05, 0.013, 1.5e-3, 0.07)}


@pytest.mark.parametrize("name,kind", CASES)
def test_trained_state_bf16_train_step_bounds(name, kind):
    z, model, data, stride = _setup(name, kind)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(data)
        losses, _ = model.loss(pred, {**pred, **data})
    losses["total"].mean().backward()
    mx, p99, mean = la_digest_error(z, pred["log_assignment"], stride)
    rel = abs(float(losses["total"].mean()) - float(z["loss.total"].mean())) / abs(float(z["loss.total"].mean()))
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    errs = significant_grads(grad_digest_errors(z, grads))
    rels = sorted(e[1] for e in errs.values())
    print(f"{name} bf16: log_assignment max|d| {mx:.4f} p99 {p99:.4f} mean {mean:.4f}; total loss rel {rel:.2e}; "
          f"gradient-sample error median {rels[len(rels) // 2]:.4f} worst {max((e[1], k) for k, e in errs.items())}")
    b = BF16[name]
    assert mx <= b[0] and p99 <= b[1] and mean <= b[2] and rel <= b[3] and max(rels) <= b[4], (mx, p99, mean, rel, max(rels))
