"""HIP-vs-reference parity AT BASELINE configs[3] and configs[4]: SuperGlue with N=2048 keypoints, the full 18-layer
GNN and 100 Sinkhorn iterations, and GlueStick with 2048 keypoints + 512 lines (3072 tokens per image), B=1, whole
train step (forward + loss + backward) -- in fp32 against the reference-generated compact goldens at north_star's
1e-4, and in bf16 (the dtype bench.py times for `other_configs`) with stated, measured bounds incl. per-tensor
gradient error.  Goldens: tests/golden/superglue_config4.npz, gluestick_config5.npz (oracle/gen_golden.py).

Round 5: the DECISIVE goldens superglue_sharp.npz / gluestick_sharp.npz (B=2, N=2048 [+ 512 lines], reference-generated
from oracle/{superglue,gluestick}_oracle.sharp_case): every mutual-NN decision of superglue.py:300-320 and
gluestick.py:321-376 -- matched or -1, points and lines -- is taken with a margin of >= 1.5 nats in the reference's own
output, so matches0/1 and line_matches0/1 are compared with assert_array_equal in fp32 AND in bf16 (the mode bench.py
times), in train mode (BatchNorm batch statistics over both pairs) and in eval mode; and train-mode-BatchNorm steps at
B=4, N=2048 against the CPU oracle (superglue.py:70-79, gluestick.py:465-474)."""
import numpy as np
import pytest
import torch

from config_golden import (assert_disagreements_are_ties, check_la_digest, grad_digest_errors, gs_config_inputs,
                           la_digest_error, sg_config_inputs, significant_grads)

pytestmark = pytest.mark.gpu

# ---- stated bf16 bounds: 1.5 x the LARGER of the values measured on MI355X in rounds 3-5 (the tests print the measured ones) ----
# SuperGlue measured (round 3 / round 5 box): log_assignment max 0.32 / 0.25, p99 0.136 / 0.132, mean 0.027 / 0.027, worst loss
# entry 3.8e-4 / 2.9e-4, per-tensor gradient error median 2.9 % / 2.8 %, worst 9.5 % / 7.8 % (gnn.layers.0.attn.proj.0.bias:
# the query bias, whose gradient is a sum of cancelling terms).  The reference's OWN mixed precision measures 1.71 / 0.43 /
# 0.081 and 13.9 % against its fp32 run (profiles/r04_reference_amp_*.txt).
SG_BF16 = {"la_max": 0.48, "la_p99": 0.21, "la_mean": 0.042, "loss_rel": 6e-4, "grad_rel": 0.145}
# GlueStick measured: log_assignment max 0.40 / 0.36, p99 0.22 / 0.195, mean 0.044 / 0.041, lines 0.41 / 0.35, 0.20 / 0.18,
# 0.055 / 0.055, worst loss entry 4.2e-3 / 3.3e-3, per-tensor gradient error median 7.4 % / 6.7 %, worst 20 % / 19 %
# (lenc.encoder.4.bias / gnn.layers.3.update.attn.proj.0.bias); the reference's own AMP: 1.60 / 0.50 / 0.107 and 33 %.
GS_BF16 = {"la_max": 0.62, "la_p99": 0.33, "la_mean": 0.083, "loss_rel": 6.5e-3, "grad_rel": 0.30}


def _cuda(d):
    from glue_factory_amd.synthetic import to_device
    return to_device(d, "cuda")


def _sg_step(bf16):
    from glue_factory_amd.matchers.superglue import SuperGlue
    z, params, data, nl, iters = sg_config_inputs()
    model = SuperGlue({"num_sinkhorn_iterations": iters})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.cuda().train()
    cdata = _cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())
    return z, pred, losses, grads


def _gs_step(bf16):
    from glue_factory_amd.matchers.gluestick import GlueStick
    z, params, data, nl = gs_config_inputs()
    model = GlueStick({})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.cuda().train()
    cdata = _cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    return z, pred, losses, grads


def _check_losses(z, losses, tol):
    for k in [k[5:] for k in z if k.startswith("loss.")]:
        np.testing.assert_allclose(losses[k].detach().float().cpu().numpy(), z["loss." + k], rtol=tol, atol=tol, err_msg=k)


def _fp32_grads(z, grads, norm_tol, sample_tol):
    errs = significant_grads(grad_digest_errors(z, grads))
    worst_n = max((e[0], k) for k, e in errs.items())
    worst_s = max((e[1], k) for k, e in errs.items())
    print("worst gradient-norm error", worst_n, "worst gradient-sample error", worst_s)
    assert worst_n[0] <= norm_tol, worst_n
    assert worst_s[0] <= sample_tol, worst_s


def _bf16_report(tag, z, la_items, losses, grads, bounds):
    for name, la, stride, prefix in la_items:
        mx, p99, mean = la_digest_error(z, la, stride, prefix)
        print(f"{tag} bf16 {name}: max|d| {mx:.4f}  p99 {p99:.4f}  mean {mean:.4f}")
        assert mx <= bounds["la_max"] and p99 <= bounds["la_p99"] and mean <= bounds["la_mean"], (name, mx, p99, mean)
    worst_loss = 0.0
    for k in [k[5:] for k in z if k.startswith("loss.")]:
        r = z["loss." + k]
        got = losses[k].detach().float().cpu().numpy()
        rel = float(np.max(np.abs(got - r) / np.maximum(np.abs(r), 1e-3)))
        worst_loss = max(worst_loss, rel)
    big = significant_grads(grad_digest_errors(z, grads))   # (without the analytically-zero gradients)
    rels = sorted(e[1] for e in big.values())
    worst = max((e[1], k) for k, e in big.items())
    print(f"{tag} bf16: worst loss entry rel {worst_loss:.2e}; per-tensor gradient-sample error: worst {worst}, "
          f"median {rels[len(rels) // 2]:.4f}, tensors {len(rels)}")
    assert worst_loss <= bounds["loss_rel"], worst_loss
    assert worst[0] <= bounds["grad_rel"], worst


def test_superglue_config4_fp32_train_step_vs_reference():
    z, pred, losses, grads = _sg_step(bf16=False)
    stride = int(z["meta"][5])
    check_la_digest(z, pred["log_assignment"], stride, tol=1e-4)
    np.testing.assert_allclose(pred["sinkhorn_cost"].detach().cpu().flatten(1)[:, ::stride].numpy(), z["train.cost_sample"],
                               rtol=1e-4, atol=1e-4)
    # random weights, no decisive margins (the sharp goldens below are the bit-exact ones): a row may differ from the
    # reference's only where our own log-assignment rates the decision a near-tie
    n = assert_disagreements_are_ties(pred["log_assignment"], pred["matches0"], z["train.matches0"], 0.2)
    print("superglue config4 fp32: matches0 rows that differ from the reference's, all near-ties:", n)
    assert n <= 2
    _check_losses(z, losses, 1e-4)
    _fp32_grads(z, grads, norm_tol=6e-4, sample_tol=2e-3)        # 2x the measured 3.1e-4 / 1.0e-3


def test_superglue_config4_bf16_train_step_bounds():
    z, pred, losses, grads = _sg_step(bf16=True)
    _bf16_report("superglue config4", z, [("log_assignment", pred["log_assignment"], int(z["meta"][5]), "train.")],
                 losses, grads, SG_BF16)


def test_gluestick_config5_fp32_train_step_vs_reference():
    z, pred, losses, grads = _gs_step(bf16=False)
    stride = int(z["meta"][5])
    check_la_digest(z, pred["log_assignment"], stride, tol=1e-4)
    check_la_digest(z, pred["line_log_assignment"], 97, prefix="train.line_", tol=1e-4)
    np.testing.assert_allclose(pred["raw_line_scores"].detach().cpu().flatten(1)[:, ::97].numpy(),
                               z["train.raw_line_scores_sample"], rtol=1e-4, atol=1e-4)
    for k, la in (("matches0", "log_assignment"), ("line_matches0", "line_log_assignment")):
        n = assert_disagreements_are_ties(pred[la], pred[k], z["train." + k], 0.2)
        print("gluestick config5 fp32:", k, "rows that differ from the reference's, all near-ties:", n)
        assert n <= 3
    _check_losses(z, losses, 1e-4)
    _fp32_grads(z, grads, norm_tol=3e-4, sample_tol=1.5e-3)      # 2x the measured 1.3e-4 / 7.3e-4


def test_gluestick_config5_bf16_train_step_bounds():
    z, pred, losses, grads = _gs_step(bf16=True)
    _bf16_report("gluestick config5", z,
                 [("log_assignment", pred["log_assignment"], int(z["meta"][5]), "train."),
                  ("line_log_assignment", pred["line_log_assignment"], 97, "train.line_")], losses, grads, GS_BF16)


# ------------------------------------------------------------------------------------------ decisive goldens (round 5)
def _sharp_model(kind):
    if kind == "superglue":
        from glue_factory_amd.matchers.superglue import SuperGlue
        z, params, data, nl, iters = sg_config_inputs("superglue_sharp")
        model = SuperGlue({"num_sinkhorn_iterations": iters})
    else:
        from glue_factory_amd.matchers.gluestick import GlueStick
        z, params, data, nl = gs_config_inputs("gluestick_sharp")
        model = GlueStick({})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert float(z["margins"].min()) > 1.5          # nats, in the reference's own output (train and eval)
    return z, model.cuda(), _cuda(data)


_SHARP_KEYS = {"superglue": ("matches0", "matches1"), "gluestick": ("matches0", "matches1", "line_matches0", "line_matches1")}


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("kind", ["superglue", "gluestick"])
def test_sharp_golden_integer_outputs_are_bit_exact(kind, bf16):
    """matches0/1 (and line_matches0/1) equal to the reference's on 100 % of the rows and columns -- matched AND -1
    entries -- in train mode (BatchNorm batch statistics over the B=2 batch) and eval mode; fp32 additionally holds the
    log-assignment digest, every loss entry and the gradients to the reference."""
    z, model, cdata = _sharp_model(kind)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    for k in _SHARP_KEYS[kind]:
        got = pred[k].cpu().numpy()
        assert (got == -1).any() and (got >= 0).any()
        np.testing.assert_array_equal(got, z["train." + k], err_msg=f"{kind} train {k}")
    stride = int(z["meta"][5])
    items = [("log_assignment", pred["log_assignment"], stride, "train.")]
    if kind == "gluestick":
        items.append(("line_log_assignment", pred["line_log_assignment"], 97, "train.line_"))
    worst = 0.0
    for name, la, st, prefix in items:
        mx, p99, mean = la_digest_error(z, la, st, prefix)
        worst = max(worst, mx)
        print(f"{kind} sharp {'bf16' if bf16 else 'fp32'} {name}: max|d| {mx:.2e} p99 {p99:.2e} mean {mean:.2e} "
              f"(decision margin {float(z['margins'].min()):.2f})")
    assert worst < 0.25 * float(z["margins"].min())       # the decisions cannot flip: |d| << margin / 2
    if not bf16:
        for name, la, st, prefix in items:
            check_la_digest(z, la, st, prefix=prefix, tol=1e-4)
        _check_losses(z, losses, 1e-4)
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        # (measured: SuperGlue 4.5e-5 / 1.6e-3, GlueStick 3.0e-4 / 4.3e-3 -- the damped case has tiny gradients in front of 18 layers.
        # The yardstick for the 6.5e-3: the CPU oracle -- fp32 torch ops, the reference's own arithmetic family -- differs from the
        # reference's golden by 2.98e-4 (norm) / 4.25e-3 (sample) on the SAME tensor, lenc.encoder.0.weight, at 8 and at 3 threads
        # (round 6, tests/test_oracle_golden.py's inputs): the fp32 conditioning of this gradient, not a kernel error.)
        _fp32_grads(z, grads, norm_tol=6e-4, sample_tol=2.5e-3 if kind == "superglue" else 6.5e-3)
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pe = model(cdata)
    for k in _SHARP_KEYS[kind]:
        np.testing.assert_array_equal(pe[k].cpu().numpy(), z["eval." + k], err_msg=f"{kind} eval {k}")


# ------------------------------------------------------------------------------ train-mode BatchNorm at B=4, N=2048
def _oracle_vs_hip(tag, model, params, data, oracle_step, la_keys, bf16=False):
    """One train step (BatchNorm on batch statistics) of the HIP module vs the CPU oracle on the same B=4 batch."""
    model.load_state_dict(params, strict=True)
    model = model.cuda().train()
    cdata = _cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    pred_o, loss_o, grads_o = oracle_step(params, odata)
    for k in la_keys:
        got, ref = pred[k].detach().float().cpu(), pred_o[k].detach()
        print(f"{tag}: max |d {k}| {float((got - ref).abs().max()):.2e} (values up to {float(ref.abs().max()):.1f})")
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    for k, v in loss_o.items():
        if torch.is_tensor(v) and k in losses:
            np.testing.assert_allclose(losses[k].detach().float().cpu().numpy(), v.detach().numpy(), rtol=1e-4, atol=1e-4, err_msg=k)
    errs = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        r = grads_o[k].double()
        errs[k] = (0.0, float((p.grad.double().cpu() - r).norm() / r.norm().clamp(min=1e-30)), float(r.norm()))
    sig = significant_grads(errs)
    worst = max((e[1], k) for k, e in sig.items())
    print(f"{tag}: worst relative gradient error {worst} over {len(sig)} tensors")
    assert worst[0] < 4e-3, worst


def test_superglue_train_mode_batchnorm_b4_n2048_vs_oracle():
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs
    from oracle import superglue_oracle as sgo
    iters = 20          # (the B=4 oracle's unrolled Sinkhorn keeps 2 GB of autograd state per 20 iterations)
    params = sgo.init_params(256, gnn_layers=18, seed=161)
    data = make_pairs(4, 2048, dim=256, size=(1024, 1024), seed=162)
    names = ["self", "cross"] * 9
    _oracle_vs_hip("superglue B=4 N=2048 train-BN", SuperGlue({"num_sinkhorn_iterations": iters}), params, data,
                   lambda p, d: sgo.train_step_grads(p, d, names, iters), ("log_assignment",))


def test_gluestick_train_mode_batchnorm_b4_n2048_vs_oracle():
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs
    from oracle import gluestick_oracle as gso
    params = gso.init_params(256, gnn_layers=18, inter=None, seed=163)
    data = make_point_line_pairs(4, 2048, 512, dim=256, size=(1024, 1024), seed=164)
    names = ["self", "cross"] * 9
    _oracle_vs_hip("gluestick B=4 2048+512 train-BN", GlueStick({}), params, data,
                   lambda p, d: gso.train_step_grads(p, d, names, inter=None), ("log_assignment", "line_log_assignment"))
