"""HIP-vs-reference parity ON the BASELINE.json configurations, in fp32 (1e-4, north_star) and in bf16 -- the
dtype bench.py times -- with stated per-tensor bounds.

  * config 1  (B=4, N=512,  L=4, 640x480):  full train step vs the reference-generated compact golden
    (tests/golden/lightglue_config1.npz) and, gradient by gradient, vs the CPU oracle;
  * config 2 shape (B=1, N=2048, L=9, 1024^2): the same (golden lightglue_n2048_l9.npz + oracle);
  * the same two in bf16 (autocast): log-assignment / loss / per-tensor relative gradient error printed and bounded;
  * Sinkhorn at config 4's size: N=2048, 100 iterations, forward and backward vs the fp64 oracle;
  * eval-mode matcher metrics and the adaptive depth/width outputs vs reference-generated vectors;
  * lightglue_sharp (N=2048, L=9, decisive margins): matches0 / matches1 bit-exact on 100 % of the rows, fp32 and bf16.
"""
import numpy as np
import pytest
import torch

from config_golden import check_eval, check_train, config_inputs
from conftest import load_golden

pytestmark = pytest.mark.gpu

from oracle import lightglue_oracle as lgo  # noqa: E402

# ---- stated bf16 bounds (bf16 has an 8-bit mantissa: eps = 3.9e-3; errors accumulate over L layers) ----
# measured on MI355X (round 2): config 1: max|dLA| 0.066, mean 0.0096, worst loss entry 1.5e-3, worst per-tensor gradient
# error 1.4 % (median 0.5 %); N=2048/L=9: 0.19, 0.026, 2.9e-3, 1.8 % (median 0.7 %).  Bounds = about 2x that.
BF16_LA_MAX = {"lightglue_config1": 0.09, "lightglue_n2048_l9": 0.25}     # max |d log_assignment| (measured 0.054 / 0.161)
BF16_LA_MEAN = {"lightglue_config1": 0.015, "lightglue_n2048_l9": 0.04}    # mean |d log_assignment| (0.0083 / 0.023)
BF16_LA_P99 = {"lightglue_config1": 0.045, "lightglue_n2048_l9": 0.12}     # 99th percentile (0.027 / 0.075)
BF16_LOSS_REL = 5e-3                                                        # every loss entry, relative
BF16_GRAD_REL = {"lightglue_config1": 0.025, "lightglue_n2048_l9": 0.035}  # ||g - g_ref|| / ||g_ref|| per tensor (0.014 / 0.025)


def _model(params, L, **kw):
    from glue_factory_amd.matchers.lightglue import LightGlue
    model = LightGlue({"n_layers": L, "filter_threshold": 0.0, **kw})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.cuda()


def _cuda(d):
    from glue_factory_amd.synthetic import to_device
    return to_device(d, "cuda")


_ORACLE = {}


def _oracle_step(name):
    """One oracle train step per configuration and test session (7 s at N=2048 on the box's host cores)."""
    if name not in _ORACLE:
        z, params, data, L = config_inputs(name)
        odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
        torch.set_num_threads(min(16, torch.get_num_threads()))
        _ORACLE[name] = lgo.train_step_grads(params, odata, L, 4)
    return _ORACLE[name]


def _hip_step(name, bf16):
    z, params, data, L = config_inputs(name)
    model = _model(params, L).train()
    cdata = _cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, metrics = model.loss(pred, {**pred, **cdata})
    assert metrics == {}
    losses["total"].mean().backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())
    return z, model, cdata, pred, losses, grads


def _rel(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.mark.parametrize("name", ["lightglue_config1", "lightglue_n2048_l9", "lightglue_sharp"])
def test_fp32_train_step_on_baseline_config(name):
    """grad_tol: 2x the measured errors (gradient norms 3.4e-5 / 4.8e-5 vs the reference, per tensor 6.9e-5 / 8.2e-5 vs the
    oracle).  lightglue_sharp: the decisive golden -- matches0 / matches1 equal the reference's on 100 % of the rows."""
    z, model, cdata, pred, losses, grads = _hip_step(name, bf16=False)
    worst = check_train(z, pred, losses, grads, tol=1e-4, grad_tol=2e-4, exact_matches=name == "lightglue_sharp")
    print(f"{name}: worst relative gradient-norm error vs the reference {worst}")
    pred_o, loss_o, grads_o = _oracle_step(name)
    torch.testing.assert_close(pred["log_assignment"].cpu(), pred_o["log_assignment"].detach(), rtol=1e-4, atol=1e-4)
    for k, v in loss_o.items():
        torch.testing.assert_close(losses[k].detach().cpu(), v.detach(), rtol=1e-4, atol=1e-4, msg=lambda m: f"{k}: {m}")
    rels = {k: _rel(grads[k], grads_o[k]) for k in grads_o}
    k_w = max(rels, key=rels.get)
    print(f"{name}: fp32 per-tensor relative gradient error: max {rels[k_w]:.2e} ({k_w}), "
          f"median {sorted(rels.values())[len(rels) // 2]:.2e}")
    assert rels[k_w] < 2e-4, (k_w, rels[k_w])


def test_bf16_matches_bit_exact_on_the_decisive_golden():
    """The benchmarked (bf16) mode on the golden whose every row / column decision has a margin >= 5.4 in the reference's
    own log-assignment (far above the bf16 error, printed): matches0 / matches1 of the train-mode and of the eval-mode
    forward equal the reference's on 100 % of the rows (lightglue.py:293-309)."""
    name = "lightglue_sharp"
    z, model, cdata, pred, losses, grads = _hip_step(name, bf16=True)
    pred_o, loss_o, grads_o = _oracle_step(name)
    err = (pred["log_assignment"].cpu() - pred_o["log_assignment"].detach()).abs()
    print(f"{name} bf16: |d log_assignment| max {err.max():.4f} mean {err.mean():.5f}; reference margins {z['margins'].tolist()}")
    assert float(err.max()) < 0.25 * float(z["margins"].min())
    np.testing.assert_array_equal(pred["matches0"].cpu().numpy(), z["train.matches0"])
    np.testing.assert_array_equal(pred["matches1"].cpu().numpy(), z["train.matches1"])
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pe = model(cdata)
    np.testing.assert_array_equal(pe["matches0"].cpu().numpy(), z["eval.matches0"])
    np.testing.assert_array_equal(pe["matches1"].cpu().numpy(), z["eval.matches1"])
    np.testing.assert_allclose(pe["matching_scores0"].cpu().numpy(), z["eval.matching_scores0"], rtol=0.35, atol=1e-6)


@pytest.mark.parametrize("name", ["lightglue_config1", "lightglue_n2048_l9"])
def test_bf16_train_step_on_baseline_config(name):
    """The benchmarked mode.  Reference = the fp32 oracle step (itself pinned to the reference at this size)."""
    z, model, cdata, pred, losses, grads = _hip_step(name, bf16=True)
    assert pred["ref_descriptors0"].dtype == torch.bfloat16
    pred_o, loss_o, grads_o = _oracle_step(name)
    err = (pred["log_assignment"].cpu() - pred_o["log_assignment"].detach()).abs()
    lrel = {k: float(((losses[k].detach().cpu() - v.detach()).abs() / v.detach().abs().clamp(min=1e-3)).max())
            for k, v in loss_o.items()}
    rels = {k: _rel(grads[k], grads_o[k]) for k in grads_o}
    k_w = max(rels, key=rels.get)
    srt = sorted(rels.values())
    sample = err.flatten()[:: max(1, err.numel() // 2_000_000)]                 # quantile() takes at most 16 M entries
    p99 = float(torch.quantile(sample.detach(), 0.99))
    print(f"{name} bf16: |d log_assignment| max {err.max():.4f} p99 {p99:.4f} mean {err.mean():.5f}; loss rel err "
          f"{ {k: round(v, 5) for k, v in lrel.items()} }; per-tensor relative gradient error max {rels[k_w]:.4f} "
          f"({k_w}) p90 {srt[int(0.9 * len(srt))]:.4f} median {srt[len(srt) // 2]:.4f}")
    assert err.max() < BF16_LA_MAX[name] and err.mean() < BF16_LA_MEAN[name] and p99 < BF16_LA_P99[name]
    for k, v in lrel.items():
        if k in ("num_matchable", "num_unmatchable"):
            assert v == 0.0
        else:
            assert v < BF16_LOSS_REL, (k, v)
    assert rels[k_w] < BF16_GRAD_REL[name], (k_w, rels[k_w])
    # matches: identical wherever the fp32 decision has a clear margin
    la = pred_o["log_assignment"].detach()[:, :-1, :-1]
    top2 = la.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4 * BF16_LA_MAX[name]
    hip_arg = pred["log_assignment"][:, :-1, :-1].argmax(-1).cpu()
    n_rows, n_clear = clear.numel(), int(clear.sum())
    n_diff_all = int((hip_arg != la.argmax(-1)).sum())
    print(f"{name} bf16: row arg-max compared on {n_clear} of {n_rows} rows (margin > {4 * BF16_LA_MAX[name]:.2f}; "
          f"{n_rows - n_clear} excluded as near-ties); rows whose arg-max differs over ALL rows: {n_diff_all}")
    assert torch.equal(hip_arg[clear], la.argmax(-1)[clear])
    # (random weights: almost every row of these goldens is a near-tie, so the clear-margin set is small; the statement that
    # covers ALL rows is the one below) wherever the decisions differ, the reference itself rates the two candidates within
    # twice the log-assignment bound of each other
    gap = la.max(-1).values - la.gather(-1, hip_arg[..., None])[..., 0]
    print(f"{name} bf16: largest reference gap between its own and our row arg-max: {float(gap.max()):.4f}")
    assert float(gap.max()) <= 2 * BF16_LA_MAX[name]
    assert n_diff_all <= 0.06 * n_rows, f"{n_diff_all} of {n_rows} row decisions differ"


@pytest.mark.parametrize("name", ["lightglue_config1", "lightglue_n2048_l9", "lightglue_sharp"])
def test_eval_matches_and_metrics_on_baseline_config(name):
    z, params, data, L = config_inputs(name)
    model = _model(params, L).eval()
    cdata = _cuda(data)
    with torch.no_grad():
        pred = model(cdata)
        losses, metrics = model.loss(pred, {**pred, **cdata})
    check_eval(z, pred, metrics)
    if name == "lightglue_sharp":
        np.testing.assert_array_equal(pred["matches0"].cpu().numpy(), z["eval.matches0"])
        np.testing.assert_array_equal(pred["matches1"].cpu().numpy(), z["eval.matches1"])
    for k in ("total", "nll_pos", "nll_neg", "row_norm"):
        np.testing.assert_allclose(losses[k].cpu().numpy(), z["evalloss." + k], rtol=1e-4, atol=1e-4, err_msg=k)


def test_adaptive_depth_width_vs_reference():
    """Eval-only point pruning / confidence gating (lightglue.py:461-529) against the reference's outputs.  (The
    reference cannot stop early -- it raises on the empty descriptor list -- so stops are covered by
    tests/test_gpu_lightglue.py::test_eval_adaptive_depth_and_width only.)"""
    from glue_factory_amd.synthetic import make_pairs
    z = load_golden("lightglue_adaptive")
    n0, n1, L, seed = (int(v) for v in z["meta"])
    base = lgo.init_params(L, 256, 4, seed=seed)
    data = _cuda(make_pairs(1, n0, n1, dim=256, size=(640, 480), seed=seed + 1))
    cases = sorted({k.split(".")[0] for k in z if k != "meta"})
    assert cases == ["neutral", "prune", "prune_tok"]
    for c in cases:
        params = {k: v.clone() for k, v in base.items()}
        for k in [k for k in z if k.startswith(c + ".edit.")]:
            name = k[len(c) + 6:]
            params[name] = torch.full_like(params[name], float(z[k][0]))
        dc, wc = (float(v) for v in z[c + ".conf"])
        model = _model(params, L, depth_confidence=dc, width_confidence=wc).eval()
        with torch.no_grad():
            out = model(data)
        assert tuple(out["log_assignment"].shape) == z[c + ".log_assignment"].shape, c
        np.testing.assert_array_equal(out["prune0"].cpu().numpy(), z[c + ".prune0"], err_msg=c)
        np.testing.assert_array_equal(out["prune1"].cpu().numpy(), z[c + ".prune1"], err_msg=c)
        np.testing.assert_allclose(out["log_assignment"].cpu().numpy(), z[c + ".log_assignment"], rtol=1e-4, atol=1e-4, err_msg=c)
        np.testing.assert_array_equal(out["matches0"].cpu().numpy(), z[c + ".matches0"], err_msg=c)
        np.testing.assert_array_equal(out["matches1"].cpu().numpy(), z[c + ".matches1"], err_msg=c)
        np.testing.assert_allclose(out["matching_scores0"].cpu().numpy(), z[c + ".matching_scores0"], rtol=1e-3, atol=1e-7, err_msg=c)
        np.testing.assert_allclose(out["matching_scores1"].cpu().numpy(), z[c + ".matching_scores1"], rtol=1e-3, atol=1e-7, err_msg=c)


def test_sinkhorn_config4_size_100_iterations():
    """gf_sinkhorn_fwd / gf_sinkhorn_bwd at SuperGlue's benchmark size (N=2048, 100 iterations, B=1) vs the fp64 oracle."""
    from glue_factory_amd import ops
    from oracle import sinkhorn_oracle as so
    M = N = 2048
    T = 100
    g = torch.Generator().manual_seed(5)
    scores = torch.randn(1, M, N, generator=g) * 2.0
    alpha = torch.tensor(1.0)
    Zc = so.couplings(scores.double(), alpha.double()).requires_grad_(True)
    lmu, lnu, norm = so.marginals(M, N, Zc)
    ref, _, _ = so.sinkhorn(Zc, lmu, lnu, T)
    ref = ref - norm
    G = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * G).sum().backward()
    Zd = Zc.detach().float().cuda().requires_grad_(True)
    out = ops.sinkhorn(Zd, T)
    err = (out.detach().cpu().double() - ref.detach()).abs().max().item()
    (out * G.float().cuda()).sum().backward()
    sc = Zc.grad.abs().max().item()
    gerr = ((Zd.grad.cpu().double() - Zc.grad).abs().max() / sc).item()
    grel = _rel(Zd.grad, Zc.grad)
    print(f"sinkhorn N=2048 T=100: max|out - fp64| = {err:.2e}; max|dZ - fp64|/max|dZ| = {gerr:.2e}; relative L2 {grel:.2e}")
    assert err < 1e-4 and gerr < 1e-3 and grel < 1e-3          # (measured on MI355X: 9.4e-6 forward)
    # converged transport plan: both marginals hold
    P = out.detach().exp()
    torch.testing.assert_close(P[:, :-1, :].sum(2), torch.ones(1, M, device="cuda"), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(P[:, :, :-1].sum(1), torch.ones(1, N, device="cuda"), rtol=2e-3, atol=2e-3)
