"""Shared checker for the compact BASELINE-configuration goldens (tests/golden/lightglue_config1.npz,
lightglue_n2048_l9.npz: the REFERENCE LightGlue run by oracle/gen_golden.py at config 1 -- B=4, N=512, L=4 --
and at the config-2 shape N=2048, L=9 with B=1).  The same function holds the CPU oracle (tests/test_oracle_golden.py)
and the HIP path (tests/test_gpu_baseline_configs.py) to those vectors."""
import numpy as np
import torch

from conftest import load_golden


def config_inputs(name):
    """(golden dict, params, data, n_layers): weights and synthetic pairs regenerated from the stored seed;
    checksums prove they are the tensors the reference saw."""
    from glue_factory_amd.synthetic import make_pairs
    from oracle import lightglue_oracle as lgo
    z = load_golden(name)
    batch, n, L, seed, w, h, _ = (int(v) for v in z["meta"])
    if "sharp" in z:        # the decisive case (every arg-max separated from its runner-up by z["margins"])
        params, data = lgo.sharp_case(batch, n, L, seed, (w, h), *(float(v) for v in z["sharp"]))
    else:
        params = lgo.init_params(L, 256, 4, seed=seed)
        data = make_pairs(batch, n, dim=256, size=(w, h), seed=seed + 1)
    chk = float(sum(v.double().abs().sum() for v in params.values()))
    assert abs(chk - float(z["param_checksum"][0])) < 1e-9 * chk
    dchk = float(sum(v.double().abs().sum() for v in data.values() if torch.is_tensor(v) and v.is_floating_point()))
    assert abs(dchk - float(z["data_checksum"][0])) < 1e-9 * dchk
    return z, params, data, L


def _np(t):
    return t.detach().float().cpu().numpy()


def check_train(z, pred, losses, grads, tol=1e-4, grad_tol=2e-3, tie_margin=1e-3, exact_matches=False):
    """pred / losses / grads (name -> tensor) of one train step vs the reference's.
    exact_matches: matches0 / matches1 must equal the reference's on 100 % of the rows (goldens with decisive margins)."""
    stride = int(z["meta"][6])
    la = pred["log_assignment"].detach().float().cpu()
    np.testing.assert_allclose(la.flatten(1)[:, ::stride].numpy(), z["train.la_sample"], rtol=tol, atol=tol)
    n1 = la.shape[2]
    np.testing.assert_allclose(la.double().sum(2).float().numpy() / n1, z["train.la_rowsum"] / n1, rtol=tol, atol=tol)
    np.testing.assert_allclose(la.double().sum(1).float().numpy() / n1, z["train.la_colsum"] / n1, rtol=tol, atol=tol)
    np.testing.assert_allclose(la[:, :-1, :-1].max(2).values.numpy(), z["train.rowmax"], rtol=tol, atol=tol)
    # matches: bit-exact wherever the reference's arg-max is not a near-tie
    top2 = la[:, :-1, :-1].topk(2, dim=-1).values
    clear_r = ((top2[..., 0] - top2[..., 1]) > tie_margin)
    top2c = la[:, :-1, :-1].transpose(1, 2).topk(2, dim=-1).values
    clear_c = ((top2c[..., 0] - top2c[..., 1]) > tie_margin)
    m0 = pred["matches0"].cpu().numpy()
    ref0 = z["train.matches0"]
    # a match also depends on the partner column's arg-max; only rows whose own and partner decisions are clear
    partner_ok = np.ones_like(ref0, dtype=bool)
    for b in range(ref0.shape[0]):
        j = np.clip(ref0[b], 0, None)
        partner_ok[b] = clear_c[b].numpy()[j] | (ref0[b] < 0)
    ok = clear_r.numpy() & partner_ok & clear_c.numpy().all()  # all columns clear -> every mutual check is stable
    if exact_matches:
        assert "margins" in z and float(z["margins"].min()) > 100 * tie_margin
    if exact_matches or (clear_c.numpy().all() and clear_r.numpy().all()):
        np.testing.assert_array_equal(m0, ref0)
        np.testing.assert_array_equal(pred["matches1"].cpu().numpy(), z["train.matches1"])
    else:
        agree = (m0 == ref0)
        assert agree[clear_r.numpy()].mean() > 0.999, agree[clear_r.numpy()].mean()
    np.testing.assert_allclose(_np(pred["matching_scores0"]), z["train.matching_scores0"], rtol=10 * tol, atol=1e-7)
    loss_keys = [k[5:] for k in z if k.startswith("loss.")]
    assert set(loss_keys) >= {"total", "last", "nll_pos", "nll_neg", "confidence", "row_norm"}
    for k in loss_keys:
        np.testing.assert_allclose(_np(losses[k]), z["loss." + k], rtol=tol, atol=tol, err_msg=k)
    worst = (0.0, None)
    n_norm = 0
    for k, g in grads.items():
        ref = float(z["gradnorm." + k][0])
        got = float(g.double().norm())
        rel = abs(got - ref) / max(ref, 1e-12)
        worst = max(worst, (rel, k))
        assert rel <= grad_tol, (k, got, ref)
        n_norm += 1
        if "grad." + k in z:
            r = z["grad." + k]
            sc = max(np.abs(r).max(), 1e-12)
            np.testing.assert_allclose(_np(g) / sc, r / sc, rtol=grad_tol, atol=grad_tol, err_msg=k)
    assert n_norm == sum(1 for k in z if k.startswith("gradnorm."))
    return worst


def check_eval(z, pred, metrics=None, tol=1e-4):
    """Eval-mode matches / scores / matcher_metrics vs the reference's (matches compared where stable)."""
    m0 = pred["matches0"].cpu().numpy()
    agree = (m0 == z["eval.matches0"]).mean()
    assert agree > 0.999, agree
    s0 = _np(pred["matching_scores0"])
    same = m0 == z["eval.matches0"]
    np.testing.assert_allclose(s0[same], z["eval.matching_scores0"][same], rtol=10 * tol, atol=1e-7)
    if metrics is not None:
        for k in ("match_recall", "match_precision", "accuracy", "average_precision"):
            np.testing.assert_allclose(_np(metrics[k]), z["metric." + k], rtol=1e-3, atol=2e-3, err_msg=k)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] (SuperGlue, N=2048, 18 GNN layers, 100 Sinkhorn iterations) and configs[4] (GlueStick, 2048
# keypoints + 512 lines) at B=1: compact goldens written by oracle/gen_golden.py from the REFERENCE modules
# (superglue_config4.npz, gluestick_config5.npz).
def sg_config_inputs(name="superglue_config4"):
    from glue_factory_amd.synthetic import make_pairs
    from oracle import superglue_oracle as sgo
    z = load_golden(name)
    batch, n, nl, iters, seed, _ = (int(v) for v in z["meta"])
    if "sharp" in z:        # the decisive case: every mutual-NN decision taken by z["margins"]
        params, data = sgo.sharp_case(batch, n, nl, seed, (1024, 1024), *(float(v) for v in z["sharp"]))
    else:
        params = sgo.init_params(256, gnn_layers=nl, seed=seed)
        data = make_pairs(batch, n, dim=256, size=(1024, 1024), seed=seed + 1)
    chk = float(sum(v.double().abs().sum() for v in params.values()))
    assert abs(chk - float(z["param_checksum"][0])) < 1e-9 * chk
    dchk = float(sum(v.double().abs().sum() for v in data.values() if torch.is_tensor(v) and v.is_floating_point()))
    assert abs(dchk - float(z["data_checksum"][0])) < 1e-9 * dchk
    return z, params, data, nl, iters


def gs_config_inputs(name="gluestick_config5"):
    from glue_factory_amd.synthetic import make_point_line_pairs
    from oracle import gluestick_oracle as gso
    z = load_golden(name)
    batch, n_kpts, n_lines, nl, seed, _ = (int(v) for v in z["meta"])
    if "sharp" in z:
        params, data = gso.sharp_case(batch, n_kpts, n_lines, nl, seed, (1024, 1024), *(float(v) for v in z["sharp"]))
    else:
        params = gso.init_params(256, gnn_layers=nl, inter=None, seed=seed)
        data = make_point_line_pairs(batch, n_kpts, n_lines, dim=256, size=(1024, 1024), seed=seed + 1)
    chk = float(sum(v.double().abs().sum() for v in params.values()))
    assert abs(chk - float(z["param_checksum"][0])) < 1e-9 * chk
    dchk = float(sum(v.double().abs().sum() for v in data.values() if torch.is_tensor(v) and v.is_floating_point()))
    assert abs(dchk - float(z["data_checksum"][0])) < 1e-9 * dchk
    return z, params, data, nl


def check_la_digest(z, la, stride, prefix="train.", tol=1e-4):
    """Strided sample, row / column sums (per entry) and row maxima of a log-assignment vs the reference's."""
    la = la.detach().float().cpu()
    np.testing.assert_allclose(la.flatten(1)[:, ::stride].numpy(), z[prefix + "la_sample"], rtol=tol, atol=tol)
    n1 = la.shape[2]
    np.testing.assert_allclose(la.double().sum(2).float().numpy() / n1, z[prefix + "la_rowsum"] / n1, rtol=tol, atol=tol)
    np.testing.assert_allclose(la.double().sum(1).float().numpy() / la.shape[1], z[prefix + "la_colsum"] / la.shape[1],
                               rtol=tol, atol=tol)
    np.testing.assert_allclose(la[:, :-1, :-1].max(2).values.numpy(), z[prefix + "rowmax"], rtol=tol, atol=tol)


def la_digest_error(z, la, stride, prefix="train."):
    """(max, p99, mean) of |d log_assignment| on the reference's strided sample (for the bf16 bounds)."""
    d = np.abs(la.detach().float().cpu().flatten(1)[:, ::stride].numpy() - z[prefix + "la_sample"])
    return float(d.max()), float(np.quantile(d, 0.99)), float(d.mean())


def grad_digest_errors(z, grads, sample=2048):
    """name -> (relative gradient-norm error, relative error of the strided gradient sample) vs the reference's."""
    out = {}
    for k, g in grads.items():
        ref = float(z["gradnorm." + k][0])
        got = float(g.double().norm())
        flat = g.detach().float().cpu().flatten()
        st = max(1, flat.numel() // sample)
        s, r = flat[::st][:sample].double().numpy(), z["gradsample." + k].astype(np.float64)
        rn = np.linalg.norm(r)
        out[k] = (abs(got - ref) / max(ref, 1e-30), float(np.linalg.norm(s - r) / max(rn, 1e-30)), ref)
    assert len(out) == sum(1 for k in z if k.startswith("gradnorm."))
    return out


def significant_grads(errs):
    """Drop the gradients that are analytically zero and hold only rounding noise in the reference too (a conv bias in
    front of a train-mode BatchNorm, key / value / merge biases that BatchNorm or the softmax cancels): those whose
    reference norm is below 1e-3 of their own layer's weight-gradient norm (the noise sits at 1e-7 .. 2e-4 of it: the
    damped `*_sharp` cases have small weight gradients; a genuine bias gradient is of the order of its weight's)."""
    keep = {}
    for k, e in errs.items():
        if k.endswith(".bias"):
            w = errs.get(k[:-5] + ".weight")
            if w is not None and e[2] < 1e-3 * w[2]:
                continue
        keep[k] = e
    return keep


def matches_agreement(m, ref):
    m = m.cpu().numpy() if torch.is_tensor(m) else m
    return float((m == ref).mean())


def assert_disagreements_are_ties(la, m, ref, threshold, margin=1e-3):
    """Integer outputs against a golden whose weights are RANDOM (no decisive margins): every row on which our match index
    differs from the reference's must be a decision OUR OWN log-assignment rates a near-tie -- the row's top-2 gap, the top-2
    gap of a candidate column (ours, the reference's, the row's arg-max: the mutual check), or the distance of a candidate's
    probability from the filter threshold is below `margin` (10 x the 1e-4 the log-assignment itself is held to).  A
    disagreement that is none of these fails.  Returns the number of (explained) disagreements."""
    la = la.detach().float().cpu()
    m = m.cpu().numpy() if torch.is_tensor(m) else m
    core = la[:, :-1, :-1]
    t2r = core.topk(2, dim=-1).values
    gap_r = (t2r[..., 0] - t2r[..., 1]).numpy()
    t2c = core.transpose(1, 2).topk(2, dim=-1).values
    gap_c = (t2c[..., 0] - t2c[..., 1]).numpy()
    arg_r = core.argmax(-1).numpy()
    bad = np.argwhere(m != ref)
    for b, i in bad:
        cands = {int(arg_r[b, i])} | {int(j) for j in (m[b, i], ref[b, i]) if j >= 0}
        ok = gap_r[b, i] < margin
        for j in cands:
            ok = ok or gap_c[b, j] < margin or abs(float(core[b, i, j].exp()) - threshold) < margin
        assert ok, (int(b), int(i), int(m[b, i]), int(ref[b, i]), float(gap_r[b, i]), [float(gap_c[b, j]) for j in cands])
    return len(bad)
