"""The CPU oracles (oracle/*_oracle.py) against the REFERENCE modules, live, over a sweep of small configurations (build
container only: skipped where /root/reference is absent).  tests/test_oracle_golden.py pins the oracles to committed
reference vectors at a handful of configurations; the GPU tests then use the oracles as the checker at many more shapes
(ragged counts, other head widths, batch sizes).  This sweep closes that gap from the other side: for every configuration
below the reference module and the oracle, fed the same seeded parameters and batch, must agree on the eval and train-mode
log-assignments, every loss entry and every parameter's gradient (fp32, north_star's 1e-4; gradient norms 2e-3)."""
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gluefactory")),
                                reason="reference checkout not present (GPU box)")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def ref_path():
    stubs = os.path.join(ROOT, "oracle", "stubs")
    added = [p for p in (stubs, REF) if p not in sys.path]
    sys.path[:0] = [stubs]
    sys.path.append(REF)
    yield
    for p in added:
        if p in sys.path:
            sys.path.remove(p)


def _ref_step(model, data):
    model.eval()
    with torch.no_grad():
        pe = model(data)
    model.train()
    pred = model(data)
    losses = model.loss(pred, {**pred, **data})
    losses = losses[0] if isinstance(losses, tuple) else losses
    losses["total"].mean().backward()
    return pe, pred, losses, {k: p.grad for k, p in model.named_parameters() if p.grad is not None}


def _check(pe_o, pred_o, losses_o, grads_o, pe, pred, losses, grads, la_keys, grad_tol=2e-3):
    for k in la_keys:
        np.testing.assert_allclose(pe_o[k].numpy(), pe[k].numpy(), **TOL, err_msg="eval " + k)
        np.testing.assert_allclose(pred_o[k].detach().numpy(), pred[k].detach().numpy(), **TOL, err_msg="train " + k)
    for k, v in losses.items():
        if torch.is_tensor(v):
            np.testing.assert_allclose(losses_o[k].detach().numpy().reshape(v.shape), v.detach().numpy(), **TOL, err_msg=k)
    assert set(grads_o) == set(grads)
    scale = max(float(g.double().norm()) for g in grads.values())
    for k, g in grads.items():
        ref, got = float(g.double().norm()), float(grads_o[k].double().norm())
        assert abs(got - ref) <= grad_tol * ref + 1e-6 * scale, (k, got, ref)


@pytest.mark.parametrize("layers,dim,heads,batch,n0,n1,seed", [(1, 256, 4, 2, 40, 40, 1), (2, 256, 4, 1, 70, 53, 2), (3, 128, 4, 3, 33, 64, 3),
                                                            (2, 512, 4, 2, 48, 31, 4), (2, 256, 2, 2, 25, 25, 5)])
def test_lightglue_oracle_equals_the_reference(ref_path, layers, dim, heads, batch, n0, n1, seed):
    from gluefactory.models.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs
    from oracle import lightglue_oracle as lgo
    params = lgo.init_params(layers, dim, heads, seed=seed)
    data = make_pairs(batch, n0, n1, dim=dim, size=(640, 480), seed=seed + 10)
    model = LightGlue({"n_layers": layers, "descriptor_dim": dim, "input_dim": dim, "num_heads": heads, "weights": None,
                       "flash": False, "checkpointed": False, "filter_threshold": 0.0})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    pe, pred, losses, grads = _ref_step(model, data)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        pe_o = lgo.forward(params, odata, layers, heads, filter_threshold=0.0, training=False)
    pred_o, losses_o, grads_o = lgo.train_step_grads(params, odata, layers, heads)
    _check(pe_o, pred_o, losses_o, grads_o, pe, pred, losses, grads, ["log_assignment"])
    np.testing.assert_array_equal(pe_o["matches0"].numpy(), pe["matches0"].numpy())


@pytest.mark.parametrize("names,iters,batch,n0,n1,seed", [(["self", "cross"], 5, 2, 40, 40, 6), (["self", "cross"] * 2, 12, 1, 61, 47, 7),
                                                          (["cross", "self", "cross"], 30, 3, 24, 35, 8)])
def test_superglue_oracle_equals_the_reference(ref_path, names, iters, batch, n0, n1, seed):
    from gluefactory_nonfree.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs
    from oracle import superglue_oracle as sgo
    params = sgo.init_params(256, gnn_layers=len(names), seed=seed)
    data = make_pairs(batch, n0, n1, dim=256, size=(640, 480), seed=seed + 10)
    data["view0"]["image"] = torch.zeros(batch, 1, 8, 8)
    data["view1"]["image"] = torch.zeros(batch, 1, 8, 8)
    model = SuperGlue({"weights": None, "GNN_layers": names, "num_sinkhorn_iterations": iters, "filter_threshold": 0.0})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    pe, pred, losses, grads = _ref_step(model, data)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        pe_o = sgo.forward(params, odata, names, iters, filter_threshold=0.0, training=False)
    pred_o, losses_o, grads_o = sgo.train_step_grads(params, odata, names, iters)
    _check(pe_o, pred_o, losses_o, grads_o, pe, pred, losses, grads, ["log_assignment"], grad_tol=3e-3)


@pytest.mark.parametrize("nl,inter,batch,nk,nlines,seed", [(2, None, 2, 30, 8, 9), (4, [0], 1, 44, 12, 10), (4, [0, 1], 2, 20, 5, 11)])
def test_gluestick_oracle_equals_the_reference(ref_path, nl, inter, batch, nk, nlines, seed):
    from gluefactory.models.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs
    from oracle import gluestick_oracle as gso
    names = ["self", "cross"] * (nl // 2)
    params = gso.init_params(256, gnn_layers=nl, inter=inter, seed=seed)
    data = make_point_line_pairs(batch, nk, nlines, dim=256, size=(320, 240), seed=seed + 10)
    model = GlueStick({"weights": None, "GNN_layers": names, "inter_supervision": inter, "filter_threshold": 0.0})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    pe, pred, losses, grads = _ref_step(model, data)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        pe_o = gso.forward(params, odata, names, filter_threshold=0.0, training=False, inter=inter)
    pred_o, losses_o, grads_o = gso.train_step_grads(params, odata, names, inter=inter)
    keys = ["log_assignment", "line_log_assignment"]
    _check(pe_o, pred_o, losses_o, grads_o, pe, pred, losses, grads, keys, grad_tol=3e-3)


@pytest.mark.parametrize("gamma,balancing", [(0.7, 0.3), (0.0, 0.8), (2.0, 0.5)])
def test_lightglue_oracle_loss_options_equal_the_reference(ref_path, gamma, balancing):
    """`loss.gamma` (gamma^(L-i-1) deep-supervision weights; <= 0 selects i + 1) and `loss.nll_balancing`
    (lightglue.py:328-332, 598-628; losses.py:9-46) through the oracle's loss()."""
    from gluefactory.models.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs
    from oracle import lightglue_oracle as lgo
    L = 3
    params = lgo.init_params(L, 256, 4, seed=21)
    data = make_pairs(2, 50, 37, dim=256, size=(640, 480), seed=22)
    model = LightGlue({"n_layers": L, "weights": None, "flash": False, "filter_threshold": 0.0,
                       "loss": {"gamma": gamma, "fn": "nll", "nll_balancing": balancing}}).train()
    model.load_state_dict(params, strict=True)
    pred = model(data)
    losses, _ = model.loss(pred, {**pred, **data})
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    pred_o = lgo.forward(params, odata, L, 4, filter_threshold=0.0, training=True)
    losses_o = lgo.loss(params, pred_o, odata, gamma=gamma, balancing=balancing)
    for k in ("total", "last", "confidence", "nll_pos", "nll_neg", "row_norm"):
        np.testing.assert_allclose(losses_o[k].detach().numpy(), losses[k].detach().numpy(), **TOL, err_msg=k)
