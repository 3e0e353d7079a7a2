"""GlueStick plugin module (points + lines) vs the reference-generated golden vectors (fp32: 1e-4)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _data(z, device):
    d = {k[5:]: torch.from_numpy(z[k]).to(device) for k in z if k.startswith("data.")}
    d["view0"] = {"image_size": d.pop("image_size0")}
    d["view1"] = {"image_size": d.pop("image_size1")}
    return d


@pytest.mark.parametrize("name", ["gluestick_d256", "gluestick_lineattn"])
def test_gluestick_module_vs_reference_golden(name):
    """Whole GlueStick step (HIP line kernels: gf_line_csr / gather / segsum / expand; gf_gemm MLPs) vs the reference's
    outputs, losses and gradient norms; ``gluestick_lineattn`` runs LineLayer with line_attention=True."""
    from glue_factory_amd.base_model import get_model
    from oracle import gluestick_oracle as gso
    z = load_golden(name)
    nl, seed = int(z["meta"][3]), int(z["meta"][4])
    inter = [int(v) for v in z["meta"][5:]]
    params = gso.init_params(256, gnn_layers=nl, inter=inter, seed=seed)
    attn = name.endswith("lineattn")
    if attn:
        gso.add_line_attention_params(params, seed=seed + 2)
    chk = float(sum(v.double().abs().sum() for v in params.values()))
    assert abs(chk - float(z["param_checksum"][0])) < 1e-6 * chk
    GS = get_model("glue_factory_amd.matchers.gluestick")
    model = GS({"GNN_layers": ["self", "cross"] * (nl // 2), "inter_supervision": inter, "line_attention": attn})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.cuda()
    data = _data(z, "cuda")
    model.eval()
    with torch.no_grad():
        pe = model(data)
        _, metrics = model.loss(pe, {**pe, **data})
    assert {"match_recall", "line_match_precision", f"line_{inter[0]}_accuracy"} <= set(metrics)
    for k in ("log_assignment", "line_log_assignment", "raw_line_scores"):
        np.testing.assert_allclose(pe[k].cpu().numpy(), z["eval." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    np.testing.assert_array_equal(pe["matches0"].cpu().numpy(), z["eval.matches0"])
    np.testing.assert_array_equal(pe["line_matches0"].cpu().numpy(), z["eval.line_matches0"])
    model.train()
    pred = model(data)
    losses, _ = model.loss(pred, {**pred, **data})
    losses["total"].mean().backward()
    for k in ("log_assignment", "line_log_assignment", f"line_{inter[0]}_log_assignment"):
        np.testing.assert_allclose(pred[k].detach().cpu().numpy(), z["train." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    for k in [k[5:] for k in z if k.startswith("loss.")]:
        np.testing.assert_allclose(losses[k].detach().cpu().numpy(), z["loss." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        ref = float(z["gradnorm." + k][0])
        assert abs(float(p.grad.double().norm()) - ref) <= 5e-3 * ref + 1e-5, (k, float(p.grad.norm()), ref)


def test_gluestick_unequal_counts_and_bf16():
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs, to_device
    from oracle import gluestick_oracle as gso
    names = ["self", "cross"]
    params = gso.init_params(256, gnn_layers=2, seed=6)
    data = make_point_line_pairs(2, 70, 20, dim=256, size=(640, 480), seed=8)
    # drop a few keypoints / lines from image 1 so the counts differ
    for k in ("keypoints1", "descriptors1", "keypoint_scores1"):
        data[k] = data[k][:, :-5]
    for k in ("lines1", "lines_junc_idx1", "line_scores1"):
        data[k] = data[k][:, :-3]
    data["lines_junc_idx1"] = data["lines_junc_idx1"].clamp(max=2 * 17 - 1)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        ref = gso.forward(params, odata, names, training=False)
    model = GlueStick({"GNN_layers": names}).cuda().eval()
    model.load_state_dict(params)
    cdata = to_device(data, "cuda")
    with torch.no_grad():
        pred = model(cdata)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pb = model(cdata)
    torch.testing.assert_close(pred["log_assignment"].cpu(), ref["log_assignment"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pred["line_log_assignment"].cpu(), ref["line_log_assignment"], rtol=1e-4, atol=1e-4)
    err = (pb["log_assignment"].cpu() - ref["log_assignment"]).abs().max().item()
    print("gluestick bf16 max|dlog_assignment| =", err)
    assert err < 0.5


def test_gluestick_other_descriptor_dim():
    """descriptor_dim 128 = 4 heads of 32 channels (generic attention kernels, non-256 GEMM shapes): fp32 forward against the
    oracle, then a bf16 train step with finite loss and gradients for every parameter."""
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs, to_device
    from oracle import gluestick_oracle as gso
    names, dim = ["self", "cross"], 128
    params = gso.init_params(dim, gnn_layers=2, seed=6)
    data = make_point_line_pairs(2, 70, 20, dim=dim, size=(640, 480), seed=8)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        ref = gso.forward(params, odata, names, training=False)
    model = GlueStick({"GNN_layers": names, "descriptor_dim": dim, "input_dim": dim}).cuda().eval()
    model.load_state_dict(params)
    cdata = to_device(data, "cuda")
    with torch.no_grad():
        pred = model(cdata)
    torch.testing.assert_close(pred["log_assignment"].cpu(), ref["log_assignment"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pred["line_log_assignment"].cpu(), ref["line_log_assignment"], rtol=1e-4, atol=1e-4)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pt = model(cdata)
        losses, _ = model.loss(pt, {**pt, **cdata})
    losses["total"].mean().backward()
    assert torch.isfinite(losses["total"]).all()
    for k, p_ in model.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), k


def test_gluestick_train_step_hipgraph_replay_equals_eager():
    """GlueStick (points + lines, HIP line layers and line head) captured as one hipGraph: replay == eager."""
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    from oracle import gluestick_oracle as gso
    params = gso.init_params(256, gnn_layers=4, seed=9)
    batches = [to_device(make_point_line_pairs(2, 96, 24, dim=256, size=(640, 480), seed=50 + i), "cuda") for i in range(5)]
    assert "gt_line_assignment_col0" in batches[0]
    results = []
    for use_graph in (False, True):
        model = GlueStick({"GNN_layers": ["self", "cross"] * 2})
        model.load_state_dict(params)
        model = model.cuda().train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
        step = TrainStep(model, opt, amp_dtype=torch.bfloat16, graph=use_graph, graph_warmup=2)
        losses = [step(b)["total"].clone() for b in batches]
        assert (step._g is not None) == use_graph
        results.append((losses, {k: p.detach().clone() for k, p in model.named_parameters()}))
    for i, (a, b) in enumerate(zip(results[0][0], results[1][0])):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m: f"step {i}: {m}")
    for k in results[0][1]:
        torch.testing.assert_close(results[0][1][k], results[1][1][k], rtol=1e-5, atol=1e-6, msg=lambda m: f"{k}: {m}")


def test_line_graph_of_any_size_equals_the_lds_kernel():
    """ops.line_graph: graphs beyond gf_line_csr's LDS capacity (4096 endpoints / 8192 junctions) are built by a stable sort;
    on a graph both paths take, they agree entry for entry -- and a 6000-endpoint graph goes through the line kernels."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    idx = torch.randint(0, 300, (3, 1024), device="cuda", generator=g)
    o1, s1 = ops.line_graph(idx, 300)
    o2, s2 = ops._line_graph_sorted(idx, 300)
    assert torch.equal(o1, o2) and torch.equal(s1, s2)
    big = torch.randint(0, 9000, (2, 6000), device="cuda", generator=g)
    order, seg = ops.line_graph(big, 9000)
    x = torch.randn(2, 9000, 64, device="cuda", generator=g).requires_grad_(True)
    out = ops.rows_gather(x, big, order, seg)
    torch.testing.assert_close(out, x.gather(1, big[..., None].expand(-1, -1, 64)))
    out.sum().backward()
    ref = torch.zeros(2, 9000, device="cuda").scatter_add_(1, big, torch.ones(2, 6000, device="cuda"))
    torch.testing.assert_close(x.grad[..., 0], ref)
