"""End-to-end parity of the HIP LightGlue (through the plugin surface) against the CPU oracle and
the reference-generated golden vectors.  fp32 mode: 1e-4; bf16 mode: loose, reported."""
import numpy as np
import pytest
import torch

from conftest import golden_data, load_golden

pytestmark = pytest.mark.gpu

from oracle import lightglue_oracle as lgo  # noqa: E402


def _model(params, n_layers, **kw):
    from glue_factory_amd.base_model import get_model
    M = get_model("glue_factory_amd.matchers.lightglue")
    model = M({"n_layers": n_layers, "filter_threshold": 0.0, **kw})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.cuda()


def _to_cuda(d):
    from glue_factory_amd.synthetic import to_device
    return to_device(d, "cuda")


def _margin_mask(la, k):
    """rows whose top-2 gap in the oracle exceeds k (arg-max is discontinuous at ties)."""
    top2 = la.topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]) > k


def test_golden_d256_eval_and_train_step():
    z = load_golden("lightglue_d256")
    meta = z["meta"]
    L, dim, heads, seed = int(meta[3]), int(meta[4]), int(meta[5]), int(meta[6])
    params = lgo.init_params(L, dim, heads, seed=seed)
    data = golden_data(z)
    model = _model(params, L)
    cdata = _to_cuda(data)
    model.eval()
    with torch.no_grad():
        pe = model(cdata)
    np.testing.assert_allclose(pe["log_assignment"].cpu().numpy(), z["eval.log_assignment"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(pe["matches0"].cpu().numpy(), z["eval.matches0"])
    np.testing.assert_array_equal(pe["matches1"].cpu().numpy(), z["eval.matches1"])
    np.testing.assert_allclose(pe["matching_scores0"].cpu().numpy(), z["eval.matching_scores0"], rtol=1e-3, atol=1e-6)

    model.train()
    pred = model(cdata)
    losses, metrics = model.loss(pred, {**pred, **cdata})
    assert metrics == {}
    losses["total"].mean().backward()
    for k in ("log_assignment", "ref_descriptors0", "ref_descriptors1"):
        np.testing.assert_allclose(pred[k].detach().cpu().numpy(), z["train." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    for k in [k[5:] for k in z if k.startswith("loss.")]:
        np.testing.assert_allclose(losses[k].detach().cpu().numpy(), z["loss." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    checked = 0
    for k, p in model.named_parameters():
        assert p.grad is not None, f"{k} got no gradient (DDP would hang)"
        if "gradnorm." + k in z:
            ref = float(z["gradnorm." + k][0])
            assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * ref + 1e-7, (k, float(p.grad.norm()), ref)
            checked += 1
        if "grad." + k in z:
            ref = z["grad." + k]
            sc = max(np.abs(ref).max(), 1e-9)
            np.testing.assert_allclose(p.grad.cpu().numpy() / sc, ref / sc, rtol=1e-3, atol=1e-3, err_msg=k)
    assert checked > 40


@pytest.mark.parametrize("n0,n1", [(192, 192), (150, 201)])
def test_train_step_vs_oracle_fp32(n0, n1):
    from glue_factory_amd.synthetic import make_pairs
    L = 3
    params = lgo.init_params(L, 256, 4, seed=n0)
    data = make_pairs(2, n0, n1, dim=256, size=(640, 480), seed=n1)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    pred_o, loss_o, grads_o = lgo.train_step_grads(params, odata, L, 4)
    model = _model(params, L).train()
    cdata = _to_cuda(data)
    pred = model(cdata)
    losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    torch.testing.assert_close(pred["log_assignment"].cpu(), pred_o["log_assignment"].detach(), rtol=1e-4, atol=1e-4)
    ok = _margin_mask(pred_o["log_assignment"].detach()[:, :-1, :-1], 1e-3)
    assert torch.equal(pred["matches0"].cpu()[ok], pred_o["matches0"][ok])
    for k, v in loss_o.items():
        torch.testing.assert_close(losses[k].detach().cpu(), v.detach(), rtol=1e-4, atol=1e-4, msg=lambda m: f"{k}: {m}")
    for k, p in model.named_parameters():
        ref = grads_o[k]
        sc = max(ref.abs().max().item(), 1e-9)
        torch.testing.assert_close(p.grad.cpu() / sc, ref / sc, rtol=2e-3, atol=2e-3, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("bf16", [False, True])
def test_sift_style_configuration_vs_reference_golden(bf16):
    """configs/sift+lightglue_{homography,megadepth}.yaml: `input_dim: 128` (a 128 -> 256 `input_proj` in front of the
    transformer, lightglue.py:343-346) and `add_scale_ori: true` (keypoint scale and orientation join the positional
    encoding: posenc.Wr is [32, 4], lightglue.py:348-350, 426-443) -- eval and train step against the vectors the
    reference produced (tests/golden/lightglue_sift.npz), fp32 at 1e-4, bf16 with the small-configuration bounds."""
    from conftest import load_golden
    z = load_golden("lightglue_sift")
    batch, n0, n1, L, seed = (int(v) for v in z["meta"])
    params = lgo.init_params(L, 256, 4, input_dim=128, seed=seed, pos_dim=4)
    from glue_factory_amd.matchers.lightglue import LightGlue
    model = LightGlue({"n_layers": L, "input_dim": 128, "add_scale_ori": True, "filter_threshold": 0.0})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys and model.posenc.Wr.weight.shape == (32, 4)
    model = model.cuda()
    data = {k[5:]: torch.from_numpy(z[k]).cuda() for k in z if k.startswith("data.") and "image_size" not in k}
    data["view0"] = {"image_size": torch.from_numpy(z["data.image_size0"]).cuda()}
    data["view1"] = {"image_size": torch.from_numpy(z["data.image_size1"]).cuda()}
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pe = model(data)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(data)
        losses, _ = model.loss(pred, {**pred, **data})
    losses["total"].mean().backward()
    la = pred["log_assignment"].detach().float().cpu().numpy()
    if not bf16:
        np.testing.assert_allclose(pe["log_assignment"].float().cpu().numpy(), z["eval.log_assignment"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(la, z["train.log_assignment"], rtol=1e-4, atol=1e-4)
        ok = _margin_mask(torch.from_numpy(z["train.log_assignment"])[:, :-1, :-1], 1e-3).numpy()
        np.testing.assert_array_equal(pred["matches0"].cpu().numpy()[ok], z["train.matches0"][ok])
        for k in [k[5:] for k in z if k.startswith("loss.")]:
            np.testing.assert_allclose(losses[k].detach().float().cpu().numpy(), z["loss." + k], rtol=1e-4, atol=1e-4, err_msg=k)
        for k, p in model.named_parameters():
            ref = float(z["gradnorm." + k][0])
            assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * ref + 1e-7, (k, float(p.grad.norm()), ref)
            if "grad." + k in z:
                r = z["grad." + k]
                sc = max(np.abs(r).max(), 1e-9)
                np.testing.assert_allclose(p.grad.cpu().numpy() / sc, r / sc, rtol=1e-3, atol=1e-3, err_msg=k)
    else:
        err = np.abs(la - z["train.log_assignment"])
        print(f"sift-style bf16: max |d log_assignment| {err.max():.3f} mean {err.mean():.4f}")
        assert err.max() <= 0.04 and err.mean() <= 0.007          # (measured 0.021 / 0.0032)
        for k in [k[5:] for k in z if k.startswith("loss.")]:
            np.testing.assert_allclose(losses[k].detach().float().cpu().numpy(), z["loss." + k], rtol=6e-3, atol=6e-3, err_msg=k)
        for k, p in model.named_parameters():
            ref = float(z["gradnorm." + k][0])
            assert abs(float(p.grad.double().norm()) - ref) <= 0.04 * ref + 1e-6, (k, float(p.grad.norm()), ref)


@pytest.mark.parametrize("bf16", [False, True])
def test_train_step_large_ragged_keypoint_counts_vs_oracle(bf16):
    """N0 = 2000, N1 = 1777 (neither a multiple of 64 nor equal): the non-EVEN attention instantiations, the
    register-resident GEMM for M % 64 != 0, the ragged tiles of the head / loss kernels and the unstacked (two-image) layer
    code AT benchmark-like sizes, where the small ragged cases above only touch a tile or two.  fp32 at 1e-4; bf16 with the
    bounds of the N = 2048 configuration (tests/test_gpu_baseline_configs.py)."""
    from glue_factory_amd.synthetic import make_pairs
    L, n0, n1 = 2, 2000, 1777
    params = lgo.init_params(L, 256, 4, seed=91)
    data = make_pairs(1, n0, n1, dim=256, size=(1024, 1024), seed=92)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    pred_o, loss_o, grads_o = lgo.train_step_grads(params, odata, L, 4)
    model = _model(params, L).train()
    cdata = _to_cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    la, la_o = pred["log_assignment"].detach().float().cpu(), pred_o["log_assignment"].detach()
    assert la.shape == (1, n0 + 1, n1 + 1)
    if not bf16:
        torch.testing.assert_close(la, la_o, rtol=1e-4, atol=1e-4)
        ok = _margin_mask(la_o[:, :-1, :-1], 1e-3)
        assert torch.equal(pred["matches0"].cpu()[ok], pred_o["matches0"][ok])
        for k, v in loss_o.items():
            torch.testing.assert_close(losses[k].detach().cpu(), v.detach(), rtol=1e-4, atol=1e-4, msg=lambda m: f"{k}: {m}")
        for k, p in model.named_parameters():
            ref = grads_o[k]
            sc = max(ref.abs().max().item(), 1e-9)
            torch.testing.assert_close(p.grad.cpu() / sc, ref / sc, rtol=2e-3, atol=2e-3, msg=lambda m: f"{k}: {m}")
    else:
        err = (la - la_o).abs()
        print(f"ragged 2000 x 1777 bf16: max |d log_assignment| {float(err.max()):.3f} mean {float(err.mean()):.4f}")
        assert float(err.max()) <= 0.05 and float(err.mean()) <= 0.008       # (measured 0.022 / 0.0034 at L = 2)
        for k, v in loss_o.items():
            torch.testing.assert_close(losses[k].detach().float().cpu(), v.detach(), rtol=5e-3, atol=5e-3, msg=lambda m: f"{k}: {m}")
        worst = 0.0
        for k, p in model.named_parameters():
            ref = grads_o[k].double()
            worst = max(worst, float((p.grad.double().cpu() - ref).norm() / ref.norm().clamp(min=1e-30)))
        print(f"   worst per-parameter gradient error {worst:.4f}")
        assert worst <= 0.035


@pytest.mark.parametrize("dim,heads", [(128, 4), (256, 2)])
@pytest.mark.parametrize("bf16", [False, True])
def test_train_step_other_head_dims_vs_oracle(dim, heads, bf16):
    """head_dim 32 (descriptor_dim 128 / 4 heads) and 128 (256 / 2): the generic attention kernels under the same model code --
    fp32 against the oracle at 1e-4, bf16 with the bounds of the 64-wide configuration.  (fp32 at head_dim 128 does not fit the
    LDS of the generic kernels: that combination raises from the library and is skipped here.)"""
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs
    if dim // heads == 128 and not bf16:
        pytest.skip("fp32 at head_dim 128: GF_ERR_UNSUPPORTED (LDS), see tests/test_gpu_kernels.py::test_attention_head_dim_limits")
    L = 2
    params = lgo.init_params(L, dim, heads, seed=dim + heads)
    data = make_pairs(2, 160, 131, dim=dim, size=(640, 480), seed=5)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    pred_o, loss_o, grads_o = lgo.train_step_grads(params, odata, L, heads)
    model = LightGlue({"n_layers": L, "descriptor_dim": dim, "input_dim": dim, "num_heads": heads, "filter_threshold": 0.0})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.cuda().train()
    cdata = _to_cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    tol = 0.1 if bf16 else 1e-4
    torch.testing.assert_close(pred["log_assignment"].float().cpu(), pred_o["log_assignment"].detach(), rtol=tol, atol=tol)
    for k, v in loss_o.items():
        torch.testing.assert_close(losses[k].detach().float().cpu(), v.detach(), rtol=5e-3 if bf16 else 1e-4, atol=5e-3 if bf16 else 1e-4,
                                   msg=lambda m: f"{k}: {m}")
    worst = 0.0
    for k, p in model.named_parameters():
        ref = grads_o[k]
        rel = float((p.grad.cpu().double() - ref.double()).norm() / ref.double().norm().clamp(min=1e-30))
        if float(ref.norm()) > 1e-6:
            worst = max(worst, rel)
    print(f"LightGlue dim {dim} / {heads} heads ({'bf16' if bf16 else 'fp32'}): worst per-tensor gradient error {worst:.2e}")
    assert worst < (0.05 if bf16 else 5e-4)


def test_train_step_bf16_close_to_oracle():
    """perf mode (autocast bf16): same plumbing, looser numbers; reports the error."""
    from glue_factory_amd.synthetic import make_pairs
    L = 3
    params = lgo.init_params(L, 256, 4, seed=2)
    data = make_pairs(2, 256, dim=256, size=(640, 480), seed=3)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    pred_o, loss_o, grads_o = lgo.train_step_grads(params, odata, L, 4)
    model = _model(params, L).train()
    cdata = _to_cuda(data)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    assert pred["ref_descriptors0"].dtype == torch.bfloat16
    err = (pred["log_assignment"].cpu() - pred_o["log_assignment"].detach()).abs().max().item()
    rel = ((losses["total"].cpu() - loss_o["total"].detach()).abs() / loss_o["total"].detach().abs()).max().item()
    print(f"bf16: max|dlog_assignment|={err:.3e} rel loss err={rel:.3e}")
    assert err < 0.5 and rel < 2e-2
    cos = []
    for k, p in model.named_parameters():
        a, b = p.grad.cpu().flatten().double(), grads_o[k].flatten().double()
        cos.append(torch.nn.functional.cosine_similarity(a, b, dim=0).item())
    assert min(cos) > 0.9, min(cos)


def test_eval_adaptive_depth_and_width():
    """Eval-only early stop / point pruning (b == 1): neutral settings reproduce the plain path; biased
    token / matchability heads trigger the stop and prune points, whose matches come back as -1."""
    from glue_factory_amd.synthetic import make_pairs
    L = 3
    params = lgo.init_params(L, 256, 4, seed=7)
    data = _to_cuda(make_pairs(1, 160, 200, dim=256, size=(640, 480), seed=8))
    plain = _model(params, L).eval()
    with torch.no_grad():
        ref = plain(data)
    # random-weight confidences (~0.5) never reach the thresholds: nothing stops, nothing is pruned
    neutral = _model(params, L, depth_confidence=0.95, width_confidence=0.99).eval()
    with torch.no_grad():
        out = neutral(data)
    assert out["stop_layer"] == L - 1
    torch.testing.assert_close(out["log_assignment"], ref["log_assignment"], rtol=1e-4, atol=1e-4)
    assert torch.equal(out["matches0"], ref["matches0"]) and torch.equal(out["prune0"], torch.full_like(out["prune0"], L))
    # confident tokens -> stop after the first layer
    p2 = dict(params)
    p2["token_confidence.0.token.0.bias"] = torch.tensor([8.0])
    stop = _model(p2, L, depth_confidence=0.9).eval()
    with torch.no_grad():
        o2 = stop(data)
    assert o2["stop_layer"] == 0 and o2["log_assignment"].shape == (1, 161, 201)
    # pruning: keep a point only if sigmoid(matchability) > 0.5  (about half of them)
    prune = _model(params, L, width_confidence=0.5).eval()
    with torch.no_grad():
        o3 = prune(data)
    kept0 = (o3["prune0"] == L).sum().item()
    assert 0 < kept0 < 160 and o3["matches0"].shape == (1, 160)
    assert (o3["matches0"][o3["prune0"] < L] == -1).all()
    valid = o3["matches0"] > -1
    assert (o3["matches1"][0, o3["matches0"][valid]] == valid.nonzero()[:, 1]).all()


def test_fixed_length_positives_equal_dense_nonzero():
    """gt_assignment_col0 (no dense scan, no host sync) and nonzero(gt_assignment) give the same step."""
    from glue_factory_amd.synthetic import make_pairs
    L = 3
    params = lgo.init_params(L, 256, 4, seed=11)
    data = _to_cuda(make_pairs(3, 320, dim=256, size=(640, 480), seed=12))
    assert "gt_assignment_col0" in data
    col = data["gt_assignment_col0"]
    dense = torch.zeros_like(data["gt_assignment"])
    dense.scatter_(2, col.clamp(min=0)[..., None], (col >= 0)[..., None])
    assert torch.equal(dense, data["gt_assignment"])
    out = []
    for drop in (False, True):
        model = _model(params, L).train()
        d = {k: v for k, v in data.items() if not (drop and k == "gt_assignment_col0")}
        pred = model(d)
        losses, _ = model.loss(pred, {**pred, **d})
        losses["total"].mean().backward()
        out.append((losses, {k: p.grad.clone() for k, p in model.named_parameters()}))
    for k in out[0][0]:
        torch.testing.assert_close(out[0][0][k], out[1][0][k], rtol=1e-5, atol=1e-6, msg=lambda m: f"{k}: {m}")
    for k in out[0][1]:
        sc = max(out[1][1][k].abs().max().item(), 1e-9)
        torch.testing.assert_close(out[0][1][k] / sc, out[1][1][k] / sc, rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")


def test_train_step_hipgraph_replay_equals_eager():
    """TrainStep(graph=True): the whole step (forward, fused loss, backward, fused Adam) captured once and replayed
    gives the same weights and losses as launching it kernel by kernel, on changing batches; a NaN batch is skipped
    inside the replay (device-side flag) and an eval forward afterwards sees the updated weights."""
    from glue_factory_amd.synthetic import make_pairs
    from glue_factory_amd.train_step import TrainStep
    L = 2
    params = lgo.init_params(L, 256, 4, seed=21)
    batches = [_to_cuda(make_pairs(2, 256, dim=256, size=(640, 480), seed=30 + i)) for i in range(6)]
    results = []
    for use_graph in (False, True):
        model = _model(params, L).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
        step = TrainStep(model, opt, amp_dtype=torch.bfloat16, graph=use_graph, graph_warmup=2)
        losses = []
        for i, b in enumerate(batches):
            if i == 4:       # poisoned batch: must not touch the weights in either mode
                b = dict(b, descriptors0=b["descriptors0"].clone())
                b["descriptors0"][0, 0, 0] = float("nan")
            out = step(b)
            losses.append(out["total"].clone())
        assert step.skipped == 1
        assert (step._g is not None) == use_graph
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            ev = model(batches[0])["log_assignment"].clone()
        results.append((losses, {k: p.detach().clone() for k, p in model.named_parameters()}, ev))
    (l0, p0, e0), (l1, p1, e1) = results
    for i, (a, b) in enumerate(zip(l0, l1)):
        if i == 4:
            assert torch.isnan(a).any() and torch.isnan(b).any()
            continue
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m: f"step {i}: {m}")
    for k in p0:
        torch.testing.assert_close(p0[k], p1[k], rtol=1e-5, atol=1e-6, msg=lambda m: f"{k}: {m}")
    torch.testing.assert_close(e0, e1, rtol=1e-4, atol=1e-4)


def test_train_step_takes_batches_from_pinned_host_memory():
    """A DataLoader hands the batch over in (pinned) host memory (the reference moves it with batch_to_device,
    train.py:462-469).  TrainStep accepts it as it is -- eager steps move it, replays copy it straight into the captured
    graph's input buffers, ONE graph for host and device batches alike -- and trains exactly as on device-resident batches."""
    from glue_factory_amd.synthetic import make_pairs
    from glue_factory_amd.train_step import TrainStep
    L = 2
    params = lgo.init_params(L, 256, 4, seed=23)
    cpu_batches = [make_pairs(2, 256, dim=256, size=(640, 480), seed=50 + i) for i in range(6)]

    def pin(d):
        return {k: pin(v) if isinstance(v, dict) else (v.pin_memory() if torch.is_tensor(v) else v) for k, v in d.items()}

    results = []
    for source in ("device", "host", "mixed"):
        model = _model(params, L).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
        step = TrainStep(model, opt, amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
        losses, graphs = [], set()
        for i, b in enumerate(cpu_batches):
            host = source == "host" or (source == "mixed" and i % 2 == 1)
            losses.append(step(pin(b) if host else _to_cuda(b))["total"].clone())
            if step._g is not None:
                graphs.add(id(step._g[1]))
        assert len(graphs) == 1                      # captured once; host batches replay the same graph
        assert all(t.is_cuda for t in step.static_inputs().values() if torch.is_tensor(t))
        results.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    for (losses, ps), source in zip(results[1:], ("host", "mixed")):
        print(f"{source} vs device batches: max loss difference {float((losses - results[0][0]).abs().max()):.2e}")
        # (same tolerance as graph vs eager above: the loss accumulators are atomic sums; another batch's loss differs by >1e-2)
        torch.testing.assert_close(losses, results[0][0], rtol=1e-5, atol=1e-5)
        for a, b in zip(ps, results[0][1]):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_hipgraph_replays_with_eager_work_between_and_no_host_sync():
    """The benchmark's pipeline pattern: eager kernels (there: the extractor) queued between replays of the captured
    step, and no host synchronisation for dozens of steps.  Every replay's reported losses must equal the eager
    run's.  (Regression: with a hipMemsetAsync inside the capture, ROCm 7.2 dropped the memset node from some
    replays in exactly this pattern and the loss accumulators read stale pool memory -- csrc/gf_common.h.)"""
    from glue_factory_amd.synthetic import make_pairs
    from glue_factory_amd.train_step import TrainStep
    L = 2
    params = lgo.init_params(L, 256, 4, seed=22)
    batch = _to_cuda(make_pairs(2, 256, dim=256, size=(640, 480), seed=40))
    img = torch.rand(8, 64, 512, 512, device="cuda")
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()

    def eager_work():
        with torch.no_grad():
            y = img
            for _ in range(6):
                y = torch.relu(conv(y))
            return y.mean()

    runs = []
    for use_graph in (False, True):
        model = _model(params, L).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
        step = TrainStep(model, opt, amp_dtype=torch.bfloat16, graph=use_graph, graph_warmup=2)
        losses = []
        for _ in range(40):
            eager_work()
            losses.append(step(batch)["total"].mean())      # no .item(): nothing synchronises inside the loop
        torch.cuda.synchronize()
        runs.append(torch.stack(losses))
    torch.testing.assert_close(runs[0], runs[1], rtol=1e-4, atol=1e-4)
