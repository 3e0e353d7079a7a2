"""Sinkhorn kernels and the SuperGlue plugin module vs the reference-generated golden vectors
and the CPU oracle (fp32: 1e-4)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_sinkhorn_fwd_bwd_vs_reference_golden():
    from glue_factory_amd import ops
    z = load_golden("superglue_ot")
    scores = torch.from_numpy(z["scores"]).cuda().requires_grad_(True)
    alpha = torch.from_numpy(z["alpha"]).cuda().requires_grad_(True)
    b, m, n = scores.shape
    Z = torch.cat([torch.cat([scores, alpha.expand(b, m, 1)], -1),
                   torch.cat([alpha.expand(b, 1, n), alpha.expand(b, 1, 1)], -1)], 1)
    out = ops.sinkhorn(Z, int(z["iters"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=1e-4, atol=1e-4)
    (out * torch.from_numpy(z["G"]).cuda()).sum().backward()
    sc = np.abs(z["gscores"]).max()
    np.testing.assert_allclose(scores.grad.cpu().numpy() / sc, z["gscores"] / sc, rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(float(alpha.grad), float(z["galpha"]), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("B,M,N,T", [(3, 130, 97, 7), (1, 1, 5, 3), (2, 300, 300, 0), (2, 2100, 2050, 2),
                                     # N % 256 == 0: the chip-resident sweeps (csrc/sinkhorn_resident.h), ragged M, one and
                                     # several pairs per launch, rows in registers only / registers + LDS
                                     (3, 300, 256, 6), (2, 1000, 1024, 4), (5, 700, 512, 25), (11, 1500, 1280, 3),
                                     (1, 2048, 2048, 1)])
def test_sinkhorn_shapes_vs_oracle(B, M, N, T, monkeypatch):
    from glue_factory_amd import ops
    monkeypatch.setenv("GF_SINKHORN_RESIDENT", "2")        # resident sweeps wherever they fit, small batches included
    from oracle import sinkhorn_oracle as so
    g = torch.Generator().manual_seed(M + N + T)
    scores = torch.randn(B, M, N, generator=g) * 3
    alpha = torch.tensor(1.0)
    Zc = so.couplings(scores.double(), alpha.double()).requires_grad_(True)
    lmu, lnu, norm = so.marginals(M, N, Zc)
    ref, _, _ = so.sinkhorn(Zc, lmu, lnu, T)
    ref = ref - norm
    G = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * G).sum().backward()
    Zd = Zc.detach().float().cuda().requires_grad_(True)
    out = ops.sinkhorn(Zd, T)
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=2e-4)
    (out * G.float().cuda()).sum().backward()
    sc = Zc.grad.abs().max().item()
    torch.testing.assert_close(Zd.grad.cpu().double() / sc, Zc.grad / sc, rtol=1e-3, atol=5e-4)
    # a transport plan: rows of exp(out) (without the bin row) sum to 1 after column-last iterations
    if T > 0:
        col = (out.detach()[:, :, :-1] ).exp().sum(1)
        torch.testing.assert_close(col, torch.ones_like(col), rtol=1e-3, atol=1e-3)


def test_sinkhorn_resident_equals_streaming(monkeypatch):
    """The chip-resident sweeps and the streaming kernels are two schedules of the same recurrence: outputs and gradients
    agree to rounding (summation order of the column partials differs), at a multi-chunk geometry with LDS-resident rows."""
    from glue_factory_amd import ops
    g = torch.Generator().manual_seed(5)
    Z = (torch.randn(9, 1801, 2049, generator=g) * 2).cuda()
    G = torch.randn(9, 1801, 2049, generator=g).cuda()
    res = []
    for mode in ("0", "1"):                                 # 9 pairs: two launches of 5 and 4 pairs by default
        monkeypatch.setenv("GF_SINKHORN_RESIDENT", mode)
        z = Z.clone().requires_grad_(True)
        out = ops.sinkhorn(z, 20)
        (out * G).sum().backward()
        res.append((out.detach(), z.grad))
    do = float((res[0][0] - res[1][0]).abs().max())
    dg = float((res[0][1] - res[1][1]).abs().max()) / float(res[0][1].abs().max())
    print(f"resident vs streaming Sinkhorn: max |d out| {do:.2e}, max |d dZ| / max|dZ| {dg:.2e}")
    assert do < 2e-5 and dg < 2e-5


def _sg_data(z, device):
    t = lambda k: torch.from_numpy(z["data." + k]).to(device)  # noqa: E731
    d = {k: t(k) for k in ("keypoints0", "keypoints1", "descriptors0", "descriptors1", "keypoint_scores0",
                           "keypoint_scores1", "gt_assignment", "gt_matches0", "gt_matches1")}
    d["view0"] = {"image_size": t("image_size0")}
    d["view1"] = {"image_size": t("image_size1")}
    return d


def test_superglue_module_vs_reference_golden():
    from glue_factory_amd.base_model import get_model
    from oracle import superglue_oracle as sgo
    z = load_golden("superglue_d256")
    nl, iters, seed = int(z["meta"][3]), int(z["meta"][4]), int(z["meta"][5])
    params = sgo.init_params(256, gnn_layers=nl, seed=seed)
    SG = get_model("glue_factory_amd.matchers.superglue")
    model = SG({"GNN_layers": ["self", "cross"] * (nl // 2), "num_sinkhorn_iterations": iters})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.cuda()
    data = _sg_data(z, "cuda")
    model.eval()
    with torch.no_grad():
        pe = model(data)
    np.testing.assert_allclose(pe["log_assignment"].cpu().numpy(), z["eval.log_assignment"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(pe["matches0"].cpu().numpy(), z["eval.matches0"])
    np.testing.assert_allclose(pe["matching_scores0"].cpu().numpy(), z["eval.matching_scores0"], rtol=1e-3, atol=1e-6)
    model.train()
    pred = model(data)
    losses, metrics = model.loss(pred, {**pred, **data})
    assert metrics == {}
    losses["total"].mean().backward()
    np.testing.assert_allclose(pred["log_assignment"].detach().cpu().numpy(), z["train.log_assignment"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pred["sinkhorn_cost"].detach().cpu().numpy(), z["train.sinkhorn_cost"], rtol=1e-4, atol=1e-4)
    for k in [k[5:] for k in z if k.startswith("loss.")]:
        np.testing.assert_allclose(losses[k].detach().cpu().numpy(), z["loss." + k], rtol=1e-4, atol=1e-4, err_msg=k)
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        ref = float(z["gradnorm." + k][0])
        assert abs(float(p.grad.double().norm()) - ref) <= 3e-3 * ref + 1e-6, (k, float(p.grad.norm()), ref)
        if "grad." + k in z and np.abs(z["grad." + k]).max() > 1e-5:
            # (a conv bias in front of a train-mode BatchNorm has an exactly-zero gradient: noise only)
            sc = np.abs(z["grad." + k]).max()
            # the query-bias gradient sums dS rows that cancel exactly (softmax Jacobian): fp32 noise
            tol = 2e-2 if k.endswith("attn.proj.0.bias") else 5e-3
            a, r = p.grad.cpu().numpy() / sc, z["grad." + k] / sc
            bad = np.abs(a - r) > tol * (1 + np.abs(r))
            # a ReLU input within fp32 noise of 0 may flip its mask: allow isolated outliers
            assert bad.mean() <= 0.01, (k, float(np.abs(a - r).max()))
    # BatchNorm running statistics were updated like the reference's (two calls per layer)
    assert int(model.kenc.encoder[1].num_batches_tracked) == 2


def test_superglue_unequal_counts_and_bf16():
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import superglue_oracle as sgo
    names = ["self", "cross"]
    params = sgo.init_params(256, gnn_layers=2, seed=4)
    data = make_pairs(2, 90, 70, dim=256, size=(640, 480), seed=9)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        ref = sgo.forward(params, odata, names, 10, training=False)
    model = SuperGlue({"GNN_layers": names, "num_sinkhorn_iterations": 10}).cuda().eval()
    model.load_state_dict(params)
    cdata = to_device(data, "cuda")
    with torch.no_grad():
        pred = model(cdata)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pb = model(cdata)
    torch.testing.assert_close(pred["log_assignment"].cpu(), ref["log_assignment"], rtol=1e-4, atol=1e-4)
    err = (pb["log_assignment"].cpu() - ref["log_assignment"]).abs().max().item()
    print("superglue bf16 max|dlog_assignment| =", err)
    assert err < 0.5


@pytest.mark.parametrize("dim", [128])
def test_superglue_other_descriptor_dim(dim):
    """descriptor_dim 128 = 4 heads of 32 channels: the generic attention kernels and the non-256 GEMM shapes under the same
    module code; fp32 forward against the oracle, then a bf16 train step (finite loss and gradients for every parameter)."""
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import superglue_oracle as sgo
    names = ["self", "cross"]
    params = sgo.init_params(dim, gnn_layers=2, seed=4)
    data = make_pairs(2, 96, 80, dim=dim, size=(640, 480), seed=9)
    odata = dict(data, image_size0=data["view0"]["image_size"], image_size1=data["view1"]["image_size"])
    with torch.no_grad():
        ref = sgo.forward(params, odata, names, 10, training=False)
    model = SuperGlue({"GNN_layers": names, "num_sinkhorn_iterations": 10, "descriptor_dim": dim}).cuda().eval()
    model.load_state_dict(params)
    cdata = to_device(data, "cuda")
    with torch.no_grad():
        pred = model(cdata)
    torch.testing.assert_close(pred["log_assignment"].cpu(), ref["log_assignment"], rtol=1e-4, atol=1e-4)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pt = model(cdata)
        losses, _ = model.loss(pt, {**pt, **cdata})
    losses["total"].mean().backward()
    assert torch.isfinite(losses["total"]).all()
    for k, p_ in model.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), k


def test_superglue_train_step_hipgraph_replay_equals_eager():
    """SuperGlue's loss gathers its positives through gt_assignment_col0 (no nonzero() scan, no host read), so the
    whole step captures: replay == kernel-by-kernel on changing batches, and fixed-length == dense-scan losses."""
    from glue_factory_amd import ops
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    from oracle import superglue_oracle as sgo
    params = sgo.init_params(256, gnn_layers=4, seed=3)
    batches = [to_device(make_pairs(2, 192, dim=256, size=(640, 480), seed=40 + i), "cuda") for i in range(5)]
    results = []
    for use_graph in (False, True):
        model = SuperGlue({"GNN_layers": ["self", "cross"] * 2, "num_sinkhorn_iterations": 10})
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
    # the fixed-length positives are the dense matrix's
    d = batches[0]
    la = torch.randn(2, 193, 193, device="cuda")
    s1, n1 = ops.nll_positive_terms(la, d)
    s2, n2 = ops.nll_positive_terms(la, {k: v for k, v in d.items() if k != "gt_assignment_col0"})
    torch.testing.assert_close(s1, s2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(n1, n2)


def test_nll_node_gradient_and_sums_handed_to_sinkhorn():
    """ops.nll_terms (positives + dustbin terms of superglue.py:322-352 as ONE autograd node): same values and the same
    dense gradient as the gather / slice expressions it replaces, and the row / column sums it hands to the Sinkhorn
    backward (instead of two more sweeps of the dense gradient) are the sums of that gradient."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, M, N = 3, 37, 41
    la = torch.randn(B, M + 1, N + 1, device="cuda", generator=g, requires_grad=True)
    col0 = torch.randint(-1, N, (B, M), device="cuda", generator=g)
    for b in range(B):                                   # one positive per column at most
        seen = set()
        for i in range(M):
            c = int(col0[b, i])
            if c >= 0 and c in seen:
                col0[b, i] = -1
            seen.add(c)
    neg0 = (col0 < 0).float()
    neg1 = torch.ones(B, N, device="cuda")
    for b in range(B):
        neg1[b, col0[b][col0[b] >= 0]] = 0.0
    data = {"gt_assignment_col0": col0}
    pos, npos, neg = ops.nll_terms(la, data, neg0, neg1)
    valid = col0 >= 0
    picked = la[:, :-1, :].gather(2, col0.clamp(min=0)[..., None]).squeeze(-1)
    pos_ref = (picked * valid.float()).sum(1)
    neg_ref = (la[:, :-1, -1] * neg0).sum(1) + (la[:, -1, :-1] * neg1).sum(1)
    torch.testing.assert_close(pos, pos_ref)
    torch.testing.assert_close(neg, neg_ref)
    assert torch.equal(npos, valid.sum(1).float())
    w0, w1 = torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)
    G_ref, = torch.autograd.grad((pos_ref * w0 + neg_ref * w1).sum(), la)
    seen = []
    orig = ops._known_sums

    class Probe(torch.autograd.Function):                # stands where the Sinkhorn backward stands
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, G):
            seen.append((G, orig(G)))
            return G

    la2 = la.detach().clone().requires_grad_(True)
    pos2, _, neg2 = ops.nll_terms(Probe.apply(la2), data, neg0, neg1)
    (pos2 * w0 + neg2 * w1).sum().backward()
    torch.testing.assert_close(la2.grad, G_ref)
    G, sums = seen[0]
    assert sums is not None
    torch.testing.assert_close(sums[0], G.sum(2))
    torch.testing.assert_close(sums[1], G.sum(1))


@pytest.mark.parametrize("graph", [False, True])
def test_skipped_step_updates_batchnorm_statistics_once(graph):
    """The reference `continue`s BEFORE its backward on a non-finite loss (train.py:477-480): the activation-checkpointed GNN
    blocks are not re-run, so their BatchNorm buffers take ONE update on such a step (num_batches_tracked + 2: two images) and
    two on a normal one (+ 4).  TrainStep always runs its backward; the replays are gated by the step's device-side flag
    (ops.REPLAY_GATE, gf_bn_replay_running's `skip`) -- eager and inside a replayed hipGraph."""
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    torch.manual_seed(3)
    model = SuperGlue({"GNN_layers": ["self", "cross"], "num_sinkhorn_iterations": 5}).cuda().train()
    good = to_device(make_pairs(2, 128, dim=256, size=(640, 480), seed=21), "cuda")
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in good.items()}
    # a loss that is NaN although every activation is finite: a NaN among the ground-truth weights... the NLL reads the
    # log-assignment at the positives only, so poison the couplings through ONE keypoint score instead -- and keep the
    # statistics comparable by looking at the COUNTERS, which do not depend on the values
    bad["keypoint_scores0"][0, 0] = float("nan")
    step = TrainStep(model, FusedAdam(model.parameters(), lr=1e-4), amp_dtype=None, graph=graph, graph_warmup=2)
    gnn_bn, kenc_bn = model.gnn.layers[0].mlp[1], model.kenc.encoder[1]
    n_steps = 4 if graph else 1                  # (graph: two eager steps, the capture, one replay -- all on the good batch)
    for _ in range(n_steps):
        step(good)
    torch.cuda.synchronize()
    assert step.skipped == 0
    g0, k0 = int(gnn_bn.num_batches_tracked), int(kenc_bn.num_batches_tracked)
    assert g0 == 4 * n_steps and k0 == 2 * n_steps
    step(bad)
    torch.cuda.synchronize()
    assert step.skipped == 1
    assert int(kenc_bn.num_batches_tracked) == k0 + 2
    assert int(gnn_bn.num_batches_tracked) == g0 + 2, "a skipped step must not replay the checkpointed blocks' update"
    step.close()
