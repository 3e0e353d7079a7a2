"""Size-independent properties of the HIP SuperGlue and GlueStick at BASELINE.json's full sizes (configs[3]: N = 2048, 18 GNN
layers, 100 Sinkhorn iterations; configs[4]: 2048 keypoints + 512 lines), where the reference itself (B = 1, CPU) is the
only pinned checker (tests/test_gpu_configs45.py): transport-plan marginals (superglue.py:186-214), probability bounds of the
double softmax (gluestick.py:772-783), mutual consistency of the matches (superglue.py:300-320, gluestick.py:321-376), permutation
equivariance in the second image, image swap = transpose -- in eval mode (BatchNorm on its running statistics), fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N, B = 2048, 2


def _mutual(la, m0, m1, s0, th):
    valid = m0 > -1
    rows = valid.nonzero()
    assert int(valid.sum()) > 0
    assert torch.equal(m1[rows[:, 0], m0[rows[:, 0], rows[:, 1]]], rows[:, 1])
    best = la[:, :-1, :-1].max(2).values.exp()
    torch.testing.assert_close(s0[valid], best[valid], rtol=1e-4, atol=1e-9)
    assert bool((best[valid] > th).all())


@pytest.fixture(scope="module")
def superglue():
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import superglue_oracle as sgo
    model = SuperGlue({"num_sinkhorn_iterations": 100, "filter_threshold": 0.0})     # random weights: no score clears 0.2
    model.load_state_dict(sgo.init_params(256, gnn_layers=18, seed=181), strict=True)
    model = model.cuda().eval()
    data = to_device(make_pairs(B, N, dim=256, size=(1024, 1024), seed=182), "cuda")
    with torch.no_grad():
        pred = model(data)
    return model, data, pred


def test_superglue_full_size_is_a_transport_plan_with_mutual_matches(superglue):
    _, _, pred = superglue
    la = pred["log_assignment"]
    assert la.shape == (B, N + 1, N + 1) and bool(torch.isfinite(la).all())
    P = la.double().exp()
    # out = Z + u + v - norm: after the last (column) update every column of the plan holds its marginal exactly -- 1 for a
    # keypoint, M for the dustbin --, the rows theirs up to the convergence of 100 iterations (superglue.py:206-214)
    torch.testing.assert_close(P[:, :, :-1].sum(1), torch.ones(B, N, dtype=torch.float64, device="cuda"), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(P[:, :, -1].sum(1), torch.full((B,), float(N), dtype=torch.float64, device="cuda"), rtol=1e-4, atol=0)
    torch.testing.assert_close(P[:, :-1].sum(2), torch.ones(B, N, dtype=torch.float64, device="cuda"), rtol=2e-2, atol=2e-2)
    _mutual(la, pred["matches0"], pred["matches1"], pred["matching_scores0"], 0.0)


def test_superglue_full_size_permutation_and_swap(superglue):
    model, data, pred = superglue
    g = torch.Generator().manual_seed(0)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).cuda()
    d2 = dict(data)
    d2["keypoints1"] = data["keypoints1"].gather(1, perm[..., None].expand(-1, -1, 2))
    d2["descriptors1"] = data["descriptors1"].gather(1, perm[..., None].expand(-1, -1, 256))
    d2["keypoint_scores1"] = data["keypoint_scores1"].gather(1, perm)
    with torch.no_grad():
        p2 = model(d2)
    cols = torch.cat([perm, torch.full((B, 1), N, device="cuda")], 1)
    ref = pred["log_assignment"].gather(2, cols[:, None, :].expand(-1, N + 1, -1))
    torch.testing.assert_close(p2["log_assignment"], ref, rtol=1e-3, atol=3e-3)
    d3 = dict(data)
    for k in ("keypoints", "descriptors", "keypoint_scores"):
        d3[k + "0"], d3[k + "1"] = data[k + "1"], data[k + "0"]
    d3["view0"], d3["view1"] = data["view1"], data["view0"]
    with torch.no_grad():
        p3 = model(d3)
    # (the Sinkhorn iteration ends on a column update: the swapped problem ends on what were the rows -- equal at convergence)
    torch.testing.assert_close(p3["log_assignment"], pred["log_assignment"].transpose(1, 2), rtol=1e-2, atol=3e-2)


@pytest.fixture(scope="module")
def gluestick():
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs, to_device
    from oracle import gluestick_oracle as gso
    model = GlueStick({"filter_threshold": 0.0})
    model.load_state_dict(gso.init_params(256, gnn_layers=18, inter=None, seed=183), strict=True)
    model = model.cuda().eval()
    data = to_device(make_point_line_pairs(B, N, 512, dim=256, size=(1024, 1024), seed=184), "cuda")
    with torch.no_grad():
        pred = model(data)
    return model, data, pred


def test_gluestick_full_size_normalisation_and_mutual_matches(gluestick):
    _, _, pred = gluestick
    for key, m, n in (("log_assignment", N + 1024, N + 1024), ("line_log_assignment", 512, 512)):
        la = pred[key]
        assert la.shape == (B, m + 1, n + 1) and bool(torch.isfinite(la).all())
        # A_ij = ((S_ij - r_i) + (S_ij - c_j)) / 2 with the bin inside both normalisers: exp(A_ij) = sqrt(P_row P_col) <= 1, and a
        # bin entry exp(beta - r_i) is the row softmax's bin probability; the corner is 0 (gluestick.py:772-783)
        assert float(la.max()) <= 1e-5 and float(la[:, -1, -1].abs().max()) == 0.0
    _mutual(pred["log_assignment"], pred["matches0"], pred["matches1"], pred["matching_scores0"], 0.0)
    lm0 = pred["line_matches0"]
    if int((lm0 > -1).sum()) > 0:
        _mutual(pred["line_log_assignment"], lm0, pred["line_matches1"], pred["line_matching_scores0"], 0.0)


def test_gluestick_full_size_image_swap_is_transpose(gluestick):
    model, data, pred = gluestick
    d3 = dict(data)
    for k in ("keypoints", "descriptors", "keypoint_scores", "lines", "lines_junc_idx", "line_scores"):
        d3[k + "0"], d3[k + "1"] = data[k + "1"], data[k + "0"]
    d3["view0"], d3["view1"] = data["view1"], data["view0"]
    with torch.no_grad():
        p3 = model(d3)
    torch.testing.assert_close(p3["log_assignment"], pred["log_assignment"].transpose(1, 2), rtol=1e-3, atol=3e-3)
    torch.testing.assert_close(p3["line_log_assignment"], pred["line_log_assignment"].transpose(1, 2), rtol=1e-3, atol=3e-3)
    assert torch.equal(p3["matches0"], pred["matches1"]) or float((p3["matches0"] == pred["matches1"]).float().mean()) > 0.999
