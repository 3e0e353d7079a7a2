"""Size-independent properties of the HIP LightGlue at BASELINE.json's full size (N=2048, L=9, d=256),
where the CPU oracle is too slow to be the checker: permutation equivariance, image-swap symmetry,
mutual consistency of the matches, normalisation bounds, fp32-vs-bf16 agreement."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N, L, B = 2048, 9, 2


@pytest.fixture(scope="module")
def setup():
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import lightglue_oracle as lgo
    params = lgo.init_params(L, 256, 4, seed=3)
    model = LightGlue({"n_layers": L, "filter_threshold": 0.0}).cuda().eval()
    model.load_state_dict(params)
    data = to_device(make_pairs(B, N, dim=256, seed=4), "cuda")
    with torch.no_grad():
        pred = model(data)
    return model, data, pred


def test_full_size_normalisation_and_mutual_matches(setup):
    _, _, pred = setup
    la = pred["log_assignment"]
    assert la.shape == (B, N + 1, N + 1) and torch.isfinite(la).all()
    # exp(A) is (sub-)stochastic in both directions once the dustbin is included
    assert (la[:, :-1].exp().sum(2) <= 1 + 1e-4).all() and (la[:, :, :-1].exp().sum(1) <= 1 + 1e-4).all()
    m0, m1 = pred["matches0"], pred["matches1"]
    valid = m0 > -1
    rows = valid.nonzero()
    assert valid.sum() > 0
    assert torch.equal(m1[rows[:, 0], m0[rows[:, 0], rows[:, 1]]], rows[:, 1])
    torch.testing.assert_close(pred["matching_scores0"][valid], la[:, :-1, :-1].max(2).values.exp()[valid], rtol=1e-4, atol=1e-9)


def test_full_size_permutation_equivariance(setup):
    model, data, pred = setup
    g = torch.Generator().manual_seed(0)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).cuda()
    d2 = dict(data)
    d2["keypoints1"] = data["keypoints1"].gather(1, perm[..., None].expand(-1, -1, 2))
    d2["descriptors1"] = data["descriptors1"].gather(1, perm[..., None].expand(-1, -1, 256))
    with torch.no_grad():
        p2 = model(d2)
    cols = torch.cat([perm, torch.full((B, 1), N, device="cuda")], 1)
    ref = pred["log_assignment"].gather(2, cols[:, None, :].expand(-1, N + 1, -1))
    torch.testing.assert_close(p2["log_assignment"], ref, rtol=1e-3, atol=2e-3)


def test_full_size_image_swap_is_transpose(setup):
    model, data, pred = setup
    d2 = dict(data)
    for k in ("keypoints", "descriptors"):
        d2[k + "0"], d2[k + "1"] = data[k + "1"], data[k + "0"]
    d2["view0"], d2["view1"] = data["view1"], data["view0"]
    with torch.no_grad():
        p2 = model(d2)
    torch.testing.assert_close(p2["log_assignment"], pred["log_assignment"].transpose(1, 2), rtol=1e-3, atol=2e-3)


def test_full_size_bf16_agrees_with_fp32(setup):
    model, data, pred = setup
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pb = model(data)
    err = (pb["log_assignment"] - pred["log_assignment"]).abs()
    print(f"full-size bf16 vs fp32: max {err.max().item():.3f} mean {err.mean().item():.4f}")
    assert err.mean() < 0.05 and err.max() < 1.5
    top2 = pred["log_assignment"][:, :-1, :-1].topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.5
    agree = (pb["log_assignment"][:, :-1, :-1].argmax(-1) == pred["log_assignment"][:, :-1, :-1].argmax(-1))[clear]
    assert agree.float().mean() > 0.99
