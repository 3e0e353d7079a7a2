"""TripletPipeline plumbing on the CPU, ours next to the REFERENCE's (build container only): three cached views, pair by pair
(`batch_triplets: false` -- the reference's batched mode torch.cat's nested view dicts, which it cannot do itself), a CPU toy
matcher plugin resolved by both `get_model`s, homography ground truth per pair; predictions per pair, summed losses,
concatenated metrics and the gradient of the toy parameter -- gluefactory/models/triplet_pipeline.py:23-99.  (The HIP LightGlue
in the same slot: tests/test_gpu_reference_boundary.py.)"""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gluefactory")),
                                reason="reference checkout not present (GPU box)")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_triplet_pipeline_plumbing_equals_the_reference(ref_path):
    from omegaconf import OmegaConf
    from gluefactory.models.triplet_pipeline import TripletPipeline as RefTriplet
    from glue_factory_amd.synthetic import make_pairs, similarity_homography
    from glue_factory_amd.triplet_pipeline import TripletPipeline
    conf = {"extractor": {"name": None}, "allow_no_extract": True, "batch_triplets": False,
            "matcher": {"name": "toy_models", "dim": 8},
            "ground_truth": {"name": "matchers.homography_matcher", "th_positive": 3.0, "th_negative": 5.0}}
    ref = RefTriplet(OmegaConf.create(conf)).train()
    ours = TripletPipeline(conf).train()
    with torch.no_grad():
        w = torch.eye(8) + 0.1 * torch.randn(8, 8, generator=torch.Generator().manual_seed(0))
        ref.matcher.w.copy_(w)
        ours.matcher.w.copy_(w)
    b, n = 2, 30
    base = make_pairs(b, n, dim=8, size=(320, 240), seed=1, with_gt=False)
    extra = make_pairs(b, n, dim=8, size=(320, 240), seed=2, with_gt=False)
    H = similarity_homography(320, 240)[None].repeat(b, 1, 1)
    img = torch.rand(b, 1, 24, 32, generator=torch.Generator().manual_seed(3))
    view = lambda src, i: {"image": img, "image_size": src["view" + i]["image_size"],            # noqa: E731
                           "cache": {"keypoints": src["keypoints" + i], "descriptors": src["descriptors" + i]}}
    data = {"view0": view(base, "0"), "view1": view(base, "1"), "view2": view(extra, "1"),
            "H_0to1": H, "H_0to2": H, "H_1to2": torch.eye(3)[None].repeat(b, 1, 1)}

    def step(model):
        model.zero_grad()
        pred = model(dict(data))
        losses, metrics = model.loss(pred, dict(data))
        losses["total"].mean().backward()
        return pred, losses, metrics, model.matcher.w.grad.clone()

    pr, lr, mr, gr = step(ref)
    po, lo, mo, go = step(ours)
    assert {"0to1", "0to2", "1to2"} <= set(po) and set(k for k in pr if not isinstance(pr[k], dict)) <= set(po)
    for idx in ("0to1", "0to2", "1to2"):
        assert set(pr[idx]) <= set(po[idx])
        for k, v in pr[idx].items():
            if torch.is_tensor(v):
                assert po[idx][k].dtype == v.dtype and po[idx][k].shape == v.shape, (idx, k)
                assert torch.allclose(po[idx][k].float(), v.float(), rtol=1e-5, atol=1e-6, equal_nan=True), (idx, k)
    assert set(lr) == set(lo) and set(mr) == set(mo)
    for k in lr:
        torch.testing.assert_close(lo[k], lr[k], rtol=1e-6, atol=1e-6, msg=lambda m: f"{k}: {m}")
    for k in mr:
        assert mo[k].shape == mr[k].shape == (3 * b,)                     # concatenated over the three pairs
        torch.testing.assert_close(mo[k], mr[k])
    torch.testing.assert_close(go, gr, rtol=1e-6, atol=1e-7)
