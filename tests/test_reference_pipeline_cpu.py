"""TwoViewPipeline plumbing on the CPU, ours next to the REFERENCE's (build container only: skipped where /root/reference is
absent): the frozen extractor's stock path on both views, the `0` / `1` key suffixes, cached features (`allow_no_extract`),
ground truth inside forward (`run_gt_in_forward`) or inside loss, and the loss aggregation of components that define no loss
-- gluefactory/models/two_view_pipeline.py:20-114.  The matcher slot stays empty here (the HIP matchers need the GPU:
tests/test_gpu_reference_boundary.py drives them through the reference pipeline there)."""
import os
import sys
import tempfile

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gluefactory")),
                                reason="reference checkout not present (GPU box)")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = {"max_num_keypoints": 48, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}
GT = {"th_positive": 3.0, "th_negative": 5.0}


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


def _pipelines(extra, weights):
    from omegaconf import OmegaConf
    from gluefactory.models.two_view_pipeline import TwoViewPipeline as RefPipe
    from glue_factory_amd.pipeline import TwoViewPipeline
    ref = RefPipe(OmegaConf.create({"extractor": {"name": "extractors.superpoint_open", "weights": weights, **EXT},
                                    "ground_truth": {"name": "matchers.homography_matcher", **GT}, **extra})).eval()
    ours = TwoViewPipeline({"extractor": {"name": "extractors.superpoint_open", "weights": weights, **EXT},
                            "ground_truth": {"name": "matchers.homography_matcher", **GT}, **extra}).eval()
    return ref, ours


OURS_ONLY = {"gt_assignment_col0"}      # the positives as a fixed-length column vector for the fused loss


def _same(a, b, path=""):
    """a: ours, b: the reference's -- the same keys (plus OURS_ONLY), shapes, dtypes and values."""
    assert set(b) <= set(a) and set(a) - set(b) <= OURS_ONLY, (path, set(a) ^ set(b))
    for k in b:
        if isinstance(a[k], dict):
            _same(a[k], b[k], path + k + ".")
        elif torch.is_tensor(a[k]):
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (path + k, a[k].shape, b[k].shape, a[k].dtype, b[k].dtype)
            if a[k].is_floating_point():
                torch.testing.assert_close(a[k], b[k], rtol=1e-4, atol=1e-5, equal_nan=True, msg=lambda m: f"{path}{k}: {m}")
            else:
                assert torch.equal(a[k], b[k]), path + k


@pytest.mark.parametrize("run_gt_in_forward", [False, True])
def test_two_view_pipeline_plumbing_equals_the_reference(ref_path, run_gt_in_forward):
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    from glue_factory_amd.synthetic import similarity_homography
    torch.manual_seed(3)
    sp = SuperPoint(EXT)
    for prm in sp.detector[1].parameters():
        if prm.ndim == 4:
            prm.data.mul_(40.0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sp.pth")
        torch.save(sp.state_dict(), path)
        ref, ours = _pipelines({"run_gt_in_forward": run_gt_in_forward}, path)
    g = torch.Generator().manual_seed(4)
    data = {"view0": {"image": torch.rand(2, 1, 96, 128, generator=g)}, "view1": {"image": torch.rand(2, 3, 96, 128, generator=g)},
            "H_0to1": similarity_homography(128, 96)[None].repeat(2, 1, 1)}
    with torch.no_grad():
        pr, po = ref(dict(data)), ours(dict(data))
    assert ("gt_matches0" in pr) == run_gt_in_forward
    _same(po, pr)
    with torch.no_grad():
        lr, mr = ref.loss(dict(pr), data)
        lo, mo = ours.loss(dict(po), data)
    assert float(lr["total"]) == float(lo["total"]) == 0 and mr == mo == {}      # neither component defines a loss
    # cached features in the views replace the extraction (allow_no_extract)
    cache = {k[:-1]: v for k, v in pr.items() if k.endswith("0") and not k.startswith("gt_")}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sp.pth")
        torch.save(sp.state_dict(), path)
        ref2, ours2 = _pipelines({"allow_no_extract": True}, path)
    cached = {"view0": {**data["view0"], "cache": cache}, "view1": {**data["view1"], "cache": cache}, "H_0to1": data["H_0to1"]}
    with torch.no_grad():
        pr2, po2 = ref2(dict(cached)), ours2(dict(cached))
    _same(po2, pr2)
    assert torch.equal(po2["keypoints1"], cache["keypoints"])
