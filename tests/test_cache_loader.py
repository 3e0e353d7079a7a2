"""Feature-cache path (cache_loader.py mirror): padding rules against the reference's pad_to_length, export ->
load round trip, scaling, collation, and the pipeline's allow_no_extract contract — CPU."""
import sys

import numpy as np
import pytest
import torch

from glue_factory_amd.base_model import get_model
from glue_factory_amd.cache_loader import CacheLoader, export_features, pad_local_features, pad_to_length


def test_pad_to_length_matches_reference_semantics():
    x = torch.arange(12.0).reshape(1, 4, 3)
    for mode in ("zeros", "ones"):
        y = pad_to_length(x, 7, -2, mode=mode)
        assert y.shape == (1, 7, 3) and torch.equal(y[:, :4], x)
        assert torch.equal(y[:, 4:], torch.full((1, 3, 3), 0.0 if mode == "zeros" else 1.0))
    torch.manual_seed(0)
    y = pad_to_length(x, 40, -2, mode="random_c")
    for c in range(3):      # every channel padded inside that channel's own range
        assert y[0, 4:, c].min() >= x[0, :, c].min() and y[0, 4:, c].max() <= x[0, :, c].max()
    y = pad_to_length(x, 40, -2, mode="random")
    assert y[0, 4:].min() >= x.min() and y[0, 4:].max() <= x.max()
    assert pad_to_length(x, 4, -2) is x
    with pytest.raises(AssertionError):
        pad_to_length(x, 3, -2)
    if "/root/reference" not in sys.path:
        return
    # same call on the reference (build container only): deterministic modes are identical
    try:
        from gluefactory.models.utils.misc import pad_to_length as ref
    except Exception:
        return
    assert torch.equal(ref(x, 9, -2, mode="zeros"), pad_to_length(x, 9, -2, mode="zeros"))


def test_export_load_roundtrip_with_padding_and_scaling(tmp_path):
    g = torch.Generator().manual_seed(3)
    names, feats = ["a", "b"], []
    for n, k in zip(names, (30, 17)):
        f = {"keypoints": torch.rand(k, 2, generator=g) * 100, "keypoint_scores": torch.rand(k, generator=g),
             "descriptors": torch.randn(k, 8, generator=g)}
        export_features(str(tmp_path), n, f)
        feats.append(f)
    loader = get_model("cache_loader")({"path": str(tmp_path), "padding_fn": "pad_local_features", "padding_length": 32})
    scales = torch.tensor([[2.0, 2.0], [0.5, 0.5]])
    out = loader({"name": names, "scales": scales})
    assert out["keypoints"].shape == (2, 32, 2) and out["descriptors"].shape == (2, 32, 8)
    for i, f in enumerate(feats):
        k = f["keypoints"].shape[0]
        torch.testing.assert_close(out["keypoints"][i, :k], f["keypoints"] * scales[i])
        torch.testing.assert_close(out["descriptors"][i, :k], f["descriptors"])
        torch.testing.assert_close(out["keypoint_scores"][i, :k], f["keypoint_scores"])
        assert (out["keypoint_scores"][i, k:] == 0).all()
        lo, hi = (f["keypoints"] * scales[i]).amin(0), (f["keypoints"] * scales[i]).amax(0)
        assert (out["keypoints"][i, k:] >= lo).all() and (out["keypoints"][i, k:] <= hi).all()


def test_pipeline_uses_cached_features_without_extractor(tmp_path):
    """two_view_pipeline.py:52-63: with allow_no_extract, a view's `cache` dict replaces the extractor."""
    from glue_factory_amd.pipeline import TwoViewPipeline
    pipe = TwoViewPipeline({"extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": 16},
                            "allow_no_extract": True})
    cache = {"keypoints": torch.rand(1, 16, 2), "descriptors": torch.randn(1, 16, 256)}
    data = {"view0": {"image": torch.rand(1, 1, 32, 32), "cache": cache},
            "view1": {"image": torch.rand(1, 1, 32, 32), "cache": cache}}
    pred = pipe(data)
    assert torch.equal(pred["keypoints0"], cache["keypoints"]) and torch.equal(pred["descriptors1"], cache["descriptors"])
