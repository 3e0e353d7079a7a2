"""Stock-torch components on the measured pipeline path (extractor, GT, pipeline plumbing) — CPU."""
import numpy as np
import torch

from conftest import load_golden
from glue_factory_amd.base_model import get_model


def test_superpoint_open_matches_reference():
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    z = load_golden("superpoint_open")
    torch.manual_seed(int(z["seed"]))
    model = SuperPoint({"max_num_keypoints": 100, "force_num_keypoints": True, "detection_threshold": 0.0,
                        "nms_radius": 3})
    image = torch.from_numpy(z["image"])
    for mode in ("eval", "train"):
        getattr(model, mode)()
        with torch.no_grad():
            pred = model({"image": image})
        # equal scores may be ordered differently by the batched top-k: compare as sets
        np.testing.assert_allclose(pred["keypoint_scores"].numpy(), z[mode + ".keypoint_scores"], rtol=1e-5, atol=1e-7)
        for b in range(image.shape[0]):
            ours = {tuple(k): i for i, k in enumerate(pred["keypoints"][b].tolist())}
            ref = z[mode + ".keypoints"][b].tolist()
            common = [(ours[tuple(k)], j) for j, k in enumerate(ref) if tuple(k) in ours]
            assert len(common) >= 0.95 * len(ref)
            io, ir = zip(*common)
            np.testing.assert_allclose(pred["descriptors"][b][list(io)].numpy(),
                                       z[mode + ".descriptors"][b][list(ir)], rtol=1e-4, atol=1e-5)


def test_superpoint_padding_and_state_dict_names():
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    torch.manual_seed(0)
    model = SuperPoint({"max_num_keypoints": 5000, "force_num_keypoints": True, "detection_threshold": 0.0,
                        "nms_radius": 4}).eval()
    pred = model({"image": torch.rand(2, 1, 64, 96)})
    assert pred["keypoints"].shape == (2, 5000, 2) and pred["descriptors"].shape == (2, 5000, 256)
    padded = pred["keypoint_scores"] == 0
    assert padded.any() and (~padded).any()
    assert (pred["keypoints"] >= 0).all() and (pred["keypoints"][..., 0] <= 96).all()
    names = set(model.state_dict())
    assert {"backbone.0.0.conv.weight", "backbone.3.1.bn.running_var", "detector.1.conv.bias",
            "descriptor.0.bn.weight"} <= names


def test_pipeline_plumbing_config1_shape():
    """extractor -> (no matcher on CPU) -> GT, with the reference's key conventions."""
    P = get_model("glue_factory_amd.pipeline")
    pipe = P({"extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": 64,
                            "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3},
              "ground_truth": {"name": "matchers.homography_matcher", "th_positive": 3, "th_negative": 3}})
    assert all(not p.requires_grad for p in pipe.extractor.parameters())
    torch.manual_seed(0)
    data = {"view0": {"image": torch.rand(2, 3, 120, 160)}, "view1": {"image": torch.rand(2, 3, 120, 160)},
            "H_0to1": torch.eye(3)[None].repeat(2, 1, 1)}
    pred = pipe(data)
    assert {"keypoints0", "descriptors1", "keypoint_scores0"} <= set(pred)
    losses, metrics = pipe.loss(pred, data)
    assert losses["total"] == 0 and metrics == {}
    assert pred["gt_assignment"].shape == (2, 64, 64) and pred["gt_matches0"].dtype == torch.int64
    # cached features bypass the extractor when allowed
    pipe2 = P({"extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": 64},
               "allow_no_extract": True})
    cache = {"keypoints": torch.rand(1, 7, 2), "descriptors": torch.rand(1, 7, 256)}
    out = pipe2({"view0": {"cache": cache}, "view1": {"cache": cache}})
    assert out["keypoints0"] is cache["keypoints"]
