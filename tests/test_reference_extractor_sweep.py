"""The extractors' stock (library-convolution) path against the REFERENCE modules over a sweep of configurations, on the CPU
(build container only: skipped where /root/reference is absent).  tests/golden/superpoint_*.npz pin a few configurations
to committed vectors and the GPU tests hold the fused HIP path to those; here the reference module and ours are
constructed from the same seeded weights for every combination of NMS radius / border / threshold / cap / padding /
image size below and must produce the same keypoints (as sets where scores tie), scores and descriptors."""
import os
import sys
import tempfile

import numpy as np
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


def _compare(pred, ref, thr, cap):
    assert set(pred) == set(ref)
    assert pred["keypoints"].shape[0] == ref["keypoints"].shape[0]
    for b in range(ref["keypoints"].shape[0]):
        ours = {tuple(k): i for i, k in enumerate(pred["keypoints"][b].tolist())}
        theirs = {tuple(k): j for j, k in enumerate(ref["keypoints"][b].tolist())}
        so, sr = pred["keypoint_scores"][b].numpy(), ref["keypoint_scores"][b].numpy()
        valid_r = {k for k, j in theirs.items() if sr[j] > 0} if cap else set(theirs)      # (padding entries carry score 0 and random positions)
        valid_o = {k for k, i in ours.items() if so[i] > 0} if cap else set(ours)
        edge = float(min(sr[sr > 0])) if cap and (sr > 0).sum() == cap else thr
        for k in valid_o ^ valid_r:                # only detections on the deciding edge may differ
            s_k = so[ours[k]] if k in ours else sr[theirs[k]]
            assert abs(float(s_k) - edge) < 1e-5, (k, float(s_k), edge)
        common = sorted(valid_o & valid_r)
        assert len(common) >= 0.97 * len(valid_r)
        io, ir = [ours[k] for k in common], [theirs[k] for k in common]
        np.testing.assert_allclose(so[io], sr[ir], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(pred["descriptors"][b][io].numpy(), ref["descriptors"][b][ir].numpy(), rtol=1e-4, atol=1e-5)
    if "dense_descriptors" in ref:
        np.testing.assert_allclose(pred["dense_descriptors"].numpy(), ref["dense_descriptors"].numpy(), rtol=1e-4, atol=1e-5)


OPEN_CONFS = [
    ({"max_num_keypoints": 120, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 4}, (2, 1, 96, 128)),
    ({"max_num_keypoints": 90, "force_num_keypoints": True, "detection_threshold": 0.018, "nms_radius": 2, "remove_borders": 8}, (2, 3, 120, 160)),
    ({"detection_threshold": 0.02, "nms_radius": 3, "dense_outputs": True}, (1, 1, 88, 104)),
    ({"max_num_keypoints": 40, "detection_threshold": 0.02, "nms_radius": 5, "remove_borders": 0}, (1, 3, 64, 200)),
    ({"max_num_keypoints": 64, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 1, "remove_borders": 2}, (3, 1, 72, 72)),
]


@pytest.mark.parametrize("case", range(len(OPEN_CONFS)))
def test_superpoint_open_stock_path_equals_the_reference(ref_path, case):
    from gluefactory.models.extractors.superpoint_open import SuperPoint as RefSP
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    conf, shape = OPEN_CONFS[case]
    torch.manual_seed(100 + case)
    ours = SuperPoint(conf)
    for prm in ours.detector[1].parameters():       # spread the detector logits (random weights score 1/65 everywhere)
        if prm.ndim == 4:
            prm.data.mul_(40.0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sp.pth")
        torch.save(ours.state_dict(), path)
        ref = RefSP({**conf, "weights": path}).eval()
    ours.eval()
    image = torch.rand(*shape, generator=torch.Generator().manual_seed(200 + case))
    with torch.no_grad():
        pr, po = ref({"image": image}), ours({"image": image})
    _compare(po, pr, conf["detection_threshold"], conf.get("max_num_keypoints") if conf.get("force_num_keypoints") else None)
    if conf.get("max_num_keypoints") and conf.get("force_num_keypoints"):
        assert po["keypoints"].shape[1] == conf["max_num_keypoints"] == pr["keypoints"].shape[1]


NONFREE_CONFS = [
    ({"max_num_keypoints": 80, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}, False),
    ({"max_num_keypoints": 80, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 4, "legacy_sampling": False}, True),
    ({"max_num_keypoints": 60, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 2, "refinement_radius": 2}, True),
    ({"max_num_keypoints": -1, "detection_threshold": 0.02, "nms_radius": 3, "refinement_radius": 1}, False),
]


@pytest.mark.parametrize("case", range(len(NONFREE_CONFS)))
def test_nonfree_superpoint_stock_path_equals_the_reference(ref_path, case, monkeypatch):
    from gluefactory_nonfree.superpoint import SuperPoint as RefSP
    from glue_factory_amd.extractors.superpoint import SuperPoint
    conf, sized = NONFREE_CONFS[case]
    torch.manual_seed(300 + case)
    ours = SuperPoint(conf)
    ours.convPb.weight.data.mul_(40.0)
    sd = ours.state_dict()
    monkeypatch.setattr(torch.hub, "load_state_dict_from_url", lambda *a, **k: sd)      # (the reference downloads superpoint_v1.pth)
    ref = RefSP(conf).eval()
    assert set(ref.state_dict()) == set(sd)
    ours.eval()
    batch = 1 if conf["max_num_keypoints"] == -1 else 2
    image = torch.rand(batch, 1, 104, 136, generator=torch.Generator().manual_seed(400 + case))
    data = {"image": image}
    if sized:
        data["image_size"] = torch.tensor([[136, 104], [111, 83]])[:batch]
    with torch.no_grad():
        pr, po = ref(dict(data)), ours(dict(data))
    cap = conf["max_num_keypoints"] if conf.get("force_num_keypoints") else None
    if conf.get("refinement_radius"):      # refined positions are real-valued: compare by nearest neighbour instead of exact sets
        for b in range(batch):
            keep = pr["keypoint_scores"][b] > 0
            d = torch.cdist(po["keypoints"][b][po["keypoint_scores"][b] > 0], pr["keypoints"][b][keep],
                            compute_mode="donot_use_mm_for_euclid_dist")          # (the matmul form is off by 0.06 px at |x| ~ 100)
            assert d.shape[0] == d.shape[1] and float(d.min(0).values.max()) < 1e-3
    else:
        _compare(po, pr, conf["detection_threshold"], cap)
