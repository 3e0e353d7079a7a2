"""glue_factory_amd.gt (torch ops, device-agnostic) vs the reference's
gt_matches_from_homography outputs stored in tests/golden/gt_homography.npz."""
import numpy as np
import torch

from conftest import load_golden
from glue_factory_amd.gt import gt_matches_from_homography


def test_gt_from_homography_matches_reference():
    z = load_golden("gt_homography")
    kp0, kp1, H = (torch.from_numpy(z["data." + k]) for k in ("keypoints0", "keypoints1", "H_0to1"))
    out = gt_matches_from_homography(kp0, kp1, H, pos_th=3.0, neg_th=3.0)
    for k in ("assignment", "matches0", "matches1"):
        np.testing.assert_array_equal(out[k].numpy(), z["gt." + k], err_msg=k)
    for k in ("reward", "matching_scores0", "matching_scores1", "proj_0to1", "proj_1to0"):
        np.testing.assert_allclose(out[k].numpy(), z["gt." + k], rtol=1e-5, atol=1e-4, err_msg=k)
    # assignment is exactly "gt_matches0[i] == j" (at most one positive per row/column)
    m0 = out["matches0"]
    dense = torch.zeros_like(out["assignment"])
    rows = (m0 >= 0).nonzero()
    dense[rows[:, 0], rows[:, 1], m0[rows[:, 0], rows[:, 1]]] = True
    assert torch.equal(dense, out["assignment"])


def test_gt_empty_inputs():
    out = gt_matches_from_homography(torch.zeros(2, 0, 2), torch.zeros(2, 5, 2), torch.eye(3)[None].repeat(2, 1, 1))
    assert out["assignment"].shape == (2, 0, 5) and out["matches1"].tolist() == [[-1] * 5] * 2


def _depth_data(z, device="cpu"):
    from glue_factory_amd.geometry import Camera, Pose
    t = lambda k: torch.from_numpy(z[k]).to(device)
    data = {"view0": {"camera": Camera(t("data.camera0")), "depth": t("data.depth0")},
            "view1": {"camera": Camera(t("data.camera1")), "depth": t("data.depth1")},
            "T_0to1": Pose.from_Rt(t("data.R"), t("data.t"))}
    return t("data.keypoints0"), t("data.keypoints1"), data


def test_gt_from_pose_depth_matches_reference():
    """gt_matches_from_pose_depth (+ our pinhole Camera / Pose) vs vectors produced by the reference's
    gt_generation.py:13-106 with its own wrappers: plain, cycle-consistency and epipolar-extended variants."""
    from glue_factory_amd.gt import gt_matches_from_pose_depth
    z = load_golden("gt_depth")
    kp0, kp1, data = _depth_data(z)
    for tag, kw in (("plain", {}), ("cc", {"cc_th": 4.0}), ("epi", {"epi_th": 1.0, "cc_th": 4.0})):
        out = gt_matches_from_pose_depth(kp0, kp1, data, pos_th=3.0, neg_th=5.0, **kw)
        for k in ("assignment", "matches0", "matches1", "visible0", "visible1"):
            np.testing.assert_array_equal(out[k].numpy(), z[f"{tag}.{k}"], err_msg=f"{tag}.{k}")
        for k in ("reward", "matching_scores0", "depth_keypoints0", "depth_keypoints1", "proj_0to1", "proj_1to0"):
            np.testing.assert_allclose(out[k].numpy(), z[f"{tag}.{k}"], rtol=1e-4, atol=1e-4, err_msg=f"{tag}.{k}",
                                       equal_nan=True)
        col = out["assignment_col0"]
        dense = torch.zeros_like(out["assignment"])
        dense.scatter_(2, col.clamp(min=0)[..., None], (col >= 0)[..., None])
        assert torch.equal(dense, out["assignment"])


def test_depth_matcher_plugin_surface():
    from glue_factory_amd.base_model import get_model
    z = load_golden("gt_depth")
    kp0, kp1, data = _depth_data(z)
    m = get_model("matchers.depth_matcher")({"th_consistency": 4.0})
    pred = m({**data, "keypoints0": kp0, "keypoints1": kp1})
    np.testing.assert_array_equal(pred["matches0"].numpy(), z["cc.matches0"])
    np.testing.assert_array_equal(pred["assignment"].numpy(), z["cc.assignment"])


def test_line_gt_from_homography_matches_reference():
    """gt_line_matches_from_homography vs vectors produced by the reference's gt_generation.py:409-558
    (sampled-point overlap counts + Hungarian assignment, invalid lines ignored)."""
    from glue_factory_amd.gt import gt_line_matches_from_homography
    z = load_golden("gt_lines")
    t = lambda k: torch.from_numpy(z[k])
    h, w = (int(v) for v in z["hw"])
    pos, m0, m1 = gt_line_matches_from_homography(t("lines0"), t("lines1"), t("valid0"), t("valid1"), (2, 1, h, w),
                                                  (2, 1, h, w), t("H"), npts=50, dist_th=5, overlap_th=0.2,
                                                  min_visibility_th=0.5)
    np.testing.assert_array_equal(pos.numpy(), z["assignment"])
    np.testing.assert_array_equal(m0.numpy(), z["matches0"])
    np.testing.assert_array_equal(m1.numpy(), z["matches1"])
    assert pos.sum() > 10 and (m0 == -2).sum() > 0 and (m0 == -1).sum() > 0


def test_homography_matcher_with_lines_plugin_surface():
    from glue_factory_amd.base_model import get_model
    z = load_golden("gt_lines")
    t = lambda k: torch.from_numpy(z[k])
    h, w = (int(v) for v in z["hw"])
    m = get_model("matchers.homography_matcher")({"use_points": False, "use_lines": True})
    img = torch.zeros(2, 1, h, w)
    pred = m({"H_0to1": t("H"), "lines0": t("lines0"), "lines1": t("lines1"), "valid_lines0": t("valid0"),
              "valid_lines1": t("valid1"), "view0": {"image": img}, "view1": {"image": img}})
    np.testing.assert_array_equal(pred["line_matches0"].numpy(), z["matches0"])
    np.testing.assert_array_equal(pred["line_assignment"].numpy(), z["assignment"])


def _line_depth_data(z, device="cpu"):
    from glue_factory_amd.geometry import Camera, Pose
    t = lambda k: torch.from_numpy(z[k]).to(device)
    image = torch.zeros(z["data.depth0"].shape[0], 1, *z["data.depth0"].shape[-2:], device=device)
    data = {"view0": {"camera": Camera(t("data.camera0")), "depth": t("data.depth0"), "image": image},
            "view1": {"camera": Camera(t("data.camera1")), "depth": t("data.depth1"), "image": image},
            "T_0to1": Pose.from_Rt(t("data.R"), t("data.t"))}
    return t("data.lines0"), t("data.lines1"), t("data.valid0"), t("data.valid1"), data


def test_line_gt_from_pose_depth_matches_reference():
    """gt_line_matches_from_pose_depth vs vectors produced by the reference's gt_generation.py:207-407 on a seeded depth
    scene (segments over depth holes -> ignored, reprojections leaving the image -> unmatched, invalid-flagged lines ->
    ignored), at the default thresholds and at a second set; bit-exact labels."""
    from glue_factory_amd.gt import gt_line_matches_from_pose_depth
    z = load_golden("gt_lines_depth")
    l0, l1, v0, v1, data = _line_depth_data(z)
    for tag, kw in (("default", {}), ("loose", {"npts": 30, "dist_th": 3, "overlap_th": 0.4, "min_visibility_th": 0.3})):
        pos, m0, m1 = gt_line_matches_from_pose_depth(l0, l1, v0, v1, data, **kw)
        np.testing.assert_array_equal(pos.numpy(), z[f"{tag}.assignment"], err_msg=tag)
        np.testing.assert_array_equal(m0.numpy(), z[f"{tag}.matches0"], err_msg=tag)
        np.testing.assert_array_equal(m1.numpy(), z[f"{tag}.matches1"], err_msg=tag)
        assert pos.sum() > 10 and (m0 == -2).sum() > 0 and (m0 == -1).sum() > 0
    pos, m0, m1 = gt_line_matches_from_pose_depth(l0[:, :0], l1, v0[:, :0], v1, data)          # gt_generation.py:230-242
    assert pos.shape == (2, 0, 36) and m0.shape == (2, 0) and m1.tolist() == [[-1] * 36] * 2


def test_depth_matcher_with_lines_plugin_surface():
    """depth_matcher.py:70-87: `use_lines` adds line_matches0/1 + line_assignment next to the point ground truth."""
    from glue_factory_amd.base_model import get_model
    z = load_golden("gt_lines_depth")
    l0, l1, v0, v1, data = _line_depth_data(z)
    m = get_model("matchers.depth_matcher")({"use_points": False, "use_lines": True})
    assert {"lines0", "lines1", "valid_lines0", "valid_lines1"} <= set(m.required_data_keys)
    pred = m({**data, "lines0": l0, "lines1": l1, "valid_lines0": v0, "valid_lines1": v1})
    assert set(pred) == {"line_matches0", "line_matches1", "line_assignment"}
    np.testing.assert_array_equal(pred["line_matches0"].numpy(), z["default.matches0"])
    np.testing.assert_array_equal(pred["line_matches1"].numpy(), z["default.matches1"])
    np.testing.assert_array_equal(pred["line_assignment"].numpy(), z["default.assignment"])
    zp = load_golden("gt_depth")
    kp0, kp1, _ = _depth_data(zp)
    both = get_model("matchers.depth_matcher")({"use_lines": True})
    pred = both({**data, "keypoints0": kp0[:, :20], "keypoints1": kp1[:, :20], "lines0": l0, "lines1": l1,
                 "valid_lines0": v0, "valid_lines1": v1})
    assert {"matches0", "assignment", "line_matches0", "line_assignment"} <= set(pred)
