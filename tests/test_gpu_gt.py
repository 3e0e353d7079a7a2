"""Fused ground-truth nearest-neighbour kernel vs the reference's gt_matches_from_homography outputs
(tests/golden/gt_homography.npz) and vs the torch restatement at the benchmark size."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_fused_gt_matches_reference_golden():
    from glue_factory_amd.gt import gt_matches_from_homography_fused
    z = load_golden("gt_homography")
    kp0, kp1, H = (torch.from_numpy(z["data." + k]).cuda() for k in ("keypoints0", "keypoints1", "H_0to1"))
    out = gt_matches_from_homography_fused(kp0, kp1, H, 3.0, 3.0, with_reward=True)
    for k in ("assignment", "matches0", "matches1"):
        np.testing.assert_array_equal(out[k].cpu().numpy(), z["gt." + k], err_msg=k)
    for k in ("reward", "matching_scores0", "matching_scores1", "proj_0to1", "proj_1to0"):
        np.testing.assert_allclose(out[k].cpu().numpy(), z["gt." + k], rtol=1e-5, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("b,m,n", [(3, 2048, 2048), (2, 700, 1025), (1, 1, 5)])
def test_fused_gt_equals_torch_path(b, m, n):
    from glue_factory_amd.gt import gt_matches_from_homography, gt_matches_from_homography_fused
    from glue_factory_amd.synthetic import make_pairs
    data = make_pairs(b, m, n, dim=8, size=(1024, 1024), seed=m + n, with_gt=False)
    kp0, kp1, H = (data[k].cuda() for k in ("keypoints0", "keypoints1", "H_0to1"))
    ref = gt_matches_from_homography(kp0, kp1, H, 3.0, 3.0)
    out = gt_matches_from_homography_fused(kp0, kp1, H, 3.0, 3.0)
    for k in ("assignment", "matches0", "matches1"):
        assert torch.equal(out[k], ref[k]), k
    assert "reward" not in out
    # idempotence-style property at full size: every positive is mutual and inside the threshold
    m0, m1 = out["matches0"], out["matches1"]
    rows = (m0 >= 0).nonzero()
    assert torch.equal(m1[rows[:, 0], m0[rows[:, 0], rows[:, 1]]], rows[:, 1])


@pytest.mark.parametrize("cc_th", [None, 4.0])
def test_fused_depth_gt_equals_dense_form(cc_th):
    """gt_matches_from_pose_depth_fused (gf_gt_nn, no [B,M,N] fp32 tensor) vs the dense torch form and the
    reference-generated vectors."""
    import numpy as np
    from conftest import load_golden
    from test_gt_golden import _depth_data
    from glue_factory_amd.gt import gt_matches_from_pose_depth, gt_matches_from_pose_depth_fused
    z = load_golden("gt_depth")
    kp0, kp1, data = _depth_data(z, "cuda")
    kw = {} if cc_th is None else {"cc_th": cc_th}
    dense = gt_matches_from_pose_depth(kp0, kp1, data, pos_th=3.0, neg_th=5.0, **kw)
    fused = gt_matches_from_pose_depth_fused(kp0, kp1, data, pos_th=3.0, neg_th=5.0, **kw)
    for k in ("assignment", "assignment_col0", "matches0", "matches1", "visible0", "visible1"):
        assert torch.equal(fused[k], dense[k]), k
    tag = "plain" if cc_th is None else "cc"
    np.testing.assert_array_equal(fused["matches0"].cpu().numpy(), z[f"{tag}.matches0"])
    np.testing.assert_array_equal(fused["matches1"].cpu().numpy(), z[f"{tag}.matches1"])
    np.testing.assert_array_equal(fused["assignment"].cpu().numpy(), z[f"{tag}.assignment"])


def test_line_gt_from_pose_depth_on_the_device():
    """gt_line_matches_from_pose_depth on cuda tensors (sampling, reprojection and the close-point counts on the device,
    the Hungarian step on the CPU -- as the reference; on a GPU the reference rounds the point-to-segment geometry to fp16,
    gt_generation.py:191-193, so a label on the 5 px / overlap edge may differ from the CPU golden): labels agree with the
    reference-generated CPU vectors on nearly every line, and every positive pair is mutual."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gt_golden import _line_depth_data
    from conftest import load_golden
    from glue_factory_amd.gt import gt_line_matches_from_pose_depth
    z = load_golden("gt_lines_depth")
    l0, l1, v0, v1, data = _line_depth_data(z, "cuda")
    pos, m0, m1 = gt_line_matches_from_pose_depth(l0, l1, v0, v1, data)
    assert pos.is_cuda and m0.is_cuda
    agree0 = (m0.cpu().numpy() == z["default.matches0"]).mean()
    agree1 = (m1.cpu().numpy() == z["default.matches1"]).mean()
    print(f"device vs reference CPU labels: {agree0:.3f} / {agree1:.3f}")
    assert agree0 > 0.9 and agree1 > 0.9
    b, i = torch.nonzero(m0 > -1, as_tuple=True)
    assert torch.equal(m1[b, m0[b, i]], i) and bool(pos[b, i, m0[b, i]].all())
