"""Ground-truth "matcher" from depth + relative pose, mirroring gluefactory/models/matchers/depth_matcher.py:16-89
(keys, defaults, outputs).  Points: the fused HIP nearest-neighbour kernel (gf_gt_nn) where its preconditions hold.  Lines
(`use_lines`): gt.gt_line_matches_from_pose_depth -- torch ops on the keypoints' device + the Hungarian assignment on the
CPU (scipy), exactly as gt_generation.py:207-407 does it."""
import torch

from ..base_model import BaseModel
from ..gt import gt_line_matches_from_pose_depth, gt_matches_from_pose_depth, gt_matches_from_pose_depth_fused


class DepthMatcher(BaseModel):
    default_conf = {
        "use_points": True,
        "th_positive": 3.0,
        "th_negative": 5.0,
        "th_epi": None,
        "th_consistency": None,
        "use_lines": False,
        "n_line_sampled_pts": 50,
        "line_perp_dist_th": 5,
        "overlap_th": 0.2,
        "min_visibility_th": 0.5,
        "with_reward": True,        # ours: False skips the dense [B,M,N] reward (unused by the matcher losses)
    }
    required_data_keys = ["view0", "view1", "T_0to1"]

    def _init(self, conf):
        if conf.use_points:
            self.required_data_keys = self.required_data_keys + ["keypoints0", "keypoints1"]
        if conf.use_lines:
            self.required_data_keys = self.required_data_keys + ["lines0", "lines1", "valid_lines0", "valid_lines1"]

    def _forward(self, data):
        result = self._points(data) if self.conf.use_points else {}
        if self.conf.use_lines:
            with torch.autocast(device_type=data["lines0"].device.type, enabled=False):
                assignment, m0, m1 = gt_line_matches_from_pose_depth(
                    data["lines0"].float(), data["lines1"].float(), data["valid_lines0"], data["valid_lines1"], data,
                    self.conf.n_line_sampled_pts, self.conf.line_perp_dist_th, self.conf.overlap_th,
                    self.conf.min_visibility_th)
            result = dict(result, line_matches0=m0, line_matches1=m1, line_assignment=assignment)
        return result

    def _points(self, data):
        keys = ["depth_keypoints0", "valid_depth_keypoints0", "depth_keypoints1", "valid_depth_keypoints1"]
        kw = {k: data[k] for k in keys} if "depth_keypoints0" in data else {}
        kp0, kp1 = data["keypoints0"].float(), data["keypoints1"].float()
        with torch.autocast(device_type=kp0.device.type, enabled=False):
            if kp0.is_cuda and self.conf.th_epi is None and not self.conf.with_reward and kp0.shape[1] and kp1.shape[1]:
                return gt_matches_from_pose_depth_fused(kp0, kp1, data, pos_th=self.conf.th_positive,
                                                        neg_th=self.conf.th_negative, cc_th=self.conf.th_consistency, **kw)
            return gt_matches_from_pose_depth(kp0, kp1, data, pos_th=self.conf.th_positive, neg_th=self.conf.th_negative,
                                              epi_th=self.conf.th_epi, cc_th=self.conf.th_consistency, **kw)

    def loss(self, pred, data):
        raise NotImplementedError
