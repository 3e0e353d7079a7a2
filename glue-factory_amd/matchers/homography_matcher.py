"""Ground-truth "matcher" from a homography (no-grad), the labelling step that feeds the loss.

Plugin-surface mirror of gluefactory/models/matchers/homography_matcher.py:8-66 for the point
branch (``use_points``); the assignment itself is ``glue_factory_amd.gt`` (restating
gluefactory/geometry/gt_generation.py:109-161).  Line ground truth (``use_lines``: overlap counts of sampled
line points + Hungarian assignment, gt_generation.py:409-558) runs through ``gt.gt_line_matches_from_homography``
(torch for the counts, scipy on the host for the assignment, exactly as the reference)."""
from ..base_model import BaseModel
from ..gt import gt_line_matches_from_homography, gt_matches_from_homography, gt_matches_from_homography_fused


class HomographyMatcher(BaseModel):
    default_conf = {
        "use_points": True,
        "th_positive": 3.0,
        "th_negative": 3.0,
        "use_lines": False,
        "n_line_sampled_pts": 50,
        "line_perp_dist_th": 5,
        "overlap_th": 0.2,
        "min_visibility_th": 0.5,
        "with_reward": True,   # dense `reward` [B,M,N] output (unused by the matcher losses); False skips it
    }
    required_data_keys = ["H_0to1"]

    def _init(self, conf):
        if conf.use_points:
            self.required_data_keys += ["keypoints0", "keypoints1"]
        if conf.use_lines:
            self.required_data_keys += ["lines0", "lines1", "valid_lines0", "valid_lines1"]

    def _forward(self, data):
        result = {}
        if self.conf.use_points:
            kp0, kp1 = data["keypoints0"], data["keypoints1"]
            if kp0.is_cuda and kp0.shape[1] > 0 and kp1.shape[1] > 0:    # fused HIP nearest-neighbour search
                result = gt_matches_from_homography_fused(kp0, kp1, data["H_0to1"], self.conf.th_positive,
                                                          self.conf.th_negative, with_reward=self.conf.with_reward)
            else:
                result = gt_matches_from_homography(kp0, kp1, data["H_0to1"], pos_th=self.conf.th_positive,
                                                    neg_th=self.conf.th_negative)
        if self.conf.use_lines:
            assignment, m0, m1 = gt_line_matches_from_homography(
                data["lines0"], data["lines1"], data["valid_lines0"], data["valid_lines1"],
                data["view0"]["image"].shape, data["view1"]["image"].shape, data["H_0to1"],
                self.conf.n_line_sampled_pts, self.conf.line_perp_dist_th, self.conf.overlap_th,
                self.conf.min_visibility_th)
            result["line_matches0"], result["line_matches1"], result["line_assignment"] = m0, m1, assignment
        return result

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = HomographyMatcher
