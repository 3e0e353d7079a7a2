"""Ground-truth "matcher" from a homography (no-grad), the labelling step that feeds the loss.

Plugin-surface mirror of gluefactory/models/matchers/homography_matcher.py:8-66 for the point
branch (``use_points``); the assignment itself is ``glue_factory_amd.gt`` (restating
gluefactory/geometry/gt_generation.py:109-161).  Line ground truth (Hungarian matching of sampled
line points, gt_generation.py:409-558) is CPU/scipy work outside the accelerated path."""
from ..base_model import BaseModel
from ..gt import gt_matches_from_homography, gt_matches_from_homography_fused


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
            raise NotImplementedError("line ground truth (scipy Hungarian) is outside the accelerated path")

    def _forward(self, data):
        if not self.conf.use_points:
            return {}
        kp0, kp1 = data["keypoints0"], data["keypoints1"]
        if kp0.is_cuda and kp0.shape[1] > 0 and kp1.shape[1] > 0:    # fused HIP nearest-neighbour search
            return gt_matches_from_homography_fused(kp0, kp1, data["H_0to1"], self.conf.th_positive,
                                                    self.conf.th_negative, with_reward=self.conf.with_reward)
        return gt_matches_from_homography(kp0, kp1, data["H_0to1"], pos_th=self.conf.th_positive,
                                          neg_th=self.conf.th_negative)

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = HomographyMatcher
