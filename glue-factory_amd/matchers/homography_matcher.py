"""Ground-truth "matcher" from a homography (no-grad), the labelling step that feeds the loss.

Plugin-surface mirror of gluefactory/models/matchers/homography_matcher.py:8-66 for the point
branch (``use_points``); the assignment itself is ``glue_factory_amd.gt`` (restating
gluefactory/geometry/gt_generation.py:109-161).  Line ground truth (Hungarian matching of sampled
line points, gt_generation.py:409-558) is CPU/scipy work outside the accelerated path."""
from ..base_model import BaseModel
from ..gt import gt_matches_from_homography


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
        return gt_matches_from_homography(data["keypoints0"], data["keypoints1"], data["H_0to1"],
                                          pos_th=self.conf.th_positive, neg_th=self.conf.th_negative)

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = HomographyMatcher
