"""Ground-truth "matcher" from a homography (no-grad): the labelling step that feeds the loss.

Plugin-surface mirror of gluefactory/models/matchers/homography_matcher.py:8-66 — same configuration keys,
same required inputs, same output names.  Points (``use_points``) are labelled by ``glue_factory_amd.gt``
(restating gluefactory/geometry/gt_generation.py:109-161; on a HIP device through the fused nearest-neighbour
kernel gf_gt_nn, which never builds a [B,M,N] fp32 tensor).  Lines (``use_lines``) go through
``gt.gt_line_matches_from_homography`` (gt_generation.py:409-558: torch overlap counts, Hungarian assignment
on the host with scipy, exactly as the reference does it)."""
from .. import gt as _gt
from ..base_model import BaseModel

_POINT_KEYS = ("keypoints0", "keypoints1")
_LINE_KEYS = ("lines0", "lines1", "valid_lines0", "valid_lines1")


class HomographyMatcher(BaseModel):
    default_conf = dict(
        use_points=True, th_positive=3.0, th_negative=3.0,                      # point labels
        use_lines=False, n_line_sampled_pts=50, line_perp_dist_th=5,            # line labels
        overlap_th=0.2, min_visibility_th=0.5,
        with_reward=True,     # ours: False skips the dense [B,M,N] `reward` (unused by the matcher losses)
    )
    required_data_keys = ["H_0to1"]

    def _init(self, conf):
        extra = (_POINT_KEYS if conf.use_points else ()) + (_LINE_KEYS if conf.use_lines else ())
        self.required_data_keys = list(self.required_data_keys) + list(extra)

    def _label_points(self, data):
        c = self.conf
        pts0, pts1, hom = data["keypoints0"], data["keypoints1"], data["H_0to1"]
        on_gpu = pts0.is_cuda and pts0.shape[1] > 0 and pts1.shape[1] > 0
        if on_gpu:            # fused HIP nearest-neighbour search
            return _gt.gt_matches_from_homography_fused(pts0, pts1, hom, c.th_positive, c.th_negative,
                                                        with_reward=c.with_reward)
        return _gt.gt_matches_from_homography(pts0, pts1, hom, pos_th=c.th_positive, neg_th=c.th_negative)

    def _label_lines(self, data):
        c = self.conf
        assignment, fwd, bwd = _gt.gt_line_matches_from_homography(
            data["lines0"], data["lines1"], data["valid_lines0"], data["valid_lines1"],
            data["view0"]["image"].shape, data["view1"]["image"].shape, data["H_0to1"],
            c.n_line_sampled_pts, c.line_perp_dist_th, c.overlap_th, c.min_visibility_th)
        # (ours) the single positive column of every row, -1 if none: lets the matcher losses gather their positive terms
        # with a fixed-length index instead of scanning the dense matrix (ops.nll_positive_terms)
        col0 = fwd.clamp(min=-1) if fwd.numel() else fwd
        return {"line_matches0": fwd, "line_matches1": bwd, "line_assignment": assignment, "line_assignment_col0": col0}

    def _forward(self, data):
        import torch
        # labels are geometry in pixels: always fp32, also when the caller's train loop wraps the pipeline's loss -- and
        # with it this module -- in torch.autocast (gluefactory/train.py:470-476)
        with torch.autocast(device_type=data["H_0to1"].device.type, enabled=False):
            out = self._label_points(data) if self.conf.use_points else {}
            if self.conf.use_lines:
                out.update(self._label_lines(data))
        return out

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = HomographyMatcher
