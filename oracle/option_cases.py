"""Seeded inputs for the SuperGlue / GlueStick configuration options and edge cases no other fixture touches.

TEST INFRASTRUCTURE ONLY (shared by oracle/gen_golden.py, which runs the REFERENCE on them and stores its outputs in
tests/golden/matcher_options.npz, and by tests/test_gpu_matcher_options.py, which runs the HIP modules on the same
inputs).  Never imported by the product package.
"""
import torch

from glue_factory_amd.synthetic import make_pairs


def option_cases():
    """name -> (kind, conf, params, data): seeded, small; shared by this generator and tests/test_gpu_matcher_options.py
    (the parameters are regenerated there from the same code, only checksums are stored)."""
    from glue_factory_amd.synthetic import make_point_line_pairs
    from oracle import gluestick_oracle as gso
    from oracle import superglue_oracle as sgo
    cases = {}
    # SuperGlue: `use_scores: false` (2-channel keypoint encoder input), a shorter `keypoint_encoder`, views WITHOUT image_size
    # (normalize_keypoints falls back to the image tensor's shape, superglue.py:82-93, 282-287)
    g = torch.Generator().manual_seed(201)
    p = sgo.init_params(256, kenc_layers=(32, 64), gnn_layers=2, seed=201)
    p["kenc.encoder.0.weight"] = p["kenc.encoder.0.weight"][:, :2].contiguous()
    d = make_pairs(2, 70, 64, dim=256, size=(320, 240), seed=202)
    d["view0"] = {"image": torch.zeros(2, 1, 240, 320)}
    d["view1"] = {"image": torch.zeros(2, 1, 240, 320)}
    cases["sg_noscores"] = ("superglue", {"weights": None, "use_scores": False, "keypoint_encoder": [32, 64],
                                          "GNN_layers": ["self", "cross"], "num_sinkhorn_iterations": 10,
                                          "filter_threshold": 0.0}, p, d)
    # SuperGlue: an image without keypoints (superglue.py:271-279)
    d = make_pairs(2, 30, 30, dim=256, size=(320, 240), seed=203)
    for k in ("keypoints1", "descriptors1", "keypoint_scores1"):
        d[k] = d[k][:, :0]
    d["view0"]["image"] = torch.zeros(2, 1, 240, 320)
    d["view1"]["image"] = torch.zeros(2, 1, 240, 320)
    cases["sg_empty"] = ("superglue", {"weights": None, "GNN_layers": ["self", "cross"], "num_sinkhorn_iterations": 5},
                         sgo.init_params(256, gnn_layers=2, seed=203), d)
    # GlueStick: `input_dim: 128` (a Conv1d input_proj, gluestick.py:161-167, 205-207), `num_line_iterations: 2`
    p = gso.init_params(256, gnn_layers=2, inter=None, seed=205)
    p["input_proj.weight"] = ((torch.rand(256, 128, 1, generator=g) * 2 - 1) / 128 ** 0.5)
    p["input_proj.bias"] = torch.zeros(256)
    d = make_point_line_pairs(2, 40, 12, dim=128, size=(320, 240), seed=206)
    cases["gs_inputproj"] = ("gluestick", {"weights": None, "input_dim": 128, "num_line_iterations": 2,
                                           "GNN_layers": ["self", "cross"], "filter_threshold": 0.0}, p, d)
    # GlueStick: no line segments in image 1 (gluestick.py:211-239, 286-311: zero line encodings, empty line outputs)
    d = make_point_line_pairs(2, 40, 12, dim=256, size=(320, 240), seed=207)
    nj = 24
    for k in ("keypoints1", "descriptors1", "keypoint_scores1"):
        d[k] = d[k][:, nj:]                                  # drop image 1's junction block
    d["lines1"], d["lines_junc_idx1"], d["line_scores1"] = d["lines1"][:, :0], d["lines_junc_idx1"][:, :0], d["line_scores1"][:, :0]
    from glue_factory_amd.gt import gt_matches_from_homography
    gt = gt_matches_from_homography(d["keypoints0"], d["keypoints1"], d["H_0to1"], pos_th=3.0, neg_th=3.0)
    d.update({"gt_assignment": gt["assignment"], "gt_assignment_col0": gt["assignment_col0"], "gt_matches0": gt["matches0"],
              "gt_matches1": gt["matches1"], "gt_line_assignment": torch.zeros(2, 12, 0, dtype=torch.bool),
              "gt_line_matches0": torch.full((2, 12), -1), "gt_line_matches1": torch.full((2, 0), -1)})
    d.pop("gt_line_assignment_col0", None)
    cases["gs_nolines"] = ("gluestick", {"weights": None, "GNN_layers": ["self", "cross"], "filter_threshold": 0.0},
                           gso.init_params(256, gnn_layers=2, inter=None, seed=207), d)
    # GlueStick: no keypoints at all in image 0 (gluestick.py:163-195)
    d = make_point_line_pairs(2, 20, 6, dim=256, size=(320, 240), seed=208)
    for k in ("keypoints0", "descriptors0", "keypoint_scores0"):
        d[k] = d[k][:, :0]
    d["lines0"], d["lines_junc_idx0"], d["line_scores0"] = d["lines0"][:, :0], d["lines_junc_idx0"][:, :0], d["line_scores0"][:, :0]
    cases["gs_empty"] = ("gluestick", {"weights": None, "GNN_layers": ["self", "cross"]},
                         gso.init_params(256, gnn_layers=2, inter=None, seed=208), d)
    # GlueStick: intermediate supervision after layer 1 with non-default loss weights (gluestick.py:39-43, 378-462), and
    # `skip_init: true` -- which the reference's constructor never forwards to its GNN (gluestick.py:78-85: `skip` keeps its
    # default False), so it must register NO `scaling` parameter and change nothing: the strict load below proves both sides agree
    p = gso.init_params(256, gnn_layers=4, inter=[1], seed=213)
    cases["gs_skipinit"] = ("gluestick", {"weights": None, "GNN_layers": ["self", "cross"] * 2, "skip_init": True,
                                          "inter_supervision": [1], "filter_threshold": 0.0,
                                          "loss": {"nll_weight": 0.5, "nll_balancing": 0.3, "inter_supervision": [0.2, 0.7]}},
                            p, make_point_line_pairs(2, 40, 12, dim=256, size=(320, 240), seed=214))
    # GlueStick: `checkpointed: true` (gluestick.py:724-757): same numbers as without, but the backward re-runs every GNN and
    # line layer's forward in training mode -> their BatchNorm buffers take each update twice (the fixture stores the buffers
    # after the step; "gs_skipinit" above is the un-checkpointed counterpart)
    cases["gs_checkpointed"] = ("gluestick", {"weights": None, "GNN_layers": ["self", "cross"] * 2, "checkpointed": True,
                                              "filter_threshold": 0.0},
                                gso.init_params(256, gnn_layers=4, inter=None, seed=217),
                                make_point_line_pairs(2, 40, 12, dim=256, size=(320, 240), seed=218))
    # SuperGlue: `descriptor_dim: 128` (4 heads of 32 channels) and `loss.nll_balancing: 0.8` (superglue.py:222-233, 322-342)
    d = make_pairs(2, 70, 64, dim=128, size=(320, 240), seed=216)
    d["view0"]["image"] = torch.zeros(2, 1, 240, 320)        # the reference reads view["image"].shape unconditionally
    d["view1"]["image"] = torch.zeros(2, 1, 240, 320)
    cases["sg_dim128"] = ("superglue", {"weights": None, "descriptor_dim": 128, "keypoint_encoder": [32, 64],
                                        "GNN_layers": ["self", "cross"], "num_sinkhorn_iterations": 20, "filter_threshold": 0.0,
                                        "loss": {"nll_balancing": 0.8}},
                          sgo.init_params(128, kenc_layers=(32, 64), gnn_layers=2, seed=215), d)
    # LightGlue: deep-supervision weights `loss.gamma` (gamma^(L-i-1); gamma <= 0 selects i + 1) and `loss.nll_balancing`
    # away from their defaults (lightglue.py:328-332, 598-628; gluefactory/models/utils/losses.py:9-46)
    from oracle import lightglue_oracle as lgo
    for cname, loss in (("lg_gamma07", {"gamma": 0.7, "fn": "nll", "nll_balancing": 0.3}),
                        ("lg_gamma0", {"gamma": 0.0, "fn": "nll", "nll_balancing": 0.8})):
        cases[cname] = ("lightglue", {"weights": None, "n_layers": 3, "flash": False, "filter_threshold": 0.0, "loss": loss},
                        lgo.init_params(3, 256, 4, seed=211), make_pairs(2, 96, 80, dim=256, size=(320, 240), seed=212))
    return cases


def superpoint_option_cases():
    """name -> (conf, image shape): superpoint_open configurations the main golden does not touch (superpoint_open.py:79-90,
    150-207): no keypoint cap + a real detection threshold + dense outputs (batch of one, variable count), a cap without
    padding, no border removal with a small NMS radius.  (A batch of several images WITHOUT force_num_keypoints is not a
    reference configuration: superpoint_open.py:196-208 ends in `list.transpose`.)"""
    return {"uncapped": ({"detection_threshold": 0.02, "dense_outputs": True}, (1, 1, 120, 160)),
            "capped": ({"max_num_keypoints": 60, "detection_threshold": 0.02, "nms_radius": 3}, (1, 3, 96, 128)),
            "noborder": ({"max_num_keypoints": 80, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 2,
                          "remove_borders": 0}, (2, 1, 96, 128))}
