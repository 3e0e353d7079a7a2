"""Shared by tests/test_gpu_zz_learning.py (HIP modules on the GPU) and tools/probe/ref_learning_curve.py (the unmodified
reference modules on the CPU): model configuration, seeded initial parameters and seeded batches of the learning-curve runs."""
import numpy as np
import torch

STEPS, HELD_OUT = 300, (9001, 9002, 9003)
# (at 1e-3 the BatchNorm models' loss curves are spiky -- the reference's too -- and two runs decorrelate within 150 steps)
LR = {"lightglue": 1e-3, "superglue": 2e-4, "gluestick": 2e-4}


def conf(kind):
    return {"lightglue": {"n_layers": 3, "filter_threshold": 0.1},
            "superglue": {"GNN_layers": ["self", "cross"] * 2, "num_sinkhorn_iterations": 20, "filter_threshold": 0.2},
            "gluestick": {"GNN_layers": ["self", "cross"] * 2, "filter_threshold": 0.2}}[kind]


def initial_params(kind):
    """None = the module's own initialisation under torch.manual_seed(0) (LightGlue: the reference and the HIP module create
    their parameters in the same order); SuperGlue / GlueStick: seeded state_dicts shared through the oracle's builders."""
    if kind == "superglue":
        from oracle import superglue_oracle as sgo
        return sgo.init_params(256, gnn_layers=4, seed=301)
    if kind == "gluestick":
        from oracle import gluestick_oracle as gso
        return gso.init_params(256, gnn_layers=4, inter=None, seed=302)
    return None


def batch(kind, seed):
    from glue_factory_amd.synthetic import make_pairs, make_point_line_pairs
    if kind == "gluestick":
        return make_point_line_pairs(8, 192, 32, dim=256, size=(640, 480), seed=seed)
    d = make_pairs(8, 256, dim=256, size=(640, 480), seed=seed)
    if kind == "superglue":      # the reference reads view["image"].shape unconditionally (superglue.py:280)
        d["view0"]["image"] = torch.zeros(8, 1, 8, 8)
        d["view1"]["image"] = torch.zeros(8, 1, 8, 8)
    return d


def trained_state_from_delta(init, z):
    """The state a `*_trained` golden is taken at: initial state + the stored bf16 drift (exactly reproducible on both sides)."""
    state = {}
    for k, v in init.items():
        if v.is_floating_point():
            d = torch.from_numpy(z["delta." + k].astype(np.int16)).view(torch.bfloat16).reshape(v.shape)
            state[k] = v + d.float()
        else:
            state[k] = torch.from_numpy(z["state." + k]).reshape(v.shape).to(v.dtype)
    return state
