"""GlueStick point + line matcher on the MI355X hot path (drop-in for
gluefactory.models.matchers.gluestick).

Same plugin surface as the reference (gluefactory/models/matchers/gluestick.py:25-462): BaseModel
subclass, same ``default_conf`` / ``required_data_keys`` and ``state_dict`` names (``kenc``,
``lenc``, ``gnn.layers.i.update.*``, ``gnn.line_layers.k.mlp.*``, ``final_proj``,
``final_line_proj``, ``inter_line_proj.*``, ``bin_score``, ``line_bin_score``).

What runs where
  * GNN layers: the SuperGlue building blocks of ``superglue.py`` (``gf_gemm`` over stacked
    channels-last activations, MFMA flash attention with the head-fastest channel order folded
    into the weights, one BatchNorm call per image);
  * point assignment (gluestick.py:772-783, bin-augmented averaged double softmax): the row /
    column log-sum-exp come from ``gf_rows_lse`` tiles of md0 md1^T (the bin joins through a
    logaddexp on the [B,N] vectors) and the (N+1)^2 matrix is written once by ``gf_assign_write``;
  * line layers (:589-691): junction gather, endpoint MLP input, mean scatter back on the HIP line
    kernels (``ops.line_graph`` builds the CSR once per forward; ``gf_line_gather`` / ``gf_line_segsum``
    / ``gf_line_expand``), MLP on ``gf_gemm`` + the fused BatchNorm op; ``line_attention: True`` runs the
    same kernels with a per-junction softmax weighting (the small softmax itself in stock torch);
  * the line head (:336-376, 2*Nl x 2*Nl scores, two endpoint pairings) works on <= 1024-row tensors
    and stays on stock torch ops.
"""
from pathlib import Path

import torch
from torch import nn

from .. import ops
from ..base_model import BaseModel
from ..metrics import matcher_metrics
from .superglue import MLP, AttentionalPropagation, KeypointEncoder, _conv_cl, _mlp_cl, derived_specs_of

ETH_EPS = 1e-8


def normalize_keypoints(kpts, shape_or_size):
    """Centre, divide by 0.7 * max(size) (gluestick.py:477-488)."""
    kpts = kpts.float()
    if isinstance(shape_or_size, (tuple, list, torch.Size)):
        h, w = shape_or_size[-2:]
        size = kpts.new_tensor([[w, h]])
    else:
        size = shape_or_size.float().to(kpts)
    return (kpts - size[:, None] / 2) / (size.max(1, keepdim=True).values * 0.7)[:, None]


class EndPtEncoder(nn.Module):
    def __init__(self, feature_dim, layers):
        super().__init__()
        self.encoder = MLP([5] + list(layers) + [feature_dim], do_bn=True)
        nn.init.constant_(self.encoder[-1].bias, 0.0)

    def forward(self, endpoints, scores, halves=1):
        """endpoints [B,Nl,2,2] -> [B, 2Nl, D] (xy, offset to the other endpoint, line score)."""
        b, nl = endpoints.shape[:2]
        off = endpoints[:, :, 1] - endpoints[:, :, 0]
        off = torch.stack([off, -off], 2).reshape(b, 2 * nl, 2)
        x = torch.cat([endpoints.reshape(b, 2 * nl, 2), off, scores.repeat(1, 2)[..., None]], -1)
        return _mlp_cl(self.encoder, x.float(), halves)


class GNNLayer(nn.Module):
    def __init__(self, feature_dim, layer_type, skip_init=False):
        super().__init__()
        assert layer_type in ("cross", "self")
        self.type = layer_type
        self.update = AttentionalPropagation(feature_dim, 4)
        if skip_init:
            self.update.register_parameter("scaling", nn.Parameter(torch.tensor(0.0)))
        else:
            self.update.scaling = 1.0


class LineLayer(nn.Module):
    """gluestick.py:589-691 on the HIP line kernels (csrc/line_layer.hip): gather of the two junction descriptors of
    every endpoint + the endpoint encoding -> MLP (gf_gemm + fused BatchNorm/ReLU) -> mean over the endpoints of each
    junction fused with the residual add, all on channels-last tensors; ``line_attention`` weighs the endpoints of a
    junction by a softmax over them (proj_node / proj_neigh, :623-639) instead of averaging."""

    checkpointed_in_reference = False      # set per `checkpointed` (gluestick.py:741-757: see AttentionalPropagation)

    def __init__(self, feature_dim, line_attention=False):
        super().__init__()
        self.dim = feature_dim
        self.mlp = MLP([feature_dim * 3, feature_dim * 2, feature_dim], do_bn=True)
        self.line_attention = line_attention
        if line_attention:
            self.proj_node = nn.Conv1d(feature_dim, feature_dim, kernel_size=1)
            self.proj_neigh = nn.Conv1d(2 * feature_dim, feature_dim, kernel_size=1)

    def _attention(self, msg, ldesc, junc_idx, graph):
        """Per-endpoint weights: softmax over the endpoints that share a junction of <proj_node(x_j), proj_neigh([x_other,
        enc])> / sqrt(D), with the reference's global max shift and its epsilon (gluestick.py:623-639)."""
        d = self.dim
        query = ops.rows_gather(_conv_cl(ldesc, self.proj_node).float(), junc_idx, *graph)
        key = _conv_cl(msg[..., d:].contiguous(), self.proj_neigh).float()
        prob = (query * key).sum(-1) / d ** 0.5
        prob = torch.exp(prob - prob.max())
        denom = torch.zeros_like(ldesc[..., 0], dtype=torch.float32).scatter_reduce(
            1, junc_idx, prob, reduce="sum", include_self=False)
        return prob / (denom.gather(1, junc_idx) + 1e-8)

    def forward(self, ldesc, line_enc, junc_idx, halves, graph, replay_out=None):
        """ldesc [B',N,D], line_enc [B',2Nl,D], junc_idx [B',2Nl], graph = ops.line_graph(junc_idx, N).
        replay_out: see superglue._mlp_cl (two per-image calls of a checkpointed layer replay in call order)."""
        order, seg = graph
        # ldesc feeds the gather and the residual of the aggregation: one gradient chain (the sum happens in the gather's
        # segment-sum kernel instead of an autograd add over [B', N, D])
        chain = ops.GradChain(2) if ldesc.requires_grad and torch.is_grad_enabled() else None
        msg = ops.line_gather(ldesc, line_enc.to(ldesc.dtype), junc_idx, order, seg, chain=chain)
        ck = self.checkpointed_in_reference and self.training
        upd = _mlp_cl(self.mlp, msg, halves, replay=ck and replay_out is None, replay_out=replay_out if ck else None)
        if self.line_attention:
            upd = upd * self._attention(msg, ldesc, junc_idx, graph)[..., None].to(upd.dtype)
        return ops.line_aggregate(ldesc, upd, junc_idx, order, seg, mean=not self.line_attention, chain=chain)


class AttentionalGNN(nn.Module):
    def __init__(self, feature_dim, layer_types, inter_supervision=None, num_line_iterations=1,
                 line_attention=False):
        super().__init__()
        self.inter_supervision = inter_supervision
        self.num_line_iterations = num_line_iterations
        self.layers = nn.ModuleList([GNNLayer(feature_dim, t) for t in layer_types])
        self.line_layers = nn.ModuleList([LineLayer(feature_dim, line_attention)
                                          for _ in range(len(layer_types) // 2)])


class GlueStick(BaseModel):
    default_conf = {
        "input_dim": 256,
        "descriptor_dim": 256,
        "weights": None,
        "version": "v0.1_arxiv",
        "keypoint_encoder": [32, 64, 128, 256],
        "GNN_layers": ["self", "cross"] * 9,
        "num_line_iterations": 1,
        "line_attention": False,
        "filter_threshold": 0.2,
        "checkpointed": False,     # accepted; activations are kept, never recomputed
        "skip_init": False,
        "inter_supervision": None,
        "mp": False,
        # Mixed precision only: arithmetic of the GNN's attention.  The reference pins this one function to fp32 under
        # autocast (gluestick.py:18-22, 524-529 @AMP_CUSTOM_FWD_F32: scores, softmax and weighted sum in fp32 on the
        # bf16-valued projections).  "reference": the same -- fp32-equivalent second products; "bf16": P and dS are
        # rounded to bf16 in front of the second products like every other bf16 kernel here (the only difference; measured on
        # config 5: per-tensor gradient error against the fp32 reference 7.0 % median with "bf16", 6.8 % with "reference" --
        # the error of a bf16 GlueStick step is the linear layers', not the attention's).
        "attention_precision": "reference",
        "loss": {"nll_weight": 1.0, "nll_balancing": 0.5, "inter_supervision": [0.3, 0.6]},
    }
    required_data_keys = ["view0", "view1", "keypoints0", "keypoints1", "descriptors0", "descriptors1",
                          "keypoint_scores0", "keypoint_scores1", "lines0", "lines1", "lines_junc_idx0",
                          "lines_junc_idx1", "line_scores0", "line_scores1"]

    def _init(self, conf):
        if conf.descriptor_dim not in (128, 256, 512):
            raise NotImplementedError("the HIP attention kernels exist for 4 heads of 64 channels (tuned) and of 32 / 128 (generic kernels)")
        if conf.attention_precision not in ("reference", "bf16"):
            raise ValueError(f"attention_precision: 'reference' or 'bf16', got {conf.attention_precision!r}")
        if conf.descriptor_dim == 512 and conf.attention_precision == "reference":
            raise NotImplementedError("head_dim 128 has no fp32 attention backward (LDS); use attention_precision: bf16")
        d = conf.descriptor_dim
        if conf.input_dim != d:
            self.input_proj = nn.Conv1d(conf.input_dim, d, kernel_size=1)
            nn.init.constant_(self.input_proj.bias, 0.0)
        self.kenc = KeypointEncoder(d, conf.keypoint_encoder)
        self.lenc = EndPtEncoder(d, conf.keypoint_encoder)
        inter = None if conf.inter_supervision is None else list(conf.inter_supervision)
        self.gnn = AttentionalGNN(d, conf.GNN_layers, inter, conf.num_line_iterations, conf.line_attention)
        self.final_proj = nn.Conv1d(d, d, kernel_size=1)
        self.final_line_proj = nn.Conv1d(d, d, kernel_size=1)
        for c in (self.final_proj, self.final_line_proj):
            nn.init.constant_(c.bias, 0.0)
            nn.init.orthogonal_(c.weight, gain=1)
        self.layer2idx = {}
        if inter is not None:
            self.inter_line_proj = nn.ModuleList([nn.Conv1d(d, d, kernel_size=1) for _ in inter])
            for i, layer in enumerate(inter):
                nn.init.constant_(self.inter_line_proj[i].bias, 0.0)
                nn.init.orthogonal_(self.inter_line_proj[i].weight, gain=1)
                self.layer2idx[layer] = i
        for layer in self.gnn.layers:
            layer.update.attention_fp32 = conf.attention_precision == "reference"
            layer.update.checkpointed_in_reference = bool(conf.checkpointed)
        for layer in self.gnn.line_layers:
            layer.checkpointed_in_reference = bool(conf.checkpointed)
        self.register_parameter("bin_score", nn.Parameter(torch.tensor(1.0)))
        self.register_parameter("line_bin_score", nn.Parameter(torch.tensor(1.0)))
        if conf.weights:
            path = Path(conf.weights)
            if not path.exists():
                raise FileNotFoundError(f"GlueStick weights '{conf.weights}' not found locally")
            sd = torch.load(str(path), map_location="cpu")
            if "model" in sd:
                sd = {k.replace("matcher.", "").replace("module.", ""): v for k, v in sd["model"].items()
                      if "matcher." in k}
            self.load_state_dict(sd, strict=False)

    # ------------------------------------------------------------------ heads
    def _filter(self, scores):
        with torch.no_grad():
            core = scores[:, :-1, :-1]
            max0, a0 = core.max(2)
            a1 = core.max(1).indices
            return ops.filter_matches(max0, a0, a1, self.conf.filter_threshold)

    def _point_head(self, d0, d1):
        d = self.conf.descriptor_dim
        s = d ** -0.25
        w, b = self.final_proj.weight.squeeze(-1) * s, self.final_proj.bias * s
        md0, md1 = ops.linear(d0, w, b), ops.linear(d1, w, b)       # S = md0 md1^T = scores / sqrt(d)
        r_raw, c_raw = ops.dual_lse(md0, md1)
        beta = self.bin_score.float()
        r, c = torch.logaddexp(r_raw, beta), torch.logaddexp(c_raw, beta)
        # (+ sum of exp over the rows of the matrix while it is written: the `sinkhorn_norm` statistic of the loss)
        return ops.assign_write(md0, md1, -0.5 * r, -0.5 * c, beta - r, beta - c, alpha=1.0, corner=0.0, with_expsum=True)

    def _line_head(self, ld0, ld1, idx0, idx1, graph0, graph1, proj):
        """gluestick.py:336-376: endpoint descriptors -> final_line_proj -> endpoint scores -> line scores (max over the
        two endpoint pairings) -> bin-augmented double softmax.  The projection runs on the gathered endpoint rows only
        (a 1x1 convolution commutes with the row gather); every [B, lines, lines] pass is a HIP kernel
        (csrc/line_head.hip, gf_bgemm)."""
        d = self.conf.descriptor_dim
        g0 = _conv_cl(ops.rows_gather(ld0, idx0, *graph0), proj)
        g1 = _conv_cl(ops.rows_gather(ld1, idx1, *graph1), proj)
        raw = ops.line_pair_scores(g0, g1, d ** -0.5)
        scores = ops.dense_log_double_softmax(raw, self.line_bin_score)
        return (scores, *self._filter(scores), raw)

    # ------------------------------------------------------------------ forward
    # torch.compile(model) (gluefactory/train.py:332-333): the HIP path is opaque to dynamo -- ctypes launches inside
    # autograd.Functions, host-side caches -- so forward / loss are one clean graph break and run eagerly
    @torch.compiler.disable
    def _forward(self, data):
        dev = data["keypoints0"].device
        b = len(data["keypoints0"])
        n0, n1 = data["keypoints0"].shape[1], data["keypoints1"].shape[1]
        nl0, nl1 = data["lines0"].shape[1], data["lines1"].shape[1]
        if n0 == 0 or n1 == 0:
            z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
            f = lambda *s: torch.full(s, -1, dtype=torch.int64, device=dev)          # noqa: E731
            return {"log_assignment": z(b, n0, n1), "matches0": f(b, n0), "matches1": f(b, n1),
                    "matching_scores0": z(b, n0), "matching_scores1": z(b, n1),
                    "line_log_assignment": z(b, nl0, nl1), "line_matches0": f(b, nl0), "line_matches1": f(b, nl1),
                    "line_matching_scores0": z(b, nl0), "line_matching_scores1": z(b, n1)}
        if not data["keypoints0"].is_cuda:
            raise RuntimeError("glue_factory_amd.GlueStick runs on the MI355X HIP path only (no CPU fallback)")
        T = torch.bfloat16 if (self.conf.mp or torch.is_autocast_enabled()) else torch.float32
        with torch.autocast(device_type="cuda", enabled=False):
            # one launch: compute-dtype + transposed weights + the layers' prepared (q | k | v) projections
            ops.precast(list(self.parameters()), T, key=id(self), derived=derived_specs_of(self) if T != torch.float32 else None)
            return self._forward_impl(data, T)

    def _forward_impl(self, data, T):
        conf = self.conf
        b = len(data["keypoints0"])
        v0, v1 = data["view0"], data["view1"]
        size0 = v0["image_size"] if "image_size" in v0 else v0["image"].shape
        size1 = v1["image_size"] if "image_size" in v1 else v1["image"].shape
        n0, n1 = data["keypoints0"].shape[1], data["keypoints1"].shape[1]
        nl0, nl1 = data["lines0"].shape[1], data["lines1"].shape[1]
        idx0, idx1 = data["lines_junc_idx0"].flatten(1, 2), data["lines_junc_idx1"].flatten(1, 2)
        have_lines = nl0 > 0 and nl1 > 0
        stacked = n0 == n1 and nl0 == nl1
        halves = 2 if stacked else 1

        def cat2(a, b_):
            return [torch.cat([a, b_], 0)] if stacked else [a, b_]

        kp = cat2(normalize_keypoints(data["keypoints0"], size0), normalize_keypoints(data["keypoints1"], size1))
        sc = cat2(data["keypoint_scores0"].float(), data["keypoint_scores1"].float())
        desc = cat2(data["descriptors0"].float(), data["descriptors1"].float())
        if conf.input_dim != conf.descriptor_dim:
            desc = [_conv_cl(x, self.input_proj) for x in desc]
        xs = [(x + self.kenc(k, s, halves)).to(T) for x, k, s in zip(desc, kp, sc)]
        jidx = cat2(idx0, idx1)
        if have_lines:
            ln = cat2(normalize_keypoints(data["lines0"].flatten(1, 2), size0).reshape(b, nl0, 2, 2),
                      normalize_keypoints(data["lines1"].flatten(1, 2), size1).reshape(b, nl1, 2, 2))
            lsc = cat2(data["line_scores0"].float(), data["line_scores1"].float())
            lenc = [self.lenc(l_, s_, halves).to(T) for l_, s_ in zip(ln, lsc)]     # cast once, not in every line layer
            graphs = [ops.line_graph(ji, x.shape[1]) for ji, x in zip(jidx, xs)]   # shared by all line layers
        inter_desc = {}
        inter = self.gnn.inter_supervision
        for i, layer in enumerate(self.gnn.layers):
            cross = layer.type == "cross"
            if stacked and not torch.is_tensor(layer.update.scaling) and layer.update.scaling == 1.0:
                xs = [layer.update(xs[0], cross=cross, halves=2, residual=True)]      # residual inside the last GEMM
            elif stacked:
                xs = [xs[0] + layer.update(xs[0], cross=cross, halves=2) * layer.update.scaling]
            else:
                d0, d1 = layer.update.forward_pair(xs[0], xs[1], cross=cross)
                xs = [xs[0] + d0 * layer.update.scaling, xs[1] + d1 * layer.update.scaling]
            if layer.type == "self" and have_lines:
                for _ in range(self.gnn.num_line_iterations):
                    pending = [] if len(xs) > 1 else None        # two per-image calls: their replays run in call order from one node
                    xs = [self.gnn.line_layers[i // 2](x, le, ji, halves, gr, replay_out=pending)
                          for x, le, ji, gr in zip(xs, lenc, jidx, graphs)]
                    if pending:
                        xs[0] = ops.replay_running_stats(xs[0], pending)
            if inter is not None and (i // 2) in inter and cross:
                inter_desc[i // 2] = xs
        split = (lambda t: (t[0][:b], t[0][b:])) if stacked else (lambda t: (t[0], t[1]))
        d0, d1 = split(xs)

        pred = {}
        kp_scores, kp_expsum = self._point_head(d0, d1)
        m0, m1, ms0, ms1 = self._filter(kp_scores)
        pred.update({"log_assignment": kp_scores, "_log_assignment_expsum": kp_expsum, "matches0": m0, "matches1": m1,
                     "matching_scores0": ms0, "matching_scores1": ms1})
        if have_lines:
            # junction graphs of the two images (the gather's backward is a segment sum over them)
            gr0, gr1 = (tuple(t[:b] for t in graphs[0]), tuple(t[b:] for t in graphs[0])) if stacked else (graphs[0], graphs[1])
            ls, lm0, lm1, lms0, lms1, raw = self._line_head(d0, d1, idx0, idx1, gr0, gr1, self.final_line_proj)
            for layer_id in (inter or []):
                e0, e1 = split(inter_desc[layer_id])
                li, a0, a1, s0, s1, _ = self._line_head(e0, e1, idx0, idx1, gr0, gr1,
                                                        self.inter_line_proj[self.layer2idx[layer_id]])
                pred.update({f"line_{layer_id}_log_assignment": li, f"line_{layer_id}_matches0": a0,
                             f"line_{layer_id}_matches1": a1, f"line_{layer_id}_matching_scores0": s0,
                             f"line_{layer_id}_matching_scores1": s1})
        else:
            dev = d0.device
            ls = torch.zeros(b, nl0, nl1, device=dev)
            lm0 = torch.full((b, nl0), -1, device=dev, dtype=torch.int64)
            lm1 = torch.full((b, nl1), -1, device=dev, dtype=torch.int64)
            lms0, lms1 = torch.zeros(b, nl0, device=dev), torch.zeros(b, nl1, device=dev)
            raw = torch.zeros(b, nl0, nl1, device=dev)
        pred.update({"line_log_assignment": ls, "line_matches0": lm0, "line_matches1": lm1,
                     "line_matching_scores0": lms0, "line_matching_scores1": lms1, "raw_line_scores": raw})
        return pred

    # ------------------------------------------------------------------ loss
    def sub_loss(self, pred, data, losses, bin_score, prefix="", layer=-1):
        suffix = "" if layer == -1 else f"{layer}_"
        weight = 1.0 if layer == -1 else self.conf.loss.inter_supervision[self.layer2idx[layer]]
        la = pred[prefix + suffix + "log_assignment"]
        neg0 = (data["gt_" + prefix + "matches0"] == -1).float()
        neg1 = (data["gt_" + prefix + "matches1"] == -1).float()
        pos_sum, num_pos, neg_sum = ops.nll_terms(la, data, neg0, neg1, prefix)   # one autograd node with the col0 vector
        num_pos = num_pos.clamp(min=1.0)
        num_neg = (neg0.sum(1) + neg1.sum(1)).clamp(min=1.0)
        nll_pos = -pos_sum / num_pos
        nll_neg = -neg_sum / num_neg
        bal = self.conf.loss.nll_balancing
        nll = bal * nll_pos + (1 - bal) * nll_neg
        losses[prefix + suffix + "assignment_nll"] = nll
        if self.conf.loss.nll_weight > 0:
            losses["total"] = losses["total"] + nll * self.conf.loss.nll_weight * weight
        if suffix == "":
            losses[prefix + "num_matchable"] = num_pos
            losses[prefix + "num_unmatchable"] = num_neg
            with torch.no_grad():
                es = pred.get("_" + prefix + "log_assignment_expsum")       # accumulated while the matrix was written
                losses[prefix + "sinkhorn_norm"] = (es / (la.shape[1] - 1) if es is not None
                                                    else la.exp()[:, :-1].sum(2).mean(1))
            losses[prefix + "bin_score"] = bin_score[None]
        return losses

    @torch.compiler.disable
    def loss(self, pred, data):
        losses = {"total": 0}
        if not (data["keypoints0"].shape[1] == 0 or data["keypoints1"].shape[1] == 0):
            losses = self.sub_loss(pred, data, losses, self.bin_score, prefix="")
        has_lines = ("lines0" in data and "lines1" in data and data["lines0"].shape[1] > 0
                     and data["lines1"].shape[1] > 0)
        if has_lines:
            losses = self.sub_loss(pred, data, losses, self.line_bin_score, prefix="line_")
        if self.conf.inter_supervision:
            for layer in self.conf.inter_supervision:
                losses = self.sub_loss(pred, data, losses, self.line_bin_score, prefix="line_", layer=layer)
        metrics = {}
        if not self.training:
            if "matches0" in pred and pred["matches0"].shape[1] > 0 and pred["matches1"].shape[1] > 0:
                metrics.update(matcher_metrics(pred, data, prefix=""))
            if "line_matches0" in pred and has_lines:
                metrics.update(matcher_metrics(pred, data, prefix="line_"))
            if self.conf.inter_supervision:
                for layer in self.conf.inter_supervision:
                    metrics.update(matcher_metrics(pred, data, prefix=f"line_{layer}_", prefix_gt="line_"))
        return losses, metrics


__main_model__ = GlueStick
