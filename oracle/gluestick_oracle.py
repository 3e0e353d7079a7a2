"""CPU restatement (torch) of the GlueStick point+line matcher.  TEST INFRASTRUCTURE ONLY.

Reference restated (gluefactory/models/matchers/gluestick.py): normalize_keypoints :477-488,
KeypointEncoder :491-499, EndPtEncoder :502-521, attention/MultiHeadedAttention :524-550,
AttentionalPropagation :553-566, GNNLayer :569-586, LineLayer :589-691 (mean aggregation,
``line_attention: False``), AttentionalGNN :694-769, log_double_softmax :772-783,
_forward :143-319, _get_matches :321-334, _get_line_matches :336-376, sub_loss/loss :378-462.
Channels-last functional form over the reference's state_dict names.
"""
import math

import torch
import torch.nn.functional as F

from .lightglue_oracle import filter_matches
from .superglue_oracle import _bn, _conv, propagate


def init_params(dim=256, kenc_layers=(32, 64, 128, 256), gnn_layers=18, inter=None, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    p = {}

    def conv(name, out_c, in_c, zero_bias=False):
        bound = 1.0 / math.sqrt(in_c)
        p[name + ".weight"] = ((torch.rand(out_c, in_c, 1, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
        b = (torch.rand(out_c, generator=g, dtype=torch.float64) * 2 - 1) * bound
        p[name + ".bias"] = (b * 0 if zero_bias else b).to(dtype)

    def bn(name, c):
        p[name + ".weight"] = (1 + 0.1 * torch.randn(c, generator=g, dtype=torch.float64)).to(dtype)
        p[name + ".bias"] = (0.1 * torch.randn(c, generator=g, dtype=torch.float64)).to(dtype)
        p[name + ".running_mean"] = (0.1 * torch.randn(c, generator=g, dtype=torch.float64)).to(dtype)
        p[name + ".running_var"] = (1 + 0.2 * torch.rand(c, generator=g, dtype=torch.float64)).to(dtype)
        p[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def mlp(prefix, chans):
        idx = 0
        for i in range(1, len(chans)):
            last = i == len(chans) - 1
            conv(f"{prefix}.{idx}", chans[i], chans[i - 1], zero_bias=last)
            idx += 1
            if not last:
                bn(f"{prefix}.{idx}", chans[i])
                idx += 2

    mlp("kenc.encoder", [3] + list(kenc_layers) + [dim])
    mlp("lenc.encoder", [5] + list(kenc_layers) + [dim])
    for i in range(gnn_layers):
        base = f"gnn.layers.{i}.update"
        conv(f"{base}.attn.merge", dim, dim)
        for j in range(3):
            conv(f"{base}.attn.proj.{j}", dim, dim)
        mlp(f"{base}.mlp", [2 * dim, 2 * dim, dim])
    for k in range(gnn_layers // 2):
        mlp(f"gnn.line_layers.{k}.mlp", [3 * dim, 2 * dim, dim])
    conv("final_proj", dim, dim, zero_bias=True)
    conv("final_line_proj", dim, dim, zero_bias=True)
    for i, _ in enumerate(inter or []):
        conv(f"inter_line_proj.{i}", dim, dim, zero_bias=True)
    p["bin_score"] = torch.tensor(1.0, dtype=dtype)
    p["line_bin_score"] = torch.tensor(0.8, dtype=dtype)
    return p


def add_line_attention_params(p, seed=0, dim=256):
    """Seeded parameters of LineLayer's proj_node / proj_neigh (gluestick.py:596-597) for every line layer found in
    ``p`` (used by the line_attention golden; the oracle functions themselves cover line_attention=False only)."""
    g = torch.Generator().manual_seed(seed)
    k = 0
    while f"gnn.line_layers.{k}.mlp.0.weight" in p:
        base = f"gnn.line_layers.{k}."
        p[base + "proj_node.weight"] = torch.randn(dim, dim, 1, generator=g) / dim ** 0.5
        p[base + "proj_node.bias"] = torch.randn(dim, generator=g) * 0.1
        p[base + "proj_neigh.weight"] = torch.randn(dim, 2 * dim, 1, generator=g) / (2 * dim) ** 0.5
        p[base + "proj_neigh.bias"] = torch.randn(dim, generator=g) * 0.1
        k += 1
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def normalize_keypoints(kpts, size):
    size = size.to(kpts)
    return (kpts - size[:, None] / 2) / (size.max(1, keepdim=True).values * 0.7)[:, None]


def _mlp(p, prefix, x, training):
    idxs = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in p
                   if k.startswith(prefix + ".") and k.endswith(".weight") and p[k].dim() == 3})
    for i, idx in enumerate(idxs):
        x = _conv(p, f"{prefix}.{idx}", x)
        if i < len(idxs) - 1:
            x = F.relu(_bn(p, f"{prefix}.{idx + 1}", x, training))
    return x


def endpoint_encoder(p, lines, scores, training):
    """lines [B,Nl,2,2] (normalised) -> [B, 2Nl, D]: (xy, offset to the other endpoint, score)."""
    b, nl = lines.shape[:2]
    off = lines[:, :, 1] - lines[:, :, 0]
    off = torch.stack([off, -off], 2).reshape(b, 2 * nl, 2)
    x = torch.cat([lines.reshape(b, 2 * nl, 2), off, scores.repeat(1, 2)[..., None]], -1)
    return _mlp(p, "lenc.encoder", x, training)


def line_layer(p, prefix, ldesc, line_enc, junc_idx, training):
    """ldesc [B,N,D]; line_enc [B,2Nl,D]; junc_idx [B,2Nl] -> ldesc + mean of endpoint messages."""
    b, n, d = ldesc.shape
    idx = junc_idx[..., None].expand(-1, -1, d)
    ld = ldesc.gather(1, idx)                                          # this endpoint
    ld2 = ld.reshape(b, -1, 2, d).flip(2).reshape(b, -1, d)            # the other endpoint of the line
    upd = _mlp(p, prefix + ".mlp", torch.cat([ld, ld2, line_enc], -1), training)
    out = torch.zeros_like(ldesc).scatter_reduce(1, idx, upd, reduce="mean", include_self=False)
    return ldesc + out


def log_double_softmax(scores, bin_score):
    b, m, n = scores.shape
    beta = bin_score.to(scores).reshape(1, 1, 1)
    r = torch.logsumexp(torch.cat([scores, beta.expand(b, m, 1)], 2), 2)
    c = torch.logsumexp(torch.cat([scores, beta.expand(b, 1, n)], 1), 1)
    out = scores.new_zeros(b, m + 1, n + 1)
    out[:, :m, :n] = scores - 0.5 * (r[:, :, None] + c[:, None, :])
    out[:, :m, n] = beta.reshape(1, 1) - r
    out[:, m, :n] = beta.reshape(1, 1) - c
    return out


def line_head(p, proj, d0, d1, idx0, idx1, dim):
    """d0/d1: descriptors of the 2*Nl endpoint block; returns (log assignment, raw line scores)."""
    m0, m1 = _conv(p, proj, d0), _conv(p, proj, d1)
    s = m0 @ m1.transpose(1, 2) / dim ** 0.5
    s = s.gather(2, idx1[:, None, :].expand(-1, s.shape[1], -1))
    s = s.gather(1, idx0[:, :, None].expand(-1, -1, s.shape[2]))
    b = s.shape[0]
    s = s.reshape(b, idx0.shape[1] // 2, 2, idx1.shape[1] // 2, 2)
    raw = 0.5 * torch.maximum(s[:, :, 0, :, 0] + s[:, :, 1, :, 1], s[:, :, 0, :, 1] + s[:, :, 1, :, 0])
    return log_double_softmax(raw, p["line_bin_score"]), raw


def forward(p, data, layer_names, filter_threshold=0.2, training=False, inter=None):
    k0 = normalize_keypoints(data["keypoints0"], data["image_size0"])
    k1 = normalize_keypoints(data["keypoints1"], data["image_size1"])
    b = k0.shape[0]
    nl0, nl1 = data["lines0"].shape[1], data["lines1"].shape[1]
    idx0, idx1 = data["lines_junc_idx0"].flatten(1, 2), data["lines_junc_idx1"].flatten(1, 2)
    kin = lambda k, s: _mlp(p, "kenc.encoder", torch.cat([k, s[..., None]], -1), training)  # noqa: E731
    d0 = data["descriptors0"] + kin(k0, data["keypoint_scores0"])
    d1 = data["descriptors1"] + kin(k1, data["keypoint_scores1"])
    l0 = normalize_keypoints(data["lines0"].flatten(1, 2), data["image_size0"]).reshape(b, nl0, 2, 2)
    l1 = normalize_keypoints(data["lines1"].flatten(1, 2), data["image_size1"]).reshape(b, nl1, 2, 2)
    le0 = endpoint_encoder(p, l0, data["line_scores0"], training)
    le1 = endpoint_encoder(p, l1, data["line_scores1"], training)
    inter_desc = {}
    for i, name in enumerate(layer_names):
        base = f"gnn.layers.{i}.update"
        s0, s1 = (d0, d1) if name == "self" else (d1, d0)
        delta0 = propagate(p, base, d0, s0, training)
        delta1 = propagate(p, base, d1, s1, training)
        d0, d1 = d0 + delta0, d1 + delta1
        if name == "self" and nl0 > 0 and nl1 > 0:
            d0 = line_layer(p, f"gnn.line_layers.{i // 2}", d0, le0, idx0, training)
            d1 = line_layer(p, f"gnn.line_layers.{i // 2}", d1, le1, idx1, training)
        if inter is not None and (i // 2) in inter and name == "cross":
            inter_desc[i // 2] = (d0, d1)
    dim = d0.shape[-1]
    m0, m1 = _conv(p, "final_proj", d0), _conv(p, "final_proj", d1)
    kp = log_double_softmax(m0 @ m1.transpose(1, 2) / dim ** 0.5, p["bin_score"])
    a0, a1, s0, s1 = filter_matches(kp, filter_threshold)
    pred = {"log_assignment": kp, "matches0": a0, "matches1": a1, "matching_scores0": s0, "matching_scores1": s1}
    ls, raw = line_head(p, "final_line_proj", d0[:, :2 * nl0], d1[:, :2 * nl1], idx0, idx1, dim)
    la0, la1, ls0, ls1 = filter_matches(ls, filter_threshold)
    pred.update({"line_log_assignment": ls, "line_matches0": la0, "line_matches1": la1,
                 "line_matching_scores0": ls0, "line_matching_scores1": ls1, "raw_line_scores": raw})
    for j, layer in enumerate(inter or []):
        e0, e1 = inter_desc[layer]
        li, _ = line_head(p, f"inter_line_proj.{j}", e0[:, :2 * nl0], e1[:, :2 * nl1], idx0, idx1, dim)
        pred[f"line_{layer}_log_assignment"] = li
        x0, x1, y0, y1 = filter_matches(li, filter_threshold)
        pred.update({f"line_{layer}_matches0": x0, f"line_{layer}_matches1": x1,
                     f"line_{layer}_matching_scores0": y0, f"line_{layer}_matching_scores1": y1})
    return pred


def _sub_loss(losses, la, pos, m0, m1, bin_score, prefix, suffix, weight, balancing=0.5):
    pos = pos.to(la.dtype)
    neg0, neg1 = (m0 == -1).to(la.dtype), (m1 == -1).to(la.dtype)
    num_pos = pos.sum((1, 2)).clamp(min=1.0)
    num_neg = (neg0.sum(1) + neg1.sum(1)).clamp(min=1.0)
    nll_pos = -(la[:, :-1, :-1] * pos).sum((1, 2)) / num_pos
    nll_neg = -((la[:, :-1, -1] * neg0).sum(1) + (la[:, -1, :-1] * neg1).sum(1)) / num_neg
    nll = balancing * nll_pos + (1 - balancing) * nll_neg
    losses[prefix + suffix + "assignment_nll"] = nll
    losses["total"] = losses["total"] + nll * weight
    if suffix == "":
        losses[prefix + "num_matchable"] = num_pos
        losses[prefix + "num_unmatchable"] = num_neg
        losses[prefix + "sinkhorn_norm"] = la.exp()[:, :-1].sum(2).mean(1)
        losses[prefix + "bin_score"] = bin_score[None]


def loss(p, pred, data, inter=None, inter_weights=(0.3, 0.6)):
    losses = {"total": 0}
    _sub_loss(losses, pred["log_assignment"], data["gt_assignment"], data["gt_matches0"], data["gt_matches1"],
              p["bin_score"], "", "", 1.0)
    _sub_loss(losses, pred["line_log_assignment"], data["gt_line_assignment"], data["gt_line_matches0"],
              data["gt_line_matches1"], p["line_bin_score"], "line_", "", 1.0)
    for j, layer in enumerate(inter or []):
        _sub_loss(losses, pred[f"line_{layer}_log_assignment"], data["gt_line_assignment"],
                  data["gt_line_matches0"], data["gt_line_matches1"], p["line_bin_score"], "line_",
                  f"{layer}_", inter_weights[j])
    return losses


def train_step_grads(p, data, layer_names, inter=None):
    names = trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    pred = forward(q, data, layer_names, training=True, inter=inter)
    losses = loss(q, pred, data, inter=inter)
    losses["total"].mean().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return pred, losses, grads
