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


# --------------------------------------------------------------------------- sharpened case (bit-exact matches)
def sharp_case(batch, n_kpts, n_lines, gnn_layers, seed, size=(1024, 1024), damp=0.01, sharp=16.0, noise=0.03,
               unmatched=0.125, enc_damp=0.004, share_junctions=0.25):
    """Seeded weights + a seeded point+line pair batch on which EVERY mutual-nearest-neighbour decision of
    gluestick.py:321-376 -- points and lines -- is decisive (superglue_oracle.decisiveness), so matches0/1 and
    line_matches0/1 can be compared bit for bit in fp32 and bf16.
    Image 0: 2*n_lines junction slots + n_kpts keypoints; a fraction of the line endpoints re-uses an earlier junction
    (shared nodes in the junction graph).  Image 1: the same junction graph under a permutation of the junction slots
    and of the lines (half of the lines with their endpoints swapped: the `max(s00 + s11, s01 + s10)` of :352-357),
    coordinates warped by a similarity, descriptors normalise(d + noise N(0, I)); the last `unmatched` fraction of
    the lines (which never share a junction) and of the keypoints have FRESH counterparts in image 1 (coordinates and
    descriptors) and must come out as -1, as do those lines' junction points.  The last convolution of both encoders,
    of every propagation MLP and of every line-layer MLP is damped; final_proj / final_line_proj are `sharp * I + init`.
    Returns (params, data) of CPU fp32 tensors incl. the point and line ground truth."""
    p = init_params(256, gnn_layers=gnn_layers, inter=None, seed=seed)
    for k in list(p):
        if k.endswith("mlp.3.weight"):
            p[k] = p[k] * damp
        if k in ("kenc.encoder.12.weight", "lenc.encoder.12.weight"):
            p[k] = p[k] * enc_damp
        if k in ("final_proj.weight", "final_line_proj.weight"):
            p[k] = p[k] + sharp * torch.eye(256)[:, :, None]
    g = torch.Generator().manual_seed(seed + 1)
    w, h = size
    wh = torch.tensor([w, h], dtype=torch.float32)
    nl, nj = n_lines, 2 * n_lines
    nlm = nl - int(unmatched * nl)                 # matched lines: 0 .. nlm-1 (image-0 numbering)
    nkm = n_kpts - int(unmatched * n_kpts)         # matched keypoints
    a = math.radians(10.0)
    c, s = math.cos(a) * 1.1, math.sin(a) * 1.1
    ctr = wh / 2
    rot = torch.tensor([[c, -s], [s, c]])
    warp = lambda x: (x - ctr) @ rot.T + ctr + torch.tensor([15.0, -10.0])   # noqa: E731
    unit = lambda *shape: F.normalize(torch.randn(*shape, 256, generator=g), dim=-1)   # noqa: E731
    # ---- image 0
    p0 = torch.rand(batch, nl, 2, generator=g) * (wh - 200) + 100
    ang = torch.rand(batch, nl, generator=g) * 2 * math.pi
    length = 15 + torch.rand(batch, nl, generator=g) * 60
    lines0 = torch.stack([p0, p0 + length[..., None] * torch.stack([torch.cos(ang), torch.sin(ang)], -1)], 2)
    idx0 = torch.arange(nj).reshape(1, nl, 2).repeat(batch, 1, 1)
    lnum = torch.arange(nl)[None, :, None]
    share = (torch.rand(batch, nl, 2, generator=g) < share_junctions) & (lnum >= 1) & (lnum < nlm)
    share[:, :, 0] &= ~share[:, :, 1]              # at most one shared endpoint per line: no two lines on the same junction pair
    prev = (torch.rand(batch, nl, 2, generator=g) * (2 * lnum)).long()                 # a junction slot of an EARLIER line
    idx0 = torch.where(share, prev, idx0)
    jc0 = lines0.reshape(batch, nj, 2).clone()
    jd0 = unit(batch, nj)
    kp0 = torch.rand(batch, n_kpts, 2, generator=g) * wh
    kd0 = unit(batch, n_kpts)
    # ---- image 1 in image-0 numbering, then permuted
    lines1 = warp(lines0.reshape(batch, nj, 2)).reshape(batch, nl, 2, 2)
    lines1[:, nlm:] = torch.rand(batch, nl - nlm, 2, 2, generator=g) * wh
    jc1 = warp(jc0)
    jd1 = F.normalize(jd0 + noise * torch.randn(batch, nj, 256, generator=g), dim=-1)
    jc1[:, 2 * nlm:] = lines1[:, nlm:].reshape(batch, -1, 2)
    jd1[:, 2 * nlm:] = unit(batch, nj - 2 * nlm)
    kp1 = warp(kp0)
    kd1 = F.normalize(kd0 + noise * torch.randn(batch, n_kpts, 256, generator=g), dim=-1)
    kp1[:, nkm:] = torch.rand(batch, n_kpts - nkm, 2, generator=g) * wh
    kd1[:, nkm:] = unit(batch, n_kpts - nkm)
    lperm = torch.stack([torch.randperm(nl, generator=g) for _ in range(batch)])        # new line position <- old line
    jperm = torch.stack([torch.randperm(nj, generator=g) for _ in range(batch)])        # new junction slot <- old slot
    kperm = torch.stack([torch.randperm(n_kpts, generator=g) for _ in range(batch)])
    linv, jinv, kinv = (torch.argsort(t, 1) for t in (lperm, jperm, kperm))             # old -> new
    flip = torch.rand(batch, nl, generator=g) < 0.5                                     # (in new line order)
    l1 = lines1.gather(1, lperm[:, :, None, None].expand(-1, -1, 2, 2))
    i1 = jinv.gather(1, idx0.gather(1, lperm[:, :, None].expand(-1, -1, 2)).flatten(1)).reshape(batch, nl, 2)
    l1 = torch.where(flip[:, :, None, None], l1.flip(2), l1)
    i1 = torch.where(flip[:, :, None], i1.flip(2), i1)
    g2 = lambda t, pm: t.gather(1, pm[..., None].expand(-1, -1, t.shape[-1]))            # noqa: E731
    pts0 = torch.cat([jc0, kp0], 1)
    pts1 = torch.cat([g2(jc1, jperm), g2(kp1, kperm)], 1)
    des0 = torch.cat([jd0, kd0], 1)
    des1 = torch.cat([g2(jd1, jperm), g2(kd1, kperm)], 1)
    ar = lambda n: torch.arange(n)[None]                                                 # noqa: E731
    m0 = torch.cat([torch.where(ar(nj) < 2 * nlm, jinv, -1), torch.where(ar(n_kpts) < nkm, kinv + nj, -1)], 1)
    m1 = torch.cat([torch.where(jperm < 2 * nlm, jperm, -1), torch.where(kperm < nkm, kperm + nj, -1)], 1)
    lm0 = torch.where(ar(nl) < nlm, linv, -1)
    lm1 = torch.where(lperm < nlm, lperm, -1)
    nt = nj + n_kpts
    gt = torch.zeros(batch, nt, nt, dtype=torch.bool)
    gt.scatter_(2, m0.clamp(min=0)[..., None], (m0 >= 0)[..., None])
    gtl = torch.zeros(batch, nl, nl, dtype=torch.bool)
    gtl.scatter_(2, lm0.clamp(min=0)[..., None], (lm0 >= 0)[..., None])
    data = {"keypoints0": pts0, "keypoints1": pts1, "descriptors0": des0, "descriptors1": des1,
            "keypoint_scores0": torch.rand(batch, nt, generator=g), "keypoint_scores1": torch.rand(batch, nt, generator=g),
            "lines0": lines0, "lines1": l1, "lines_junc_idx0": idx0, "lines_junc_idx1": i1,
            "line_scores0": torch.rand(batch, nl, generator=g), "line_scores1": torch.rand(batch, nl, generator=g),
            "view0": {"image_size": wh[None].repeat(batch, 1)}, "view1": {"image_size": wh[None].repeat(batch, 1)},
            "gt_assignment": gt, "gt_assignment_col0": m0.clone(), "gt_matches0": m0, "gt_matches1": m1,
            "gt_line_assignment": gtl, "gt_line_assignment_col0": lm0.clone(), "gt_line_matches0": lm0, "gt_line_matches1": lm1}
    return p, data
