"""CPU restatement (torch) of the SuperGlue matcher: keypoint-encoder MLP, attentional GNN,
log-optimal-transport assignment with dustbins, mutual-NN filter and NLL loss.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference restated (gluefactory_nonfree/superglue.py): MLP :70-79, normalize_keypoints :82-93,
KeypointEncoder :96-109, attention :112-116, MultiHeadedAttention :119-135 (head index is the
FASTEST channel index: view(b, dim, h, n)), AttentionalPropagation :138-147, AttentionalGNN
:150-183, log_optimal_transport :194-214 (-> sinkhorn_oracle), _forward :266-320, loss :322-352.
Channels-last ([B,N,C]) functional form over the reference's state_dict names; BatchNorm uses
batch statistics when ``training`` (one call per image, as the reference does).
"""
import math

import torch
import torch.nn.functional as F

from .lightglue_oracle import filter_matches
from .sinkhorn_oracle import log_optimal_transport


def init_params(dim=256, kenc_layers=(32, 64, 128, 256), gnn_layers=18, seed=0, dtype=torch.float32):
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

    chans = [3] + list(kenc_layers) + [dim]
    idx = 0
    for i in range(1, len(chans)):
        last = i == len(chans) - 1
        conv(f"kenc.encoder.{idx}", chans[i], chans[i - 1], zero_bias=last)
        idx += 1
        if not last:
            bn(f"kenc.encoder.{idx}", chans[i])
            idx += 2
    for i in range(gnn_layers):
        base = f"gnn.layers.{i}"
        conv(f"{base}.attn.merge", dim, dim)
        for j in range(3):
            conv(f"{base}.attn.proj.{j}", dim, dim)
        conv(f"{base}.mlp.0", 2 * dim, 2 * dim)
        bn(f"{base}.mlp.1", 2 * dim)
        conv(f"{base}.mlp.3", dim, 2 * dim, zero_bias=True)
    conv("final_proj", dim, dim)
    p["bin_score"] = torch.tensor(1.0, dtype=dtype)
    return p


def trainable_names(p):
    return [k for k in p if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def _conv(p, name, x):  # x [B,N,Cin], weight [Cout,Cin,1]
    return F.linear(x, p[name + ".weight"].squeeze(-1), p[name + ".bias"])


def _bn(p, name, x, training):
    b, n, c = x.shape
    y = F.batch_norm(x.reshape(b * n, c), p[name + ".running_mean"].clone(), p[name + ".running_var"].clone(),
                     p[name + ".weight"], p[name + ".bias"], training=training, momentum=0.1, eps=1e-5)
    return y.reshape(b, n, c)


def normalize_keypoints(kpts, size):
    size = size.to(kpts)
    return (kpts - size[:, None] / 2) / (size.max(1).values * 0.7)[:, None, None]


def keypoint_encoder(p, kpts, scores, training):
    x = torch.cat([kpts, scores[..., None]], -1)
    names = sorted({int(k.split(".")[2]) for k in p if k.startswith("kenc.encoder.") and k.endswith(".bias")
                    and p[k[:-4] + "weight"].dim() == 3})
    for i, idx in enumerate(names):
        x = _conv(p, f"kenc.encoder.{idx}", x)
        if i < len(names) - 1:
            x = F.relu(_bn(p, f"kenc.encoder.{idx + 1}", x, training))
    return x


def propagate(p, base, x, src, training, heads=4):
    b, n, d = x.shape
    hd = d // heads

    def split(t):  # channel c = channel_in_head * heads + head
        return t.reshape(t.shape[0], t.shape[1], hd, heads).permute(0, 3, 1, 2)  # [B,H,N,hd]

    q = split(_conv(p, f"{base}.attn.proj.0", x))
    k = split(_conv(p, f"{base}.attn.proj.1", src))
    v = split(_conv(p, f"{base}.attn.proj.2", src))
    prob = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, -1)
    o = (prob @ v).permute(0, 2, 3, 1).reshape(b, n, d)       # back to c = ch * heads + head
    msg = _conv(p, f"{base}.attn.merge", o)
    h = _conv(p, f"{base}.mlp.0", torch.cat([x, msg], -1))
    h = F.relu(_bn(p, f"{base}.mlp.1", h, training))
    return _conv(p, f"{base}.mlp.3", h)


def forward(p, data, layer_names, iters, filter_threshold=0.2, training=False):
    k0 = normalize_keypoints(data["keypoints0"], data["image_size0"])
    k1 = normalize_keypoints(data["keypoints1"], data["image_size1"])
    d0 = data["descriptors0"] + keypoint_encoder(p, k0, data["keypoint_scores0"], training)
    d1 = data["descriptors1"] + keypoint_encoder(p, k1, data["keypoint_scores1"], training)
    for i, name in enumerate(layer_names):
        base = f"gnn.layers.{i}"
        s0, s1 = (d0, d1) if name == "self" else (d1, d0)
        delta0 = propagate(p, base, d0, s0, training)
        delta1 = propagate(p, base, d1, s1, training)
        d0, d1 = d0 + delta0, d1 + delta1
    m0, m1 = _conv(p, "final_proj", d0), _conv(p, "final_proj", d1)
    cost = m0 @ m1.transpose(1, 2) / m0.shape[-1] ** 0.5
    scores = log_optimal_transport(cost, p["bin_score"], iters)
    a0, a1, s0, s1 = filter_matches(scores, filter_threshold)
    return {"sinkhorn_cost": cost, "log_assignment": scores, "matches0": a0, "matches1": a1,
            "matching_scores0": s0, "matching_scores1": s1}


def loss(p, pred, data, balancing=0.5):
    la = pred["log_assignment"]
    pos = data["gt_assignment"].to(la.dtype)
    neg0 = (data["gt_matches0"] == -1).to(la.dtype)
    neg1 = (data["gt_matches1"] == -1).to(la.dtype)
    num_pos = pos.sum((1, 2)).clamp(min=1.0)
    num_neg = (neg0.sum(1) + neg1.sum(1)).clamp(min=1.0)
    nll_pos = -(la[:, :-1, :-1] * pos).sum((1, 2)) / num_pos
    nll_neg = -((la[:, :-1, -1] * neg0).sum(1) + (la[:, -1, :-1] * neg1).sum(1)) / num_neg
    nll = balancing * nll_pos + (1 - balancing) * nll_neg
    return {"total": nll, "assignment_nll": nll, "nll_pos": nll_pos, "nll_neg": nll_neg,
            "num_matchable": num_pos, "num_unmatchable": num_neg, "bin_score": p["bin_score"][None]}


def train_step_grads(p, data, layer_names, iters):
    names = trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    pred = forward(q, data, layer_names, iters, training=True)
    losses = loss(q, pred, data)
    losses["total"].mean().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return pred, losses, grads
