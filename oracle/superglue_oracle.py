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


# --------------------------------------------------------------------------- sharpened case (bit-exact matches)
def sharp_case(batch, n, gnn_layers, seed, size=(1024, 1024), damp=0.01, sharp=16.0, noise=0.03, unmatched=0.125,
               kenc_damp=0.004, bin_score=1.0):
    """Seeded weights + a seeded pair batch on which EVERY mutual-nearest-neighbour decision of superglue.py:300-320 is
    decisive (see `decisiveness`), so matches0/1 can be compared bit for bit, in fp32 and bf16: the first
    (1 - unmatched) * n keypoints of image 1 are image 0's (warped by a similarity, then permuted) with descriptors
    normalise(d0 + noise N(0, I)), the rest are fresh points that must come out as -1; the last convolution of the
    keypoint encoder and of every propagation MLP (`mlp.3`) is damped so the residual stream stays close to the input
    descriptors, and final_proj is `sharp * I + init`.  With the filter threshold at 0.2 a matched row's margin cannot exceed
    -log 0.2 = 1.6 (its probability is at most 1); the defaults reach 1.5.  Returns (params, data) of CPU fp32 tensors."""
    p = init_params(256, gnn_layers=gnn_layers, seed=seed)
    last_kenc = max(int(k.split(".")[2]) for k in p if k.startswith("kenc.encoder.") and k.endswith(".weight") and p[k].dim() == 3)
    for k in p:
        if k.endswith("mlp.3.weight"):
            p[k] = p[k] * damp
    p[f"kenc.encoder.{last_kenc}.weight"] = p[f"kenc.encoder.{last_kenc}.weight"] * kenc_damp
    p["bin_score"] = torch.tensor(float(bin_score))
    p["final_proj.weight"] = p["final_proj.weight"] + sharp * torch.eye(256)[:, :, None]
    g = torch.Generator().manual_seed(seed + 1)
    w, h = size
    wh = torch.tensor([w, h], dtype=torch.float32)
    nm = n - int(unmatched * n)
    kp0 = torch.rand(batch, n, 2, generator=g) * wh
    a = math.radians(10.0)
    c, s = math.cos(a) * 1.1, math.sin(a) * 1.1
    ctr = wh / 2
    rot = torch.tensor([[c, -s], [s, c]])
    d0 = F.normalize(torch.randn(batch, n, 256, generator=g), dim=-1)
    kp1 = (kp0 - ctr) @ rot.T + ctr + torch.tensor([15.0, -10.0])
    d1 = F.normalize(d0 + noise * torch.randn(batch, n, 256, generator=g), dim=-1)
    kp1[:, nm:] = torch.rand(batch, n - nm, 2, generator=g) * wh                         # fresh, unmatched points
    d1[:, nm:] = F.normalize(torch.randn(batch, n - nm, 256, generator=g), dim=-1)
    perm = torch.stack([torch.randperm(n, generator=g) for _ in range(batch)])          # new position j holds old perm[j]
    kp1 = kp1.gather(1, perm[..., None].expand(-1, -1, 2))
    d1 = d1.gather(1, perm[..., None].expand(-1, -1, 256))
    inv = torch.argsort(perm, 1)                                                        # old index -> new position
    m0 = torch.where(torch.arange(n)[None] < nm, inv, -1)                               # image-0 point i <-> old index i
    m1 = torch.where(perm < nm, perm, -1)
    gt = torch.zeros(batch, n, n, dtype=torch.bool)
    gt.scatter_(2, m0.clamp(min=0)[..., None], (m0 >= 0)[..., None])
    data = {"keypoints0": kp0, "keypoints1": kp1, "descriptors0": d0, "descriptors1": d1,
            "keypoint_scores0": torch.rand(batch, n, generator=g), "keypoint_scores1": torch.rand(batch, n, generator=g),
            "view0": {"image_size": wh[None].repeat(batch, 1)}, "view1": {"image_size": wh[None].repeat(batch, 1)},
            "gt_assignment": gt, "gt_assignment_col0": m0.clone(), "gt_matches0": m0, "gt_matches1": m1}
    return p, data


def decisiveness(la, th):
    """Smallest margin by which any row / column decision of the mutual-NN filter (superglue.py:300-320,
    gluestick.py:321-334) on the log-assignment `la` [B,M+1,N+1] is taken: a row (column) either has its best core
    entry BELOW log(th) by the margin (it is unmatched whatever its arg-max is), or its best entry beats the runner-up
    AND log(th) by the margin.  Outputs computed from any `la'` with max |la' - la| < margin / 2 are identical."""
    core = la[:, :-1, :-1]
    lth = math.log(th) if th > 0 else -float("inf")
    worst = float("inf")
    for dim in (2, 1):
        t = core.topk(2, dim=dim).values
        t1, t2 = (t[..., 0], t[..., 1]) if dim == 2 else (t[:, 0], t[:, 1])
        matched = torch.minimum(t1 - t2, t1 - lth)
        unmatched = lth - t1
        worst = min(worst, float(torch.maximum(matched, unmatched).min()))
    return worst
