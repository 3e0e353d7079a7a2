"""CPU restatement (torch, fp32/fp64) of the LightGlue train-step math.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written from the math of the
reference, not copied: a purely functional model over a flat parameter dict that
uses the reference's ``state_dict`` names (SURVEY.md §8 note S), so the same
weights drive the reference, this oracle and the HIP product path.

Reference lines restated (all under /root/reference/gluefactory/models/):
  normalize_keypoints      matchers/lightglue.py:27-39
  fourier_encoding         matchers/lightglue.py:52-65
  rotary                   matchers/lightglue.py:42-49
  self_block               matchers/lightglue.py:150-163
  cross_block              matchers/lightglue.py:192-221 (non-flash branch)
  log_double_softmax       matchers/lightglue.py:256-268
  match_assignment         matchers/lightglue.py:278-287
  filter_matches           matchers/lightglue.py:293-309
  forward                  matchers/lightglue.py:412-543 (training / plain eval path)
  nll                      utils/losses.py:6-73
  token_confidence_loss    matchers/lightglue.py:81-94
  loss                     matchers/lightglue.py:578-627
Backward comes from torch autograd on this restatement (fp64-capable).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- params
def init_params(n_layers=9, dim=256, heads=4, input_dim=None, seed=0, dtype=torch.float32,
                pos_dim=2):
    """Seeded random parameters under the reference's state_dict names.

    Uses torch's default Linear init distributions (uniform +-1/sqrt(fan_in)) so the
    statistics resemble a freshly constructed reference module; exact equality with
    a reference instance is obtained by loading one state_dict into both.
    """
    g = torch.Generator().manual_seed(seed)
    hd = dim // heads
    p = {}

    def lin(name, out_f, in_f, bias=True):
        bound = 1.0 / math.sqrt(in_f)
        p[name + ".weight"] = ((torch.rand(out_f, in_f, generator=g, dtype=torch.float64) * 2 - 1)
                               * bound).to(dtype)
        if bias:
            p[name + ".bias"] = ((torch.rand(out_f, generator=g, dtype=torch.float64) * 2 - 1)
                                 * bound).to(dtype)

    if input_dim is not None and input_dim != dim:
        lin("input_proj", dim, input_dim)
    p["posenc.Wr.weight"] = torch.randn(hd // 2, pos_dim, generator=g, dtype=torch.float64).to(dtype)
    for i in range(n_layers):
        for blk, names in (("self_attn", (("Wqkv", 3 * dim, dim), ("out_proj", dim, dim))),
                           ("cross_attn", (("to_qk", dim, dim), ("to_v", dim, dim),
                                           ("to_out", dim, dim)))):
            base = f"transformers.{i}.{blk}"
            for n, o, ii in names:
                lin(f"{base}.{n}", o, ii)
            lin(f"{base}.ffn.0", 2 * dim, 2 * dim)
            p[f"{base}.ffn.1.weight"] = (1 + 0.1 * torch.randn(2 * dim, generator=g,
                                                               dtype=torch.float64)).to(dtype)
            p[f"{base}.ffn.1.bias"] = (0.1 * torch.randn(2 * dim, generator=g,
                                                         dtype=torch.float64)).to(dtype)
            lin(f"{base}.ffn.3", dim, 2 * dim)
        lin(f"log_assignment.{i}.matchability", 1, dim)
        lin(f"log_assignment.{i}.final_proj", dim, dim)
        if i < n_layers - 1:
            lin(f"token_confidence.{i}.token.0", 1, dim)
    p["confidence_thresholds"] = torch.tensor(
        [min(max(0.8 + 0.1 * math.exp(-4.0 * i / n_layers), 0.0), 1.0) for i in range(n_layers)],
        dtype=dtype)
    return p


def trainable_names(params):
    return [k for k in params if k != "confidence_thresholds"]


# --------------------------------------------------------------------------- pieces
def normalize_keypoints(kpts, size=None):
    """(k - size/2) / (max(size)/2); size defaults to the keypoints' extent + 1."""
    if size is None:
        size = 1 + kpts.max(-2).values - kpts.min(-2).values
    size = torch.as_tensor(size).to(kpts)
    if size.dim() == 1:
        size = size[None].expand(kpts.shape[0], -1)
    centre = size / 2
    half_long_side = size.max(-1).values / 2
    return (kpts - centre[:, None, :]) / half_long_side[:, None, None]


def fourier_encoding(wr, kpts):
    """Returns (cos, sin), each [B, N, hd] with every frequency duplicated pairwise."""
    proj = kpts @ wr.t()  # [B,N,hd/2]
    cos = torch.cos(proj).repeat_interleave(2, dim=-1)
    sin = torch.sin(proj).repeat_interleave(2, dim=-1)
    return cos, sin


def rotary(x, cos, sin):
    """x: [B,H,N,hd]; rotate each consecutive pair (x0,x1) by the pair's angle."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-xo, xe), dim=-1).flatten(-2)
    return x * cos[:, None] + rot * sin[:, None]


def softmax_attention(q, k, v):
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    return torch.softmax(s, -1) @ v


def _ffn(p, base, x, msg):
    h = F.linear(torch.cat([x, msg], -1), p[f"{base}.ffn.0.weight"], p[f"{base}.ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), p[f"{base}.ffn.1.weight"], p[f"{base}.ffn.1.bias"])
    h = F.gelu(h)
    return x + F.linear(h, p[f"{base}.ffn.3.weight"], p[f"{base}.ffn.3.bias"])


def self_block(p, base, x, enc, heads):
    b, n, d = x.shape
    hd = d // heads
    qkv = F.linear(x, p[f"{base}.Wqkv.weight"], p[f"{base}.Wqkv.bias"])
    # channel index = (head * hd + c) * 3 + {q,k,v}
    qkv = qkv.reshape(b, n, heads, hd, 3).permute(4, 0, 2, 1, 3)  # [3,B,H,N,hd]
    q, k, v = rotary(qkv[0], *enc), rotary(qkv[1], *enc), qkv[2]
    ctx = softmax_attention(q, k, v).transpose(1, 2).reshape(b, n, d)
    msg = F.linear(ctx, p[f"{base}.out_proj.weight"], p[f"{base}.out_proj.bias"])
    return _ffn(p, base, x, msg)


def cross_block(p, base, x0, x1, heads):
    def split(t):
        b, n, d = t.shape
        return t.reshape(b, n, heads, d // heads).transpose(1, 2)

    def merge(t):
        b, h, n, hd = t.shape
        return t.transpose(1, 2).reshape(b, n, h * hd)

    qk0, qk1 = (split(F.linear(x, p[f"{base}.to_qk.weight"], p[f"{base}.to_qk.bias"]))
                for x in (x0, x1))
    v0, v1 = (split(F.linear(x, p[f"{base}.to_v.weight"], p[f"{base}.to_v.bias"]))
              for x in (x0, x1))
    s = qk0.shape[-1] ** -0.25
    sim = (qk0 * s) @ (qk1 * s).transpose(-1, -2)  # shared similarity [B,H,M,N]
    m0 = torch.softmax(sim, -1) @ v1
    m1 = torch.softmax(sim.transpose(-1, -2), -1) @ v0
    m0, m1 = (F.linear(merge(m), p[f"{base}.to_out.weight"], p[f"{base}.to_out.bias"])
              for m in (m0, m1))
    return _ffn(p, base, x0, m0), _ffn(p, base, x1, m1)


def log_double_softmax(sim, z0, z1):
    """sim [B,M,N], z0 [B,M], z1 [B,N] -> log assignment [B,M+1,N+1] with dustbins."""
    b, m, n = sim.shape
    row = sim.logsumexp(2, keepdim=True)
    col = sim.logsumexp(1, keepdim=True)
    out = sim.new_zeros(b, m + 1, n + 1)
    out[:, :m, :n] = (2 * sim - row - col + F.logsigmoid(z0)[:, :, None]
                      + F.logsigmoid(z1)[:, None, :])
    out[:, :m, n] = F.logsigmoid(-z0)
    out[:, m, :n] = F.logsigmoid(-z1)
    return out


def match_assignment(p, i, d0, d1):
    base = f"log_assignment.{i}"
    dim = d0.shape[-1]
    md0, md1 = (F.linear(d, p[f"{base}.final_proj.weight"], p[f"{base}.final_proj.bias"])
                / dim ** 0.25 for d in (d0, d1))
    sim = md0 @ md1.transpose(-1, -2)
    z0, z1 = (F.linear(d, p[f"{base}.matchability.weight"],
                       p[f"{base}.matchability.bias"]).squeeze(-1) for d in (d0, d1))
    return log_double_softmax(sim, z0, z1), sim


def filter_matches(scores, th):
    """Mutual nearest neighbours on the core of the log-assignment."""
    core = scores[:, :-1, :-1]
    max0, m0 = core.max(2)
    m1 = core.max(1).indices
    ar0 = torch.arange(m0.shape[1], device=m0.device)[None]
    ar1 = torch.arange(m1.shape[1], device=m1.device)[None]
    mutual0 = m1.gather(1, m0) == ar0
    mutual1 = m0.gather(1, m1) == ar1
    s0 = torch.where(mutual0, max0.exp(), max0.new_zeros(()))
    s1 = torch.where(mutual1, s0.gather(1, m1), s0.new_zeros(()))
    valid0 = mutual0 & (s0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    return (torch.where(valid0, m0, -1), torch.where(valid1, m1, -1), s0, s1)


# --------------------------------------------------------------------------- model
def forward(p, data, n_layers=9, heads=4, filter_threshold=0.0, training=True):
    """LightGlue forward without early-stop / pruning (training path and plain eval)."""
    k0 = normalize_keypoints(data["keypoints0"], data.get("image_size0"))
    k1 = normalize_keypoints(data["keypoints1"], data.get("image_size1"))
    if p["posenc.Wr.weight"].shape[1] == 4:        # add_scale_ori (lightglue.py:426-443): (x, y, scale, orientation)
        cat = lambda k, sc, o: torch.cat([k, sc if sc.dim() == 3 else sc[..., None], o if o.dim() == 3 else o[..., None]], -1)   # noqa: E731
        k0 = cat(k0, data["scales0"], data["oris0"])
        k1 = cat(k1, data["scales1"], data["oris1"])
    d0, d1 = data["descriptors0"], data["descriptors1"]
    if "input_proj.weight" in p:
        d0, d1 = (F.linear(d, p["input_proj.weight"], p["input_proj.bias"]) for d in (d0, d1))
    e0 = fourier_encoding(p["posenc.Wr.weight"], k0)
    e1 = fourier_encoding(p["posenc.Wr.weight"], k1)
    all0, all1 = [], []
    for i in range(n_layers):
        base = f"transformers.{i}"
        d0 = self_block(p, base + ".self_attn", d0, e0, heads)
        d1 = self_block(p, base + ".self_attn", d1, e1, heads)
        d0, d1 = cross_block(p, base + ".cross_attn", d0, d1, heads)
        if training or i == n_layers - 1:
            all0.append(d0)
            all1.append(d1)
    scores, _ = match_assignment(p, n_layers - 1, d0, d1)
    m0, m1, s0, s1 = filter_matches(scores, filter_threshold)
    return {
        "matches0": m0, "matches1": m1, "matching_scores0": s0, "matching_scores1": s1,
        "ref_descriptors0": torch.stack(all0, 1), "ref_descriptors1": torch.stack(all1, 1),
        "log_assignment": scores,
        "prune0": torch.ones_like(s0) * n_layers, "prune1": torch.ones_like(s1) * n_layers,
    }


def nll(la, gt_assignment, gt_m0, gt_m1, balancing=0.5):
    """Sparse statement of the dense-weight NLL: positives + the two dustbin vectors."""
    pos = gt_assignment.to(la.dtype)
    neg0 = (gt_m0 == -1).to(la.dtype)
    neg1 = (gt_m1 == -1).to(la.dtype)
    num_pos = pos.sum((1, 2)).clamp(min=1.0)
    n0 = neg0.sum(1).clamp(min=1.0)
    n1 = neg1.sum(1).clamp(min=1.0)
    nll_pos = -(la[:, :-1, :-1] * pos).sum((1, 2)) / num_pos
    nll_neg = -((la[:, :-1, -1] * neg0).sum(1) + (la[:, -1, :-1] * neg1).sum(1)) / (n0 + n1)
    total = balancing * nll_pos + (1 - balancing) * nll_neg
    return total, {"assignment_nll": total, "nll_pos": nll_pos, "nll_neg": nll_neg,
                   "num_matchable": num_pos, "num_unmatchable": (n0 + n1) / 2.0}


def token_confidence_loss(p, i, d0, d1, la_now, la_final):
    base = f"token_confidence.{i}.token.0"
    l0, l1 = (F.linear(d.detach(), p[base + ".weight"], p[base + ".bias"]).squeeze(-1)
              for d in (d0, d1))
    la_now, la_final = la_now.detach(), la_final.detach()
    same0 = la_final[:, :-1, :].argmax(-1) == la_now[:, :-1, :].argmax(-1)
    same1 = la_final[:, :, :-1].argmax(-2) == la_now[:, :, :-1].argmax(-2)
    bce = F.binary_cross_entropy_with_logits
    return (bce(l0, same0.to(l0.dtype), reduction="none").mean(-1)
            + bce(l1, same1.to(l1.dtype), reduction="none").mean(-1)) / 2.0


def loss(p, pred, data, gamma=1.0, balancing=0.5, training=True):
    L = pred["ref_descriptors0"].shape[1]
    gt = (data["gt_assignment"], data["gt_matches0"], data["gt_matches1"])

    def head(i):
        return match_assignment(p, i, pred["ref_descriptors0"][:, i],
                                pred["ref_descriptors1"][:, i])[0]

    last, stats = nll(head(L - 1), *gt, balancing)
    out = {"total": last, "last": last.detach().clone(), **stats}
    out["row_norm"] = pred["log_assignment"].exp()[:, :-1].sum(2).mean(1)
    conf = torch.zeros_like(last)
    wsum = 1.0
    for i in range(L - 1):
        la = head(i)
        w = gamma ** (L - i - 1) if gamma > 0 else i + 1
        wsum += w
        out["total"] = out["total"] + nll(la, *gt, balancing)[0] * w
        conf = conf + token_confidence_loss(
            p, i, pred["ref_descriptors0"][:, i], pred["ref_descriptors1"][:, i],
            la, pred["log_assignment"]) / (L - 1)
    out["total"] = out["total"] / wsum
    if training:
        out["confidence"] = conf
        out["total"] = out["total"] + conf
    return out


def train_step_grads(p, data, n_layers, heads=4, filter_threshold=0.0):
    """One forward + loss + backward; returns (pred, losses, grads dict)."""
    names = trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    pred = forward(q, data, n_layers, heads, filter_threshold, training=True)
    losses = loss(q, pred, data)
    losses["total"].mean().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return pred, losses, grads


# --------------------------------------------------------------------------- sharpened case (bit-exact matches)
def sharp_case(batch, n, n_layers, seed, size=(1024, 1024), damp=0.1, sharp=6.0, noise=0.1):
    """Seeded weights + a seeded pair batch on which EVERY row and column of the final log-assignment has a decisive
    arg-max (so matches0/1 can be compared bit for bit on 100 % of the rows, lightglue.py:293-309): image 1's keypoints
    are a permutation of image 0's (warped by a similarity) with descriptors normalise(d0 + noise N(0, I)); the blocks'
    output linears (out_proj / to_out / ffn.3) are damped so the residual stream stays close to the input
    descriptors, and every final_proj is `sharp * I + init`.  Returns (params, data) of CPU fp32 tensors."""
    p = init_params(n_layers, 256, 4, seed=seed)
    for k in p:
        if k.endswith(("out_proj.weight", "out_proj.bias", "to_out.weight", "to_out.bias", "ffn.3.weight", "ffn.3.bias")):
            p[k] = p[k] * damp
        if k.endswith("final_proj.weight"):
            p[k] = p[k] + sharp * torch.eye(p[k].shape[0])
    g = torch.Generator().manual_seed(seed + 1)
    w, h = size
    wh = torch.tensor([w, h], dtype=torch.float32)
    kp0 = torch.rand(batch, n, 2, generator=g) * wh
    a = math.radians(10.0)
    c, s = math.cos(a) * 1.1, math.sin(a) * 1.1
    ctr = wh / 2
    rot = torch.tensor([[c, -s], [s, c]])
    d0 = F.normalize(torch.randn(batch, n, 256, generator=g), dim=-1)
    perm = torch.stack([torch.randperm(n, generator=g) for _ in range(batch)])          # kp1[j] = image of kp0[perm[j]]
    kp1 = ((kp0 - ctr) @ rot.T + ctr + torch.tensor([15.0, -10.0])).gather(1, perm[..., None].expand(-1, -1, 2))
    d1 = F.normalize(d0 + noise * torch.randn(batch, n, 256, generator=g), dim=-1).gather(1, perm[..., None].expand(-1, -1, 256))
    inv = torch.argsort(perm, 1)                                                        # gt_matches0[i] = j with perm[j] = i
    gt = torch.zeros(batch, n, n, dtype=torch.bool)
    gt.scatter_(2, inv[..., None], True)
    data = {"keypoints0": kp0, "keypoints1": kp1, "descriptors0": d0, "descriptors1": d1,
            "view0": {"image_size": wh[None].repeat(batch, 1)}, "view1": {"image_size": wh[None].repeat(batch, 1)},
            "gt_assignment": gt, "gt_assignment_col0": inv.clone(), "gt_matches0": inv, "gt_matches1": perm}
    return p, data


def decision_margins(la):
    """(min over rows, min over columns) of top-1 minus top-2 of the core block of a log-assignment [B,M+1,N+1]."""
    core = la[:, :-1, :-1]
    r = core.topk(2, dim=2).values
    c = core.topk(2, dim=1).values
    return float((r[..., 0] - r[..., 1]).min()), float((c[:, 0] - c[:, 1]).min())
