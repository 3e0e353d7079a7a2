"""LightGlue matcher on the MI355X hot path (drop-in for gluefactory.models.matchers.lightglue).

Same plugin surface as the reference module (gluefactory/models/matchers/lightglue.py:312-630):
``LightGlue(conf)``, ``forward(data) -> pred`` and ``loss(pred, data) -> (losses, metrics)``,
the same ``default_conf`` keys, ``required_data_keys`` and ``state_dict`` names/shapes
(SURVEY.md §8 note S), exported as ``__main_model__`` so that
``model.matcher.name: glue_factory_amd.matchers.lightglue`` in a glue-factory yaml is the whole
integration.

What runs where
  * linear layers: hipBLASLt through torch (plain library GEMMs);
  * rotary + self attention, bidirectional cross attention (forward and backward):
    hand-written MFMA flash kernels (csrc/attention.hip), fed by the fused projections in
    place — no [B,H,N,N] tensor exists;
  * LayerNorm+GELU of the FFN: one fused HIP kernel each way (csrc/elementwise.hip);
  * assignment heads (lightglue.py:256-309): row/column log-sum-exp, arg-max and the
    materialised log-assignment from MFMA tiles of md0 md1^T (csrc/assignment.hip).  The L
    training heads never build the (N+1)^2 matrix: the NLL (utils/losses.py:6-73) is evaluated
    on the ground-truth positives and the two dustbin vectors, which is exactly the dense
    weighted sum because the weight matrix is zero elsewhere.
Compute dtype: fp32 (exact-fp32 MFMA, parity mode) by default, bf16 with fp32 statistics when
``conf.mp`` is set or the call happens under ``torch.autocast``.  There is no CPU path.
"""
import math
from pathlib import Path

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..conf import Conf
from ..metrics import matcher_metrics


def normalize_keypoints(kpts, size=None):
    """(k - size/2) / (max(size)/2), fp32 (lightglue.py:27-39)."""
    kpts = kpts.float()
    if size is None:
        size = 1 + kpts.max(-2).values - kpts.min(-2).values
    elif not isinstance(size, torch.Tensor):
        size = torch.tensor(size, device=kpts.device, dtype=kpts.dtype)
    size = size.to(kpts)
    shift = size / 2
    scale = size.max(-1).values / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


class _PosEnc(nn.Module):
    """Holds posenc.Wr (lightglue.py:52-65); returns the pair angles and interleaved (cos, sin)."""

    def __init__(self, in_dim, head_dim):
        super().__init__()
        self.Wr = nn.Linear(in_dim, head_dim // 2, bias=False)
        nn.init.normal_(self.Wr.weight.data, mean=0.0, std=1.0)

    def forward(self, kpts):
        w = self.Wr.weight.float()
        theta = ops.small_linear(kpts, w) if kpts.is_cuda else F.linear(kpts, w)     # [B,N,hd/2] fp32
        with torch.no_grad():
            cs = torch.stack((torch.cos(theta), torch.sin(theta)), -1).flatten(-2).contiguous()
        return theta, cs


def _ffn_modules(dim):
    return nn.Sequential(nn.Linear(2 * dim, 2 * dim), nn.LayerNorm(2 * dim, elementwise_affine=True),
                         nn.GELU(), nn.Linear(2 * dim, dim))


def _lin(x, layer):
    if layer.out_features == 1 and layer.in_features % 8 == 0:   # 256 -> 1 heads: streaming row-dot kernels
        return ops.rowdot(x, layer.weight, layer.bias)[..., None]
    if layer.out_features % 8 or layer.in_features % 8:
        return F.linear(x, layer.weight.to(x.dtype), None if layer.bias is None else layer.bias.to(x.dtype))
    return ops.linear(x, layer.weight, layer.bias)


def _ffn(ffn, x, msg, chain=None, out=None, first=None):
    """x + ffn(cat(x, msg)).  ``chain``: the GradChain of x (its three consumers in a block are this residual, the
    FFN input and the block's projection): the residual and FFN-input gradients ride in GEMM epilogues.
    ``first``: (weight, bias) of the first linear when the block's output projection was folded into it (``msg`` is then
    the attention context itself, see _folded_ffn0)."""
    w0, b0 = (ffn[0].weight, ffn[0].bias) if first is None else first
    h = ops.linear_cat(x, msg, w0, b0, chain1=chain)
    h = ops.ln_gelu(h, ffn[1].weight, ffn[1].bias, ffn[1].eps)
    return ops.linear(h, ffn[3].weight, ffn[3].bias, res=x, res_chain=chain, out=out)   # residual fused into the GEMM epilogue


def _fold_spec(name, ffn, proj):
    """out_proj / to_out feed ONLY ffn.0, through a concatenation and with no non-linearity in between (lightglue.py:131-163,
    166-221): ffn.0(cat[x, Wo c + bo]) = [W0a | W0b Wo] cat[x, c] + (b0 + W0b bo).  The folded weight is prepared by the
    per-step precast launch (ops.precast / csrc/fold.hip): one forward GEMM, one input-gradient GEMM and one weight-gradient
    reduction over all tokens fewer per block, and no message tensor."""
    return (name + ".ffn0", "fold", ffn[0].weight, ffn[0].bias, proj.weight, proj.bias, ffn[0].in_features - proj.out_features, None)


def _folded_ffn0(pc, dtype, ffn, proj):
    """(weight, bias) of ffn.0 with `proj` folded in, from this forward's precast launch; None in the fp32 parity mode (the
    reference's two linears run as written)."""
    if pc is None:
        return None
    return ops.folded_linear(pc[0], dtype, pc[1] + ".ffn0", ffn[0].weight, ffn[0].bias, proj.weight, proj.bias)


class SelfBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads, self.head_dim = heads, dim // heads
        self.Wqkv = nn.Linear(dim, 3 * dim, bias=True)
        self.out_proj = nn.Linear(dim, dim, bias=True)
        self.ffn = _ffn_modules(dim)
        # reference channel order of Wqkv's output is (head, channel, {q,k,v}); the kernels want
        # ({q,k,v}, head, channel): gather the weight rows instead of shuffling activations.
        h, c = heads, self.head_dim
        perm = (torch.arange(h)[None, :, None] * (3 * c) + torch.arange(c)[None, None, :] * 3
                + torch.arange(3)[:, None, None]).reshape(-1)
        self.register_buffer("_perm", perm, persistent=False)
        # head_dim^-1/2 * log2(e) is folded into the q rows of the projection (fp32, before the GEMM's single rounding): the
        # attention kernels then read their scores as exp2 arguments (ops.attn_premul)
        rs = torch.ones(3 * dim)
        rs[:dim] = ops.attn_premul(self.head_dim)
        self.register_buffer("_rowscale", rs, persistent=False)

    _pc = None      # (precast key, name): set by the owning model, see LightGlue._derived_specs

    def derived_specs(self, name):
        """Row gather + q-row scale of the projection as entries of the model's per-step precast launch (ops.precast)."""
        return [(name + ".w", [(self.Wqkv.weight, self._perm, self._rowscale, 1.0)]),
                (name + ".b", [(self.Wqkv.bias, self._perm, self._rowscale, 1.0)]),
                _fold_spec(name, self.ffn, self.out_proj)]

    def _prepared(self, dtype):
        if self._pc is not None:
            w = ops.derived_weight(self._pc[0], dtype, self._pc[1] + ".w", self.Wqkv.weight)
            if w is not None:
                return w, ops.derived_weight(self._pc[0], dtype, self._pc[1] + ".b", self.Wqkv.bias)
        return (self.Wqkv.weight.index_select(0, self._perm) * self._rowscale[:, None],
                self.Wqkv.bias.index_select(0, self._perm) * self._rowscale)

    def forward(self, x, theta, cs, chain=None, theta_sum=None):
        b, n, d = x.shape
        w, bias = self._prepared(x.dtype)
        # x feeds the projection, the FFN input and the residual: one gradient chain, closed by the projection (a chain
        # handed in by the caller may already carry optional contributions: the loss heads of the previous layer's output)
        if chain is None:
            chain = ops.GradChain(3) if x.requires_grad and torch.is_grad_enabled() else None
        if self.head_dim == 64 and ops.gemm_takes(d, 3 * d, x.dtype):     # q, k leave the GEMM already rotated (rotary epilogue: 64-wide heads)
            qkv = ops.linear(x, w, bias, rotary_cs=cs, rot_n=2 * d, chain=chain, chain_last=True)
            qkv = qkv.view(b, n, 3, self.heads, self.head_dim)
            ctx = ops.self_attention_rotary(qkv, theta, cs, pre_rotated=True, scale=ops.LN2, theta_sum=theta_sum)      # [b,n,H,hd]
        else:
            qkv = ops.linear(x, w, bias, chain=chain, chain_last=True).view(b, n, 3, self.heads, self.head_dim)
            ctx = ops.self_attention_rotary(qkv, theta, cs, scale=ops.LN2, theta_sum=theta_sum)
        first = _folded_ffn0(self._pc, x.dtype, self.ffn, self.out_proj)
        if first is not None:          # out_proj lives inside ffn.0's weight: the FFN reads the attention context directly
            return _ffn(self.ffn, x, ctx.view(b, n, d), chain, first=first)
        msg = _lin(ctx.view(b, n, d), self.out_proj)
        return _ffn(self.ffn, x, msg, chain)


class CrossBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads, self.head_dim = heads, dim // heads
        self.to_qk = nn.Linear(dim, dim, bias=True)
        self.to_v = nn.Linear(dim, dim, bias=True)
        self.to_out = nn.Linear(dim, dim, bias=True)
        self.ffn = _ffn_modules(dim)

    _pc = None

    def derived_specs(self, name):
        sq = ops.attn_premul(self.head_dim) ** 0.5
        return [(name + ".w", [(self.to_qk.weight, None, None, sq), (self.to_v.weight, None, None, 1.0)]),
                (name + ".b", [(self.to_qk.bias, None, None, sq), (self.to_v.bias, None, None, 1.0)]),
                _fold_spec(name, self.ffn, self.to_out)]

    def _proj(self, x, chain=None):
        # sqrt(head_dim^-1/2 * log2(e)) on BOTH images' qk (each is query in one direction and key in the other)
        w = None
        if self._pc is not None:
            w = ops.derived_weight(self._pc[0], x.dtype, self._pc[1] + ".w", self.to_qk.weight, self.to_v.weight)
        if w is not None:
            bias = ops.derived_weight(self._pc[0], x.dtype, self._pc[1] + ".b", self.to_qk.bias, self.to_v.bias)
        else:
            sq = ops.attn_premul(self.head_dim) ** 0.5
            w = torch.cat([self.to_qk.weight * sq, self.to_v.weight], 0)
            bias = torch.cat([self.to_qk.bias * sq, self.to_v.bias], 0)
        return ops.linear(x, w, bias, chain=chain, chain_last=True).view(x.shape[0], x.shape[1], 2, self.heads, self.head_dim)

    def forward_stacked(self, x, out=None):
        """x [2B,N,C]: image 0 in the first half of the batch, image 1 in the second."""
        b2, n, d = x.shape
        chain = ops.GradChain(3) if x.requires_grad and torch.is_grad_enabled() else None
        m = ops.cross_attention_stacked(self._proj(x, chain), scale=ops.LN2)
        first = _folded_ffn0(self._pc, x.dtype, self.ffn, self.to_out)
        if first is not None:          # to_out lives inside ffn.0's weight
            return _ffn(self.ffn, x, m.view(b2, n, d), chain, out, first=first)
        return _ffn(self.ffn, x, _lin(m.view(b2, n, d), self.to_out), chain, out)

    def forward(self, x0, x1):
        m0, m1 = ops.cross_attention(self._proj(x0), self._proj(x1), scale=ops.LN2)
        first = _folded_ffn0(self._pc, x0.dtype, self.ffn, self.to_out)
        if first is not None:
            return (_ffn(self.ffn, x0, m0.reshape(x0.shape), first=first), _ffn(self.ffn, x1, m1.reshape(x1.shape), first=first))
        m0 = _lin(m0.view(x0.shape), self.to_out)
        m1 = _lin(m1.view(x1.shape), self.to_out)
        return _ffn(self.ffn, x0, m0), _ffn(self.ffn, x1, m1)


class TransformerLayer(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.self_attn = SelfBlock(dim, heads)
        self.cross_attn = CrossBlock(dim, heads)


class MatchAssignment(nn.Module):
    """final_proj + matchability and the double-softmax statistics (lightglue.py:271-290)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.matchability = nn.Linear(dim, 1, bias=True)
        self.final_proj = nn.Linear(dim, dim, bias=True)

    _pc = None      # (precast key, name): set by the owning model, see LightGlue._derived_specs

    def derived_specs(self, name):
        s = self.dim ** -0.25
        return [(name + ".w", [(self.final_proj.weight, None, None, s)]), (name + ".b", [(self.final_proj.bias, None, None, s)])]

    def scaled_proj(self, dtype):
        """final_proj with dim^-1/4 folded in (lightglue.py:271-274): from this forward's precast launch, else by torch ops."""
        if self._pc is not None:
            w = ops.derived_weight(self._pc[0], dtype, self._pc[1] + ".w", self.final_proj.weight)
            if w is not None:
                return w, ops.derived_weight(self._pc[0], dtype, self._pc[1] + ".b", self.final_proj.bias)
        s = self.dim ** -0.25
        return self.final_proj.weight * s, self.final_proj.bias * s

    def stats(self, d0, d1):
        """Everything the log assignment is made of, without the matrix:
        A_ij = 2 md0_i.md1_j - r_i - c_j + lz0_i + lz1_j,  A_i,n = bin0_i,  A_m,j = bin1_j."""
        w, bias = self.scaled_proj(d0.dtype)
        md0, md1 = ops.linear(d0, w, bias), ops.linear(d1, w, bias)
        z0 = _lin(d0, self.matchability).squeeze(-1).float()
        z1 = _lin(d1, self.matchability).squeeze(-1).float()
        r, c = ops.dual_lse(md0, md1)
        return {"md0": md0, "md1": md1, "r": r, "c": c,
                "lz0": F.logsigmoid(z0), "lz1": F.logsigmoid(z1),
                "bin0": F.logsigmoid(-z0), "bin1": F.logsigmoid(-z1)}

    def stats_stacked(self, x, b, chain=None, closing=False):
        """Same as ``stats`` on the batch-stacked descriptors x [2B,N,D] (image 0 first): one GEMM for both
        images and gradients that stay stacked (no slice / zero-fill / add nodes in the autograd graph).
        ``chain``: GradChain of x.  closing=True (last layer: these two heads are x's only consumers): the matchability
        head parks its rank-1 gradient, the projection's input-gradient GEMM adds it in its epilogue and returns the
        total.  closing=False: both park, uncounted, into the chain of the NEXT block (see LightGlue._loss_fused)."""
        w, bias = self.scaled_proj(x.dtype)
        md = ops.linear(x, w, bias, chain=chain, chain_last=True if closing else "extra")
        z = ops.rowdot(x, self.matchability.weight, self.matchability.bias, chain=chain, counted=closing).float()
        r, c = ops.dual_lse_stacked(md)
        lz, lnz = F.logsigmoid(z), F.logsigmoid(-z)
        return {"md": md, "z": z, "md0": md[:b], "md1": md[b:], "r": r, "c": c,
                "lz0": lz[:b], "lz1": lz[b:], "bin0": lnz[:b], "bin1": lnz[b:]}

    @staticmethod
    def materialize(h, with_expsum=False):
        return ops.assign_write(h["md0"], h["md1"], h["lz0"] - h["r"], h["lz1"] - h["c"],
                                h["bin0"], h["bin1"], alpha=2.0, corner=0.0, with_expsum=with_expsum)

    @staticmethod
    @torch.no_grad()
    def argmaxes(h):
        """Row / column maxima of the core (value in the full log-assignment, index), plus the
        arg-max over the full row / column INCLUDING its dustbin entry."""
        n, m = h["md1"].shape[1], h["md0"].shape[1]
        v0, a0 = ops.rows_argmax(h["md0"], h["md1"], h["lz1"] - h["c"], alpha=2.0)
        v1, a1 = ops.rows_argmax(h["md1"], h["md0"], h["lz0"] - h["r"], alpha=2.0)
        max0 = v0 - h["r"] + h["lz0"]
        max1 = v1 - h["c"] + h["lz1"]
        full0 = torch.where(h["bin0"] > max0, torch.full_like(a0, n), a0)
        full1 = torch.where(h["bin1"] > max1, torch.full_like(a1, m), a1)
        return {"max0": max0, "arg0": a0, "arg1": a1, "full0": full0, "full1": full1}

    def get_matchability(self, desc):
        return torch.sigmoid(_lin(desc, self.matchability)).squeeze(-1)


class TokenConfidence(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.token = nn.Sequential(nn.Linear(dim, 1), nn.Sigmoid())

    def logits(self, desc):
        return _lin(desc.detach(), self.token[0]).squeeze(-1).float()

    def forward(self, desc0, desc1):
        return torch.sigmoid(self.logits(desc0)), torch.sigmoid(self.logits(desc1))


class _RefDescriptors(torch.autograd.Function):
    """The public ``ref_descriptors0/1`` [B, L, N, C] of a training forward as VIEWS of the per-layer buffer the layers'
    last GEMMs wrote (no stack of L copies), differentiable w.r.t. the batch-stacked layer outputs: a loss built from
    them by somebody else (a pred dict that went through the reference's stack / unstack helpers,
    gluefactory/utils/misc.py:31-46, or a filtered copy) trains the model exactly like the fused loss does.  The
    backward only runs in that case."""

    @staticmethod
    def forward(ctx, holder, b, *layer_x):
        ctx.b = b
        d = holder[0].detach()
        return d[:, :b].transpose(0, 1), d[:, b:].transpose(0, 1)

    @staticmethod
    def backward(ctx, g0, g1):
        n_layers = (g0 if g0 is not None else g1).shape[1]
        out = []
        for i in range(n_layers):
            a = g0[:, i] if g0 is not None else torch.zeros_like(g1[:, i])
            c = g1[:, i] if g1 is not None else torch.zeros_like(g0[:, i])
            out.append(torch.cat([a, c], 0))
        return (None, None, *out)


_PRIVATE = "_gf_private"       # attribute of the returned log_assignment TENSOR that carries the fused loss's state


class LightGlue(nn.Module):
    default_conf = {
        "name": "lightglue",
        "input_dim": 256,
        "add_scale_ori": False,
        "descriptor_dim": 256,
        "n_layers": 9,
        "num_heads": 4,
        "flash": False,            # accepted for yaml compatibility: attention is always flash-style
        "mp": False,               # bf16 compute with fp32 statistics
        "depth_confidence": -1,
        "width_confidence": -1,
        "filter_threshold": 0.0,
        "checkpointed": False,     # accepted; activations are kept (288 GB HBM), never recomputed
        "weights": None,
        "weights_from_version": "v0.1_arxiv",
        "loss": {"gamma": 1.0, "fn": "nll", "nll_balancing": 0.5},
    }
    required_data_keys = ["keypoints0", "keypoints1", "descriptors0", "descriptors1"]

    def __init__(self, conf=None):
        super().__init__()
        self.conf = conf = Conf.merge(self.default_conf, conf or {})
        d, h, n = conf.descriptor_dim, conf.num_heads, conf.n_layers
        if d % h or d // h not in (32, 64, 128):
            raise NotImplementedError("the HIP attention kernels exist for head_dim 64 (tuned) and 32 / 128 (generic kernels)")
        self.input_proj = (nn.Linear(conf.input_dim, d, bias=True) if conf.input_dim != d
                           else nn.Identity())
        self.posenc = _PosEnc(2 + 2 * conf.add_scale_ori, d // h)
        self.transformers = nn.ModuleList([TransformerLayer(d, h) for _ in range(n)])
        self.log_assignment = nn.ModuleList([MatchAssignment(d) for _ in range(n)])
        self.token_confidence = nn.ModuleList([TokenConfidence(d) for _ in range(n - 1)])
        self.register_buffer("confidence_thresholds", torch.Tensor(
            [min(max(0.8 + 0.1 * math.exp(-4.0 * i / n), 0.0), 1.0) for i in range(n)]))
        if conf.weights is not None:
            self._load_weights(conf.weights)

    def _load_weights(self, weights):
        path = Path(weights)
        if not path.exists():
            raise FileNotFoundError(f"weights file {weights} not found (no network on this target)")
        sd = torch.load(str(path), map_location="cpu")
        for i in range(self.conf.n_layers):  # legacy key layout (lightglue.py:384-391)
            sd = {k.replace(f"self_attn.{i}", f"transformers.{i}.self_attn")
                  if not k.startswith("transformers.") else k: v for k, v in sd.items()}
            sd = {k.replace(f"cross_attn.{i}", f"transformers.{i}.cross_attn")
                  if not k.startswith("transformers.") else k: v for k, v in sd.items()}
        self.load_state_dict(sd, strict=False)

    # ------------------------------------------------------------------ forward
    def _derived_specs(self):
        specs = []
        for i, layer in enumerate(self.transformers):
            for blk, nm in ((layer.self_attn, f"self{i}"), (layer.cross_attn, f"cross{i}")):
                blk._pc = (id(self), nm)
                specs += blk.derived_specs(nm)
        for i, la in enumerate(self.log_assignment):
            la._pc = (id(self), f"head{i}")
            specs += la.derived_specs(f"head{i}")
        return specs

    def _compute_dtype(self):
        if self.conf.mp or torch.is_autocast_enabled():
            return torch.bfloat16
        return torch.float32

    # torch.compile(model) (gluefactory/train.py:332-333): the HIP path is opaque to dynamo -- ctypes launches inside
    # autograd.Functions, host-side caches -- so forward / loss are one clean graph break and run eagerly
    @torch.compiler.disable
    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        if not data["keypoints0"].is_cuda:
            raise RuntimeError("glue_factory_amd.LightGlue runs on the MI355X HIP path only "
                               "(move the batch to the GPU; there is no CPU fallback)")
        T = self._compute_dtype()  # read the autocast state before switching it off
        if T != torch.float32:
            # one launch per step: every parameter in the compute dtype (+ transposed copies) AND the blocks' prepared
            # projections (row order / softmax scale folded in)
            ops.precast(list(self.parameters()), T, key=id(self), derived=self._derived_specs())
        with torch.autocast(device_type="cuda", enabled=False):
            return self._forward(data, T)

    def _forward(self, data, T):
        conf = self.conf
        kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
        b, m, _ = kpts0.shape
        n = kpts1.shape[1]
        size0 = size1 = None
        if "view0" in data and "view1" in data:
            size0 = data["view0"].get("image_size")
            size1 = data["view1"].get("image_size")
        kpts0 = normalize_keypoints(kpts0, size0)
        kpts1 = normalize_keypoints(kpts1, size1)
        if conf.add_scale_ori:
            def cat_so(k, sc, o):
                return torch.cat([k, sc if sc.dim() == 3 else sc[..., None],
                                  o if o.dim() == 3 else o[..., None]], -1)
            kpts0 = cat_so(kpts0, data["scales0"].float(), data["oris0"].float())
            kpts1 = cat_so(kpts1, data["scales1"].float(), data["oris1"].float())
        desc0, desc1 = data["descriptors0"], data["descriptors1"]
        assert desc0.shape[-1] == conf.input_dim and desc1.shape[-1] == conf.input_dim
        x_in = None
        if (m == n and isinstance(self.input_proj, nn.Identity) and not (desc0.requires_grad or desc1.requires_grad)):
            # cast straight into the batch-stacked residual stream (no separate casts + concatenation)
            x_in = torch.empty((2 * b, m, conf.input_dim), dtype=T, device=desc0.device)
            x_in[:b].copy_(desc0)
            x_in[b:].copy_(desc1)
            desc0, desc1 = x_in[:b], x_in[b:]
        else:
            desc0, desc1 = desc0.to(T).contiguous(), desc1.to(T).contiguous()
            if not isinstance(self.input_proj, nn.Identity):
                desc0, desc1 = _lin(desc0, self.input_proj), _lin(desc1, self.input_proj)

        do_early_stop = conf.depth_confidence > 0 and not self.training
        do_point_pruning = conf.width_confidence > 0 and not self.training
        if do_early_stop or do_point_pruning:
            return self._forward_adaptive(kpts0, kpts1, desc0, desc1, do_early_stop, do_point_pruning)

        stacked = m == n
        all0, all1, layer_x = [], [], []
        if stacked:   # both images share every GEMM / kernel launch
            x = x_in if x_in is not None else torch.cat([desc0, desc1], 0)
            theta, cs = self.posenc(torch.cat([kpts0, kpts1], 0))
            # training: every layer's output is written straight into its slice of ONE [L, 2B, N, C] buffer by the
            # layer's last GEMM, so the public ref_descriptors are views of it (no stack of L copies)
            lbuf = torch.empty((conf.n_layers, 2 * b, m, x.shape[-1]), dtype=x.dtype, device=x.device) \
                if self.training and x.is_cuda else None
            # gradient chains: chains[i] collects every gradient of the tensor ENTERING layer i (= layer i-1's output):
            # the self block's three consumers and, optionally, that output's loss heads (added in _loss_fused)
            grad = torch.is_grad_enabled() and self.training and x.is_cuda
            chains = []
            # the L self blocks share the rotary angles: their gradient is summed inside the rotary-backward launches
            tsum = ops.SharedGradSum(conf.n_layers) if grad and theta.requires_grad else None
            for i, layer in enumerate(self.transformers):
                ch = ops.GradChain(3) if grad and x.requires_grad else None
                chains.append(ch)
                x = layer.self_attn(x, theta, cs, chain=ch, theta_sum=tsum)
                x = layer.cross_attn.forward_stacked(x, out=None if lbuf is None else lbuf[i])
                if self.training or i == conf.n_layers - 1:
                    layer_x.append(x)
                    if lbuf is None:
                        with torch.no_grad():      # public copies; gradients flow through the stacked list
                            all0.append(x[:b])
                            all1.append(x[b:])
            desc0, desc1 = x[:b], x[b:]
        else:
            th0, cs0 = self.posenc(kpts0)
            th1, cs1 = self.posenc(kpts1)
            for i, layer in enumerate(self.transformers):
                desc0 = layer.self_attn(desc0, th0, cs0)
                desc1 = layer.self_attn(desc1, th1, cs1)
                desc0, desc1 = layer.cross_attn(desc0, desc1)
                if self.training or i == conf.n_layers - 1:
                    all0.append(desc0)
                    all1.append(desc1)

        if stacked:
            fch = ops.GradChain(2) if grad and x.requires_grad else None      # the last output feeds only its two heads
            head = self.log_assignment[conf.n_layers - 1].stats_stacked(x, b, chain=fch, closing=True)
        else:
            head = self.log_assignment[conf.n_layers - 1].stats(desc0, desc1)
        # the `row_norm` statistic of the loss (lightglue.py:602) is accumulated while the matrix is written
        scores, expsum = MatchAssignment.materialize(head, with_expsum=True)
        am = MatchAssignment.argmaxes(head)
        m0, m1, ms0, ms1 = ops.filter_matches(am["max0"], am["arg0"], am["arg1"], conf.filter_threshold)
        if stacked and self.training:
            if lbuf is not None:
                if grad and layer_x[0].requires_grad:
                    rd0, rd1 = _RefDescriptors.apply([lbuf], b, *layer_x)           # [B, L, N, C] views, differentiable
                else:
                    rd0, rd1 = lbuf.detach()[:, :b].transpose(0, 1), lbuf.detach()[:, b:].transpose(0, 1)
            else:
                rd0, rd1 = torch.stack([x_[:b] for x_ in layer_x], 1), torch.stack([x_[b:] for x_ in layer_x], 1)
            # State of the fused loss (batch-stacked layer outputs, the last head's statistics, the gradient chains).  It is
            # NOT part of the pred dict -- the reference's pipelines slice / concatenate every entry of it along the batch
            # axis (triplet_pipeline.py:62-71), which only tensors survive -- but rides on the log_assignment tensor OBJECT:
            # loss() takes the fused path when it is handed that very tensor, and the reference's dense formulation on the
            # (differentiable) ref_descriptors otherwise.
            setattr(scores, _PRIVATE, {"layer_desc": layer_x, "final_head": head, "row_norm": expsum / (scores.shape[1] - 1),
                                       "layer_chain": chains[1:] + [None]})  # chain of layer_x[i] = the one entering layer i + 1
        else:
            rd0, rd1 = torch.stack(all0, 1), torch.stack(all1, 1)
        return {
            "matches0": m0, "matches1": m1,
            "matching_scores0": ms0, "matching_scores1": ms1,
            "ref_descriptors0": rd0, "ref_descriptors1": rd1,
            "log_assignment": scores,
            "prune0": torch.ones_like(ms0) * conf.n_layers,
            "prune1": torch.ones_like(ms1) * conf.n_layers,
            # private: final-layer arg-maxes incl. dustbins, reused by loss() for the confidence targets
            "_final_argmax0": am["full0"], "_final_argmax1": am["full1"],
        }

    # ------------------------------------------------------------------ eval-only adaptive depth / width
    def _forward_adaptive(self, kpts0, kpts1, desc0, desc1, do_early_stop, do_point_pruning):
        """Inference-time early stopping (token confidences) and point pruning (matchability), batch size 1
        (lightglue.py:461-529, 545-570).  Stock index_select compaction between layers; every layer still
        runs on the HIP kernels with the shrinking keypoint counts."""
        conf = self.conf
        b, m = kpts0.shape[:2]
        n = kpts1.shape[1]
        assert b == 1, "adaptive depth/width needs batch size 1 (as in the reference)"
        dev = kpts0.device
        th0, cs0 = self.posenc(kpts0)
        th1, cs1 = self.posenc(kpts1)
        ind0 = torch.arange(m, device=dev)[None]
        ind1 = torch.arange(n, device=dev)[None]
        prune0, prune1 = torch.ones_like(ind0), torch.ones_like(ind1)
        last = conf.n_layers - 1
        for i, layer in enumerate(self.transformers):
            desc0 = layer.self_attn(desc0, th0, cs0)
            desc1 = layer.self_attn(desc1, th1, cs1)
            desc0, desc1 = layer.cross_attn(desc0, desc1)
            last = i
            if i == conf.n_layers - 1:
                break
            token0 = token1 = None
            if do_early_stop:
                token0, token1 = self.token_confidence[i](desc0, desc1)
                confidences = torch.cat([token0, token1], -1)
                ratio = 1.0 - (confidences < self.confidence_thresholds[i]).float().sum() / (m + n)
                if ratio > conf.depth_confidence:
                    break
            if do_point_pruning:
                def prune(desc, th, cs, ind, pr, token):
                    keep = self.log_assignment[i].get_matchability(desc) > (1 - conf.width_confidence)
                    if token is not None:       # low-confidence points are never pruned
                        keep |= token <= self.confidence_thresholds[i]
                    k = torch.where(keep)[1]
                    ind = ind.index_select(1, k)
                    pr[:, ind[0]] += 1
                    return (desc.index_select(1, k).contiguous(), th.index_select(1, k).contiguous(),
                            cs.index_select(1, k).contiguous(), ind)
                desc0, th0, cs0, ind0 = prune(desc0, th0, cs0, ind0, prune0, token0)
                desc1, th1, cs1, ind1 = prune(desc1, th1, cs1, ind1, prune1, token1)
                if desc0.shape[1] == 0 or desc1.shape[1] == 0:
                    break
        if desc0.shape[1] == 0 or desc1.shape[1] == 0:
            scores = desc0.new_zeros((b, desc0.shape[1] + 1, desc1.shape[1] + 1), dtype=torch.float32)
            m0 = torch.full((b, desc0.shape[1]), -1, device=dev, dtype=torch.int64)
            m1 = torch.full((b, desc1.shape[1]), -1, device=dev, dtype=torch.int64)
            ms0, ms1 = scores.new_zeros((b, desc0.shape[1])), scores.new_zeros((b, desc1.shape[1]))
        else:
            head = self.log_assignment[last].stats(desc0, desc1)
            scores = MatchAssignment.materialize(head)
            am = MatchAssignment.argmaxes(head)
            m0, m1, ms0, ms1 = ops.filter_matches(am["max0"], am["arg0"], am["arg1"], conf.filter_threshold)
        if do_point_pruning:     # scatter back to the original keypoint indexing
            m0_ = torch.full((b, m), -1, device=dev, dtype=m0.dtype)
            m1_ = torch.full((b, n), -1, device=dev, dtype=m1.dtype)
            m0_[:, ind0[0]] = torch.where(m0 == -1, -1, ind1.gather(1, m0.clamp(min=0)))
            m1_[:, ind1[0]] = torch.where(m1 == -1, -1, ind0.gather(1, m1.clamp(min=0)))
            ms0_, ms1_ = ms0.new_zeros((b, m)), ms1.new_zeros((b, n))
            ms0_[:, ind0[0]] = ms0
            ms1_[:, ind1[0]] = ms1
            m0, m1, ms0, ms1 = m0_, m1_, ms0_, ms1_
        else:
            prune0 = torch.ones_like(ms0) * conf.n_layers
            prune1 = torch.ones_like(ms1) * conf.n_layers
        return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
                "ref_descriptors0": desc0[:, None], "ref_descriptors1": desc1[:, None],
                "log_assignment": scores, "prune0": prune0, "prune1": prune1, "stop_layer": last}

    # ------------------------------------------------------------------ loss
    @staticmethod
    def _gt_sparse(data, fixed=False):
        """COO positives and dustbin masks of the dense weight matrix of
        utils/losses.py:62-73 (weights = gt_assignment, (gt_matches0==-1), (gt_matches1==-1)).
        fixed=True (fused loss only): when the ground-truth producer also supplied
        ``gt_assignment_col0`` (the single positive column of each row, -1 if none) the list has the fixed
        length B*M with -1 marking "no positive": no scan of the dense matrix and no host sync."""
        neg0 = (data["gt_matches0"] == -1).float()
        neg1 = (data["gt_matches1"] == -1).float()
        bsz, m = neg0.shape
        col0 = data.get("gt_assignment_col0") if fixed else None
        if col0 is not None:
            dev = neg0.device
            pos = (torch.arange(bsz, device=dev).repeat_interleave(m),
                   torch.arange(m, device=dev).repeat(bsz), col0.reshape(-1).long())
            num_pos = (col0 >= 0).sum(-1).float().clamp(min=1.0)
        else:
            pos = data["gt_assignment"].nonzero(as_tuple=True)     # (b, i, j), one sync per step
            num_pos = torch.zeros(bsz, device=neg0.device).index_add_(
                0, pos[0], torch.ones_like(pos[0], dtype=torch.float32)).clamp(min=1.0)
        return {"pos": pos, "neg0": neg0, "neg1": neg1, "num_pos": num_pos,
                "n0": neg0.sum(-1).clamp(min=1.0), "n1": neg1.sum(-1).clamp(min=1.0)}

    def _nll(self, h, gt):
        """weight_loss of utils/losses.py:6-25 on the non-zero weights only."""
        bi, ii, ji = gt["pos"]
        m, n = h["md0"].shape[1], h["md1"].shape[1]
        f0, f1 = bi * m + ii, bi * n + ji     # flat row ids: index_select's backward is a sort-free index_add
        if "md" in h:                         # stacked: one gather / one scatter-add for both images
            rows = h["md"].flatten(0, 1).index_select(0, torch.cat([f0, f1 + h["md0"].shape[0] * m]))
            p_ = f0.shape[0]
            dot = (rows[:p_].float() * rows[p_:].float()).sum(-1)
        else:
            dot = (h["md0"].flatten(0, 1).index_select(0, f0).float()
                   * h["md1"].flatten(0, 1).index_select(0, f1).float()).sum(-1)
        a_pos = (2.0 * dot - h["r"].flatten().index_select(0, f0) - h["c"].flatten().index_select(0, f1)
                 + h["lz0"].flatten().index_select(0, f0) + h["lz1"].flatten().index_select(0, f1))
        nll_pos = -torch.zeros_like(gt["num_pos"]).index_add_(0, bi, a_pos) / gt["num_pos"]
        nll_neg = -((h["bin0"] * gt["neg0"]).sum(-1) + (h["bin1"] * gt["neg1"]).sum(-1)) / (gt["n0"] + gt["n1"])
        bal = self.conf.loss.nll_balancing
        nll = bal * nll_pos + (1 - bal) * nll_neg
        return nll, {"assignment_nll": nll, "nll_pos": nll_pos, "nll_neg": nll_neg,
                     "num_matchable": gt["num_pos"], "num_unmatchable": (gt["n0"] + gt["n1"]) / 2.0}

    @torch.compiler.disable
    def loss(self, pred, data):
        with torch.autocast(device_type="cuda", enabled=False):
            return self._loss(pred, data)

    def _layer_weights(self, L, gamma, device):
        """Deep-supervision weights of lightglue.py:596-606 as a device tensor, built once (a host -> device copy
        inside the step would break hipGraph capture and costs a launch per step)."""
        key = (L, float(gamma), str(device))
        cache = self.__dict__.setdefault("_w_cache", {})
        if key not in cache:
            w = [gamma ** (L - i - 1) if gamma > 0.0 else i + 1 for i in range(L - 1)] + [1.0]
            cache[key] = torch.tensor(w, device=device, dtype=torch.float32)
        return cache[key]

    def _loss_fused(self, pred, priv, data, gt):
        """Training loss on the batch-stacked per-layer descriptors: one fused HIP node per layer
        (ops.lg_layer_loss) and a handful of [L,B] tensor ops for the whole step."""
        layer_x = priv["layer_desc"]
        L = len(layer_x)
        b = pred["ref_descriptors0"].shape[0]
        n = layer_x[0].shape[1]
        fin0, fin1 = pred["_final_argmax0"], pred["_final_argmax1"]
        accs = []
        for i in range(L):
            x = layer_x[i]
            la = self.log_assignment[i]
            fh = priv["final_head"] if i == L - 1 else None
            t, t_done = None, False
            if fh is not None:           # the forward pass already projected the last layer
                md, z, rc = fh["md"], fh["z"], (fh["r"], fh["c"])
            else:
                # the gradients of x from these two heads ride in the next block's chain (GEMM / row-dot epilogues)
                ch = priv["layer_chain"][i]
                w, bias = la.scaled_proj(x.dtype)
                md = ops.linear(x, w, bias, chain=ch, chain_last="extra")
                tok = self.token_confidence[i].token[0] if i < L - 1 else None
                if tok is not None and x.is_cuda and x.shape[-1] % 8 == 0:
                    # matchability and token-confidence logits of this layer's output with one read of it
                    z, t = ops.rowdot2(x, la.matchability.weight, la.matchability.bias, tok.weight, tok.bias, chain=ch, counted=False)
                    t_done = True
                else:
                    z = ops.rowdot(x, la.matchability.weight, la.matchability.bias, chain=ch, counted=False)
                rc = None
            if not t_done:
                t = self.token_confidence[i].logits(x) if i < L - 1 else None
            accs.append(ops.lg_layer_loss(md, z, t, rc, gt["pos"], gt["neg0"], gt["neg1"], fin0, fin1))
        acc = torch.stack(accs)                                       # [L, B, 4]
        nll_pos = -acc[..., 0] / gt["num_pos"]
        nll_neg = -acc[..., 1] / (gt["n0"] + gt["n1"])
        bal = self.conf.loss.nll_balancing
        nll = bal * nll_pos + (1 - bal) * nll_neg                     # [L, B]
        gamma = self.conf.loss.gamma
        w = self._layer_weights(L, gamma, acc.device)
        total = (nll * w[:, None]).sum(0) / w.sum()
        losses = {"total": total, "last": nll[-1].detach(), "assignment_nll": nll[-1], "nll_pos": nll_pos[-1],
                  "nll_neg": nll_neg[-1], "num_matchable": gt["num_pos"],
                  "num_unmatchable": (gt["n0"] + gt["n1"]) / 2.0}
        if L > 1:
            losses["confidence"] = (acc[:-1, :, 2] + acc[:-1, :, 3]).sum(0) / (2.0 * n * (L - 1))
        else:
            losses["confidence"] = torch.zeros_like(total)
        with torch.no_grad():
            losses["row_norm"] = priv["row_norm"]
        losses["total"] = losses["total"] + losses["confidence"]
        return losses, {}

    def _loss(self, pred, data):
        rd0, rd1 = pred["ref_descriptors0"], pred["ref_descriptors1"]
        L = rd0.shape[1]
        priv = getattr(pred["log_assignment"], _PRIVATE, None)
        if priv is not None and self.training and "_final_argmax0" in pred and not priv.get("used"):
            # the tensors forward() returned, untouched: one fused node per layer.  (Once: the gradient chains of the step
            # are single-use; a second loss() on the same pred takes the dense formulation below.)
            priv["used"] = True
            return self._loss_fused(pred, priv, data, self._gt_sparse(data, fixed=True))
        # anything else -- eval mode, a pred dict that was sliced / concatenated / rebuilt by the caller (the reference's
        # TripletPipeline does both), different keypoint counts: the reference's own formulation on ref_descriptors
        gt = self._gt_sparse(data)

        def head(i):
            # i = -1 is the LAST assignment module on the LAST stored descriptors, as the reference's
            # loss_params(pred, -1) (lightglue.py:579-590): in eval mode only one layer is stored (L == 1).
            return self.log_assignment[i].stats(rd0[:, i], rd1[:, i])

        def conf_inputs(i):
            return rd0[:, i], rd1[:, i]

        nll, stats = self._nll(head(-1), gt)
        losses = {"total": nll, "last": nll.clone().detach(), **stats}
        if self.training:
            losses["confidence"] = 0.0
        with torch.no_grad():
            losses["row_norm"] = pred["log_assignment"].exp()[:, :-1].sum(2).mean(1)
        if "_final_argmax0" in pred:
            fin0, fin1 = pred["_final_argmax0"], pred["_final_argmax1"]
        else:  # pred built elsewhere: read the arg-maxes off the materialised matrix
            la = pred["log_assignment"].detach()
            fin0, fin1 = la[:, :-1, :].max(-1).indices, la[:, :, :-1].max(-2).indices

        sum_weights = 1.0
        gamma = self.conf.loss.gamma
        for i in range(L - 1):
            h = head(i)
            nll_i, _ = self._nll(h, gt)
            weight = gamma ** (L - i - 1) if gamma > 0.0 else i + 1
            sum_weights += weight
            losses["total"] = losses["total"] + nll_i * weight
            am = MatchAssignment.argmaxes(h)
            tc = self.token_confidence[i]
            bce = F.binary_cross_entropy_with_logits
            c0, c1 = conf_inputs(i)
            conf_i = (bce(tc.logits(c0), (am["full0"] == fin0).float(), reduction="none").mean(-1)
                      + bce(tc.logits(c1), (am["full1"] == fin1).float(), reduction="none").mean(-1)) / 2.0
            losses["confidence"] = losses["confidence"] + conf_i / (L - 1)
        losses["total"] = losses["total"] / sum_weights
        if self.training:
            losses["total"] = losses["total"] + losses["confidence"]
            metrics = {}
        else:
            metrics = matcher_metrics(pred, data)
        return losses, metrics


__main_model__ = LightGlue
