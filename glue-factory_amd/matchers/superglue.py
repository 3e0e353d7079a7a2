"""SuperGlue matcher on the MI355X hot path (drop-in for gluefactory_nonfree.superglue).

Same plugin surface as the reference (gluefactory_nonfree/superglue.py:221-355): ``BaseModel``
subclass, same ``default_conf`` keys / ``required_data_keys`` and the same ``state_dict``
names and shapes (Conv1d weights stay ``(out, in, 1)``, BatchNorm running statistics included).

What runs where
  * Conv1d(k=1) layers are GEMMs over channels-last activations ``[2B, N, C]`` (both images
    stacked) on ``gf_gemm`` (forward and input gradient, fused bias / two-source concat) and
    ``gf_linear_dw`` (weight gradient); only the 3-channel input layer of the keypoint
    encoder (K = 3) falls outside the kernel's plans and uses the library;
  * attention (superglue.py:112-135): the MFMA flash kernels of csrc/attention.hip.  The
    reference puts the head index FASTEST in the channel dimension (``view(b, dim, h, n)``);
    the projection weight rows (and the merge weight columns) are gathered once so the kernels
    see ``[.., head, channel]`` with contiguous channels — no activation shuffles;
  * BatchNorm (+ReLU) is the fused HIP op ``ops.batch_norm_act`` on the ``nn.BatchNorm1d`` modules'
    parameters / running statistics (SyncBatchNorm-convertible: the sums are all-reduced inside
    the op), one call per image exactly like the reference (batch statistics per image set);
  * couplings ``[[scores/sqrt(d), a],[a, a]]`` are written in one pass by ``gf_assign_write``
    from MFMA tiles (no ``sim`` tensor, no torch.cat), then the log-domain Sinkhorn iterations
    and their hand-derived reverse sweep run in csrc/sinkhorn.hip (superglue.py:186-214);
  * ``loss`` returns ``(losses, {})`` — the reference returns a bare dict, which breaks
    ``TwoViewPipeline.loss``'s tuple unpack; the values match the dict's.
"""
from pathlib import Path

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..base_model import BaseModel


def MLP(channels, do_bn=True):
    layers = []
    n = len(channels)
    for i in range(1, n):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < n - 1:
            if do_bn:
                layers.append(nn.BatchNorm1d(channels[i]))
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def _conv_cl(x, conv, rows=None, cols=None):
    """Conv1d(k=1) on channels-last x [..., Cin]; optional gather of output rows / input columns."""
    pc = getattr(conv, "_pc", None)
    if pc is not None and rows is None and cols is not None:
        wd = ops.derived_weight(pc[0], x.dtype, pc[1] + ".w", conv.weight)      # column gather done by the precast launch
        if wd is not None:
            return ops.linear(x, wd, conv.bias)
    w = conv.weight.squeeze(-1)
    b = conv.bias
    if rows is not None:
        w, b = w.index_select(0, rows), b.index_select(0, rows)
    if cols is not None:
        w = w.index_select(1, cols)
    if w.shape[1] <= 8 and w.shape[0] * w.shape[1] <= 256 and x.dtype == torch.float32 and x.is_cuda:
        # the keypoint / endpoint encoders' first layer (K = 3 or 5 input columns, superglue.py:82-91, gluestick.py:489-521): its weight gradient
        # is a 1e5-deep reduction, not a GEMM (gf_small_dw)
        return ops.small_linear(x, w) + b
    if w.shape[0] % 8 or w.shape[1] % 8:
        return F.linear(x, w.to(x.dtype), b.to(x.dtype))
    return ops.linear(x, w, b)


def _mlp_cl(seq, x, halves, x2=None, res=None, chain=None, first_wb=None, replay=False, replay_out=None):
    """Run an MLP Sequential (Conv1d / BatchNorm1d / ReLU) on channels-last x [B',N,C].
    BatchNorm (+ the ReLU that follows it) is one fused HIP pass pair, applied per image set
    (``halves`` = 2 when two images are stacked on the batch axis), reproducing the reference's
    one-call-per-image statistics and running-stat updates.  ``res``: added to the output inside the last
    convolution's GEMM epilogue; ``chain``: ops.GradChain of x (= res) for the first convolution and the residual.
    ``first_wb``: (weight, bias) replacing the first convolution's (the merge convolution folded into it: x2 is then the
    attention output itself, AttentionalPropagation.forward).  ``replay``: the MLP sits inside an activation-checkpointed
    block of the reference (its backward re-runs the forward in training mode), so the BatchNorm running statistics take
    every update TWICE per step -- the second one is applied from our backward (ops.batch_norm_act_sets); with ``replay_out``
    (a list) the replays are collected for ops.replay_running_stats instead (several calls, replayed in call order)."""
    layers = list(seq)
    i = 0
    if x2 is not None:      # first conv on cat[x, x2] without building the concatenation
        first = layers[0]
        w0, b0 = (first.weight.squeeze(-1), first.bias) if first_wb is None else first_wb
        x = ops.linear_cat(x, x2, w0, b0, chain1=chain)
        i = 1
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Conv1d) and i == len(layers) - 1 and res is not None \
                and layer.out_channels % 8 == 0 and layer.in_channels % 8 == 0:
            x = ops.linear(x, layer.weight.squeeze(-1), layer.bias, res=res, res_chain=chain)
            res = None
        elif isinstance(layer, nn.Conv1d):
            x = _conv_cl(x, layer)
        elif isinstance(layer, nn.modules.batchnorm._BatchNorm):
            relu = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            b, n, c = x.shape
            if c % 8:
                parts = x.float().reshape(halves, (b // halves) * n, c)
                y = torch.stack([layer(parts[k]) for k in range(halves)]).reshape(b, n, c).to(x.dtype)
                x = torch.relu(y) if relu else y
            else:
                parts = x.reshape(halves, (b // halves) * n, c)
                x = ops.batch_norm_act_sets(parts, layer, relu, replay=replay, stats_out=replay_out).reshape(b, n, c)
            i += int(relu)
        else:
            x = layer(x)
        i += 1
    return x if res is None else x + res


def normalize_keypoints(kpts, size=None, shape=None):
    """Centre and divide by 0.7 * max(size) (superglue.py:82-93), fp32."""
    kpts = kpts.float()
    if size is None:
        assert shape is not None
        _, _, h, w = shape
        size = kpts.new_tensor([[w, h]])
    size = size.float().to(kpts)
    return (kpts - size[:, None] / 2) / (size.max(1).values * 0.7)[:, None, None]


class KeypointEncoder(nn.Module):
    def __init__(self, feature_dim, layers, use_scores=True):
        super().__init__()
        self.use_scores = use_scores
        self.encoder = MLP([3 if use_scores else 2] + list(layers) + [feature_dim])
        nn.init.constant_(self.encoder[-1].bias, 0.0)

    def forward(self, kpts, scores, halves=1):
        x = torch.cat([kpts, scores[..., None]], -1) if self.use_scores else kpts
        return _mlp_cl(self.encoder, x.float(), halves)


class MultiHeadedAttention(nn.Module):
    def __init__(self, h, d_model):
        super().__init__()
        assert d_model % h == 0
        self.dim, self.h = d_model // h, h
        self.merge = nn.Conv1d(d_model, d_model, kernel_size=1)
        self.proj = nn.ModuleList([nn.Conv1d(d_model, d_model, kernel_size=1) for _ in range(3)])
        # reference channel c = channel_in_head * h + head  ->  kernel order head * dim + channel
        perm = (torch.arange(self.dim)[None, :] * h + torch.arange(h)[:, None]).reshape(-1)
        self.register_buffer("_perm", perm, persistent=False)

    _pc = None      # (precast key, name): set by the owning model (derived_specs_of)

    def derived_specs(self, name):
        """The stacked (q | k | v) projection in kernel channel order, head_dim^-1/2 log2(e) folded into the q rows, as
        entries of the model's per-step precast launch (ops.precast(derived=...))."""
        qs = ops.attn_premul(self.dim)
        self.merge._pc = (self._pc[0], name + ".merge") if self._pc is not None else None
        return [(name + ".w", [(p.weight, self._perm, None, qs if i == 0 else 1.0) for i, p in enumerate(self.proj)]),
                (name + ".b", [(p.bias, self._perm, None, qs if i == 0 else 1.0) for i, p in enumerate(self.proj)]),
                # the merge convolution reads the attention output in kernel channel order: its columns gathered
                (name + ".merge.w", [(self.merge.weight, None, None, 1.0, self._perm)])]

    def fold_spec(self, name, first):
        """merge feeds ONLY the MLP's first convolution, through a concatenation (superglue.py:137-160): mlp.0(cat[x, Wm o +
        bm]) = [W0a | W0b Wm] cat[x, o] + (b0 + W0b bm) -- prepared by the per-step precast launch (csrc/fold.hip), with the
        merge columns gathered into kernel channel order on the way."""
        return (name + ".mlp0", "fold", first.weight, first.bias, self.merge.weight, self.merge.bias,
                first.in_channels - self.merge.out_channels, self._perm)

    def fused_projection(self, x, chain=None, premul=True):
        """-> (qkv [B', N, 3, H, D], softmax scale the attention op has to use)."""
        w = None
        if premul and self._pc is not None:
            w = ops.derived_weight(self._pc[0], x.dtype, self._pc[1] + ".w", *[p.weight for p in self.proj])
        if w is not None:
            b = ops.derived_weight(self._pc[0], x.dtype, self._pc[1] + ".b", *[p.bias for p in self.proj])
            scale = ops.LN2
        else:
            w = torch.cat([p.weight.squeeze(-1).index_select(0, self._perm) for p in self.proj], 0)
            b = torch.cat([p.bias.index_select(0, self._perm) for p in self.proj], 0)
            scale = None
        qkv = ops.linear(x, w, b, chain=chain, chain_last=True)
        return qkv.view(x.shape[0], x.shape[1], 3, self.h, self.dim), scale


def derived_specs_of(model):
    """Every module of `model` that offers prepared weights, named by its module path."""
    specs = []
    for name, mod in model.named_modules():
        if hasattr(mod, "derived_specs") and mod is not model:
            mod._pc = (id(model), name)
            specs += mod.derived_specs(name)
        if isinstance(mod, AttentionalPropagation):      # (its attention module's specs follow under "<name>.attn")
            specs.append(mod.attn.fold_spec(name + ".attn", mod.mlp[0]))
    return specs


class AttentionalPropagation(nn.Module):
    attention_fp32 = False      # GlueStick sets it per its `attention_precision` configuration
    # SuperGlue's GNN wraps every layer in torch.utils.checkpoint while training (superglue.py:160-169): its backward re-runs the
    # layer's forward in training mode, so the BatchNorm inside updates its running statistics TWICE per call.  To leave the
    # same buffers behind (eval after training = the reference's), the second update is applied from our backward.
    # GlueStick sets this per its `checkpointed` configuration (gluestick.py:724-757).
    checkpointed_in_reference = True

    def __init__(self, num_dim, num_heads):
        super().__init__()
        self.attn = MultiHeadedAttention(num_heads, num_dim)
        self.mlp = MLP([num_dim * 2, num_dim * 2, num_dim])
        nn.init.constant_(self.mlp[-1].bias, 0.0)

    def forward(self, x, cross, halves, residual=False):
        """x [B',N,C] (stacked images when halves == 2); returns the residual delta, or with ``residual`` the updated
        x + delta (the addition rides in the last GEMM's epilogue, and the three gradients that meet in x -- residual,
        MLP input, projection -- are summed in GEMM epilogues: ops.GradChain)."""
        b, n, d = x.shape
        chain = ops.GradChain(3) if residual and x.requires_grad and torch.is_grad_enabled() else None
        qkv, scale = self.attn.fused_projection(x, chain)
        # attention_fp32 (GlueStick, `attention_precision: reference`): the reference forces THIS attention to fp32 under mixed
        # precision (gluestick.py:18-22, 524-529 @AMP_CUSTOM_FWD_F32) -- scores, softmax and the weighted sum in fp32 on the
        # bf16-valued projections: the bf16 kernels with the softmax weights / score gradients split into hi + lo pairs
        split = self.attention_fp32 and qkv.dtype != torch.float32
        if split and self.attn.dim != 64:      # no split instantiation for this head width: the reference's own form, the
            o = ops.attention_qkv(qkv.float(), cross=cross, scale=scale).to(qkv.dtype)   # projections cast up to fp32
        else:
            o = ops.attention_qkv(qkv, cross=cross, scale=scale, split=split)
        pc = self.attn._pc
        first = None if pc is None else ops.folded_linear(pc[0], x.dtype, pc[1] + ".mlp0", self.mlp[0].weight, self.mlp[0].bias,
                                                          self.attn.merge.weight, self.attn.merge.bias)
        replay = self.checkpointed_in_reference and self.training
        if first is not None:       # merge lives inside mlp.0's weight: the MLP reads the attention output directly
            return _mlp_cl(self.mlp, x, halves, x2=o.view(b, n, d), res=x if residual else None, chain=chain, first_wb=first,
                           replay=replay)
        msg = _conv_cl(o.view(b, n, d), self.attn.merge, cols=self.attn._perm)
        return _mlp_cl(self.mlp, x, halves, x2=msg, res=x if residual else None, chain=chain, replay=replay)

    def forward_pair(self, x0, x1, cross):
        """Different keypoint counts: one projection per image, generic attention op."""
        outs = []
        # (the reference's checkpoint re-runs image 0's call, then image 1's: the replays are collected and attached to ONE node,
        # because the two calls' own autograd nodes run in the opposite order)
        pending = [] if self.checkpointed_in_reference and self.training else None
        p0, p1 = self.attn.fused_projection(x0, premul=False)[0], self.attn.fused_projection(x1, premul=False)[0]
        for x, pq, ps in ((x0, p0, p1 if cross else p0), (x1, p1, p0 if cross else p1)):
            split = self.attention_fp32 and pq.dtype != torch.float32
            if split and self.attn.dim != 64:
                o = ops.attention(pq[:, :, 0].float(), ps[:, :, 1].float(), ps[:, :, 2].float()).to(pq.dtype)
            else:
                o = ops.attention(pq[:, :, 0], ps[:, :, 1], ps[:, :, 2], split=split)
            msg = _conv_cl(o.reshape(x.shape), self.attn.merge, cols=self.attn._perm)
            outs.append(_mlp_cl(self.mlp, x, 1, x2=msg, replay_out=pending))
        if pending:
            outs[0] = ops.replay_running_stats(outs[0], pending)
        return outs


class AttentionalGNN(nn.Module):
    def __init__(self, feature_dim, layer_names):
        super().__init__()
        self.layers = nn.ModuleList([AttentionalPropagation(feature_dim, 4) for _ in layer_names])
        self.names = list(layer_names)


class SuperGlue(BaseModel):
    default_conf = {
        "descriptor_dim": 256,
        "weights": None,           # path to a state_dict (the reference default "outdoor" needs network)
        "keypoint_encoder": [32, 64, 128, 256],
        "GNN_layers": ["self", "cross"] * 9,
        "num_sinkhorn_iterations": 50,
        "filter_threshold": 0.2,
        "use_scores": True,
        "mp": False,
        "loss": {"nll_balancing": 0.5},
    }
    required_data_keys = ["view0", "view1", "keypoints0", "keypoints1", "descriptors0", "descriptors1",
                          "keypoint_scores0", "keypoint_scores1"]

    def _init(self, conf):
        if conf.descriptor_dim not in (128, 256, 512):
            raise NotImplementedError("the HIP attention kernels exist for 4 heads of 64 channels (tuned) and of 32 / 128 (generic kernels)")
        self.kenc = KeypointEncoder(conf.descriptor_dim, conf.keypoint_encoder, conf.use_scores)
        self.gnn = AttentionalGNN(conf.descriptor_dim, conf.GNN_layers)
        self.final_proj = nn.Conv1d(conf.descriptor_dim, conf.descriptor_dim, kernel_size=1, bias=True)
        self.register_parameter("bin_score", nn.Parameter(torch.tensor(1.0)))
        if conf.weights:
            path = Path(conf.weights)
            if not path.exists():
                raise FileNotFoundError(f"SuperGlue weights '{conf.weights}' not found locally "
                                        "(pretrained downloads need network access)")
            self.load_state_dict(torch.load(str(path), map_location="cpu"))

    # torch.compile(model) (gluefactory/train.py:332-333): the HIP path is opaque to dynamo -- ctypes launches inside
    # autograd.Functions, host-side caches -- so forward / loss are one clean graph break and run eagerly
    @torch.compiler.disable
    def _forward(self, data):
        kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:
            s0, s1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return {"matches0": kpts0.new_full(s0, -1, dtype=torch.int),
                    "matches1": kpts1.new_full(s1, -1, dtype=torch.int),
                    "matching_scores0": kpts0.new_zeros(s0), "matching_scores1": kpts1.new_zeros(s1)}
        if not kpts0.is_cuda:
            raise RuntimeError("glue_factory_amd.SuperGlue runs on the MI355X HIP path only (no CPU fallback)")
        T = torch.bfloat16 if (self.conf.mp or torch.is_autocast_enabled()) else torch.float32
        with torch.autocast(device_type="cuda", enabled=False):
            # one launch: compute-dtype + transposed weights + the layers' prepared (q | k | v) projections
            ops.precast(list(self.parameters()), T, key=id(self), derived=derived_specs_of(self) if T != torch.float32 else None)
            return self._forward_impl(data, T)

    def _forward_impl(self, data, T):
        conf = self.conf
        view0, view1 = data["view0"], data["view1"]
        kpts0 = normalize_keypoints(data["keypoints0"], size=view0.get("image_size"),
                                    shape=view0["image"].shape if "image" in view0 else None)
        kpts1 = normalize_keypoints(data["keypoints1"], size=view1.get("image_size"),
                                    shape=view1["image"].shape if "image" in view1 else None)
        b, m = kpts0.shape[:2]
        n = kpts1.shape[1]
        sc0, sc1 = data["keypoint_scores0"].float(), data["keypoint_scores1"].float()
        d = conf.descriptor_dim
        if m == n:
            enc = self.kenc(torch.cat([kpts0, kpts1], 0), torch.cat([sc0, sc1], 0), halves=2)
            x = (torch.cat([data["descriptors0"], data["descriptors1"]], 0).float() + enc).to(T)
            for layer, name in zip(self.gnn.layers, self.gnn.names):
                if name not in ("self", "cross"):
                    raise ValueError(name)
                x = layer(x, cross=(name == "cross"), halves=2, residual=True)
            md = _conv_cl(x, self.final_proj)
            md0, md1 = md[:b], md[b:]
        else:
            x0 = (data["descriptors0"].float() + self.kenc(kpts0, sc0)).to(T)
            x1 = (data["descriptors1"].float() + self.kenc(kpts1, sc1)).to(T)
            for layer, name in zip(self.gnn.layers, self.gnn.names):
                if name not in ("self", "cross"):
                    raise ValueError(name)
                d0, d1 = layer.forward_pair(x0, x1, cross=(name == "cross"))
                x0, x1 = x0 + d0, x1 + d1
            md0, md1 = _conv_cl(x0, self.final_proj), _conv_cl(x1, self.final_proj)

        alpha = self.bin_score.float()
        # couplings Z = [[md0.md1^T / sqrt(d), a], [a, a]] in one MFMA pass (superglue.py:200-204)
        zero_m, zero_n = md0.new_zeros((b, m), dtype=torch.float32), md0.new_zeros((b, n), dtype=torch.float32)
        Z = ops.assign_write(md0.contiguous(), md1.contiguous(), zero_m, zero_n, alpha.expand(b, m),
                             alpha.expand(b, n), alpha=d ** -0.5, corner=alpha)
        scores = ops.sinkhorn(Z, conf.num_sinkhorn_iterations)

        with torch.no_grad():
            core = scores[:, :-1, :-1]
            max0, a0 = core.max(2)
            a1 = core.max(1).indices
            m0, m1, ms0, ms1 = ops.filter_matches(max0, a0, a1, conf.filter_threshold)
        return {"sinkhorn_cost": Z[:, :-1, :-1], "log_assignment": scores, "matches0": m0, "matches1": m1,
                "matching_scores0": ms0, "matching_scores1": ms1}

    @torch.compiler.disable
    def loss(self, pred, data):
        la = pred["log_assignment"]
        neg0 = (data["gt_matches0"] == -1).float()
        neg1 = (data["gt_matches1"] == -1).float()
        pos_sum, num_pos, neg_sum = ops.nll_terms(la, data, neg0, neg1)   # one autograd node when gt_assignment_col0 is there
        num_pos = num_pos.clamp(min=1.0)
        num_neg = (neg0.sum(1) + neg1.sum(1)).clamp(min=1.0)
        nll_pos = -pos_sum / num_pos
        nll_neg = -neg_sum / num_neg
        bal = self.conf.loss.nll_balancing
        nll = bal * nll_pos + (1 - bal) * nll_neg
        losses = {"total": nll, "assignment_nll": nll, "nll_pos": nll_pos, "nll_neg": nll_neg,
                  "num_matchable": num_pos, "num_unmatchable": num_neg, "bin_score": self.bin_score[None]}
        return losses, {}


__main_model__ = SuperGlue
