"""torch.autograd.Function wrappers around the HIP launchers of libgf_amd.so.

PyTorch is plumbing here (device memory, streams, autograd graph); every op below calls the
C ABI of include/gf_amd.h through ctypes with raw device pointers and the current HIP stream.
There is no CPU or eager fallback: a non-CUDA tensor or a missing library raises.
"""
import os
import weakref

import torch

from . import lib as _lib

F32, BF16 = 0, 1


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError(f"glue_factory_amd kernels take float32 or bfloat16, got {t.dtype}")


def _chk(*ts):
    """Every launcher enqueues on the CURRENT device's current stream with raw pointers: tensors on another
    GPU would be touched from the wrong stream (faults or silent races), so that is an error, not a fallback."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("glue_factory_amd ops need tensors on a HIP device (no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: call "
                               "torch.cuda.set_device (one process per GPU) before using glue_factory_amd ops")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _s3(t):
    """(batch, token, head) element strides of a [B,N,H,D] view with contiguous D."""
    assert t.dim() == 4 and t.stride(3) == 1, "attention operands must be [B,N,H,D] with contiguous D"
    return _lib.strides(t.stride(0), t.stride(1), t.stride(2))


# ------------------------------------------------------------------------------ attention
ATTN_SPLIT = 4       # GF_ATTN_SPLIT (include/gf_amd.h): P / dS as hi + lo bf16 pairs = fp32-equivalent second products


def attn_fwd_raw(q, k, v, scale, out=None, lse=None, split=False, o32=None):
    """split (bf16 only): fp32-equivalent second products; o32: [B, Nq, H, D] fp32 contiguous buffer that receives the
    un-rounded output next to `out` (attn_bwd_raw(split=True) takes it as its `o`)."""
    _chk(q, k, v)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    o = torch.empty((B, Nq, H, D), dtype=q.dtype, device=q.device) if out is None else out
    if lse is None:
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
    split = bool(split) and q.dtype == torch.bfloat16
    assert o32 is None or (o32.dtype == torch.float32 and o32.is_contiguous() and tuple(o32.shape) == (B, Nq, H, D))
    _lib.check(_lib.load().gf_attn_fwd_ex(_p(q), _p(k), _p(v), _p(o), _p(lse), B, H, Nq, Nk, D,
                                          _s3(q), _s3(k), _s3(v), _s3(o), float(scale), _dt(q), ATTN_SPLIT if split else 0,
                                          _p(o32) if split else None, _stream()), "gf_attn_fwd_ex")
    return o, lse


def attn_bwd_raw(q, k, v, o, do, lse, dq, dk, dv, scale, acc_dq=False, acc_dk=False, split=False):
    """split (bf16 only): `o` must be the fp32 copy attn_fwd_raw(split=True, o32=...) wrote."""
    B, Nq, H, D = q.shape
    split = bool(split) and q.dtype == torch.bfloat16
    assert not split or o.dtype == torch.float32
    Nk = k.shape[1]
    if do.stride(3) != 1:
        do = do.contiguous()
    delta = lse.new_empty((2,) + tuple(lse.shape))      # scratch: the two per-row vectors the dQ kernel hands to dK/dV
    _lib.check(_lib.load().gf_attn_bwd_acc(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta),
                                       _p(dq), _p(dk), _p(dv), B, H, Nq, Nk, D,
                                       _s3(q), _s3(k), _s3(v), _s3(o), _s3(do), _s3(dq), _s3(dk),
                                       _s3(dv), float(scale), _dt(q),
                                       int(acc_dq) | 2 * int(acc_dk) | (ATTN_SPLIT if split else 0), _stream()),
               "gf_attn_bwd_acc")


class _Attention(torch.autograd.Function):
    """o = softmax(scale q k^T) v on [B,N,H,D] views (generic entry, used by SuperGlue/GlueStick)."""

    @staticmethod
    def forward(ctx, q, k, v, scale, split=False):
        split = bool(split) and q.dtype == torch.bfloat16
        o32 = torch.empty(q.shape, dtype=torch.float32, device=q.device) if split else None
        o, lse = attn_fwd_raw(q, k, v, scale, split=split, o32=o32)
        ctx.save_for_backward(q, k, v, o32 if split else o, lse)
        ctx.scale, ctx.split = scale, split
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dq, dk, dv = (t.contiguous() for t in (dq, dk, dv))
        attn_bwd_raw(q, k, v, o, do, lse, dq, dk, dv, ctx.scale, split=ctx.split)
        return dq, dk, dv, None, None


def attention(q, k, v, scale=None, split=False):
    """split: fp32-equivalent second products on bf16 operands (GF_ATTN_SPLIT; no effect on fp32 tensors)."""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    return _Attention.apply(q, k, v, scale, split)


class _SelfAttentionRotary(torch.autograd.Function):
    """Rotary(q,k) + self attention on the fused projection qkv [B,N,3,H,D].

    qkv is the private output buffer of the Wqkv GEMM: it is rotated IN PLACE (nobody else
    reads it) and kept for the backward.  theta [B,N,D/2] are the pair angles (differentiable,
    they carry the gradient to posenc.Wr); cs [B,N,D] = interleaved (cos, sin) of theta."""

    @staticmethod
    def forward(ctx, qkv, theta, cs, pre_rotated=False, scale=None, theta_sum=None):
        _chk(qkv, cs)
        ctx.theta_sum = theta_sum
        B, N, three, H, D = qkv.shape
        assert three == 3 and qkv.is_contiguous() and cs.is_contiguous() and cs.dtype == torch.float32
        L = _lib.load()
        if not pre_rotated:      # else: q and k left the Wqkv GEMM already rotated (gf_gemm's rotary epilogue)
            _lib.check(L.gf_rotary_qk(_p(qkv), _p(cs), B, N, H, D, 0, _dt(qkv), _stream()), "gf_rotary_qk")
        ctx.scale = D ** -0.5 if scale is None else scale
        o, lse = attn_fwd_raw(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], ctx.scale)
        ctx.save_for_backward(qkv, cs, o, lse)
        ctx.theta_dtype = theta.dtype
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, cs, o, lse = ctx.saved_tensors
        B, N, _, H, D = qkv.shape
        dqkv = torch.empty_like(qkv)
        attn_bwd_raw(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, do, lse,
                     dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], ctx.scale)
        dtheta = torch.empty((B, N, D // 2), dtype=torch.float32, device=qkv.device)
        ts = ctx.theta_sum
        base = None if ts is None else ts.acc
        _lib.check(_lib.load().gf_rotary_qk_bwd(_p(dqkv), _p(qkv), _p(cs), _p(dtheta), _p(base), B, N, H, D,
                                                _dt(qkv), _stream()), "gf_rotary_qk_bwd")
        if ts is not None:          # the layers share theta: only the LAST backward returns the (complete) sum
            dtheta = ts.add(dtheta)
            if dtheta is None:
                return dqkv, None, None, None, None, None
        return dqkv, dtheta.to(ctx.theta_dtype), None, None, None, None


class SharedGradSum:
    """Gradient of ONE tensor consumed by `consumers` nodes of the same kind (the rotary angles of LightGlue's L self
    blocks): every node's backward kernel adds the running sum of the nodes that ran before it (`acc`, passed as the
    kernel's base operand) and only the last one hands the total to autograd -- L - 1 elementwise adds fewer.  A backward
    pass that does not visit all consumers (torch.autograd.grad on an intermediate layer, a loss on layer k < L - 1 only)
    would silently drop the visited consumers' share: `check()` raises in that case; TrainStep calls `check_all()` after its
    backward (callers that drive autograd themselves on a partial graph can do the same)."""
    __slots__ = ("acc", "got", "expected", "__weakref__")
    live = None           # the sums armed by the last forward (weak): TrainStep checks them after its backward

    def __init__(self, consumers):
        import weakref
        self.acc, self.got, self.expected = None, 0, int(consumers)
        if SharedGradSum.live is None:
            SharedGradSum.live = weakref.WeakSet()
        SharedGradSum.live.add(self)

    def add(self, g):
        self.got += 1
        if self.got < self.expected:
            self.acc = g
            return None
        self.acc, self.got = None, 0
        return g

    def check(self):
        if 0 < self.got < self.expected:
            got = self.got
            self.acc, self.got = None, 0
            raise RuntimeError(f"SharedGradSum: the backward visited {got} of {self.expected} consumers of a shared tensor -- "
                               "its gradient is incomplete (differentiate through all layers, or build the model without the shared sum)")

    @classmethod
    def check_all(cls):
        for s_ in list(cls.live or ()):
            s_.check()


LN2 = 0.6931471805599453
# "Pre-multiplied operands": a caller that folds head_dim^-1/2 * log2(e) into the projection that PRODUCES q (one rounding,
# in the GEMM's fp32 epilogue) passes scale = LN2, i.e. softmax(ln2 * q'.k) = 2^(q'.k) / sum: the kernels then take the
# scores straight from the matrix pipe as exp2 arguments (no multiply per score; csrc/attn_common.h host_split_scale).
def attn_premul(head_dim):
    """The factor to fold into q (self attention) -- or its square root into both operands (cross attention, where the
    same tensor is query in one direction and key in the other) -- when calling the attention ops with scale=LN2."""
    return head_dim ** -0.5 * 1.4426950408889634


def self_attention_rotary(qkv, theta, cs, pre_rotated=False, scale=None, theta_sum=None):
    """theta_sum: a SharedGradSum over all layers that share `theta` (None: every call returns its own angle gradient)."""
    if theta_sum is not None and not (torch.is_grad_enabled() and theta.requires_grad):
        theta_sum = None
    return _SelfAttentionRotary.apply(qkv, theta, cs, pre_rotated, scale, theta_sum)


class _CrossAttention(torch.autograd.Function):
    """Bidirectional cross attention with shared qk projection.

    p0, p1: [B,N_i,2,H,D] fused (to_qk, to_v) projections of image 0 / 1.
    m0 = softmax(qk0 qk1^T / sqrt(D)) v1,  m1 = softmax(qk1 qk0^T / sqrt(D)) v0."""

    @staticmethod
    def forward(ctx, p0, p1, scale=None):
        D = p0.shape[-1]
        sc = ctx.scale = D ** -0.5 if scale is None else scale
        m0, lse0 = attn_fwd_raw(p0[:, :, 0], p1[:, :, 0], p1[:, :, 1], sc)
        m1, lse1 = attn_fwd_raw(p1[:, :, 0], p0[:, :, 0], p0[:, :, 1], sc)
        ctx.save_for_backward(p0, p1, m0, m1, lse0, lse1)
        return m0, m1

    @staticmethod
    def backward(ctx, dm0, dm1):
        p0, p1, m0, m1, lse0, lse1 = ctx.saved_tensors
        D = p0.shape[-1]
        d0, d1 = torch.empty_like(p0), torch.empty_like(p1)
        # direction 0->1: q = qk0, k = qk1, v = v1: writes d qk0 (as query) and d qk1 (as key)
        attn_bwd_raw(p0[:, :, 0], p1[:, :, 0], p1[:, :, 1], m0, dm0, lse0,
                     d0[:, :, 0], d1[:, :, 0], d1[:, :, 1], ctx.scale)
        # direction 1->0: q = qk1, k = qk0, v = v0: ADDS d qk1 (as query) and d qk0 (as key) in the kernels' epilogues
        attn_bwd_raw(p1[:, :, 0], p0[:, :, 0], p0[:, :, 1], m1, dm1, lse1,
                     d1[:, :, 0], d0[:, :, 0], d0[:, :, 1], ctx.scale, acc_dq=True, acc_dk=True)
        return d0, d1, None


class _CrossAttentionStacked(torch.autograd.Function):
    """Same as _CrossAttention for equal keypoint counts, on the batch-stacked projection
    p [2B,N,2,H,D] (image 0 in the first half); returns the stacked messages [2B,N,H,D] so
    the following to_out GEMM runs once over both images without a concat."""

    @staticmethod
    def forward(ctx, p, scale=None):
        B2, N, _, H, D = p.shape
        B = B2 // 2
        sc = ctx.scale = D ** -0.5 if scale is None else scale
        m = torch.empty((B2, N, H, D), dtype=p.dtype, device=p.device)
        lse = torch.empty((B2, H, N), dtype=torch.float32, device=p.device)
        p0, p1 = p[:B], p[B:]
        attn_fwd_raw(p0[:, :, 0], p1[:, :, 0], p1[:, :, 1], sc, out=m[:B], lse=lse[:B])
        attn_fwd_raw(p1[:, :, 0], p0[:, :, 0], p0[:, :, 1], sc, out=m[B:], lse=lse[B:])
        ctx.save_for_backward(p, m, lse)
        return m

    @staticmethod
    def backward(ctx, dm):
        p, m, lse = ctx.saved_tensors
        B2, N, _, H, D = p.shape
        B = B2 // 2
        if not dm.is_contiguous():
            dm = dm.contiguous()
        d = torch.empty_like(p)
        if XBWD_ENABLED and p.dtype == torch.bfloat16 and D == 64 and H <= 4 and N % 64 == 0:
            # both directions from ONE score tile per image side (csrc/attention_xbwd.hip): 10 MFMA products instead of 14
            stat = torch.empty((2, B2, H, N), dtype=torch.float32, device=p.device)
            _lib.check(_lib.load().gf_attn_cross_bwd(_p(p[:, :, 0]), _p(p[:, :, 1]), _p(m), _p(dm), _p(lse), _p(stat),
                                                     _p(d[:, :, 0]), _p(d[:, :, 1]), B2, B, H, N, D,
                                                     _s3(p[:, :, 0]), _s3(p[:, :, 1]), _s3(m), _s3(dm), _s3(d[:, :, 0]),
                                                     _s3(d[:, :, 1]), float(ctx.scale), _dt(p), _stream()), "gf_attn_cross_bwd")
            return d, None
        p0, p1, d0, d1 = p[:B], p[B:], d[:B], d[B:]
        attn_bwd_raw(p0[:, :, 0], p1[:, :, 0], p1[:, :, 1], m[:B], dm[:B], lse[:B],
                     d0[:, :, 0], d1[:, :, 0], d1[:, :, 1], ctx.scale)
        # the second direction adds its query / key gradients to the first one's in the kernels' epilogues
        attn_bwd_raw(p1[:, :, 0], p0[:, :, 0], p0[:, :, 1], m[B:], dm[B:], lse[B:],
                     d1[:, :, 0], d0[:, :, 0], d0[:, :, 1], ctx.scale, acc_dq=True, acc_dk=True)
        return d, None


XBWD_ENABLED = True      # tools/probe/ab_matcher.py --switch XBWD_ENABLED: same-process A/B against two gf_attn_bwd_acc calls


def cross_attention(p0, p1, scale=None):
    return _CrossAttention.apply(p0, p1, scale)


def cross_attention_stacked(p, scale=None):
    return _CrossAttentionStacked.apply(p, scale)


# ------------------------------------------------------------------------------ linear layer
# ---- per-step low-precision copies of the fp32 master parameters -------------------------------
# Every linear casts its weight / bias to the compute dtype; done one by one that is two tiny kernels per
# layer per step.  precast() converts a whole parameter list with one multi-tensor copy into a flat buffer
# and _lp() serves the views made by the precast() of the CURRENT forward.
_LP_CACHE = {}        # id(param) -> (param._version, dtype, view)
_LP_PTR = {}          # (data_ptr, numel) of a parameter -> the same record (serves reshaped views of it)
_LP_T = {}            # data_ptr of a compute-dtype weight -> (its transposed copy [K,N], weakref(param), dtype)
_LP_FLAT = {}         # (key, dtype) -> (flat buffer, [views], [params])


def precast(params, dtype, key="default", derived=None):
    """One launch per forward: every fp32 master parameter -> the compute dtype (flat buffer, served by _lp()), and the
    transposed copy W^T of every matrix (served by _wt_t(): the weight of the input-gradient GEMM dx = dy W), by
    gf_multi_cast_transpose.  Always re-done: a fused / capturable optimiser step, ``param.data = ...`` or a replayed
    hipGraph change the values without bumping the version counter (measured: stale bf16 weights in an eval forward
    after fused Adam steps), so skipping it "when nothing moved" is not safe.

    derived: [(name, [(src_param, perm | None, rscale | None, scale[, cperm]), ...]), ...] -- prepared weights built from row blocks
    of parameters (rows gathered by the int32 vector `perm` -- columns by `cperm` --, scaled per row by the fp32 vector `rscale` and by the float
    `scale`, in fp32 before the single rounding), written by the SAME launch: matrices in the compute dtype with their
    transposed copy, vectors (biases) in fp32.  derived_weight(key, name) hands them to the linears."""
    params = [p_ for p_ in params if p_.is_cuda and p_.dtype == torch.float32]
    if not params or dtype not in (torch.bfloat16, torch.float32):
        return
    derived = derived or []
    assert not derived or dtype != torch.float32, "derived weights are a compute-dtype (cast) feature"
    # (name, "fold", W0, b0, Wo, bo, c0, cperm): the FOLDED weight [W0[:, :c0] | W0[:, c0:] Wo[:, cperm]] and bias
    # b0 + W0[:, c0:] bo of a linear that consumes cat[x, Wo ctx + bo] (csrc/fold.hip; served by folded_linear())
    folds = [d_ for d_ in derived if len(d_) > 2 and d_[1] == "fold"]
    derived = [d_ for d_ in derived if not (len(d_) > 2 and d_[1] == "fold")]
    dkey = (tuple((name, tuple(id(b_[0]) for b_ in blocks)) for name, blocks in derived)
            + tuple((f[0], "fold", id(f[2]), id(f[4]), f[6]) for f in folds))
    slot = _LP_FLAT.get((key, dtype))
    ptrs = tuple(p_.data_ptr() for p_ in params)          # (`p.data = ...` / module.to() move the storage under the same object)
    if (slot is None or len(slot["params"]) != len(params) or any(a is not b for a, b in zip(slot["params"], params))
            or slot["dkey"] != dkey or slot["ptrs"] != ptrs):
        import struct
        cast = dtype != torch.float32
        dev = params[0].device
        al = lambda n_: (n_ + 7) // 8 * 8                      # noqa: E731  (16-byte aligned views)
        sizes = [al(p_.numel()) for p_ in params]
        dmat = [(name, blocks) for name, blocks in derived if blocks[0][0].dim() >= 2]
        dvec = [(name, blocks) for name, blocks in derived if blocks[0][0].dim() < 2]
        dsize = lambda blocks: sum(b_[0].numel() for b_ in blocks)   # noqa: E731
        fgeo = []                                               # (R, ldw0, K, N, c0) of every folded linear
        for f in folds:
            w0, wo, c0 = f[2], f[4], int(f[6])
            R, K = w0.shape[0], wo.shape[0]
            ldw0, N = w0.numel() // R, wo.numel() // K
            assert w0.is_contiguous() and wo.is_contiguous() and ldw0 == c0 + K, "fold: W0 must be [R, c0 + K], Wo [K, N]"
            fgeo.append((R, ldw0, K, N, c0))
        fsize = sum(al(R * (c0 + N)) for R, _, _, N, c0 in fgeo)
        flat = torch.empty((sum(sizes) if cast else 0) + sum(al(dsize(bl)) for _, bl in dmat) + fsize, dtype=dtype, device=dev)
        mats = [p_ for p_ in params if p_.dim() >= 2]
        flat_t = torch.empty(sum(al(p_.numel()) for p_ in mats) + sum(al(dsize(bl)) for _, bl in dmat) + fsize, dtype=dtype, device=dev)
        fold32 = torch.empty(sum(al(R * N) + al(R) for R, _, _, N, _ in fgeo), dtype=torch.float32, device=dev)
        flat32 = torch.empty(sum(al(dsize(bl)) for _, bl in dvec), dtype=torch.float32, device=dev)
        views, tviews, rec, off, toff, tile0 = [], {}, b"", 0, 0, 0

        def entry(src, dst, dst_t, rows, cols, perm=None, rscale=None, scale=1.0, ldt=None, flags=0, cperm=None, lds=0, ldd=0):
            nonlocal rec, tile0
            tx = (cols + 31) // 32
            rec += struct.pack("<QQQiiiiQQfiiiQii", src, dst, dst_t, rows, cols, tile0, tx, 0 if perm is None else perm.data_ptr(),
                               0 if rscale is None else rscale.data_ptr(), float(scale), rows if ldt is None else ldt, flags, ldd,
                               0 if cperm is None else cperm.data_ptr(), lds, 0)
            tile0 += tx * ((rows + 31) // 32)

        for p_, sz in zip(params, sizes):
            v = flat[off:off + p_.numel()].view(p_.shape) if cast else p_
            off += sz if cast else 0
            views.append(v)
            rows = p_.shape[0] if p_.dim() >= 2 else 1
            cols = p_.numel() // rows
            vt = None
            if p_.dim() >= 2:
                vt = flat_t[toff:toff + p_.numel()].view(cols, rows)
                toff += al(p_.numel())
                tviews[id(p_)] = vt
            elif not cast:
                continue                                       # fp32 vector: nothing to do
            entry(p_.data_ptr(), v.data_ptr() if cast else 0, 0 if vt is None else vt.data_ptr(), rows, cols)
        dslot, off32, keep = {}, 0, []
        for name, blocks in derived:
            mat = blocks[0][0].dim() >= 2
            cols = blocks[0][0].numel() // blocks[0][0].shape[0]      # (a Conv1d(k=1) weight [O, I, 1] is the matrix [O, I])
            rows_all = sum(b_[0].shape[0] for b_ in blocks)
            if mat:
                v = flat[off:off + rows_all * cols].view(rows_all, cols)
                vt = flat_t[toff:toff + rows_all * cols].view(cols, rows_all)
                off += al(rows_all * cols)
                toff += al(rows_all * cols)
                handle = torch.empty((rows_all, cols), dtype=torch.float32, device=dev)    # never written: _lp() maps it to v
            else:
                v = flat32[off32:off32 + rows_all]
                off32 += al(rows_all)
                vt, handle = None, v
            r0, meta = 0, []
            for blk in blocks:
                src, perm, rscale, scale = blk[:4]
                cperm = blk[4] if len(blk) > 4 else None              # optional column gather (SuperGlue's merge weight)
                assert src.is_contiguous() and src.dtype == torch.float32 and src.numel() == src.shape[0] * cols
                rows = src.shape[0]
                perm = None if perm is None else perm.to(device=dev, dtype=torch.int32).contiguous()
                cperm = None if cperm is None else cperm.to(device=dev, dtype=torch.int32).contiguous()
                rscale = None if rscale is None else rscale.to(device=dev, dtype=torch.float32).contiguous()
                keep += [perm, rscale, cperm]
                esz = v.element_size()
                entry(src.data_ptr(), v.data_ptr() + r0 * cols * esz, 0 if vt is None else vt.data_ptr() + r0 * vt.element_size(),
                      rows, cols, perm, rscale, scale, rows_all, 0 if mat else 1, cperm)
                meta.append((r0, rows, perm, rscale, float(scale), tuple(src.shape), cperm))
                r0 += rows
            dslot[name] = {"view": v, "view_t": vt, "handle": handle, "meta": meta, "cols": cols}
        # folded linears: fp32 products by gf_fold_linear_fwd (its own table, launched first), stacked / cast / transposed by
        # two column-block entries of the cast table
        frec, ftile0, foff = b"", 0, 0
        for f, (R, ldw0, K, N, c0) in zip(folds, fgeo):
            name, _, w0, b0, wo, bo, _, cperm = f
            cperm = None if cperm is None else cperm.to(device=dev, dtype=torch.int32).contiguous()
            keep.append(cperm)
            wc = fold32[foff:foff + R * N].view(R, N)
            foff += al(R * N)
            bc = fold32[foff:foff + R] if (b0 is not None or bo is not None) else None
            foff += al(R)
            wid = c0 + N
            v = flat[off:off + R * wid].view(R, wid)
            vt = flat_t[toff:toff + R * wid].view(wid, R)
            off += al(R * wid)
            toff += al(R * wid)
            esz_ = v.element_size()
            entry(w0.data_ptr(), v.data_ptr(), vt.data_ptr(), R, c0, ldt=R, lds=ldw0, ldd=wid)
            entry(wc.data_ptr(), v.data_ptr() + c0 * esz_, vt.data_ptr() + c0 * R * esz_, R, N, ldt=R, ldd=wid)
            ftx = (N + 63) // 64
            frec += struct.pack("<QQQQQQQiiiiiiii", w0.data_ptr(), wo.data_ptr(), 0 if b0 is None else b0.data_ptr(),
                                0 if bo is None else bo.data_ptr(), 0 if cperm is None else cperm.data_ptr(), wc.data_ptr(),
                                0 if bc is None else bc.data_ptr(), R, K, N, ldw0, c0, ftile0, ftx, 0)
            ftile0 += (ftx + 1) * ((R + 63) // 64)           # (+ 1: the bias tile column of every row block)
            handle = torch.empty((R, wid), dtype=torch.float32, device=dev)      # never written: _lp() maps it to v
            dslot[name] = {"view": v, "view_t": vt, "handle": handle, "bias": bc, "fold": (R, ldw0, K, N, c0), "cperm": cperm}
        esz = _lib.load().gf_cast_entry_bytes()
        assert len(rec) % esz == 0 and esz == 88
        table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev) if rec else None
        ftable = None
        if frec:
            assert len(frec) == len(folds) * _lib.load().gf_fold_entry_bytes()
            ftable = torch.frombuffer(bytearray(frec), dtype=torch.uint8).to(dev)
        slot = {"flat": flat, "flat_t": flat_t, "flat32": flat32, "views": views, "tviews": tviews, "params": params, "table": table,
                "n": len(rec) // esz, "tiles": tile0, "dkey": dkey, "derived": dslot, "keep": keep, "ptrs": ptrs,
                "fold32": fold32, "ftable": ftable, "fn": len(folds), "ftiles": ftile0}
        _LP_FLAT[(key, dtype)] = slot
    if slot["ftable"] is not None:
        _lib.check(_lib.load().gf_fold_linear_fwd(_p(slot["ftable"]), slot["fn"], slot["ftiles"], _stream()), "gf_fold_linear_fwd")
    if slot["table"] is not None:
        _lib.check(_lib.load().gf_multi_cast_transpose(_p(slot["table"]), slot["n"], slot["tiles"], BF16 if dtype == torch.bfloat16 else F32, _stream()),
                   "gf_multi_cast_transpose")
    for p_, v in zip(params, slot["views"]):
        if v is not p_:
            _LP_CACHE[id(p_)] = (p_._version, dtype, v, weakref.ref(p_))
            _LP_PTR[(p_.data_ptr(), p_.numel())] = (p_._version, dtype, v, weakref.ref(p_))   # views (conv weight.squeeze(-1))
        vt = slot["tviews"].get(id(p_))
        if vt is not None:
            _LP_T[v.data_ptr()] = (vt, weakref.ref(p_), dtype)
    for d in slot["derived"].values():
        h, v = d["handle"], d["view"]
        if d["view_t"] is not None:             # matrices: the fp32 handle stands for the compute-dtype view
            _LP_PTR[(h.data_ptr(), h.numel())] = (h._version, dtype, v, weakref.ref(h))
            _LP_T[v.data_ptr()] = (d["view_t"], weakref.ref(h), dtype)


class _DerivedWeight(torch.autograd.Function):
    """The autograd face of a derived weight (precast(derived=...)): forward hands out the prepared tensor -- for a matrix
    an fp32 HANDLE that _lp() / _wt_t() resolve to the compute-dtype copy and its transpose written by this forward's
    precast launch, for a vector the fp32 values themselves --, backward sends the gradient of each row block back to its
    source parameter (gf_weight_grad_map: un-gather, scales)."""

    @staticmethod
    def forward(ctx, d, *srcs):
        ctx.d = d
        return d["handle"].view(d["handle"].shape)          # a fresh alias: the cached tensor keeps no autograd state

    @staticmethod
    def backward(ctx, g):
        d = ctx.d
        g = g.float().contiguous()
        outs = []
        for i, (r0, rows, perm, rscale, scale, shape, cperm) in enumerate(d["meta"]):
            if not ctx.needs_input_grad[1 + i]:
                outs.append(None)
                continue
            gi = g[r0:r0 + rows]
            if perm is None and rscale is None and scale == 1.0 and cperm is None:
                outs.append(gi.reshape(shape))
                continue
            out = torch.empty(shape, dtype=torch.float32, device=g.device)
            _lib.check(_lib.load().gf_weight_grad_map(_p(gi), _p(out), None if perm is None else _p(perm),
                                                      None if cperm is None else _p(cperm),
                                                      None if rscale is None else _p(rscale), scale, rows, d["cols"], _stream()),
                       "gf_weight_grad_map")
            outs.append(out)
        return (None, *outs)


def derived_weight(key, dtype, name, *srcs):
    """The prepared weight `name` of this forward's precast(key=..., derived=...) launch, differentiable w.r.t. the source
    parameters of its row blocks (passed again here, in block order, so autograd sees them), or None when this forward
    did not precast it (fp32 parity mode: the caller builds the weight with torch ops)."""
    slot = _LP_FLAT.get((key, dtype))
    d = None if slot is None else slot["derived"].get(name)
    if d is None:
        return None
    return _DerivedWeight.apply(d, *srcs)


class _FoldedLinear(torch.autograd.Function):
    """The autograd face of a folded linear (precast(derived=[(name, "fold", ...)]), csrc/fold.hip): forward hands out the
    fp32 HANDLE of the stacked weight [W0a | W0b Wo] (resolved by _lp() / _wt_t() to the compute-dtype copy and its transpose
    this forward's precast launch wrote) and the folded bias b0 + W0b bo (fp32 values); backward turns their gradients into
    those of W0, b0, Wo, bo with ONE launch (gf_fold_linear_bwd)."""

    @staticmethod
    def forward(ctx, d, w0, b0, wo, bo):
        ctx.d = d
        ctx.save_for_backward(w0, wo, bo)
        ctx.has = (b0 is not None, bo is not None)
        bias = d["bias"]
        return d["handle"].view(d["handle"].shape), (None if bias is None else bias.view(bias.shape))

    @staticmethod
    def backward(ctx, gw, gb):
        w0, wo, bo = ctx.saved_tensors
        R, ldw0, K, N, c0 = ctx.d["fold"]
        if gw is None:
            gw = torch.zeros((R, c0 + N), dtype=torch.float32, device=w0.device)
        gw = gw.float().contiguous()
        gb = None if gb is None else gb.float().contiguous()
        dw0, dwo = torch.empty_like(w0), torch.empty_like(wo)
        dbo = torch.empty_like(bo) if ctx.has[1] else None
        cperm = ctx.d["cperm"]
        _lib.check(_lib.load().gf_fold_linear_bwd(_p(gw), _p(gb), _p(w0), _p(wo), _p(bo) if ctx.has[1] else None,
                                                  None if cperm is None else _p(cperm), _p(dw0), _p(dwo), _p(dbo),
                                                  R, K, N, ldw0, c0, _stream()), "gf_fold_linear_bwd")
        if dbo is not None and gb is None:
            dbo = None
        return None, dw0, (gb if ctx.has[0] else None), dwo, dbo


FOLD_ENABLED = True      # tools/probe/ab_matcher.py switches it off for a same-process A/B of the folded blocks


def folded_linear(key, dtype, name, w0, b0, wo, bo):
    """(weight handle, bias) of the folded linear `name` of this forward's precast(key=..., derived=...) launch --
    y = linear_cat(x, ctx, weight, bias) then equals W0 cat[x, Wo ctx + bo] + b0 -- differentiable w.r.t. the four
    parameters, or None when this forward did not prepare it (fp32 parity mode: the caller runs the two linears)."""
    slot = _LP_FLAT.get((key, dtype)) if FOLD_ENABLED else None
    d = None if slot is None else slot["derived"].get(name)
    if d is None or "fold" not in d:
        return None
    return _FoldedLinear.apply(d, w0, b0, wo, bo)


def invalidate_precast():
    """Forget the per-parameter cache entries (the flat buffers stay): needed when parameters change without a
    version bump, e.g. after a captured optimiser step is replayed from a hipGraph."""
    _LP_CACHE.clear()
    _LP_PTR.clear()
    _LP_T.clear()


def _lp(t, dtype):
    """t in `dtype`: the precast copy when it is current, else a fresh cast."""
    if t is None or t.dtype == dtype:
        return t
    hit = _LP_CACHE.get(id(t))
    if hit is not None and hit[3]() is t and hit[0] == t._version and hit[1] == dtype:
        return hit[2]
    if t.is_contiguous():          # a reshaped view of a precast parameter (e.g. a Conv1d weight without its kernel axis)
        hit = _LP_PTR.get((t.data_ptr(), t.numel()))
        if hit is not None and hit[3]() is not None and hit[0] == t._version and hit[1] == dtype:
            return hit[2].view(t.shape)
    return t.to(dtype)


# ---- GEMMs of the linear layers: the hand-written weight-streaming kernel gf_gemm (csrc/gemm_ws.hip) is the path of
# every forward and input-gradient GEMM it supports (N % 32 == 0, K a power of two in [32, 512] -- [32, 256] in fp32 --
# or a sum of such pieces, which are accumulated through the fused residual input).  Anything else (3- or 5-channel
# encoder inputs, 1-channel heads) is a tiny library call.
_GEMM_KMAX = {torch.bfloat16: 512, torch.float32: 256}


def _k_pieces(k, dtype):
    """Split K into power-of-two pieces gf_gemm takes (largest first); None if impossible."""
    kmax = _GEMM_KMAX.get(dtype)
    if kmax is None or k < 32:
        return None
    out, rest = [], k
    while rest:
        piece = min(kmax, 1 << (rest.bit_length() - 1))
        if piece < 32:
            return None
        out.append(piece)
        rest -= piece
    return out


def _row_ok(t, dtype):
    al = 8 if dtype == torch.bfloat16 else 4
    return t.stride(1) == 1 and t.stride(0) % al == 0 and t.data_ptr() % 16 == 0


def gemm(x2, wt, bias=None, res2=None, out=None, x2b=None, cs=None, rot_n=0, res3=None):
    """y [M,N] = [x2 | x2b] wt^T (+ bias) (+ res2) (+ res3), optional rotary epilogue; x2 / x2b / wt / res2 / res3 2-D with unit
    inner stride, all in the compute dtype (bias: any float dtype).  ``out`` may alias ``res2``.  Returns y.
    res3: a second residual, fused on the streamed kernel's shapes (gf_gemm_res2), added beforehand otherwise."""
    M, K0 = x2.shape
    N = wt.shape[0]
    dtype = x2.dtype
    K1 = 0 if x2b is None else x2b.shape[1]
    if res3 is not None:
        if res2 is None:
            res2, res3 = res3, None
        else:
            if res3.dim() != 2:
                res3 = res3.reshape(M, N)
            fused = (dtype == torch.bfloat16 and cs is None and x2.is_cuda and wt.dtype == dtype and (K0 + K1) in (256, 512)
                     and (K1 == 0 or K1 == K0) and N % 256 == 0 and M % 64 == 0 and res2.dtype == dtype and res3.dtype == dtype
                     and all(_row_ok(t_, dtype) for t_ in (x2, wt, res2, res3) + ((x2b,) if x2b is not None else ())
                             + ((out,) if out is not None else ())))
            if fused:
                y = torch.empty((M, N), dtype=dtype, device=x2.device) if out is None else out
                b32 = None if bias is None else bias.detach().float().contiguous()
                rc = _lib.load().gf_gemm_res2(_p(x2), _p(x2b), _p(wt), _p(b32), _p(res2), _p(res3), _p(y), M, N, K0, K1,
                                              x2.stride(0), 0 if x2b is None else x2b.stride(0), wt.stride(0), res2.stride(0),
                                              res3.stride(0), y.stride(0), _dt(x2), _stream())
                if rc == 0:
                    return y
                if rc != -1:
                    _lib.check(rc, "gf_gemm_res2")
            res2, res3 = res2 + res3, None
    ok = (x2.is_cuda and dtype in _GEMM_KMAX and wt.dtype == dtype and N % 32 == 0 and _row_ok(x2, dtype)
          and _row_ok(wt, dtype) and (x2b is None or (x2b.dtype == dtype and _row_ok(x2b, dtype)))
          and (res2 is None or (res2.dtype == dtype and _row_ok(res2, dtype)))
          and (out is None or _row_ok(out, dtype)))
    plan = None
    if ok:
        if x2b is not None and K1 == K0 and _k_pieces(K0 + K1, dtype) == [K0 + K1]:
            plan = "two"
        else:
            p0 = _k_pieces(K0, dtype)
            p1 = _k_pieces(K1, dtype) if x2b is not None else []
            if p0 is not None and p1 is not None:
                plan = "pieces"
    if plan is None:                                  # library fallback for the odd shapes: counted and reported once per shape
        _note_library_gemm(M, N, K0 + K1, dtype)
        xx = x2 if x2b is None else torch.cat([x2, x2b], 1)
        y = torch.nn.functional.linear(xx, wt, None if bias is None else _lp(bias, dtype))
        if res2 is not None:
            y = y + res2
        if cs is not None:
            raise RuntimeError("rotary epilogue needs the gf_gemm path")
        if out is not None:
            out.copy_(y)
            return out
        return y
    L = _lib.load()
    y = torch.empty((M, N), dtype=dtype, device=x2.device) if out is None else out
    b32 = None if bias is None else bias.detach().float().contiguous()
    st = _stream()
    dt = _dt(x2)
    if plan == "two":
        _lib.check(L.gf_gemm(_p(x2), _p(x2b), _p(wt), _p(b32), _p(res2), _p(y), _p(cs), rot_n, M, N, K0, K1,
                             x2.stride(0), x2b.stride(0), wt.stride(0), 0 if res2 is None else res2.stride(0),
                             y.stride(0), dt, st), "gf_gemm")
        return y
    # K pieces: the first call carries bias / residual, the others accumulate into y through the residual input;
    # a rotary epilogue rides on the last one
    pieces = [(x2, k0, n) for k0, n in _offsets(_k_pieces(K0, dtype))]
    if x2b is not None:
        pieces += [(x2b, k0, n) for k0, n in _offsets(_k_pieces(K1, dtype))]
    wofs = 0
    for i, (src, k0, n) in enumerate(pieces):
        first, last = i == 0, i == len(pieces) - 1
        xs = src[:, k0:k0 + n]
        ws = wt[:, wofs:wofs + n]
        r = res2 if first else y
        _lib.check(L.gf_gemm(_p(xs), None, _p(ws), _p(b32) if first else None, _p(r), _p(y),
                             _p(cs) if last else None, rot_n if last else 0, M, N, n, 0,
                             xs.stride(0), 0, ws.stride(0), 0 if r is None else r.stride(0), y.stride(0), dt, st),
                   "gf_gemm")
        wofs += n
    return y


LIBRARY_GEMMS = {}       # (M, N, K, dtype) -> number of products that left the hand-written path (ops.gemm's fallback)


def _note_library_gemm(M, N, K, dtype):
    """A product outside gf_gemm's plans (N % 32, K not a sum of powers of two >= 32, misaligned rows, fp32 K > 256 pieces ...)
    runs on the vendor library.  Correct, but not the path the rooflines describe: say so once per shape instead of silently."""
    key = (int(M), int(N), int(K), str(dtype))
    n = LIBRARY_GEMMS.get(key, 0)
    LIBRARY_GEMMS[key] = n + 1
    if n == 0:
        import warnings
        warnings.warn(f"glue_factory_amd.ops.gemm: [{M} x {K}] x [{N} x {K}]^T in {dtype} is outside gf_gemm's plans and runs on the "
                      "vendor library (ops.LIBRARY_GEMMS counts these calls)", RuntimeWarning, stacklevel=3)


def gemm_takes(k, n, dtype):
    """True when a [*, k] x [n, k]^T product runs as ONE gf_gemm launch (needed for its rotary epilogue)."""
    return n % 32 == 0 and _k_pieces(k, dtype) == [k]


def _offsets(sizes):
    out, o = [], 0
    for n in sizes:
        out.append((o, n))
        o += n
    return out


def _linear_fwd(x2, wt, bias, res2=None, out=None, cs=None, rot_n=0):
    """y [M,N] = x2 [M,K] wt[N,K]^T + bias (+ res2).  bias: the fp32 master."""
    return gemm(x2, wt, bias, res2, out, cs=cs, rot_n=rot_n)


def _wt_t(wt, k0=None, k1=None):
    """Columns [k0, k1) of the [N,K] compute-dtype weight, transposed and contiguous ([k1-k0, N]: the "weight" of the
    input-gradient GEMM dx = dy W).  Served from this forward's precast() when there is one (no kernel), else copied."""
    hit = _LP_T.get(wt.data_ptr())
    n = wt.shape[0]
    kk = wt.numel() // n
    if hit is not None and hit[1]() is not None and hit[2] == wt.dtype and hit[0].shape == (kk, n):
        wt_t = hit[0]
        return wt_t if k0 is None else wt_t[k0:k1]
    w2 = wt.reshape(n, kk)
    return (w2 if k0 is None else w2[:, k0:k1]).t().contiguous()


class GradChain:
    """Sums the gradient contributions of ONE tensor that feeds several linears of a block (residual, FFN input,
    projection) inside the GEMM epilogues instead of leaving them to autograd's accumulation (one 3-pass add kernel
    per extra consumer): every consumer but the designated last one parks its contribution here and returns no
    gradient; the last one's input-gradient GEMM adds the parked sum in its residual epilogue and returns the total.
    The last consumer must be the one whose backward runs last -- true by data dependence for the transformer blocks
    (the projection's gradient needs the attention backward, which needs the FFN's) and checked by the counter."""
    __slots__ = ("acc", "got", "expected", "closed", "extra")

    def __init__(self, consumers):
        self.acc, self.got, self.expected, self.closed = None, 0, consumers - 1, False
        # a second parked tensor that has not been added to `acc` yet: it rides as the SECOND residual of the next link's
        # GEMM (gf_gemm_res2) -- the place where a loss head's gradient meets the block's residual gradient
        self.extra = None

    def pop_extra(self):
        e, self.extra = self.extra, None
        return e

    def park(self, g, counted=True):
        """counted=False: an OPTIONAL contribution (the per-layer loss heads of LightGlue, which exist only when the
        fused loss is evaluated and whose backward always runs before the blocks': they were created later)."""
        if self.closed:
            raise RuntimeError("GradChain: a contribution arrived after the last consumer closed the chain (it would be lost)")
        self.acc = g
        if counted:
            self.got += 1

    def take(self):
        if self.got != self.expected:
            raise RuntimeError(f"GradChain: {self.got} of {self.expected} contributions arrived before the last consumer")
        acc, self.acc, self.got, self.closed = self.acc, None, 0, True
        extra = self.pop_extra()
        if extra is not None:           # no link in between took it along: one explicit add after all
            acc = extra if acc is None else acc + extra
        return acc


def _flat2(t, n):
    t2 = t.reshape(-1, n)
    return t2 if t2.is_contiguous() else t2.contiguous()


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ res) (+ rotary epilogue): forward and input-gradient GEMM on gf_gemm (library GEMM only for
    shapes outside its plans), weight / bias gradient (a tiny-output, 1e5-deep reduction) on gf_linear_dw, which
    returns fp32 gradients for the fp32 master parameters directly.  ``res`` is a fused residual (gradient = dy)."""

    @staticmethod
    def forward(ctx, x, w, b, res=None, cs=None, rot_n=0, chain=None, chain_last=False, res_chain=None, out=None):
        ctx.chain, ctx.chain_last, ctx.res_chain = chain, chain_last, res_chain
        wt = _lp(w, x.dtype)
        k = x.shape[-1]
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        res2 = None
        if res is not None:
            res2 = res.reshape(-1, wt.shape[0])
            if not res2.is_contiguous():
                res2 = res2.contiguous()
        out2 = None if out is None else out.view(-1, wt.shape[0])       # caller-owned destination (contiguous rows)
        y = _linear_fwd(x2, wt, b, res2, out=out2, cs=cs, rot_n=rot_n).view(*x.shape[:-1], wt.shape[0])
        ctx.save_for_backward(x, wt)
        ctx.wdtype = w.dtype
        ctx.has_bias = b is not None
        ctx.bdtype = None if b is None else b.dtype
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        nout, k = wt.shape
        dy2 = dy.reshape(-1, nout)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        dres = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        if dres is not None and ctx.res_chain is not None:       # first link of the residual tensor's chain
            d2 = _flat2(dres, nout)
            rc = ctx.res_chain
            if rc.acc is not None:      # optional contributions got here first (the previous output's loss heads): they wait
                if rc.extra is not None:                     # in `extra` for the next link's two-residual GEMM
                    d2 = d2 + rc.pop_extra()
                rc.extra = rc.acc
            rc.park(d2)
            dres = None
        if ctx.needs_input_grad[0]:
            ch = ctx.chain
            if ch is None:
                dx = gemm(dy2, _wt_t(wt)).view(x.shape)
            elif ctx.chain_last is True:
                dx = gemm(dy2, _wt_t(wt), res2=ch.take()).view(x.shape)
            else:                                       # chain_last == "extra": an optional, uncounted contribution
                ch.park(gemm(dy2, _wt_t(wt), res2=ch.acc, res3=ch.pop_extra()), counted=ctx.chain_last != "extra")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            x2 = x.reshape(-1, k)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            m = x2.shape[0]
            L = _lib.load()
            ws = torch.empty(int(L.gf_linear_dw_ws_bytes(m, nout, k)), dtype=torch.uint8, device=x.device)
            dw32 = torch.empty((nout, k), dtype=torch.float32, device=x.device)
            db32 = torch.empty((nout,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
            _lib.check(L.gf_linear_dw(_p(dy2), _p(x2), _p(dw32), _p(db32), _p(ws), m, nout, k, _dt(x2),
                                      _stream()), "gf_linear_dw")
            dw = dw32.to(ctx.wdtype)
            db = None if db32 is None else db32.to(ctx.bdtype)
        return dx, dw, db, dres, None, None, None, None, None, None


def linear(x, w, b=None, res=None, rotary_cs=None, rot_n=0, chain=None, chain_last=False, res_chain=None, out=None):
    """w, b: fp32 master parameters (or differentiable functions of them); x (and the optional fused residual
    ``res``, same shape as the output) in the compute dtype.  ``rotary_cs`` [.., 64] fp32 interleaved (cos, sin):
    the output channels [0, rot_n) leave the GEMM already rotated (the buffer then belongs to
    self_attention_rotary(pre_rotated=True), whose backward hands the UN-rotated gradient back to this node).
    ``chain`` / ``res_chain``: GradChain of x / of res (see there); only used when that tensor requires grad.
    ``out``: optional destination with the output's shape (e.g. a slice of a per-layer buffer): written, and returned."""
    _chk(x)
    if rotary_cs is not None:
        rotary_cs = rotary_cs.reshape(-1, rotary_cs.shape[-1])
        assert rotary_cs.shape[-1] == 64 and rotary_cs.dtype == torch.float32 and rotary_cs.is_contiguous()
    if not x.requires_grad:
        chain = None
    if res is None or not res.requires_grad:
        res_chain = None
    return _Linear.apply(x, w, b, res, rotary_cs, rot_n, chain, chain_last, res_chain, out)


def _dw(dy2, x2, nout, k, with_bias):
    L = _lib.load()
    m = x2.shape[0]
    ws = torch.empty(int(L.gf_linear_dw_ws_bytes(m, nout, k)), dtype=torch.uint8, device=x2.device)
    dw32 = torch.empty((nout, k), dtype=torch.float32, device=x2.device)
    db32 = torch.empty((nout,), dtype=torch.float32, device=x2.device) if with_bias else None
    _lib.check(L.gf_linear_dw(_p(dy2), _p(x2), _p(dw32), _p(db32), _p(ws), m, nout, k, _dt(x2), _stream()),
               "gf_linear_dw")
    return dw32, db32


class _LinearCat(torch.autograd.Function):
    """y = [x1 | x2] W^T + b without building the concatenation: two accumulating GEMMs forward, two
    input-gradient GEMMs and two gf_linear_dw calls backward (the FFN input cat[x, message] of
    lightglue.py:163 / :219-220)."""

    @staticmethod
    def forward(ctx, x1, x2, w, b, chain1=None):
        ctx.chain1 = chain1
        k1 = x1.shape[-1]
        wt = _lp(w, x1.dtype)
        a2 = x1.reshape(-1, k1)
        c2 = x2.reshape(-1, x2.shape[-1])
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        c2 = c2 if c2.is_contiguous() else c2.contiguous()
        y = gemm(a2, wt, b, x2b=c2).view(*x1.shape[:-1], wt.shape[0])
        ctx.save_for_backward(x1, x2, wt)
        ctx.wdtype = w.dtype
        ctx.bdtype = None if b is None else b.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, wt = ctx.saved_tensors
        nout, k = wt.shape
        k1 = x1.shape[-1]
        dy2 = dy.reshape(-1, nout)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx1 = None
        if ctx.needs_input_grad[0]:
            ch = ctx.chain1
            if ch is None:
                dx1 = gemm(dy2, _wt_t(wt, 0, k1)).view(x1.shape)
            else:                                   # a middle link: the parked residual gradient rides in the epilogue
                ch.park(gemm(dy2, _wt_t(wt, 0, k1), res2=ch.acc, res3=ch.pop_extra()))
        dx2 = gemm(dy2, _wt_t(wt, k1, wt.shape[1])).view(x2.shape) if ctx.needs_input_grad[1] else None
        dw = db = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            a = x1.reshape(-1, k1)
            c = x2.reshape(-1, k - k1)
            a = a if a.is_contiguous() else a.contiguous()
            c = c if c.is_contiguous() else c.contiguous()
            if a.dtype == torch.bfloat16 and nout % 128 == 0 and k1 % 128 == 0 and (k - k1) % 128 == 0:
                # ONE launch over the virtual concatenation (gf_linear_dw2): dY streamed once, dw comes out whole
                L = _lib.load()
                m = a.shape[0]
                ws = torch.empty(int(L.gf_linear_dw_ws_bytes(m, nout, k)), dtype=torch.uint8, device=a.device)
                dw32 = torch.empty((nout, k), dtype=torch.float32, device=a.device)
                db32 = torch.empty((nout,), dtype=torch.float32, device=a.device) if ctx.bdtype is not None else None
                _lib.check(L.gf_linear_dw2(_p(dy2), _p(a), _p(c), k1, _p(dw32), _p(db32), _p(ws), m, nout, k, _dt(a),
                                           _stream()), "gf_linear_dw2")
                dw = dw32.to(ctx.wdtype)
            else:
                dwa, db32 = _dw(dy2, a, nout, k1, ctx.bdtype is not None)
                dwb, _ = _dw(dy2, c, nout, k - k1, False)
                dw = torch.cat([dwa, dwb], 1).to(ctx.wdtype)
            db = None if db32 is None else db32.to(ctx.bdtype)
        return dx1, dx2, dw, db, None


def linear_cat(x1, x2, w, b=None, chain1=None):
    _chk(x1, x2)
    return _LinearCat.apply(x1, x2, w, b, chain1 if x1.requires_grad else None)


class _RowDot(torch.autograd.Function):
    """z = x w^T + b for a single output channel (w [1,C], b [1]); returns fp32 [..]."""

    @staticmethod
    def forward(ctx, x, w, b, chain=None, counted=True):
        ctx.chain, ctx.counted = chain, counted
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        w32 = w.reshape(-1).float().contiguous()
        z = torch.empty(M, dtype=torch.float32, device=x.device)
        # the bias stays on the device (no .item() sync): the kernel reads the parameter itself
        b32 = None if b is None else b.detach().reshape(-1).float()
        _lib.check(_lib.load().gf_rowdot_fwd(_p(x2), _p(w32), 0.0, _p(b32), _p(z), M, C, _dt(x2), _stream()), "gf_rowdot_fwd")
        ctx.save_for_backward(x2, w32)
        ctx.meta = (x.shape, w.shape, w.dtype, None if b is None else b.dtype)
        return z.view(x.shape[:-1])

    @staticmethod
    def backward(ctx, dz):
        x2, w32 = ctx.saved_tensors
        xshape, wshape, wdt, bdt = ctx.meta
        M, C = x2.shape
        L = _lib.load()
        dz = dz.reshape(-1).float().contiguous()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        ch = ctx.chain if dx is not None else None
        base = None if ch is None else ch.acc           # the chain's running sum rides in this kernel (dx = base + dz w)
        if ch is not None and ch.extra is not None:
            e = ch.pop_extra()
            base = e if base is None else base + e
        part = torch.empty((L.gf_rowdot_nblk(M), C + 1), dtype=torch.float32, device=x2.device)
        _lib.check(L.gf_rowdot_bwd(_p(x2), _p(dz), _p(w32), _p(dx), _p(base), _p(part), M, C, _dt(x2), _stream()),
                   "gf_rowdot_bwd")
        s = part.sum(0)
        dw = s[:C].reshape(wshape).to(wdt)
        db = None if bdt is None else s[C:].to(bdt)
        if ch is not None:
            ch.park(dx, counted=ctx.counted)
            dx = None
        return (None if dx is None else dx.view(xshape)), dw, db, None, None


class _RowDot2(torch.autograd.Function):
    """(z0, z1) = (x w0^T + b0, x.detach() w1^T + b1) for two single-output heads on the same rows with ONE read of x
    (gf_rowdot2_*): a LightGlue layer's matchability (differentiable w.r.t. x: the rank-1 term joins x's gradient chain) and
    token-confidence logits (the reference feeds that head desc.detach(), lightglue.py:81-94)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, chain=None, counted=True):
        ctx.chain, ctx.counted = chain, counted
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        w0f, w1f = w0.reshape(-1).float().contiguous(), w1.reshape(-1).float().contiguous()
        z0 = torch.empty(M, dtype=torch.float32, device=x.device)
        z1 = torch.empty(M, dtype=torch.float32, device=x.device)
        b0f = None if b0 is None else b0.detach().reshape(-1).float()
        b1f = None if b1 is None else b1.detach().reshape(-1).float()
        _lib.check(_lib.load().gf_rowdot2_fwd(_p(x2), _p(w0f), _p(w1f), _p(b0f), _p(b1f), _p(z0), _p(z1), M, C, _dt(x2),
                                              _stream()), "gf_rowdot2_fwd")
        ctx.save_for_backward(x2, w0f)
        ctx.meta = (x.shape, w0.shape, w0.dtype, None if b0 is None else b0.dtype, w1.shape, w1.dtype,
                    None if b1 is None else b1.dtype)
        return z0.view(x.shape[:-1]), z1.view(x.shape[:-1])

    @staticmethod
    def backward(ctx, dz0, dz1):
        x2, w0f = ctx.saved_tensors
        xshape, w0shape, w0dt, b0dt, w1shape, w1dt, b1dt = ctx.meta
        M, C = x2.shape
        L = _lib.load()
        dz0 = torch.zeros(M, dtype=torch.float32, device=x2.device) if dz0 is None else dz0.reshape(-1).float().contiguous()
        dz1 = torch.zeros(M, dtype=torch.float32, device=x2.device) if dz1 is None else dz1.reshape(-1).float().contiguous()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        ch = ctx.chain if dx is not None else None
        base = None if ch is None else ch.acc
        if ch is not None and ch.extra is not None:
            e = ch.pop_extra()
            base = e if base is None else base + e
        part = torch.empty((L.gf_rowdot_nblk(M), 2, C + 1), dtype=torch.float32, device=x2.device)
        _lib.check(L.gf_rowdot2_bwd(_p(x2), _p(dz0), _p(dz1), _p(w0f), _p(dx), _p(base), _p(part), M, C, _dt(x2), _stream()),
                   "gf_rowdot2_bwd")
        s = part.sum(0)                                        # [2, C + 1]: ONE reduction for both heads
        dw0 = s[0, :C].reshape(w0shape).to(w0dt)
        db0 = None if b0dt is None else s[0, C:].to(b0dt)
        dw1 = s[1, :C].reshape(w1shape).to(w1dt)
        db1 = None if b1dt is None else s[1, C:].to(b1dt)
        if ch is not None:
            ch.park(dx, counted=ctx.counted)
            dx = None
        return (None if dx is None else dx.view(xshape)), dw0, db0, dw1, db1, None, None


def rowdot2(x, w0, b0, w1, b1, chain=None, counted=True):
    """See _RowDot2.  ``chain``: GradChain of x (the input gradient of head 0 is parked there)."""
    _chk(x)
    if not x.requires_grad:
        chain = None
    return _RowDot2.apply(x, w0, b0, w1, b1, chain, counted)


def rowdot(x, w, b=None, chain=None, counted=True):
    """``chain``: GradChain of x -- the input gradient is parked there (added to the chain's running sum inside the
    kernel) instead of being returned."""
    _chk(x)
    if not x.requires_grad:
        chain = None
    return _RowDot.apply(x, w, b, chain, counted)


# ------------------------------------------------------------------------------ LN + GELU
class _LnGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _chk(x, gamma, beta)
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        R = x2.shape[0]
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        y = torch.empty_like(x2)
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().gf_ln_gelu_fwd(_p(x2), _p(g32), _p(b32), _p(y), _p(mean), _p(rstd),
                                              R, C, float(eps), _dt(x2), _stream()), "gf_ln_gelu_fwd")
        ctx.save_for_backward(x2, g32, b32, mean, rstd)
        ctx.shape = x.shape
        ctx.pdtypes = (gamma.dtype, beta.dtype)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g32, b32, mean, rstd = ctx.saved_tensors
        R, C = x2.shape
        dy2 = dy.reshape(R, C)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        L = _lib.load()
        nblk = L.gf_ln_gelu_nblk(R)
        dx = torch.empty_like(x2)
        part = torch.empty((2, nblk, C), dtype=torch.float32, device=x2.device)        # per-block dgamma | dbeta partials
        _lib.check(L.gf_ln_gelu_bwd(_p(x2), _p(g32), _p(b32), _p(mean), _p(rstd), _p(dy2), _p(dx),
                                    _p(part[0]), _p(part[1]), R, C, _dt(x2), _stream()), "gf_ln_gelu_bwd")
        sums = colsum(part)                                                             # one deterministic reduction
        return (dx.view(ctx.shape), sums[0].to(ctx.pdtypes[0]), sums[1].to(ctx.pdtypes[1]), None)


def colsum(x):
    """out[g, c] = sum_r x[g, r, c] for contiguous fp32 x [G, R, C] (deterministic two-stage reduction)."""
    _chk(x)
    G, R, C = x.shape
    L = _lib.load()
    ws = torch.empty(L.gf_colsum_ws_floats(G, C), dtype=torch.float32, device=x.device)
    out = torch.empty((G, C), dtype=torch.float32, device=x.device)
    _lib.check(L.gf_colsum_f32(_p(x), _p(ws), _p(out), G, R, C, _stream()), "gf_colsum_f32")
    return out


class _SmallLinear(torch.autograd.Function):
    """theta = x W^T for a tall fp32 x [M, K] with K <= 8 input columns (the Fourier positional encoding's Wr,
    lightglue.py:52-65): the forward is the stock product (tiny), the weight gradient a dedicated reduction
    (gf_small_dw) instead of a skinny library GEMM over M = 131072 rows."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        O, K = w.shape
        if not x.is_cuda or K > 8 or x.dtype != torch.float32 or w.dtype != torch.float32:
            return torch.nn.functional.linear(x, w)
        x2 = x.reshape(-1, K).contiguous()
        y = torch.empty((x2.shape[0], O), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().gf_small_fwd(_p(x2), _p(w.contiguous()), _p(y), x2.shape[0], O, K, _stream()), "gf_small_fwd")
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        O, K = w.shape
        g2, x2 = g.reshape(-1, O).float().contiguous(), x.reshape(-1, K).float().contiguous()
        dx = g @ w if ctx.needs_input_grad[0] else None       # (keypoints / scores carry no gradient on the train path)
        L = _lib.load()
        if not g2.is_cuda or K > 8 or O * K > 256:
            _note_library_gemm(g2.shape[0], O, K, "fp32 (small_linear weight gradient)")
            return dx, (g2.t() @ x2).to(w.dtype)
        ws = torch.empty(L.gf_small_dw_ws_floats(O, K), dtype=torch.float32, device=g2.device)
        dw = torch.empty((O, K), dtype=torch.float32, device=g2.device)
        _lib.check(L.gf_small_dw(_p(g2), _p(x2), _p(ws), _p(dw), g2.shape[0], O, K, _stream()), "gf_small_dw")
        return dx, dw.to(w.dtype)


def small_linear(x, w):
    return _SmallLinear.apply(x, w)


def ln_gelu(x, gamma, beta, eps=1e-5):
    return _LnGelu.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------ assignment head
def _mat3(t):
    assert t.dim() == 3
    return t if t.is_contiguous() else t.contiguous()


def rows_lse(a, b, colbias=None):
    """lse[b,i] = log sum_j exp(a_i . b_j + colbias_j); no autograd (see dual_lse)."""
    _chk(a, b, colbias)
    a, b = _mat3(a), _mat3(b)
    B, M, D = a.shape
    N = b.shape[1]
    out = torch.empty((B, M), dtype=torch.float32, device=a.device)
    cb = None if colbias is None else colbias.float().contiguous()
    _lib.check(_lib.load().gf_rows_lse(_p(a), _p(b), _p(cb), _p(out), B, M, N, D, _dt(a), _stream()),
               "gf_rows_lse")
    return out


@torch.no_grad()
def rows_argmax(a, b, colbias=None, alpha=1.0):
    """max_j / argmax_j of alpha * a_i . b_j + colbias_j  ->  ([B,M] float, [B,M] int64)."""
    _chk(a, b, colbias)
    a, b = _mat3(a), _mat3(b)
    B, M, D = a.shape
    N = b.shape[1]
    vmax = torch.empty((B, M), dtype=torch.float32, device=a.device)
    arg = torch.empty((B, M), dtype=torch.int64, device=a.device)
    cb = None if colbias is None else colbias.float().contiguous()
    _lib.check(_lib.load().gf_rows_argmax(_p(a), _p(b), _p(cb), float(alpha), _p(vmax), _p(arg),
                                          B, M, N, D, _dt(a), _stream()), "gf_rows_argmax")
    return vmax, arg


def bgemm(a, b, out=None, alpha=1.0):
    """out[bt] = alpha * a[bt] @ b[bt] for 3-d a [B,M,K], b [B,K,N] with ARBITRARY strides (transposed views cost
    nothing); fp32 operands on the exact-fp32 MFMA.  No autograd (callers own their backward)."""
    _chk(a, b)
    assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[1] and a.dtype == b.dtype
    B, M, K = a.shape
    N = b.shape[2]
    if out is None:
        out = torch.empty((B, M, N), dtype=a.dtype, device=a.device)
    st = lambda t: _lib.strides(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    _lib.check(_lib.load().gf_bgemm(_p(a), _p(b), _p(out), B, M, N, K, st(a), st(b), st(out), float(alpha), _dt(a),
                                    _stream()), "gf_bgemm")
    return out


def _head_bwd(a, b, r, c, gr, gc, da, db):
    """da = dS b, db = dS^T a for dS = P_row * gr + P_col * gc (module docstring of _DualLSE), written into the given
    buffers.  bf16 / D = 256: the fused gf_head_bwd (no dS tensor); otherwise dS is written once and two batched
    products (gf_bgemm) follow."""
    B, M, D = a.shape
    N = b.shape[1]
    if a.dtype == torch.bfloat16 and D == 256 and da.is_contiguous() and db.is_contiguous():
        _lib.check(_lib.load().gf_head_bwd(_p(a), _p(b), _p(r), _p(c), _p(gr), _p(gc), _p(da), _p(db), B, M, N, D,
                                           _dt(a), _stream()), "gf_head_bwd")
        return
    dS = torch.empty((B, M, N), dtype=a.dtype, device=a.device)
    _lib.check(_lib.load().gf_dual_softmax_bwd(_p(a), _p(b), _p(r), _p(c), _p(gr), _p(gc), None, 0,
                                               0.0, _p(dS), B, M, N, D, _dt(a), _stream()), "gf_dual_softmax_bwd")
    bgemm(dS, b, out=da)                       # exact-fp32 MFMA in the fp32 parity mode: no library product on the path
    bgemm(dS.transpose(1, 2), a, out=db)


class _DualLSE(torch.autograd.Function):
    """(r, c) = (LSE_j S_ij, LSE_i S_ij) for S = a b^T, never materialising S.
    Backward: dS = P_row * gr + P_col * gc (written once in the compute dtype), then two GEMMs."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _mat3(a), _mat3(b)
        r = rows_lse(a, b)
        c = rows_lse(b, a)
        ctx.save_for_backward(a, b, r, c)
        return r, c

    @staticmethod
    def backward(ctx, gr, gc):
        a, b, r, c = ctx.saved_tensors
        B, M, D = a.shape
        N = b.shape[1]
        gr = torch.zeros_like(r) if gr is None else gr.float().contiguous()
        gc = torch.zeros_like(c) if gc is None else gc.float().contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        _head_bwd(a, b, r, c, gr, gc, da, db)
        return da, db


def dual_lse(a, b):
    return _DualLSE.apply(a, b)


class _DualLSEStacked(torch.autograd.Function):
    """dual_lse on a batch-stacked md [2B,N,D] (image 0 = first half): the gradient comes back as ONE
    stacked tensor (the two GEMMs write into its halves), so autograd needs no slice/zero-fill/add."""

    @staticmethod
    def forward(ctx, md):
        B = md.shape[0] // 2
        a, b = md[:B], md[B:]
        r = rows_lse(a, b)
        c = rows_lse(b, a)
        ctx.save_for_backward(md, r, c)
        return r, c

    @staticmethod
    def backward(ctx, gr, gc):
        md, r, c = ctx.saved_tensors
        B2, N, D = md.shape
        B = B2 // 2
        a, b = md[:B], md[B:]
        gr = torch.zeros_like(r) if gr is None else gr.float().contiguous()
        gc = torch.zeros_like(c) if gc is None else gc.float().contiguous()
        d = torch.empty_like(md)
        _head_bwd(a, b, r, c, gr, gc, d[:B], d[B:])
        return d


def dual_lse_stacked(md):
    return _DualLSEStacked.apply(md)


class _LGLayerLoss(torch.autograd.Function):
    """Partial sums acc [B,4] of one layer's deep-supervision loss (gf_lg_loss_fwd) from the batch-stacked
    head inputs md [2B,N,D], z [2B,N] (matchability logits), t [2B,N] (token-confidence logits or None).
    One node in the autograd graph: its backward writes d(md) (dense double-softmax part + sparse positives),
    dz and dt directly."""

    @staticmethod
    def forward(ctx, md, z, t, rc, pos, neg0, neg1, fin0, fin1):
        _chk(md, z, t)
        lib = _lib.load()
        md = _mat3(md)
        B2, N, D = md.shape
        B = B2 // 2
        a, b = md[:B], md[B:]
        z = z.float().contiguous()
        if rc is None:
            c = rows_lse(b, a)
            r = None
        else:
            r, c = (x.detach().float().contiguous() for x in rc)
        pb, pi, pj = (x.contiguous() for x in pos)
        P = pb.shape[0]
        neg0, neg1 = neg0.float().contiguous(), neg1.float().contiguous()
        acc = torch.empty((B, 4), dtype=torch.float32, device=md.device)
        tgt = None
        if t is not None or r is None:      # three passes: c, then (r, row arg-max), then column arg-max
            st = torch.empty((5, B, N), dtype=torch.float32, device=md.device)
            ar = torch.empty((2, B, N), dtype=torch.int64, device=md.device)
            v0, v1, a0, a1 = st[0], st[1], ar[0], ar[1]
            want_r = r is None
            if want_r:
                r = st[2]
            _lib.check(lib.gf_rows_lse_argmax(_p(a), _p(b), _p(z[B:]), _p(c), 2.0, _p(r) if want_r else None,
                                              _p(v0), _p(a0), B, N, N, D, _dt(md), _stream()), "gf_rows_lse_argmax")
            if t is not None:
                _lib.check(lib.gf_rows_lse_argmax(_p(b), _p(a), _p(z[:B]), _p(r), 2.0, None, _p(v1), _p(a1),
                                                  B, N, N, D, _dt(md), _stream()), "gf_rows_lse_argmax")
        if t is not None:
            t = t.float().contiguous()
            tgt = torch.empty((B2, N), dtype=torch.float32, device=md.device)
            fin0, fin1 = fin0.contiguous(), fin1.contiguous()
            extra = (_p(t[:B]), _p(t[B:]), _p(v0), _p(a0), _p(v1), _p(a1), _p(fin0), _p(fin1), _p(tgt[:B]), _p(tgt[B:]))
        else:
            extra = (None,) * 10
        _lib.check(lib.gf_lg_loss_fwd(_p(a), _p(b), _p(z[:B]), _p(z[B:]), _p(r), _p(c), _p(pb), _p(pi), _p(pj), P,
                                      _p(neg0), _p(neg1), *extra, _p(acc), B, N, N, D, _dt(md), _stream()),
                   "gf_lg_loss_fwd")
        ctx.save_for_backward(md, z, t, r, c, tgt, pb, pi, pj, neg0, neg1)
        return acc

    @staticmethod
    def backward(ctx, gacc):
        md, z, t, r, c, tgt, pb, pi, pj, neg0, neg1 = ctx.saved_tensors
        lib = _lib.load()
        B2, N, D = md.shape
        B = B2 // 2
        P = pb.shape[0]
        a, b = md[:B], md[B:]
        gacc = gacc.float().contiguous()
        dz = torch.empty_like(z)
        dt = None if t is None else torch.empty_like(t)
        grc = torch.empty((2, B, N), dtype=torch.float32, device=md.device)
        tp = (None,) * 4 if t is None else (_p(t[:B]), _p(t[B:]), _p(tgt[:B]), _p(tgt[B:]))
        dtp = (None, None) if t is None else (_p(dt[:B]), _p(dt[B:]))
        _lib.check(lib.gf_lg_loss_bwd_tokens(_p(z[:B]), _p(z[B:]), _p(neg0), _p(neg1), *tp, _p(pb), _p(pi), _p(pj), P,
                                             _p(gacc), _p(dz[:B]), _p(dz[B:]), *dtp, _p(grc[0]), _p(grc[1]),
                                             B, N, N, _stream()), "gf_lg_loss_bwd_tokens")
        d = torch.empty_like(md)
        _head_bwd(a, b, r, c, grc[0], grc[1], d[:B], d[B:])
        _lib.check(lib.gf_lg_loss_bwd_rows(_p(a), _p(b), _p(pb), _p(pi), _p(pj), P, _p(gacc), _p(d[:B]), _p(d[B:]),
                                           B, N, N, D, _dt(md), _stream()), "gf_lg_loss_bwd_rows")
        return d, dz, dt, None, None, None, None, None, None


def lg_layer_loss(md, z, t, rc, pos, neg0, neg1, fin0, fin1):
    """acc [B,4] = (sum_pos A_ij, sum of weighted dustbin terms, sum bce image 0, sum bce image 1)."""
    return _LGLayerLoss.apply(md, z, t, rc, pos, neg0, neg1, fin0, fin1)


class _AssignWrite(torch.autograd.Function):
    """out[b,i,j] = alpha a_i.b_j + rowbias_i + colbias_j, plus dustbin column/row/corner."""

    @staticmethod
    def forward(ctx, a, b, rowbias, colbias, bin_col, bin_row, alpha, corner, expsum=None):
        # corner: python float, or a 0-d / [B] tensor (differentiable, e.g. SuperGlue's bin_score)
        # expsum: optional [B] fp32 buffer, filled with sum_{i<M, j<=N} exp(out) (not differentiable)
        _chk(a, b, rowbias, colbias, bin_col, bin_row)
        corner_t = corner if torch.is_tensor(corner) else None
        corner = 0.0 if corner_t is not None else corner
        a, b = _mat3(a), _mat3(b)
        B, M, D = a.shape
        N = b.shape[1]
        rb, cb, bc, br = (t.float().contiguous() for t in (rowbias, colbias, bin_col, bin_row))
        out = torch.empty((B, M + 1, N + 1), dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().gf_assign_write(_p(a), _p(b), _p(rb), _p(cb), _p(bc), _p(br),
                                               float(alpha), float(corner), _p(out), _p(expsum), B, M, N, D,
                                               _dt(a), _stream()), "gf_assign_write")
        if corner_t is not None:
            out[:, -1, -1] = corner_t.detach().float()
        ctx.corner_shape = None if corner_t is None else corner_t.shape
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        ctx.dts = (rowbias.dtype, colbias.dtype, bin_col.dtype, bin_row.dtype)
        return out

    @staticmethod
    def backward(ctx, G):
        # Dense upstream gradient: only reached when somebody differentiates through the
        # materialised matrix (never in the training step, whose loss heads are sparse).
        a, b = ctx.saved_tensors
        sp = _known_sparse(G)
        if sp is not None:
            # the upstream node was the NLL of the matrix itself (GlueStick's point head): G holds one positive per row at
            # most plus the dustbin row / column, so the two products are a row gather and a row scatter -- O((M + N) D)
            # instead of two [M, N] x [N, D] products, a cast pass and two reductions over the dense gradient
            idx, vpos, n0, n1 = sp
            d = ctx.dts
            wv = (ctx.alpha * vpos)[..., None]
            da = db = grow = gcol = None
            if ctx.needs_input_grad[0]:
                da = (wv * b.gather(1, idx[..., None].expand(-1, -1, b.shape[2])).float()).to(a.dtype)
            if ctx.needs_input_grad[1]:
                db = torch.zeros(b.shape, dtype=torch.float32, device=b.device).scatter_add_(
                    1, idx[..., None].expand(-1, -1, a.shape[2]), wv * a.float()).to(b.dtype)
            if ctx.needs_input_grad[2]:
                grow = vpos.to(d[0])
            if ctx.needs_input_grad[3]:
                gcol = torch.zeros((b.shape[0], b.shape[1]), dtype=torch.float32, device=b.device).scatter_add_(1, idx, vpos).to(d[1])
            gcorner = None
            if ctx.corner_shape is not None:
                gcorner = G[:, -1, -1].sum() if len(ctx.corner_shape) == 0 else G[:, -1, -1].reshape(ctx.corner_shape)
            return (da, db, grow, gcol, n0.to(d[2]), n1.to(d[3]), None, gcorner, None)
        core = G[:, :-1, :-1]
        g = core.to(a.dtype, memory_format=torch.contiguous_format)      # ONE pass over the dense gradient; alpha rides in the products
        da = bgemm(g, b, alpha=ctx.alpha) if ctx.needs_input_grad[0] else None
        db = bgemm(g.transpose(1, 2), a, alpha=ctx.alpha) if ctx.needs_input_grad[1] else None
        d = ctx.dts
        gcorner = None
        if ctx.corner_shape is not None:
            gcorner = G[:, -1, -1].sum() if len(ctx.corner_shape) == 0 else G[:, -1, -1].reshape(ctx.corner_shape)
        grow = core.sum(2).to(d[0]) if ctx.needs_input_grad[2] else None     # (SuperGlue's couplings have no row / column bias)
        gcol = core.sum(1).to(d[1]) if ctx.needs_input_grad[3] else None
        return (da, db, grow, gcol, G[:, :-1, -1].to(d[2]), G[:, -1, :-1].to(d[3]), None, gcorner, None)


def assign_write(a, b, rowbias, colbias, bin_col, bin_row, alpha=2.0, corner=0.0, with_expsum=False):
    """-> out [B,M+1,N+1]; with_expsum: (out, expsum [B]) where expsum = exp(out)[:, :-1].sum((1, 2)), detached."""
    if not with_expsum:
        return _AssignWrite.apply(a, b, rowbias, colbias, bin_col, bin_row, alpha, corner)
    expsum = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    return _AssignWrite.apply(a, b, rowbias, colbias, bin_col, bin_row, alpha, corner, expsum), expsum


@torch.no_grad()
def filter_matches(max0, arg0, arg1, th):
    """Mutual-NN filter from the row/column arg-max vectors -> (m0, m1, s0, s1)."""
    _chk(max0, arg0, arg1)
    B, M = arg0.shape
    N = arg1.shape[1]
    max0, arg0, arg1 = max0.float().contiguous(), arg0.contiguous(), arg1.contiguous()
    m0 = torch.empty((B, M), dtype=torch.int64, device=arg0.device)
    m1 = torch.empty((B, N), dtype=torch.int64, device=arg0.device)
    s0 = torch.empty((B, M), dtype=torch.float32, device=arg0.device)
    s1 = torch.empty((B, N), dtype=torch.float32, device=arg0.device)
    _lib.check(_lib.load().gf_filter_matches(_p(max0), _p(arg0), _p(arg1), float(th), _p(m0), _p(m1),
                                             _p(s0), _p(s1), B, M, N, _stream()), "gf_filter_matches")
    return m0, m1, s0, s1


# ------------------------------------------------------------------------------ generic fused-qkv attention
class _AttentionQKV(torch.autograd.Function):
    """Attention on a fused projection qkv [B',N,3,H,D] (no rotary; SuperGlue / GlueStick GNN).

    cross=False: every image attends to itself.  cross=True: B' = 2B stacked images, image b
    attends to the keys/values of image (b + B) mod 2B.  Every q/k/v slot is consumed by exactly
    one call, so the backward writes dq/dk/dv straight into one dqkv buffer."""

    @staticmethod
    def forward(ctx, qkv, cross, scale=None, split=False):
        B2, N, _, H, D = qkv.shape
        sc = ctx.scale = D ** -0.5 if scale is None else scale       # LN2: the caller folded head_dim^-1/2 log2(e) into q
        split = bool(split) and qkv.dtype == torch.bfloat16
        o = torch.empty((B2, N, H, D), dtype=qkv.dtype, device=qkv.device)
        o32 = torch.empty((B2, N, H, D), dtype=torch.float32, device=qkv.device) if split else None   # kept for the backward's delta
        lse = torch.empty((B2, H, N), dtype=torch.float32, device=qkv.device)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        if not cross:
            attn_fwd_raw(q, k, v, sc, out=o, lse=lse, split=split, o32=o32)
        else:
            B = B2 // 2
            attn_fwd_raw(q[:B], k[B:], v[B:], sc, out=o[:B], lse=lse[:B], split=split, o32=None if o32 is None else o32[:B])
            attn_fwd_raw(q[B:], k[:B], v[:B], sc, out=o[B:], lse=lse[B:], split=split, o32=None if o32 is None else o32[B:])
        ctx.save_for_backward(qkv, o32 if split else o, lse)
        ctx.cross, ctx.split = cross, split
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B2, N, _, H, D = qkv.shape
        if not do.is_contiguous():
            do = do.contiguous()
        d = torch.empty_like(qkv)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        dq, dk, dv = d[:, :, 0], d[:, :, 1], d[:, :, 2]
        sp = ctx.split
        if not ctx.cross:
            attn_bwd_raw(q, k, v, o, do, lse, dq, dk, dv, ctx.scale, split=sp)
        else:
            B = B2 // 2
            attn_bwd_raw(q[:B], k[B:], v[B:], o[:B], do[:B], lse[:B], dq[:B], dk[B:], dv[B:], ctx.scale, split=sp)
            attn_bwd_raw(q[B:], k[:B], v[:B], o[B:], do[B:], lse[B:], dq[B:], dk[:B], dv[:B], ctx.scale, split=sp)
        return d, None, None, None


def attention_qkv(qkv, cross=False, scale=None, split=False):
    """split: fp32-equivalent second products on bf16 operands (GF_ATTN_SPLIT) -- the reference's fp32-pinned attention of
    GlueStick under mixed precision (gluestick.py:524-529)."""
    return _AttentionQKV.apply(qkv, cross, scale, split)


# ------------------------------------------------------------------------------ Sinkhorn optimal transport
class _Sinkhorn(torch.autograd.Function):
    """out = Z + u + v - norm after `iters` log-domain Sinkhorn iterations on the couplings
    Z [B,M+1,N+1] (fp32).  Only the u/v iterates are kept for the backward."""

    @staticmethod
    def forward(ctx, Z, iters, schedule):
        _chk(Z)
        assert Z.dtype == torch.float32 and Z.dim() == 3
        Z = Z.contiguous()
        B, R, C = Z.shape
        M, N = R - 1, C - 1
        L = _lib.load()
        nbytes = L.gf_sinkhorn_ws_bytes(B, M, N, iters)
        if nbytes < 0:
            _lib.check(int(nbytes), "gf_sinkhorn_ws_bytes")
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=Z.device)
        out = torch.empty_like(Z)
        uh = torch.empty((max(iters, 1), B, R), dtype=torch.float32, device=Z.device)
        vh = torch.empty((max(iters, 1), B, C), dtype=torch.float32, device=Z.device)
        _lib.check(L.gf_sinkhorn_fwd(_p(Z), _p(out), _p(uh), _p(vh), _p(ws), B, M, N, iters, schedule, _stream()),
                   "gf_sinkhorn_fwd")
        ctx.save_for_backward(Z, uh, vh)
        ctx.iters, ctx.schedule = iters, schedule
        return out

    @staticmethod
    def backward(ctx, G):
        Z, uh, vh = ctx.saved_tensors
        B, R, C = Z.shape
        M, N = R - 1, C - 1
        G = G.float().contiguous()
        L = _lib.load()
        ws = torch.empty(int(L.gf_sinkhorn_ws_bytes(B, M, N, ctx.iters)), dtype=torch.uint8, device=Z.device)
        gZ = torch.empty_like(Z)
        known = _known_sums(G)                   # the fused NLL node hands over the sums of its sparse gradient
        gr, gc = known if known is not None else (G.sum(2).contiguous(), G.sum(1).contiguous())
        _lib.check(L.gf_sinkhorn_bwd(_p(Z), _p(G), _p(gr), _p(gc), _p(uh), _p(vh), _p(gZ), _p(ws),
                                     B, M, N, ctx.iters, ctx.schedule, _stream()), "gf_sinkhorn_bwd")
        return gZ, None, None


def sinkhorn_schedule(mode=None, wait_ms=None, safe_handoff=None):
    """The `schedule` argument of gf_sinkhorn_fwd / _bwd (include/gf_amd.h): mode 0 = streaming kernels only, 1 = chip-resident
    sweeps from 5 pairs per launch (default), 2 = resident whenever the problem fits (csrc/sinkhorn_resident.h); wait_ms = bound
    of every inter-workgroup wait of the resident kernel (default 10 s; a pair whose wait expires comes out as NaN).
    Host-side knobs GF_SINKHORN_RESIDENT / GF_SINKHORN_WAIT_MS fill what the caller leaves open (the library itself reads no
    environment and keeps no setting)."""
    if mode is None:
        m = os.environ.get("GF_SINKHORN_RESIDENT", "1")
        mode = int(m) if m in ("0", "1", "2") else 1
    if wait_ms is None:
        w = os.environ.get("GF_SINKHORN_WAIT_MS", "0")
        wait_ms = int(w) if w.isdigit() else 0
    if mode not in (0, 1, 2) or not 0 <= wait_ms < (1 << 23):
        raise ValueError("sinkhorn_schedule: mode in {0, 1, 2}, 0 <= wait_ms < 2^23")
    if safe_handoff is None:        # GF_SINKHORN_SAFE_HANDOFF=1: never take the same-XCD (shared-L2) hand-off path
        safe_handoff = os.environ.get("GF_SINKHORN_SAFE_HANDOFF", "0") == "1"
    return int(mode) | (4 if safe_handoff else 0) | (int(wait_ms) << 8)


def sinkhorn(Z, iters, schedule=None):
    return _Sinkhorn.apply(Z, iters, sinkhorn_schedule() if schedule is None else int(schedule))


# ------------------------------------------------------------------------------ BatchNorm1d (+ReLU)
class _BatchNormAct(torch.autograd.Function):
    """act(BatchNorm(x)) on channels-last x [M,C]; batch statistics in training (optionally summed
    across ranks = SyncBatchNorm), given statistics in eval.  Returns (y, mean, biased var, count);
    only y is differentiable."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean_in, rstd_in, eps, training, relu, sync, run_mean=None, run_var=None,
                momentum=0.0):
        """run_mean / run_var (fp32, contiguous) given: updated in place by the finalize kernel (single-process
        training); otherwise the caller updates the running statistics from the returned mean / var / n."""
        _chk(x)
        M, C = x.shape
        L = _lib.load()
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        if training:
            nblk = L.gf_bn_nblk(M)
            part = torch.empty((nblk, 2, C), dtype=torch.float32, device=x.device)
            _lib.check(L.gf_bn_stats(_p(x), _p(part), M, C, _dt(x), _stream()), "gf_bn_stats")
            if sync:
                import torch.distributed as dist
                s = part.sum(0)
                n_t = torch.full((), float(M), dtype=s.dtype, device=s.device)      # device-side fill: capturable
                packed = torch.cat([s.flatten(), n_t[None]])
                dist.all_reduce(packed)
                s, n_t = packed[:-1].view(2, C), packed[-1]
                mean = (s[0] / n_t).contiguous()
                var = (s[1] / n_t - mean * mean).clamp(min=0.0)
                rstd = torch.rsqrt(var + eps).contiguous()
            else:       # one kernel: block sums -> mean / var / rstd (+ running statistics)
                mvr = torch.empty((3, C), dtype=torch.float32, device=x.device)
                mean, var, rstd = mvr[0], mvr[1], mvr[2]
                _lib.check(L.gf_bn_finalize_fwd(_p(part), nblk, C, float(M), float(eps), float(momentum), _p(mean),
                                                _p(var), _p(rstd), _p(run_mean), _p(run_var), _stream()),
                           "gf_bn_finalize_fwd")
                n_t = float(M)
        else:
            mean, rstd = mean_in.float().contiguous(), rstd_in.float().contiguous()
            var, n_t = mean.new_zeros(C), float(M)
        y = torch.empty_like(x)
        _lib.check(L.gf_bn_act_fwd(_p(x), _p(mean), _p(rstd), _p(g32), _p(b32), _p(y), M, C, int(relu), _dt(x),
                                   _stream()), "gf_bn_act_fwd")
        if torch.is_tensor(n_t):
            ctx.save_for_backward(x, mean, rstd, g32, b32, n_t)
            ctx.n = None
        else:
            ctx.save_for_backward(x, mean, rstd, g32, b32)
            ctx.n = n_t
            n_t = torch.empty(0, device=x.device)          # placeholder output (the count is a host constant here)
        ctx.cfg = (training, relu, sync, gamma.dtype, beta.dtype)
        ctx.mark_non_differentiable(mean, var, n_t)
        return y, mean, var, n_t

    @staticmethod
    def backward(ctx, dy, _gm, _gv, _gn):
        if ctx.n is None:
            x, mean, rstd, g32, b32, n_t = ctx.saved_tensors
        else:
            x, mean, rstd, g32, b32 = ctx.saved_tensors
            n_t = None
        training, relu, sync, gdt, bdt = ctx.cfg
        M, C = x.shape
        if not dy.is_contiguous():
            dy = dy.contiguous()
        L = _lib.load()
        nblk = L.gf_bn_nblk(M)
        part = torch.empty((nblk, 2, C), dtype=torch.float32, device=x.device)
        _lib.check(L.gf_bn_bwd_stats(_p(x), _p(dy), _p(mean), _p(rstd), _p(g32), _p(b32), _p(part), M, C,
                                     int(relu), _dt(x), _stream()), "gf_bn_bwd_stats")
        if training and not sync:
            out = torch.empty((4, C), dtype=torch.float32, device=x.device)
            dbeta, dgamma, m1, m2 = out[0], out[1], out[2], out[3]
            _lib.check(L.gf_bn_finalize_bwd(_p(part), nblk, C, float(M), _p(dbeta), _p(dgamma), _p(m1), _p(m2),
                                            _stream()), "gf_bn_finalize_bwd")
        else:
            s = part.sum(0)
            dbeta, dgamma = s[0].clone(), s[1].clone()          # local sums: DDP averages parameter grads
            if training:
                import torch.distributed as dist
                s = s.contiguous()
                dist.all_reduce(s)
                m1, m2 = (s[0] / n_t).contiguous(), (s[1] / n_t).contiguous()
            else:
                m1 = m2 = torch.zeros(C, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        _lib.check(L.gf_bn_bwd_dx(_p(x), _p(dy), _p(mean), _p(rstd), _p(g32), _p(b32), _p(m1), _p(m2), _p(dx),
                                  M, C, int(relu), _dt(x), _stream()), "gf_bn_bwd_dx")
        return dx, dgamma.to(gdt), dbeta.to(bdt), None, None, None, None, None, None, None, None, None


class _BatchNormActSets(torch.autograd.Function):
    """act(BatchNorm(x[h])) for h = 0..H-1 on x [H,M,C]: H independent statistics sets through the SAME BatchNorm
    module (the reference calls its MLP once per image: superglue.py:70-79 via :276-283), single-process training.
    One node: no select / stack copies around the per-image calls, running statistics updated set after set."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, run_mean, run_var, momentum, replay=None):
        """replay: None, or the module's num_batches_tracked buffer -- the BACKWARD then applies the H running-statistics
        updates a second time (gf_bn_replay_running) and counts them, as an activation-checkpointed reference does when its
        backward re-runs the forward in training mode (superglue.py:160-169; gluestick.py:724-757 with `checkpointed`)."""
        _chk(x)
        H, M, C = x.shape
        L = _lib.load()
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        nblk = L.gf_bn_nblk(M)
        part = torch.empty((nblk, 2, C), dtype=torch.float32, device=x.device)
        mvr = torch.empty((H, 3, C), dtype=torch.float32, device=x.device)
        ctx.replay = None if replay is None else (run_mean, run_var, replay, float(momentum))
        y = torch.empty_like(x)
        st, dt = _stream(), _dt(x)
        for h in range(H):
            _lib.check(L.gf_bn_stats(_p(x[h]), _p(part), M, C, dt, st), "gf_bn_stats")
            _lib.check(L.gf_bn_finalize_fwd(_p(part), nblk, C, float(M), float(eps), float(momentum), _p(mvr[h, 0]),
                                            _p(mvr[h, 1]), _p(mvr[h, 2]), _p(run_mean), _p(run_var), st),
                       "gf_bn_finalize_fwd")
            _lib.check(L.gf_bn_act_fwd(_p(x[h]), _p(mvr[h, 0]), _p(mvr[h, 2]), _p(g32), _p(b32), _p(y[h]), M, C,
                                       int(relu), dt, st), "gf_bn_act_fwd")
        ctx.save_for_backward(x, mvr, g32, b32)
        ctx.cfg = (relu, gamma.dtype, beta.dtype)
        ctx.mark_non_differentiable(mvr)
        return y, mvr

    @staticmethod
    def backward(ctx, dy, _gmvr):
        x, mvr, g32, b32 = ctx.saved_tensors
        relu, gdt, bdt = ctx.cfg
        H, M, C = x.shape
        if not dy.is_contiguous():
            dy = dy.contiguous()
        L = _lib.load()
        nblk = L.gf_bn_nblk(M)
        part = torch.empty((nblk, 2, C), dtype=torch.float32, device=x.device)
        out = torch.empty((H, 4, C), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        st, dt = _stream(), _dt(x)
        for h in range(H):
            mean, rstd = mvr[h, 0], mvr[h, 2]
            _lib.check(L.gf_bn_bwd_stats(_p(x[h]), _p(dy[h]), _p(mean), _p(rstd), _p(g32), _p(b32), _p(part), M, C,
                                         int(relu), dt, st), "gf_bn_bwd_stats")
            _lib.check(L.gf_bn_finalize_bwd(_p(part), nblk, C, float(M), _p(out[h, 0]), _p(out[h, 1]), _p(out[h, 2]),
                                            _p(out[h, 3]), st), "gf_bn_finalize_bwd")
            _lib.check(L.gf_bn_bwd_dx(_p(x[h]), _p(dy[h]), _p(mean), _p(rstd), _p(g32), _p(b32), _p(out[h, 2]),
                                      _p(out[h, 3]), _p(dx[h]), M, C, int(relu), dt, st), "gf_bn_bwd_dx")
        dbeta, dgamma = (out[0, 0], out[0, 1]) if H == 1 else (out[:, 0].sum(0), out[:, 1].sum(0))
        if ctx.replay is not None:
            run_mean, run_var, nbt, momentum = ctx.replay
            gate = REPLAY_GATE
            _lib.check(L.gf_bn_replay_running(_p(mvr), H, C, float(M), momentum, _p(run_mean), _p(run_var), _p(gate), st),
                       "gf_bn_replay_running")
            nbt.add_(H if gate is None else (gate == 0).to(nbt.dtype) * H)
        return dx, dgamma.to(gdt), dbeta.to(bdt), None, None, None, None, None, None


# The reference `continue`s BEFORE its backward when the loss is non-finite or not differentiable (train.py:477-488): the
# activation-checkpointed blocks are then not re-run and their BatchNorm statistics take ONE update.  TrainStep always runs its
# backward (the gradient reducer needs every rank's), so it parks its device-side "bad" flag here for the duration of the
# backward and the replays below become no-ops on such a step (fp32 scalar tensor, non-zero = skip; None = always replay).
REPLAY_GATE = None

FORCE_SYNC_BN = False     # tests: take the SyncBatchNorm exchange in a one-rank group too (TrainStep(force_distributed=True))

COLLECTIVES = {"syncbn": 0}       # collectives issued by the ops of this module (bench.py / tests count them per step)


def _all_reduce_sum(t):
    import torch.distributed as dist
    COLLECTIVES["syncbn"] += 1
    dist.all_reduce(t)


class _BatchNormActSetsSync(torch.autograd.Function):
    """_BatchNormActSets under SyncBatchNorm (train.py:338): the H statistics sets of a call -- both images through the SAME
    BatchNorm module (superglue.py:70-79 via :276-283) -- share ONE all-reduce per direction instead of one per set: the block
    sums of all sets are packed with their row counts (gf_bn_pack_sums), reduced across ranks, and the fused finalize kernels
    read the reduced buffer (mean / var / rstd of every set + the running statistics, set after set; m1 / m2 in the backward).
    Row counts may differ between ranks (they are reduced with the sums).  dgamma / dbeta stay LOCAL sums: the gradient reducer
    averages parameter gradients across ranks."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, run_mean, run_var, momentum, replay=None):
        _chk(x)
        H, M, C = x.shape
        L = _lib.load()
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        nblk = L.gf_bn_nblk(M)
        part = torch.empty((H, nblk, 2, C), dtype=torch.float32, device=x.device)
        packed = torch.empty(H * 2 * C + H, dtype=torch.float32, device=x.device)
        mvr = torch.empty((H, 3, C), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        st, dt = _stream(), _dt(x)
        for h in range(H):
            _lib.check(L.gf_bn_stats(_p(x[h]), _p(part[h]), M, C, dt, st), "gf_bn_stats")
        _lib.check(L.gf_bn_pack_sums(_p(part), H, nblk, C, float(M), _p(packed), None, st), "gf_bn_pack_sums")
        _all_reduce_sum(packed)                                   # ONE exchange for the H sets
        _lib.check(L.gf_bn_finalize_sets_fwd(_p(packed), H, C, float(eps), float(momentum), _p(mvr), _p(run_mean), _p(run_var),
                                             st), "gf_bn_finalize_sets_fwd")
        counts = packed[H * 2 * C:]                               # the reduced row count of every set
        for h in range(H):
            _lib.check(L.gf_bn_act_fwd(_p(x[h]), _p(mvr[h, 0]), _p(mvr[h, 2]), _p(g32), _p(b32), _p(y[h]), M, C,
                                       int(relu), dt, st), "gf_bn_act_fwd")
        ctx.replay = None if replay is None else (run_mean, run_var, replay, float(momentum))
        ctx.save_for_backward(x, mvr, g32, b32, counts)
        ctx.cfg = (relu, gamma.dtype, beta.dtype)
        ctx.mark_non_differentiable(mvr, counts)
        return y, mvr, counts

    @staticmethod
    def backward(ctx, dy, _gmvr, _gc):
        x, mvr, g32, b32, counts = ctx.saved_tensors
        relu, gdt, bdt = ctx.cfg
        H, M, C = x.shape
        if not dy.is_contiguous():
            dy = dy.contiguous()
        L = _lib.load()
        nblk = L.gf_bn_nblk(M)
        part = torch.empty((H, nblk, 2, C), dtype=torch.float32, device=x.device)
        packed = torch.empty(H * 2 * C + H, dtype=torch.float32, device=x.device)
        local = torch.empty((H, 2, C), dtype=torch.float32, device=x.device)
        m12 = torch.empty((H, 2, C), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        st, dt = _stream(), _dt(x)
        for h in range(H):
            _lib.check(L.gf_bn_bwd_stats(_p(x[h]), _p(dy[h]), _p(mvr[h, 0]), _p(mvr[h, 2]), _p(g32), _p(b32), _p(part[h]), M, C,
                                         int(relu), dt, st), "gf_bn_bwd_stats")
        _lib.check(L.gf_bn_pack_sums(_p(part), H, nblk, C, 0.0, _p(packed), _p(local), st), "gf_bn_pack_sums")
        _all_reduce_sum(packed)
        _lib.check(L.gf_bn_finalize_sets_bwd(_p(packed), _p(counts), H, C, _p(m12), st), "gf_bn_finalize_sets_bwd")
        for h in range(H):
            _lib.check(L.gf_bn_bwd_dx(_p(x[h]), _p(dy[h]), _p(mvr[h, 0]), _p(mvr[h, 2]), _p(g32), _p(b32), _p(m12[h, 0]),
                                      _p(m12[h, 1]), _p(dx[h]), M, C, int(relu), dt, st), "gf_bn_bwd_dx")
        dbeta, dgamma = (local[0, 0], local[0, 1]) if H == 1 else (local[:, 0].sum(0), local[:, 1].sum(0))
        if ctx.replay is not None:
            run_mean, run_var, nbt, momentum = ctx.replay
            gate = REPLAY_GATE
            _lib.check(L.gf_bn_replay_running_n(_p(mvr), _p(counts), H, C, momentum, _p(run_mean), _p(run_var), _p(gate), st),
                       "gf_bn_replay_running_n")
            nbt.add_(H if gate is None else (gate == 0).to(nbt.dtype) * H)
        return dx, dgamma.to(gdt), dbeta.to(bdt), None, None, None, None, None, None


class _ReplayRunningStats(torch.autograd.Function):
    """Identity on y whose BACKWARD gives BatchNorm modules' running statistics one more update from the forward's batch
    statistics, group after group and set after set (the generic form of _BatchNormActSets' `replay`: single-set /
    SyncBatchNorm / torch-fallback paths, and several module calls whose replays must run in CALL order although their
    own autograd nodes run in reverse).  groups: [(bn module, [(mean, unbiased var), ...]), ...]."""

    @staticmethod
    def forward(ctx, y, groups):
        ctx.groups = [(bn, [(m.detach(), v.detach()) for m, v in stats]) for bn, stats in groups]
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        gate = REPLAY_GATE
        with torch.no_grad():
            for bn, stats in ctx.groups:
                for mean, unbiased in stats:
                    if gate is None:
                        bn.num_batches_tracked += 1
                        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                        bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
                        bn.running_var.mul_(1 - mom).add_(unbiased.to(bn.running_var.dtype), alpha=mom)
                        continue
                    go = (gate == 0)                              # device-side: a skipped step replays nothing
                    bn.num_batches_tracked += go.to(bn.num_batches_tracked.dtype)
                    mom = bn.momentum if bn.momentum is not None else 1.0 / bn.num_batches_tracked.clamp(min=1).float()
                    rm, rv = bn.running_mean, bn.running_var
                    rm.copy_(torch.where(go, rm * (1 - mom) + mean.to(rm.dtype) * mom, rm))
                    rv.copy_(torch.where(go, rv * (1 - mom) + unbiased.to(rv.dtype) * mom, rv))
        return dy, None


def replay_running_stats(y, groups):
    """y, with the replays collected in ``groups`` (``stats_out`` of batch_norm_act_sets) attached to its backward."""
    return _ReplayRunningStats.apply(y, groups) if groups else y


def batch_norm_act_sets(x, bn, relu=True, replay=False, stats_out=None):
    """x [H,M,C]: ``batch_norm_act`` applied to each of the H sets in turn (set h sees the running statistics already
    updated by set h-1, exactly like H consecutive module calls), as ONE autograd node where that is possible.
    ``replay``: the backward repeats the H running-statistics updates (see _BatchNormActSets.forward); with ``stats_out`` (a
    list) the replay is NOT attached here: (bn, [(mean, unbiased var) per set]) is appended for replay_running_stats, which
    lets a caller replay several calls in call order from one node."""
    assert x.dim() == 3
    if not x.is_contiguous():
        x = x.contiguous()
    import torch.distributed as dist
    sync = (isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized()
            and (dist.get_world_size() > 1 or FORCE_SYNC_BN))
    want = (replay or stats_out is not None) and bn.training and bn.track_running_stats and torch.is_grad_enabled() and x.requires_grad
    if (bn.training and bn.track_running_stats and bn.momentum is not None
            and bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous()
            and bn.running_var.is_contiguous()):
        in_node = want and stats_out is None
        if sync:        # one all-reduce per direction for the H sets, fused finalize kernels kept
            y, mvr, counts = _BatchNormActSetsSync.apply(x, bn.weight, bn.bias, bn.eps, relu, bn.running_mean, bn.running_var,
                                                         float(bn.momentum), bn.num_batches_tracked if in_node else None)
        else:
            y, mvr = _BatchNormActSets.apply(x, bn.weight, bn.bias, bn.eps, relu, bn.running_mean, bn.running_var,
                                             float(bn.momentum), bn.num_batches_tracked if in_node else None)
            counts = None
        with torch.no_grad():
            bn.num_batches_tracked += x.shape[0]
        if want and stats_out is not None:
            if counts is None:
                unb = [x.shape[1] / max(x.shape[1] - 1.0, 1.0)] * x.shape[0]
            else:
                unb = [counts[h] / (counts[h] - 1.0).clamp(min=1.0) for h in range(x.shape[0])]
            stats_out.append((bn, [(mvr[h, 0], mvr[h, 1] * unb[h]) for h in range(x.shape[0])]))
        return y
    stats = [] if want else None
    y = torch.stack([batch_norm_act(x[h], bn, relu, stats_out=stats) for h in range(x.shape[0])])
    if not stats:
        return y
    if stats_out is not None:
        stats_out.append((bn, stats))
        return y
    return _ReplayRunningStats.apply(y, [(bn, stats)])        # (one node: the sets replay in call order)


def batch_norm_act(x, bn, relu=True, replay=False, stats_out=None):
    """x [M,C] channels-last through ``bn`` (an nn.BatchNorm1d / SyncBatchNorm that owns the affine
    parameters and running statistics), then ReLU when ``relu``.  Training mode uses batch statistics
    (summed across ranks when ``bn`` was converted to SyncBatchNorm) and updates the running
    statistics like torch (momentum, unbiased variance, num_batches_tracked); eval uses them.
    ``replay``: the backward repeats that update (the module sits inside an activation-checkpointed block of the
    reference: see _BatchNormActSets.forward); ``stats_out``: a list that receives this call's (mean, unbiased var)
    instead (batch_norm_act_sets replays several calls in order from one node)."""
    assert x.dim() == 2
    if not x.is_contiguous():
        x = x.contiguous()
    if bn.training or not bn.track_running_stats:
        import torch.distributed as dist
        sync = (isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size() > 1 or FORCE_SYNC_BN))
        track = bn.training and bn.track_running_stats
        fused_running = (track and not sync and bn.momentum is not None and bn.running_mean.dtype == torch.float32
                         and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous())
        replay = replay and track and torch.is_grad_enabled() and x.requires_grad
        if fused_running:       # the finalize kernel updates the running statistics in place
            y, mean, var, _ = _BatchNormAct.apply(x, bn.weight, bn.bias, None, None, bn.eps, True, relu, sync,
                                                  bn.running_mean, bn.running_var, float(bn.momentum))
            with torch.no_grad():
                bn.num_batches_tracked += 1
            if replay or stats_out is not None:
                stat = (mean, var * (x.shape[0] / max(x.shape[0] - 1.0, 1.0)))
                if stats_out is not None:
                    stats_out.append(stat)
                else:
                    y = _ReplayRunningStats.apply(y, [(bn, [stat])])
            return y
        y, mean, var, n_t = _BatchNormAct.apply(x, bn.weight, bn.bias, None, None, bn.eps, True, relu, sync)
        if track:
            with torch.no_grad():
                if not torch.is_tensor(n_t) or n_t.numel() == 0:
                    n_t = torch.full((), float(x.shape[0]), dtype=torch.float32, device=x.device)
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                unbiased = var * (n_t / (n_t - 1).clamp(min=1.0))
                bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
                bn.running_var.mul_(1 - mom).add_(unbiased.to(bn.running_var.dtype), alpha=mom)
            if stats_out is not None:
                stats_out.append((mean, unbiased))
            elif replay:
                y = _ReplayRunningStats.apply(y, [(bn, [(mean, unbiased)])])
        return y
    rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
    return _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, rstd, bn.eps, False, relu, False)[0]


# ------------------------------------------------------------------------------ GlueStick line message passing
@torch.no_grad()
def _line_graph_sorted(idx, n):
    """line_graph by a stable sort (any size): endpoints grouped by junction in their original order + segment starts."""
    B = idx.shape[0]
    order = torch.argsort(idx, dim=1, stable=True)
    sorted_idx = idx.gather(1, order).contiguous()
    seg = torch.searchsorted(sorted_idx, torch.arange(n + 1, device=idx.device).expand(B, -1).contiguous())
    return order.to(torch.int32).contiguous(), seg.to(torch.int32).contiguous()


@torch.no_grad()
def line_graph(idx, n):
    """idx [B,E] int64 junction of every line endpoint -> (order [B,E] int32: endpoints grouped by junction, stable;
    seg [B,n+1] int32: segment starts).  Built once per forward: all line layers (and the backward) share it."""
    _chk(idx)
    idx = idx.contiguous()
    B, E = idx.shape
    # a junction index outside [0, n) would corrupt LDS in the kernels below; the torch gather they replace raises too
    # (device-side assert: no host synchronisation)
    torch._assert_async(((idx >= 0) & (idx < n)).all())
    if E > 4096 or n > 8192:
        # gf_line_csr keeps one image's junction graph in LDS (4096 endpoints, 8192 junctions): larger graphs -- far beyond the
        # 250-512 lines of the shipped configurations -- are built by a stable sort instead (same order / segment arrays)
        return _line_graph_sorted(idx, n)
    order = torch.empty((B, E), dtype=torch.int32, device=idx.device)
    seg = torch.empty((B, n + 1), dtype=torch.int32, device=idx.device)
    _lib.check(_lib.load().gf_line_csr(_p(idx), _p(order), _p(seg), B, E, n, _stream()), "gf_line_csr")
    return order, seg


def _segsum(s0, s1, order, seg, base, B, E, N, D, mode):
    out = torch.empty((B, N, D), dtype=s0.dtype, device=s0.device)
    _lib.check(_lib.load().gf_line_segsum(_p(s0), s0.stride(1), _p(s1), 0 if s1 is None else s1.stride(1), _p(order),
                                          _p(seg), _p(base), _p(out), B, E, N, D, mode, _dt(s0), _stream()),
               "gf_line_segsum")
    return out


class _LineGather(torch.autograd.Function):
    """msg [B,E,3D] = [x[idx[e]] | x[idx[e^1]] | enc[e]] (gluestick.py:609-621); backward: deterministic segment sums."""

    @staticmethod
    def forward(ctx, x, enc, idx, order, seg, chain=None):
        _chk(x, enc, idx)
        x, enc, idx = x.contiguous(), enc.contiguous(), idx.contiguous()
        B, N, D = x.shape
        E = idx.shape[1]
        msg = torch.empty((B, E, 3 * D), dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().gf_line_gather(_p(x), _p(idx), _p(enc), _p(msg), B, E, N, D, _dt(x), _stream()),
                   "gf_line_gather")
        ctx.save_for_backward(order, seg)
        ctx.dims = (B, E, N, D)
        ctx.chain = chain
        return msg

    @staticmethod
    def backward(ctx, dmsg):
        order, seg = ctx.saved_tensors
        B, E, N, D = ctx.dims
        if not dmsg.is_contiguous():
            dmsg = dmsg.contiguous()
        # chain: the residual gradient of the same x (parked by _LineAggregate.backward) is the base of the segment sum
        base = ctx.chain.take() if ctx.chain is not None else None
        dx = _segsum(dmsg[:, :, :D], dmsg[:, :, D:2 * D], order, seg, base, B, E, N, D, 0)
        return dx, dmsg[:, :, 2 * D:], None, None, None, None


class _LineAggregate(torch.autograd.Function):
    """x + (mean | sum) over the endpoints on each junction of upd [B,E,D] (gluestick.py:660-700)."""

    @staticmethod
    def forward(ctx, x, upd, idx, order, seg, mean, chain=None):
        _chk(x, upd, idx)
        x, upd = x.contiguous(), upd.contiguous()
        B, N, D = x.shape
        E = upd.shape[1]
        out = _segsum(upd, None, order, seg, x, B, E, N, D, 1 if mean else 0)
        ctx.save_for_backward(idx.contiguous(), seg)
        ctx.dims = (B, E, N, D, mean)
        ctx.chain = chain
        return out

    @staticmethod
    def backward(ctx, g):
        idx, seg = ctx.saved_tensors
        B, E, N, D, mean = ctx.dims
        if not g.is_contiguous():
            g = g.contiguous()
        dupd = torch.empty((B, E, D), dtype=g.dtype, device=g.device)
        _lib.check(_lib.load().gf_line_expand(_p(g), _p(idx), _p(seg), _p(dupd), B, E, N, D, 1 if mean else 0, _dt(g),
                                              _stream()), "gf_line_expand")
        if ctx.chain is not None:                  # x's residual gradient rides in the gather's segment sum (see GradChain)
            ctx.chain.park(g)
            g = None
        return g, dupd, None, None, None, None, None


def line_gather(x, enc, idx, order, seg, chain=None):
    """chain: ops.GradChain(2) shared with the line_aggregate of the same x (a LineLayer reads its descriptors twice: the
    endpoint gather and the residual of the aggregation): the two gradients meet in the gather's segment-sum kernel."""
    return _LineGather.apply(x, enc, idx, order, seg, chain if x.requires_grad else None)


def line_aggregate(x, upd, idx, order, seg, mean=True, chain=None):
    return _LineAggregate.apply(x, upd, idx, order, seg, mean, chain if x.requires_grad else None)


# ------------------------------------------------------------------------------ GlueStick line head (dense scores)
class _RowsGather(torch.autograd.Function):
    """out[b,e,:] = x[b, idx[b,e], :]; backward: deterministic segment sum over the junction graph (order, seg)."""

    @staticmethod
    def forward(ctx, x, idx, order, seg):
        _chk(x, idx)
        x = x.contiguous()
        B, N, D = x.shape
        E = idx.shape[1]
        out = torch.empty((B, E, D), dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().gf_rows_gather(_p(x), _p(idx), _p(out), B, E, N, D, _dt(x), _stream()), "gf_rows_gather")
        ctx.save_for_backward(order, seg)
        ctx.n = N
        return out

    @staticmethod
    def backward(ctx, g):
        order, seg = ctx.saved_tensors
        g = g.contiguous()
        B, E, D = g.shape
        return _segsum(g, None, order, seg, None, B, E, ctx.n, D, 0), None, None, None


def rows_gather(x, idx, order, seg):
    return _RowsGather.apply(x, idx.contiguous(), order, seg)


class _LinePairScores(torch.autograd.Function):
    """raw[a,c] = scale/2 * max(S[2a,2c] + S[2a+1,2c+1], S[2a,2c+1] + S[2a+1,2c]) with S = g0 g1^T the endpoint
    scores (gluestick.py:345-354), fp32.  S is kept for the backward (16 MB per pair at 512 lines); the gradient
    reaches g0 / g1 through two gf_bgemm products."""

    @staticmethod
    def forward(ctx, g0, g1, scale):
        g0, g1 = g0.float().contiguous(), g1.float().contiguous()
        B, E0, D = g0.shape
        E1 = g1.shape[1]
        S = bgemm(g0, g1.transpose(1, 2), alpha=scale)
        raw = torch.empty((B, E0 // 2, E1 // 2), dtype=torch.float32, device=g0.device)
        _lib.check(_lib.load().gf_line_pair_scores(_p(S), None, _p(raw), B, E0 // 2, E1 // 2, 0, _stream()),
                   "gf_line_pair_scores")
        ctx.save_for_backward(g0, g1, S)
        ctx.scale = scale
        return raw

    @staticmethod
    def backward(ctx, draw):
        g0, g1, S = ctx.saved_tensors
        B, E0, _ = g0.shape
        E1 = g1.shape[1]
        dS = torch.empty_like(S)
        draw = draw.contiguous()
        _lib.check(_lib.load().gf_line_pair_scores(_p(S), _p(draw), _p(dS), B, E0 // 2, E1 // 2, 1, _stream()),
                   "gf_line_pair_scores")
        return bgemm(dS, g1, alpha=ctx.scale), bgemm(dS.transpose(1, 2), g0, alpha=ctx.scale), None


def line_pair_scores(g0, g1, scale):
    return _LinePairScores.apply(g0, g1, scale)


def _dense_rowcol(z, M, N, mode):
    B = z.shape[0]
    rows = torch.empty((B, M), dtype=torch.float32, device=z.device)
    cols = torch.empty((B, N), dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().gf_dense_rowcol(_p(z), z.stride(0), z.stride(1), _p(rows), _p(cols), B, M, N, mode, _stream()),
               "gf_dense_rowcol")
    return rows, cols


class _DenseLogDoubleSoftmax(torch.autograd.Function):
    """gluestick.py:772-783 on a dense fp32 [B,M,N] score matrix with a learnable bin score beta:
    out[:, :M, :N] = raw - (r_i + c_j) / 2, out[:, :M, N] = beta - r_i, out[:, M, :N] = beta - c_j, out[:, M, N] = 0 with
    r_i = log(sum_j exp raw_ij + exp beta), c_j likewise over rows.  The [B,M,N] passes are HIP kernels
    (csrc/line_head.hip); the [B,M] / [B,N] vector algebra stays in torch."""

    @staticmethod
    def forward(ctx, raw, beta):
        _chk(raw)
        raw = raw.float().contiguous()
        B, M, N = raw.shape
        beta = beta.float().reshape(())
        r0, c0 = _dense_rowcol(raw, M, N, 0)
        r, c = torch.logaddexp(r0, beta), torch.logaddexp(c0, beta)
        out = torch.empty((B, M + 1, N + 1), dtype=torch.float32, device=raw.device)
        rb, cb, br, bc = (-0.5 * r).contiguous(), (-0.5 * c).contiguous(), (beta - r).contiguous(), (beta - c).contiguous()
        _lib.check(_lib.load().gf_dense_assign(_p(raw), _p(rb), _p(cb), _p(br), _p(bc), 0.0, _p(out),
                                               B, M, N, _stream()), "gf_dense_assign")
        ctx.save_for_backward(raw, r, c, beta)
        return out

    @staticmethod
    def backward(ctx, G):
        raw, r, c, beta = ctx.saved_tensors
        B, M, N = raw.shape
        G = G.float().contiguous()
        gs_r, gs_c = _dense_rowcol(G, M, N, 1)                    # sums of the core block of G ([B,M+1,N+1] view strides)
        A = 0.5 * gs_r + G[:, :M, N]                              # gradient arriving at -r_i
        Bv = 0.5 * gs_c + G[:, M, :N]                             # ... at -c_j
        draw = torch.empty_like(raw)
        A, Bv = A.contiguous(), Bv.contiguous()      # (named: a temporary's storage could be reused before the launch reads it)
        _lib.check(_lib.load().gf_dense_assign_bwd(_p(raw), _p(r), _p(c), _p(A), _p(Bv), _p(G),
                                                   _p(draw), B, M, N, _stream()), "gf_dense_assign_bwd")
        dbeta = (G[:, :M, N] - A * torch.exp(beta - r)).sum() + (G[:, M, :N] - Bv * torch.exp(beta - c)).sum()
        return draw, dbeta


def dense_log_double_softmax(raw, beta):
    return _DenseLogDoubleSoftmax.apply(raw, beta)


# ------------------------------------------------------------------------------ sparse positives of the NLL losses
_SPARSE_SUMS = {}        # data_ptr of the dense gradient _NllTerms just wrote -> (weakref, row sums, column sums)


def _known_sums(G):
    hit = _SPARSE_SUMS.pop(G.data_ptr(), None)
    if (hit is not None and hit[0]() is G and hit[4] == G._version and hit[1].shape == G.shape[:2]
            and hit[2].shape == (G.shape[0], G.shape[2])):
        return hit[1].contiguous(), hit[2].contiguous()
    return None


def _known_sparse(G):
    """(col index of each row's positive, its gradient value (0 where none), dustbin-column values, dustbin-row values) when
    G is the gradient _NllTerms just wrote (so: nothing else anywhere), else None."""
    hit = _SPARSE_SUMS.pop(G.data_ptr(), None)
    if hit is not None and hit[0]() is G and hit[4] == G._version and hit[3][0].shape == (G.shape[0], G.shape[1] - 1):
        return hit[3]
    return None


class _NllTerms(torch.autograd.Function):
    """(sum over the positives of la[b, i, col0[b, i]], sum over the unmatched rows / columns of their dustbin entries) of a
    log assignment la [B, M+1, N+1] (superglue.py:322-352, gluestick.py:378-414).  The gradient is written ONCE: one fill of
    the dense matrix + three sparse writes -- autograd's own backward of the gather and the two dustbin slices builds three
    dense tensors and adds them (1.4 ms per SuperGlue step at 32 x 2049^2)."""

    @staticmethod
    def forward(ctx, la, col0, neg0, neg1):
        valid = col0 >= 0
        idx = col0.clamp(min=0).long()[..., None]
        picked = la[:, :-1, :].gather(2, idx).squeeze(-1)
        pos = (picked * valid.to(picked.dtype)).sum(1)
        neg = (la[:, :-1, -1] * neg0).sum(1) + (la[:, -1, :-1] * neg1).sum(1)
        ctx.save_for_backward(idx, valid, neg0, neg1)
        ctx.shape, ctx.dtype = la.shape, la.dtype
        return pos, neg

    @staticmethod
    def backward(ctx, gpos, gneg):
        idx, valid, neg0, neg1 = ctx.saved_tensors
        G = torch.zeros(ctx.shape, dtype=ctx.dtype, device=idx.device)
        vpos = gpos[:, None] * valid.to(ctx.dtype)
        n0, n1 = gneg[:, None] * neg0, gneg[:, None] * neg1
        G[:, :-1, :].scatter_(2, idx, vpos[..., None])
        G[:, :-1, -1] = n0                               # (a positive never sits in the dustbin column)
        G[:, -1, :-1] = n1
        # row / column sums of this sparse matrix, for a consumer that wants them (the Sinkhorn backward): known here
        # from O(M + N) values instead of two more sweeps of the dense tensor
        gr = torch.cat([vpos + n0, n1.sum(1, keepdim=True)], 1)
        gc = torch.cat([n1, n0.sum(1, keepdim=True)], 1).scatter_add_(1, idx.squeeze(-1), vpos)
        _SPARSE_SUMS.clear()
        # G._version: autograd's input buffer ACCUMULATES a second gradient of the same tensor in place (version bump) --
        # the sparse description would then be incomplete, and the consumers fall back to the dense tensor
        _SPARSE_SUMS[G.data_ptr()] = (weakref.ref(G), gr, gc, (idx.squeeze(-1), vpos, n0, n1), G._version)
        return G, None, None, None


def nll_terms(la, data, neg0, neg1, prefix=""):
    """-> (sum of la over the positives, number of positives, sum of the dustbin entries of the unmatched rows and columns),
    per pair.  With the ground truth's ``gt_<prefix>assignment_col0`` vector: one autograd node (see _NllTerms)."""
    col0 = data.get("gt_" + prefix + "assignment_col0")
    if col0 is not None:
        pos, neg = _NllTerms.apply(la, col0, neg0, neg1)
        return pos, (col0 >= 0).sum(1).float(), neg
    pos, num_pos = nll_positive_terms(la, data, prefix)
    return pos, num_pos, (la[:, :-1, -1] * neg0).sum(1) + (la[:, -1, :-1] * neg1).sum(1)


def nll_positive_terms(la, data, prefix=""):
    """(sum over the positives of la[b,i,j], number of positives) per pair, for the NLL of superglue.py:322-352 and
    gluestick.py:378-414 (weights = gt_assignment).  When the ground-truth producer supplied
    ``gt_<prefix>assignment_col0`` (the single positive column of each row, -1 if none: ours do) the terms are one
    fixed-length gather -- no scan of the dense matrix, no host synchronisation, capturable in a hipGraph; otherwise
    the dense matrix is scanned with nonzero() (same numbers, one host read)."""
    bsz = la.shape[0]
    col0 = data.get("gt_" + prefix + "assignment_col0")
    if col0 is not None:
        valid = col0 >= 0
        picked = la[:, :-1, :].gather(2, col0.clamp(min=0).long()[..., None]).squeeze(-1)
        return (picked * valid.to(picked.dtype)).sum(1), valid.sum(1).float()
    bi, ii, ji = data["gt_" + prefix + "assignment"].nonzero(as_tuple=True)
    zeros = torch.zeros(bsz, device=la.device)
    return zeros.index_add(0, bi, la[bi, ii, ji]), zeros.index_add(0, bi, torch.ones_like(bi, dtype=torch.float32))
