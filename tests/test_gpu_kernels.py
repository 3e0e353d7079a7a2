"""HIP kernels (through the C ABI + autograd wrappers) vs fp64 torch restatements of the same
math.  fp32 mode: 1e-4 (north_star tolerance); bf16 mode: loose bounds, reported."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from glue_factory_amd import ops

DEV = "cuda"


def _attn_ref(q, k, v, scale):
    # q [B,Nq,H,D] fp64
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    p = torch.softmax(s, -1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v), torch.logsumexp(s, -1)


def _tols(dtype):
    return dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 4, 128, 128, 64), (1, 2, 100, 77, 64), (2, 4, 300, 513, 64), (1, 1, 1, 1, 64),
                                         # head_dim 32 / 128: the generic kernels (capability path)
                                         (2, 4, 128, 128, 32), (1, 2, 100, 77, 32), (2, 2, 300, 513, 128), (1, 3, 65, 200, 128)])
def test_attention_fwd_bwd(dtype, B, H, Nq, Nk, D):
    if D == 128 and dtype == torch.float32:
        pytest.skip("fp32 at head_dim 128 does not fit the LDS (GF_ERR_UNSUPPORTED, checked below)")
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q, k, v, do = (torch.randn(B, n, H, D, generator=g, dtype=torch.float64)
                   for n in (Nq, Nk, Nk, Nq))
    # asymmetric content (transpose-detecting): scale rows / channels differently
    k = k * (1 + torch.arange(D, dtype=torch.float64) / D)
    v = v + torch.arange(Nk, dtype=torch.float64)[None, :, None, None] / Nk
    qd, kd, vd, dod = (t.to(DEV, dtype) for t in (q, k, v, do))
    qr, kr, vr = (t.to(torch.float64).cpu().requires_grad_(True) for t in (qd, kd, vd))
    oref, lse_ref = _attn_ref(qr, kr, vr, D ** -0.5)
    (oref * dod.cpu().double()).sum().backward()

    qd, kd, vd = (t.requires_grad_(True) for t in (qd, kd, vd))
    o = ops.attention(qd, kd, vd)
    (o * dod).sum().backward()
    torch.testing.assert_close(o.detach().cpu().double(), oref.detach(), **_tols(dtype))
    for name, a, b in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        scale = max(b.abs().max().item(), 1e-2)
        torch.testing.assert_close(a.cpu().double() / scale, b / scale, msg=lambda m: f"{name}: {m}",
                                   **_tols(dtype))
    _, lse = ops.attn_fwd_raw(qd.detach(), kd.detach(), vd.detach(), D ** -0.5)
    torch.testing.assert_close(lse.cpu().double(), lse_ref.detach(), rtol=1e-4,
                               atol=1e-4 if dtype == torch.float32 else 3e-2)


def test_attention_head_dim_limits():
    q = torch.zeros(1, 64, 2, 128, device=DEV, requires_grad=True)
    o = ops.attention(q, q, q)                                 # fp32, head_dim 128: the forward fits the LDS ...
    with pytest.raises(RuntimeError):
        o.sum().backward()                                     # ... the backward's staging (208 KB) does not
    q = torch.zeros(1, 64, 2, 48, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.attn_fwd_raw(q, q, q, 0.1)                         # head_dim 48: no kernel


def test_attention_strided_views_and_spike():
    """q,k,v as strided views of one fused buffer; one key spikes (forces the online rescale)."""
    B, N, H, D = 2, 200, 4, 64
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, N, 3, H, D, generator=g, dtype=torch.float64)
    qkv[:, 150, 1] *= 12.0   # late spike: running max jumps in the third 64-key tile
    d = qkv.to(DEV, torch.float32)
    o = ops.attention(d[:, :, 0], d[:, :, 1], d[:, :, 2])
    ref, _ = _attn_ref(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5)
    torch.testing.assert_close(o.cpu().double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 4, 256, 256), (1, 2, 100, 333)])
def test_attention_premultiplied_operands(B, H, Nq, Nk):
    """scale = ln 2 with head_dim^-1/2 * log2(e) already folded into q (ops.attn_premul: what LightGlue's projections do):
    the bf16 kernels take the no-multiply instantiations (rr == 1).  Forward, lse and all three gradients vs fp64 on the
    SAME pre-multiplied bf16 operands; and the result equals plain attention on the un-multiplied q up to q's rounding."""
    D = 64
    g = torch.Generator().manual_seed(Nq + Nk)
    q, k, v, do = (torch.randn(B, n, H, D, generator=g, dtype=torch.float64) for n in (Nq, Nk, Nk, Nq))
    c = ops.attn_premul(D)
    qd = (q * c).to(DEV, torch.bfloat16)
    kd, vd, dod = (t.to(DEV, torch.bfloat16) for t in (k, v, do))
    qr, kr, vr = (t.double().cpu().requires_grad_(True) for t in (qd, kd, vd))
    oref, lse_ref = _attn_ref(qr, kr, vr, ops.LN2)
    (oref * dod.cpu().double()).sum().backward()
    qd, kd, vd = (t.requires_grad_(True) for t in (qd, kd, vd))
    o = ops.attention(qd, kd, vd, scale=ops.LN2)
    (o * dod).sum().backward()
    torch.testing.assert_close(o.detach().cpu().double(), oref.detach(), rtol=2e-2, atol=2e-2)
    for name, a, b in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        sc = max(b.abs().max().item(), 1e-2)
        torch.testing.assert_close(a.cpu().double() / sc, b / sc, rtol=3e-2, atol=3e-2, msg=lambda m: f"{name}: {m}")
    _, lse = ops.attn_fwd_raw(qd.detach(), kd.detach(), vd.detach(), ops.LN2)
    torch.testing.assert_close(lse.cpu().double(), lse_ref.detach(), rtol=1e-4, atol=2e-2)
    plain = ops.attention(q.to(DEV, torch.bfloat16), kd.detach(), vd.detach())
    torch.testing.assert_close(o.detach().float(), plain.float(), rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("case", ["random2048", "late_spike", "ramp", "ramp_steep", "huge_logits", "ragged_tail", "first_key_dominates"])
def test_attention_bf16_forward_reference_cases(case):
    """The bf16 forward keeps a REFERENCE m instead of a running max (csrc/attention_fwd3.hip): scores leave the MFMA as
    exp2 arguments relative to m, and a tile is redone conventionally when its row sum exceeds 2^16.  Cases that walk
    every branch: maxima that creep up below the threshold (P >> 1 without a redo), that jump across it late, logits
    far outside exp's range, a ragged last tile, a first key that dominates (everything after underflows)."""
    B, H, D = 2, 2, 64
    N = {"random2048": 2048, "ragged_tail": 333}.get(case, 512)
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(B, N, H, D, generator=g, dtype=torch.float64) for _ in range(3))
    pos = torch.arange(N, dtype=torch.float64)[None, :, None, None]
    if case == "late_spike":
        k[:, 450] *= 12.0
        k[:, 200] *= 5.0
    elif case == "ramp":            # logit scale grows slowly with the key index: the row maximum rises in every tile
        k = k * (1 + 1.5 * pos / N)
    elif case == "ramp_steep":      # ... and fast enough to cross the 2^16 limit several times
        k = k * (1 + 12.0 * pos / N)
    elif case == "huge_logits":
        q = q * 6.0
        k = k * 6.0
    elif case == "first_key_dominates":
        k[:, 0] = q[:, 7] * 40.0
    qd, kd, vd = (t.to(DEV, torch.bfloat16) for t in (q, k, v))
    ref, lse_ref = _attn_ref(*(t.double().cpu() for t in (qd, kd, vd)), D ** -0.5)
    o, lse = ops.attn_fwd_raw(qd, kd, vd, D ** -0.5)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    # P is rounded to bf16 (2^-9 relative), O is stored as bf16: errors scale with |v| ~ 4
    torch.testing.assert_close(o.double().cpu(), ref, rtol=2e-2, atol=3e-2)
    # lse: the bf16 rounding of q * scale * log2(e) moves a logit by up to 2^-9 of its magnitude
    smax = (torch.einsum("bqhd,bkhd->bhqk", qd.double().cpu(), kd.double().cpu()) * D ** -0.5).abs().amax(-1)
    assert ((lse.double().cpu() - lse_ref).abs() <= 2e-2 + 6e-3 * smax).all(), (lse.double().cpu() - lse_ref).abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_self_attention_rotary(dtype):
    B, N, H, D = 2, 96, 4, 64
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, N, 3, H, D, generator=g).to(DEV, dtype)
    theta = (torch.randn(B, N, D // 2, generator=g) * 2).to(DEV).requires_grad_(True)
    do = torch.randn(B, N, H, D, generator=g).to(DEV, dtype)
    # reference in fp64 on the CPU
    x = qkv.detach().cpu().double().requires_grad_(True)
    th = theta.detach().cpu().double().requires_grad_(True)
    cos, sin = torch.cos(th).repeat_interleave(2, -1), torch.sin(th).repeat_interleave(2, -1)

    def rot(t):  # t [B,N,H,D]
        te, to = t[..., 0::2], t[..., 1::2]
        r = torch.stack((-to, te), -1).flatten(-2)
        return t * cos[:, :, None] + r * sin[:, :, None]

    oref, _ = _attn_ref(rot(x[:, :, 0]), rot(x[:, :, 1]), x[:, :, 2], D ** -0.5)
    (oref * do.cpu().double()).sum().backward()

    qkv_in = qkv.clone().requires_grad_(True)
    work = qkv_in * 1.0   # non-leaf private buffer, rotated in place by the op
    cs = torch.stack((torch.cos(theta), torch.sin(theta)), -1).flatten(-2).detach().contiguous()
    o = ops.self_attention_rotary(work, theta, cs)
    (o * do).sum().backward()
    tol = _tols(dtype)
    torch.testing.assert_close(o.detach().cpu().double(), oref.detach(), **tol)
    sc = x.grad.abs().max().item()
    torch.testing.assert_close(qkv_in.grad.cpu().double() / sc, x.grad / sc, **tol)
    sc = th.grad.abs().max().item()
    torch.testing.assert_close(theta.grad.cpu().double() / sc, th.grad / sc, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cross_attention_both_forms(dtype):
    B, N, H, D = 2, 130, 4, 64
    g = torch.Generator().manual_seed(7)
    p = torch.randn(2 * B, N, 2, H, D, generator=g).to(DEV, dtype)
    dm = torch.randn(2 * B, N, H, D, generator=g).to(DEV, dtype)
    x = p.detach().cpu().double().requires_grad_(True)
    m0, _ = _attn_ref(x[:B, :, 0], x[B:, :, 0], x[B:, :, 1], D ** -0.5)
    m1, _ = _attn_ref(x[B:, :, 0], x[:B, :, 0], x[:B, :, 1], D ** -0.5)
    mref = torch.cat([m0, m1], 0)
    (mref * dm.cpu().double()).sum().backward()
    ps = p.clone().requires_grad_(True)
    m = ops.cross_attention_stacked(ps)
    (m * dm).sum().backward()
    tol = _tols(dtype)
    torch.testing.assert_close(m.detach().cpu().double(), mref.detach(), **tol)
    sc = x.grad.abs().max().item()
    torch.testing.assert_close(ps.grad.cpu().double() / sc, x.grad / sc, **tol)
    # unstacked form with different keypoint counts
    p0 = torch.randn(B, 70, 2, H, D, generator=g).to(DEV, dtype).requires_grad_(True)
    p1 = torch.randn(B, 129, 2, H, D, generator=g).to(DEV, dtype).requires_grad_(True)
    a0, a1 = ops.cross_attention(p0, p1)
    (a0.sum() + 2 * a1.sum()).backward()
    x0 = p0.detach().cpu().double().requires_grad_(True)
    x1 = p1.detach().cpu().double().requires_grad_(True)
    r0, _ = _attn_ref(x0[:, :, 0], x1[:, :, 0], x1[:, :, 1], D ** -0.5)
    r1, _ = _attn_ref(x1[:, :, 0], x0[:, :, 0], x0[:, :, 1], D ** -0.5)
    (r0.sum() + 2 * r1.sum()).backward()
    torch.testing.assert_close(a0.detach().cpu().double(), r0.detach(), **tol)
    torch.testing.assert_close(a1.detach().cpu().double(), r1.detach(), **tol)
    for a, b in ((p0.grad, x0.grad), (p1.grad, x1.grad)):
        sc = b.abs().max().item()
        torch.testing.assert_close(a.cpu().double() / sc, b / sc, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C", [(37, 512), (256, 512), (5, 128)])
def test_ln_gelu(dtype, R, C):
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g) * 2 + 0.3).to(DEV, dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(R, C, generator=g).to(DEV, dtype)
    xr = x.detach().cpu().double().requires_grad_(True)
    gr, br = (t.detach().cpu().double().requires_grad_(True) for t in (gamma, beta))
    yref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-5))
    (yref * dy.cpu().double()).sum().backward()
    xs = x.clone().requires_grad_(True)
    y = ops.ln_gelu(xs, gamma, beta, 1e-5)
    (y * dy).sum().backward()
    tol = _tols(dtype)
    torch.testing.assert_close(y.detach().cpu().double(), yref.detach(), **tol)
    for a, b in ((xs.grad, xr.grad), (gamma.grad, gr.grad), (beta.grad, br.grad)):
        sc = b.abs().max().item()
        torch.testing.assert_close(a.cpu().double() / sc, b / sc, **tol)


def _head_inputs(B, M, N, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(B, M, D, generator=g) * 0.6).to(DEV, dtype)
    b = (torch.randn(B, N, D, generator=g) * 0.6).to(DEV, dtype)
    z0 = torch.randn(B, M, generator=g).to(DEV)
    z1 = torch.randn(B, N, generator=g).to(DEV)
    return a, b, z0, z1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,M,N,D", [(2, 128, 128, 256), (1, 100, 171, 256), (2, 65, 40, 64), (1, 1, 3, 128)])
def test_assignment_head(dtype, B, M, N, D):
    from oracle.lightglue_oracle import log_double_softmax
    a, b, z0, z1 = _head_inputs(B, M, N, D, dtype, 11 + M + N)
    ar, br = (t.detach().cpu().double().requires_grad_(True) for t in (a, b))
    z0r, z1r = (t.detach().cpu().double().requires_grad_(True) for t in (z0, z1))
    sim = ar @ br.transpose(1, 2)
    ref = log_double_softmax(sim, z0r, z1r)

    ad, bd = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z0d, z1d = z0.clone().requires_grad_(True), z1.clone().requires_grad_(True)
    r, c = ops.dual_lse(ad, bd)
    lz0, lz1 = torch.nn.functional.logsigmoid(z0d), torch.nn.functional.logsigmoid(z1d)
    out, expsum = ops.assign_write(ad, bd, lz0 - r, lz1 - c, torch.nn.functional.logsigmoid(-z0d),
                                   torch.nn.functional.logsigmoid(-z1d), alpha=2.0, corner=0.0, with_expsum=True)
    # the fused "row_norm" accumulator = exp of what was written, dustbin column in, dustbin row out (lightglue.py:602)
    torch.testing.assert_close(expsum, out.detach().exp()[:, :-1].sum((1, 2)), rtol=1e-5, atol=1e-6)
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=8e-2)
    torch.testing.assert_close(r.detach().cpu().double(), sim.logsumexp(2).detach(), **tol)
    torch.testing.assert_close(c.detach().cpu().double(), sim.logsumexp(1).detach(), **tol)
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), **tol)
    # dense backward through the materialised matrix + the two normalisers
    G = torch.randn(B, M + 1, N + 1, generator=torch.Generator().manual_seed(1)).to(DEV)
    (out * G).sum().backward()
    (ref * G.cpu().double()).sum().backward()
    for name, x, y in (("a", ad.grad, ar.grad), ("b", bd.grad, br.grad), ("z0", z0d.grad, z0r.grad),
                       ("z1", z1d.grad, z1r.grad)):
        sc = max(y.abs().max().item(), 1e-2)
        torch.testing.assert_close(x.cpu().double() / sc, y / sc, msg=lambda m: f"{name}: {m}",
                                   **(_tols(dtype)))
    # arg-maxes of the core and mutual-NN filter
    if dtype == torch.float32:
        core = ref[:, :-1, :-1].detach()
        h = {"md0": a, "md1": b, "r": r.detach(), "c": c.detach(), "lz0": lz0.detach(),
             "lz1": lz1.detach(), "bin0": torch.nn.functional.logsigmoid(-z0),
             "bin1": torch.nn.functional.logsigmoid(-z1)}
        from glue_factory_amd.matchers.lightglue import MatchAssignment
        am = MatchAssignment.argmaxes(h)
        torch.testing.assert_close(am["max0"].cpu().double(), core.max(2).values, rtol=1e-4, atol=1e-4)
        assert torch.equal(am["arg0"].cpu(), core.max(2).indices)
        assert torch.equal(am["arg1"].cpu(), core.max(1).indices)
        assert torch.equal(am["full0"].cpu(), ref[:, :-1, :].detach().max(-1).indices)
        assert torch.equal(am["full1"].cpu(), ref[:, :, :-1].detach().max(-2).indices)
        from oracle.lightglue_oracle import filter_matches
        for th in (0.0, 1e-4):
            m0, m1, s0, s1 = ops.filter_matches(am["max0"], am["arg0"], am["arg1"], th)
            rm0, rm1, rs0, rs1 = filter_matches(ref.detach(), th)
            assert torch.equal(m0.cpu(), rm0) and torch.equal(m1.cpu(), rm1)
            torch.testing.assert_close(s0.cpu().double(), rs0, rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(s1.cpu().double(), rs1, rtol=1e-4, atol=1e-6)


def test_rows_lse_colbias_and_large_logits():
    """colbias path (GlueStick bin) and +-large logits: LSE must stay finite and exact."""
    B, M, N, D = 1, 70, 90, 64
    a, b, _, _ = _head_inputs(B, M, N, D, torch.float32, 99)
    a = a * 6.0   # |S| up to ~ 100
    cb = torch.randn(B, N, device=DEV) * 3
    out = ops.rows_lse(a, b, cb)
    ref = (a.cpu().double() @ b.cpu().double().transpose(1, 2) + cb.cpu().double()[:, None]).logsumexp(2)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-3)


def test_ops_reject_cpu_tensors():
    with pytest.raises(RuntimeError):
        ops.attention(torch.zeros(1, 4, 1, 64), torch.zeros(1, 4, 1, 64), torch.zeros(1, 4, 1, 64))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C", [(1000, 256), (131072, 256), (37, 64)])
def test_rowdot2_two_heads_one_read(dtype, M, C):
    """gf_rowdot2: matchability + token-confidence logits of the same rows in one pass; head 0 differentiable w.r.t. x
    (dx = dz0 w0), head 1 on x.detach() (lightglue.py:81-94, 275-276, 285-286) == two nn.Linear(dim, 1) in fp64."""
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).to(DEV, dtype).requires_grad_(True)
    w0 = (torch.randn(1, C, generator=g) / C ** 0.5).to(DEV).requires_grad_(True)
    w1 = (torch.randn(1, C, generator=g) / C ** 0.5).to(DEV).requires_grad_(True)
    b0 = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    b1 = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    g0 = torch.randn(M, generator=g).to(DEV)
    g1 = torch.randn(M, generator=g).to(DEV)
    z0, z1 = ops.rowdot2(x, w0, b0, w1, b1)
    ((z0 * g0).sum() + (z1 * g1).sum()).backward()
    xr = x.detach().double().requires_grad_(True)
    w0r, w1r, b0r, b1r = (t.detach().double().requires_grad_(True) for t in (w0, w1, b0, b1))
    z0r = (xr @ w0r.t()).squeeze(-1) + b0r
    z1r = (xr.detach() @ w1r.t()).squeeze(-1) + b1r
    ((z0r * g0.double()).sum() + (z1r * g1.double()).sum()).backward()
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(z0.double(), z0r.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(z1.double(), z1r.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad.double(), xr.grad, **tol)
    for a, r in ((w0.grad, w0r.grad), (w1.grad, w1r.grad), (b0.grad, b0r.grad), (b1.grad, b1r.grad)):
        sc = float(r.abs().max())
        torch.testing.assert_close(a.double() / sc, r / sc, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K,two", [(4096, 256, 512, False), (8192, 256, 256, False), (4096, 512, 512, True), (64, 256, 256, False),
                                       (1000, 256, 512, False)])
def test_gemm_with_two_residuals(M, N, K, two):
    """gf_gemm_res2: y = [x0 | x1] W^T + b + r1 + r2 in one launch (the input-gradient GEMM where a loss head's gradient meets
    the block's residual gradient); shapes outside the streamed kernel (M % 64 != 0) take the explicit add -- same numbers."""
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    x = torch.randn(M, K, generator=g).to(DEV, bf)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV, bf)
    b = torch.randn(N, generator=g).to(DEV)
    r1 = torch.randn(M, N, generator=g).to(DEV, bf)
    r2 = torch.randn(M, N, generator=g).to(DEV, bf)
    if two:
        y = ops.gemm(x[:, :K // 2].contiguous(), w, b, res2=r1, x2b=x[:, K // 2:].contiguous(), res3=r2)
    else:
        y = ops.gemm(x, w, b, res2=r1, res3=r2)
    ref = x.double() @ w.double().t() + b.double() + r1.double() + r2.double()
    torch.testing.assert_close(y.double(), ref, rtol=1.2e-2, atol=1.2e-2)
    # one rounding of the fp32 sum: closer to the reference than "add the residuals in bf16 first"
    two_step = ops.gemm(x, w, b, res2=(r1 + r2))
    assert float((y.double() - ref).abs().mean()) <= float((two_step.double() - ref).abs().mean()) * 1.001


@pytest.mark.parametrize("M,Nout,K1,K2", [(5000, 512, 256, 256), (4133, 256, 128, 384), (131072, 512, 256, 256), (300, 128, 128, 128)])
def test_linear_cat_weight_gradient_in_one_launch(M, Nout, K1, K2):
    """gf_linear_dw2: the weight gradient of y = [x1 | x2] W^T + b over the virtual concatenation (ffn.0(cat[x, message]),
    lightglue.py:140-148,196-221) in one launch == gf_linear_dw per source == fp64 autograd."""
    from glue_factory_amd import lib as L_
    g = torch.Generator().manual_seed(M + Nout)
    x1 = torch.randn(M, K1, generator=g).to(DEV, torch.bfloat16).requires_grad_(True)
    x2 = torch.randn(M, K2, generator=g).to(DEV, torch.bfloat16).requires_grad_(True)
    w = (torch.randn(Nout, K1 + K2, generator=g) / (K1 + K2) ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(Nout, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(M, Nout, generator=g).to(DEV, torch.bfloat16)
    y = ops.linear_cat(x1, x2, w, b)
    (y * dy).sum().backward()
    # the two single-source launches this replaces (bit-comparable: same kernel, same slices per k-tile? no -- the slice
    # count follows the tile count; compared at fp32 rounding)
    L = L_.load()
    parts = []
    for x in (x1, x2):
        k = x.shape[1]
        ws = torch.empty(int(L.gf_linear_dw_ws_bytes(M, Nout, k)), dtype=torch.uint8, device=DEV)
        dwp = torch.empty(Nout, k, device=DEV)
        dbp = torch.empty(Nout, device=DEV)
        L_.check(L.gf_linear_dw(dy.data_ptr(), x.detach().data_ptr(), dwp.data_ptr(), dbp.data_ptr(), ws.data_ptr(), M, Nout, k, 1,
                                torch.cuda.current_stream().cuda_stream), "gf_linear_dw")
        parts.append((dwp, dbp))
    sc = float(w.grad.abs().max())
    torch.testing.assert_close(w.grad / sc, torch.cat([parts[0][0], parts[1][0]], 1) / sc, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(b.grad, parts[0][1], rtol=2e-5, atol=2e-5 * float(b.grad.abs().max()))
    ref_w = dy.double().t() @ torch.cat([x1.detach(), x2.detach()], 1).double()
    torch.testing.assert_close(w.grad.double() / sc, ref_w / sc, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.grad.double(), dy.double().sum(0), rtol=1e-4, atol=1e-4 * float(b.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,Nout,K", [(1000, 256, 256), (4096, 768, 256), (777, 512, 512), (64, 8, 8), (5000, 264, 136)])
def test_linear_dw(dtype, M, Nout, K):
    g = torch.Generator().manual_seed(M + Nout)
    x = torch.randn(M, K, generator=g).to(DEV, dtype).requires_grad_(True)
    w = (torch.randn(Nout, K, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(Nout, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(M, Nout, generator=g).to(DEV, dtype)
    y = ops.linear(x, w, b)
    (y * dy).sum().backward()
    xr, wr, br = (t.detach().cpu().double().requires_grad_(True) for t in (x, w.to(dtype), b.to(dtype)))
    yr = torch.nn.functional.linear(xr, wr, br)
    (yr * dy.cpu().double()).sum().backward()
    tol = _tols(dtype)
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), **tol)
    for name, a, r in (("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad), ("db", b.grad, br.grad)):
        sc = r.abs().max().item()
        torch.testing.assert_close(a.cpu().double() / sc, r / sc, msg=lambda m: f"{name}: {m}",
                                   rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-4 if dtype == torch.float32 else 2e-2)
    assert w.grad.dtype == torch.float32 and b.grad.dtype == torch.float32


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,relu", [(1000, 512, True), (333, 32, True), (4096, 256, False)])
def test_batch_norm_act(dtype, M, C, relu):
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 1.5 + 0.2).to(DEV, dtype)
    dy = torch.randn(M, C, generator=g).to(DEV, dtype)
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C, generator=g))
    ref = torch.nn.BatchNorm1d(C).double()
    ref.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in bn.state_dict().items()})
    act = torch.relu if relu else (lambda t: t)
    for mode in ("train", "eval"):
        getattr(bn, mode)()
        getattr(ref, mode)()
        xs = x.clone().requires_grad_(True)
        y = ops.batch_norm_act(xs, bn, relu)
        (y * dy).sum().backward()
        xr = x.detach().cpu().double().requires_grad_(True)
        yr = act(ref(xr))
        (yr * dy.cpu().double()).sum().backward()
        tol = _tols(dtype)
        torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), **tol)
        sc = xr.grad.abs().max().item()
        torch.testing.assert_close(xs.grad.cpu().double() / sc, xr.grad / sc, **tol)
        for a, b in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
            sc = b.abs().max().item()
            torch.testing.assert_close(a.cpu().double() / sc, b / sc, **tol)
        bn.zero_grad(); ref.zero_grad()
    torch.testing.assert_close(bn.running_mean.cpu().double(), ref.running_mean, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(bn.running_var.cpu().double(), ref.running_var, rtol=1e-3, atol=1e-3)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C", [(131072, 32), (131072, 256), (131072, 512), (196608, 32), (196608, 256), (196608, 512)])
def test_batch_norm_act_at_benchmarked_row_counts(dtype, M, C):
    """Train-mode BatchNorm + ReLU at the row counts bench.py's `other_configs` run (SuperGlue: 64 images x 2048
    keypoints = 131 072 rows, GlueStick: 64 x 3072 = 196 608; csrc/batchnorm.hip caps the statistics grid at 512
    workgroups there, every thread strides over >= 2 row groups) against an fp64 torch BatchNorm1d
    (superglue.py:70-79, gluestick.py:465-474); channel means up to 4x the spread (the statistics kernel forms
    E[x^2] - E[x]^2 from fp32 sums: its relative variance error grows as eps * (1 + (mean / std)^2))."""
    g = torch.Generator().manual_seed(M // 1024 + C)
    x = (torch.randn(M, C, generator=g) * 0.7 + torch.linspace(-1.0, 2.8, C)).to(DEV, dtype)
    dy = torch.randn(M, C, generator=g).to(DEV, dtype)
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C, generator=g))
    ref = torch.nn.BatchNorm1d(C).to(DEV).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bn.train(); ref.train()
    xs = x.clone().requires_grad_(True)
    y = ops.batch_norm_act(xs, bn, True)
    (y * dy).sum().backward()
    xr = x.detach().double().requires_grad_(True)
    zr = ref(xr)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)   # bf16: output rounding
    torch.testing.assert_close(y.detach().double(), torch.relu(zr).detach(), **tol)
    # the ReLU gate is discontinuous: an element whose pre-activation is within rounding of 0 may be gated either way (a
    # handful in 1e8 in fp32, and one of them moves a channel's dgamma by 3e-3).  The forward comparison above pins the
    # gate up to that rounding; the backward is compared under the SAME gate (taken from the kernel's own output)
    gate = y.detach() > 0
    flipped = float((gate != (zr.detach() > 0)).float().mean())
    assert flipped < (1e-6 if dtype == torch.float32 else 2e-2), flipped
    (zr * gate * dy.double()).sum().backward()
    sc = xr.grad.abs().max().item()
    torch.testing.assert_close(xs.grad.double() / sc, xr.grad / sc, **tol)
    # the statistics themselves (fp32 sums of 131 072+ terms in both modes): tight, whatever the activation dtype
    for name, a, b in (("dgamma", bn.weight.grad, ref.weight.grad), ("dbeta", bn.bias.grad, ref.bias.grad),
                       ("running_mean", bn.running_mean, ref.running_mean), ("running_var", bn.running_var, ref.running_var)):
        sc = b.abs().max().item()
        stat_tol = 2e-5 if dtype == torch.float32 or name.startswith("running") else 4e-3
        torch.testing.assert_close(a.double() / sc, b / sc, rtol=stat_tol, atol=stat_tol, msg=lambda m: f"{name}: {m}")


def test_linear_cat_equals_linear_of_cat():
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn(3, 50, 256, generator=g).to(DEV).requires_grad_(True)
    x2 = torch.randn(3, 50, 256, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(512, 512, generator=g) / 22).to(DEV).requires_grad_(True)
    b = torch.randn(512, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(3, 50, 512, generator=g).to(DEV)
    y = ops.linear_cat(x1, x2, w, b)
    (y * dy).sum().backward()
    got = [t.grad.clone() for t in (x1, x2, w, b)]
    for t in (x1, x2, w, b):
        t.grad = None
    yr = torch.nn.functional.linear(torch.cat([x1, x2], -1), w, b)
    (yr * dy).sum().backward()
    torch.testing.assert_close(y, yr, rtol=1e-4, atol=1e-4)
    for a, t in zip(got, (x1, x2, w, b)):
        torch.testing.assert_close(a, t.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C", [(5000, 256), (33, 64)])
def test_rowdot(dtype, M, C):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, C, generator=g).to(DEV, dtype).requires_grad_(True)
    w = (torch.randn(1, C, generator=g) / C ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    dz = torch.randn(M, generator=g).to(DEV)
    z = ops.rowdot(x, w, b)
    (z * dz).sum().backward()
    xr, wr, br = (t.detach().cpu().double().requires_grad_(True) for t in (x, w, b))
    zr = torch.nn.functional.linear(xr, wr, br).squeeze(-1)
    (zr * dz.cpu().double()).sum().backward()
    tol = _tols(dtype)
    torch.testing.assert_close(z.detach().cpu().double(), zr.detach(), **tol)
    for a, r in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        sc = r.abs().max().item()
        torch.testing.assert_close(a.cpu().double() / sc, r / sc, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(200, 200), (130, 77)])
def test_rows_lse_argmax_fused(dtype, M, N):
    """gf_rows_lse_argmax == (gf_rows_lse, arg-max of 2 S + logsigmoid(z) - n) of a torch fp32 reference."""
    from glue_factory_amd import lib as L_
    from glue_factory_amd.ops import _p, _dt, _stream
    g = torch.Generator(device="cuda").manual_seed(M + N)
    B, D = 3, 256
    a = (torch.randn(B, M, D, device="cuda", generator=g) * 0.3).to(dtype)
    b = (torch.randn(B, N, D, device="cuda", generator=g) * 0.3).to(dtype)
    z = torch.randn(B, N, device="cuda", generator=g)
    n = torch.randn(B, N, device="cuda", generator=g)
    S = torch.einsum("bmd,bnd->bmn", a.float(), b.float())
    ref_lse = torch.logsumexp(S, -1)
    full = 2.0 * S + (torch.nn.functional.logsigmoid(z) - n)[:, None, :]
    ref_max, ref_arg = full.max(-1)
    lse = torch.empty(B, M, device="cuda")
    vmax = torch.empty(B, M, device="cuda")
    arg = torch.empty(B, M, dtype=torch.int64, device="cuda")
    lib = L_.load()
    for with_lse in (True, False):
        lse.fill_(-7.0)
        L_.check(lib.gf_rows_lse_argmax(_p(a), _p(b), _p(z), _p(n), 2.0, _p(lse) if with_lse else None, _p(vmax), _p(arg),
                                        B, M, N, D, _dt(a), _stream()), "gf_rows_lse_argmax")
        tol = 1e-4 if dtype == torch.float32 else 2e-2
        torch.testing.assert_close(vmax, ref_max, rtol=tol, atol=tol)
        picked = full.gather(-1, arg[..., None]).squeeze(-1)
        torch.testing.assert_close(picked, ref_max, rtol=tol, atol=tol)      # ties / near-ties may pick a neighbour
        if dtype == torch.float32:
            assert (arg == ref_arg).float().mean() > 0.999
        if with_lse:
            torch.testing.assert_close(lse, ref_lse, rtol=tol, atol=tol)
        else:
            assert (lse == -7.0).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (4096, 768, 256), (777, 256, 512), (128, 128, 32), (333, 64, 64),
                                   (5000, 512, 128), (31, 32, 256)])
def test_gemm_kernel(M, N, K, dtype):
    """gf_gemm (weight-streaming GEMM) through the C ABI: fused bias / residual (incl. y aliasing res), strided weight
    slice, against an fp64 torch reference.  fp32 mode must be fp32-exact (1e-5), bf16 within bf16 rounding."""
    from glue_factory_amd import lib as L_
    from glue_factory_amd.ops import _p, _stream
    if dtype == torch.float32 and K > 256:
        pytest.skip("fp32 K > 256 goes through the K-split of ops.gemm (covered below)")
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    wide = (torch.randn(N, K + 64, device="cuda", generator=g) / K ** 0.5).to(dtype)
    w = wide[:, 32:32 + K]                                  # column slice: row stride K + 64
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g).to(dtype)
    lib = L_.load()
    dt = 1 if dtype == torch.bfloat16 else 0

    def run(b, r, y):
        L_.check(lib.gf_gemm(_p(x), None, _p(w), _p(b), _p(r), _p(y), None, 0, M, N, K, 0, x.stride(0), 0, w.stride(0),
                             0 if r is None else r.stride(0), y.stride(0), dt, _stream()), "gf_gemm")
        return y

    ref = x.double() @ w.double().t()
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    new = lambda: torch.full((M, N), float("nan"), device="cuda", dtype=dtype)   # noqa: E731
    torch.testing.assert_close(run(None, None, new()).double(), ref, **tol)
    torch.testing.assert_close(run(bias, None, new()).double(), ref + bias.double(), **tol)
    y = res.clone()
    torch.testing.assert_close(run(bias, y, y).double(), ref + bias.double() + res.double(), **tol)     # y aliases res
    if dtype == torch.bfloat16:   # tighter: against the bf16-rounded result the library produces
        lib_y = torch.nn.functional.linear(x, w, bias.bfloat16())
        assert (run(bias, None, new()).float() - lib_y.float()).abs().max() < 0.08


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_two_sources_rotary_and_k_pieces(dtype):
    """ops.gemm: the two-source form (FFN cat input), K split into power-of-two pieces (768 = 512 + 256; fp32 512 =
    256 + 256) accumulated through the residual input, and the rotary epilogue vs apply_cached_rotary_emb."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    tol = dict(rtol=2e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=2e-5)
    M = 1500
    a = torch.randn(M, 256, device="cuda", generator=g).to(dtype)
    b = torch.randn(M, 256, device="cuda", generator=g).to(dtype)
    w = (torch.randn(512, 512, device="cuda", generator=g) / 22).to(dtype)
    bias = torch.randn(512, device="cuda", generator=g)
    ref = torch.cat([a, b], 1).double() @ w.double().t() + bias.double()
    torch.testing.assert_close(ops.gemm(a, w, bias, x2b=b).double(), ref, **tol)
    # K = 768 (GlueStick line MLP, Wqkv input gradient): pieces 512 + 256 (bf16) / 256 x 3 (fp32)
    x = torch.randn(M, 768, device="cuda", generator=g).to(dtype)
    w2 = (torch.randn(256, 768, device="cuda", generator=g) / 27).to(dtype)
    r = torch.randn(M, 256, device="cuda", generator=g).to(dtype)
    ref = x.double() @ w2.double().t() + bias[:256].double() + r.double()
    tol2 = dict(rtol=3e-2, atol=5e-2) if dtype == torch.bfloat16 else tol      # bf16: one extra rounding per piece
    torch.testing.assert_close(ops.gemm(x, w2, bias[:256], res2=r).double(), ref, **tol2)
    # rotary epilogue on the first 512 of 768 output channels
    x = torch.randn(M, 256, device="cuda", generator=g).to(dtype)
    w3 = (torch.randn(768, 256, device="cuda", generator=g) / 16).to(dtype)
    b3 = torch.randn(768, device="cuda", generator=g)
    theta = torch.randn(M, 32, device="cuda", generator=g, dtype=torch.float64)
    cs = torch.stack((torch.cos(theta), torch.sin(theta)), -1).flatten(-2).float().contiguous()     # [M, 64]
    y = (x.double() @ w3.double().t() + b3.double()).view(M, 3, 4, 32, 2)
    c, s_ = torch.cos(theta)[:, None, None, :], torch.sin(theta)[:, None, None, :]
    rot = torch.stack((y[..., 0] * c - y[..., 1] * s_, y[..., 1] * c + y[..., 0] * s_), -1)
    ref = torch.cat([rot[:, :2], y[:, 2:]], 1).reshape(M, 768)
    torch.testing.assert_close(ops.gemm(x, w3, b3, cs=cs, rot_n=512).double(), ref, **tol)


def test_linear_and_ffn_residual_match_torch():
    """ops.linear(..., res=) / ops.linear_cat forward and all gradients (gf_gemm forward and input-gradient GEMMs,
    gf_linear_dw) vs the same graph in stock torch fp32."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2, 300, 256, device="cuda", generator=g).bfloat16().requires_grad_(True)
    m = torch.randn(2, 300, 256, device="cuda", generator=g).bfloat16().requires_grad_(True)
    w = (torch.randn(512, 512, device="cuda", generator=g) / 22).requires_grad_(True)
    b = torch.randn(512, device="cuda", generator=g).requires_grad_(True)
    w3 = (torch.randn(256, 512, device="cuda", generator=g) / 22).requires_grad_(True)
    outs = []
    for ours in (True, False):
        for t in (x, m, w, b, w3):
            t.grad = None
        if ours:
            h = ops.linear_cat(x, m, w, b)
            y = ops.linear(h, w3, None, res=x)
        else:
            h = torch.nn.functional.linear(torch.cat([x, m], -1).float(), w, b)
            y = torch.nn.functional.linear(h.bfloat16().float(), w3) + x.float()
        y.float().square().mean().backward()
        outs.append([y.detach().float()] + [t.grad.detach().float().clone() for t in (x, m, w, b, w3)])
    for a, r in zip(*outs):
        sc = r.abs().max().item() + 1e-6
        torch.testing.assert_close(a / sc, r / sc, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mean", [True, False])
def test_line_message_passing_kernels(dtype, mean):
    """gf_line_csr / gather / segsum / expand (GlueStick LineLayer, gluestick.py:609-700) vs torch gather + flip + cat +
    scatter_reduce, forward and backward, with shared junctions, junctions without lines and keypoint-only rows."""
    from glue_factory_amd import ops
    g = torch.Generator().manual_seed(3)
    B, NL, N, D = 3, 37, 120, 256
    E = 2 * NL
    idx = torch.randint(0, 50, (B, E), generator=g)               # junctions 0..49 shared a lot, 50..119 never used
    idx[0, :10] = 7                                               # one crowded junction
    x = torch.randn(B, N, D, generator=g)
    enc = torch.randn(B, E, D, generator=g)
    wt = torch.randn(B, E, D, generator=g)                         # stands in for the MLP: upd = msg-part product
    gout = torch.randn(B, N, D, generator=g)

    def ref(x, enc):
        ix = idx[..., None].expand(-1, -1, D)
        ld = x.gather(1, ix)
        ld2 = ld.reshape(B, NL, 2, D).flip(2).reshape(B, E, D)
        msg = torch.cat([ld, ld2, enc], -1)
        upd = (msg[..., :D] - 0.5 * msg[..., D:2 * D] + msg[..., 2 * D:]) * wt
        agg = torch.zeros_like(x).scatter_reduce(1, ix, upd, reduce="mean" if mean else "sum", include_self=False)
        return msg, x + agg

    xr, er = x.double().requires_grad_(True), enc.double().requires_grad_(True)
    wt = wt.double()
    msg_r, out_r = ref(xr, er)
    (out_r * gout.double()).sum().backward()
    wt = wt.float()

    xc = x.to(dtype).cuda().requires_grad_(True)
    ec = enc.to(dtype).cuda().requires_grad_(True)
    ic = idx.cuda()
    order, seg = ops.line_graph(ic, N)
    # the segment table is a stable grouping of the endpoints by junction
    o, sg = order.cpu().long(), seg.cpu().long()
    for b in range(B):
        assert sg[b, 0] == 0 and sg[b, -1] == E
        grouped = idx[b][o[b]]
        assert torch.equal(grouped, grouped.sort(stable=True).values)
        assert torch.equal(o[b], idx[b].sort(stable=True).indices)
    msg = ops.line_gather(xc, ec, ic, order, seg)
    upd = (msg[..., :D] - 0.5 * msg[..., D:2 * D] + msg[..., 2 * D:]) * wt.cuda().to(dtype)
    out = ops.line_aggregate(xc, upd, ic, order, seg, mean=mean)
    (out.float() * gout.cuda()).sum().backward()
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=6e-2)
    if dtype == torch.float32:
        assert torch.equal(msg.detach().cpu(), msg_r.detach().float())
    torch.testing.assert_close(out.detach().float().cpu().double(), out_r.detach(), **tol)
    sc = xr.grad.abs().max().item()
    torch.testing.assert_close(xc.grad.float().cpu().double() / sc, xr.grad / sc, **tol)
    torch.testing.assert_close(ec.grad.float().cpu().double(), er.grad, **tol)
    # the same through a gradient chain (LineLayer: the residual gradient of x rides in the gather's segment sum)
    x2 = xc.detach().clone().requires_grad_(True)
    chain = ops.GradChain(2)
    msg2 = ops.line_gather(x2, ec.detach(), ic, order, seg, chain=chain)
    upd2 = (msg2[..., :D] - 0.5 * msg2[..., D:2 * D] + msg2[..., 2 * D:]) * wt.cuda().to(dtype)
    out2 = ops.line_aggregate(x2, upd2, ic, order, seg, mean=mean, chain=chain)
    (out2.float() * gout.cuda()).sum().backward()
    assert torch.equal(out2, out)
    torch.testing.assert_close(x2.grad.float().cpu().double() / sc, xr.grad / sc, **tol)


@pytest.mark.parametrize("shape", [(2, 128, 128), (3, 200, 333), (1, 64, 1), (2, 2048, 2048)])
def test_head_bwd_fused(shape):
    """gf_head_bwd (assignment-head backward without the dS tensor: da = dS b, db = dS^T a with
    dS = exp(S - r) gr + exp(S - c) gc, lightglue.py:256-290 autograd) vs an fp64 restatement on the same bf16 inputs;
    ragged row / column counts, a single column, the benchmark size.  Tolerance: the kernel's own rounding model, entry by
    entry (one bf16 rounding of dS, fp32 accumulation, one bf16 rounding of the output), and 2 u = 7.8e-3 of the largest entry
    (was 1.5e-2; measured 2.5e-3 ... 4.5e-3, worst error / bound 0.23 ... 0.88)."""
    from glue_factory_amd import lib as L_
    B, M, N = shape
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = (torch.randn(B, M, 256, device="cuda", generator=g) * 0.25).to(torch.bfloat16)
    b = (torch.randn(B, N, 256, device="cuda", generator=g) * 0.25).to(torch.bfloat16)
    gr = torch.randn(B, M, device="cuda", generator=g)
    gc = torch.randn(B, N, device="cuda", generator=g)
    S = torch.bmm(a.double(), b.double().transpose(1, 2))
    r, c = S.logsumexp(2), S.logsumexp(1)
    dS = (S - r[:, :, None]).exp() * gr.double()[:, :, None] + (S - c[:, None, :]).exp() * gc.double()[:, None, :]
    da_ref, db_ref = torch.bmm(dS, b.double()), torch.bmm(dS.transpose(1, 2), a.double())
    da = torch.full((B, M, 256), float("nan"), device="cuda", dtype=torch.bfloat16)
    db = torch.full((B, N, 256), float("nan"), device="cuda", dtype=torch.bfloat16)
    r32, c32 = r.float().contiguous(), c.float().contiguous()
    L_.check(L_.load().gf_head_bwd(a.data_ptr(), b.data_ptr(), r32.data_ptr(), c32.data_ptr(),
                                   gr.data_ptr(), gc.data_ptr(), da.data_ptr(), db.data_ptr(), B, M, N, 256, 1,
                                   torch.cuda.current_stream().cuda_stream), "gf_head_bwd")
    # rounding model of the kernel, entry by entry: dS is rounded to bf16 once (unit roundoff u = 2^-8: 8 significant
    # bits; fp32 before), the products accumulate in fp32, the output is rounded to bf16 once:
    #     |x - y| <= u (|y| + (|dS| |b|))   (+ 5 % for the second-order terms)
    u = 2.0 ** -8
    bound_a = 1.05 * u * (da_ref.abs() + torch.bmm(dS.abs(), b.double().abs()))
    bound_b = 1.05 * u * (db_ref.abs() + torch.bmm(dS.abs().transpose(1, 2), a.double().abs()))
    for name, x, y, bd in (("da", da, da_ref, bound_a), ("db", db, db_ref, bound_b)):
        ratio = ((x.double() - y).abs() / (bd + 1e-9 * y.abs().max())).max().item()
        rel = (x.double() - y).abs().max().item() / y.abs().max().item()
        print(f"head_bwd {shape} {name}: worst |error| / rounding bound = {ratio:.3f}, max error {rel:.2e} of the largest entry")
        assert ratio <= 1.0, f"{name}: error {ratio:.3f} x the bf16 rounding bound"
        assert rel < 2 * u, f"{name}: max error {rel:.3e} of the largest entry"


@pytest.mark.parametrize("case", [(4096, 256, 256, False, False), (4096, 256, 256, False, True), (192, 512, 512, True, False),
                                  (8192, 512, 512, True, True), (2048, 256, 512, False, True), (640, 768, 256, False, False),
                                  (131072, 256, 256, False, True)])
def test_gemm_streamed_kernel(case):
    """The activation-streaming GEMM kernel behind gf_gemm (bf16, K in {256, 512}, N % 256 == 0, M % 64 == 0): one and
    several tiles per workgroup, fewer tiles than workgroups, two sources, residual incl. y aliasing it, strided weight
    rows, the benchmark's row count -- against fp64 on the same bf16 inputs, and against the register-resident kernel's
    result for shapes both can run (M + 1 rows force that one)."""
    from glue_factory_amd import lib as L_
    from glue_factory_amd.ops import _p, _stream
    M, N, K, two, with_res = case
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M + 1, K, device="cuda", generator=g).to(torch.bfloat16)
    wide = (torch.randn(N, K + 64, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    w = wide[:, 32:32 + K]
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M + 1, N, device="cuda", generator=g).to(torch.bfloat16) if with_res else None
    lib = L_.load()

    def run(rows, y):
        if two:
            x0, x1, k0, k1 = x[:, :K // 2], x[:, K // 2:], K // 2, K // 2
        else:
            x0, x1, k0, k1 = x, None, K, 0
        L_.check(lib.gf_gemm(_p(x0), _p(x1), _p(w), _p(bias), _p(y if with_res else None), _p(y), None, 0, rows, N, k0, k1,
                             x.stride(0), x.stride(0) if two else 0, w.stride(0), y.stride(0) if with_res else 0, y.stride(0),
                             1, _stream()), "gf_gemm")
        return y

    ref = x.double() @ w.double().t() + bias.double()
    if with_res:
        ref = ref + res.double()
    y_st = run(M, res.clone() if with_res else torch.full((M + 1, N), float("nan"), device="cuda", dtype=torch.bfloat16))
    torch.testing.assert_close(y_st[:M].double(), ref[:M], rtol=2e-2, atol=3e-2)
    y_ws = run(M + 1, res.clone() if with_res else torch.full((M + 1, N), float("nan"), device="cuda", dtype=torch.bfloat16))
    assert (y_st[:M].float() - y_ws[:M].float()).abs().max().item() <= 0.0625      # same fp32 sums, one bf16 rounding each


@pytest.mark.parametrize("M,N,K,rot_n", [(4096, 768, 256, 512), (128, 512, 512, 512), (131072, 768, 256, 512)])
def test_gemm_streamed_kernel_rotary(M, N, K, rot_n):
    """Rotary epilogue of the activation-streaming GEMM (the fused Wqkv projection of lightglue.py:159-160: channels
    [0, rot_n) rotated pairwise by the per-token (cos, sin) table, the rest plain) vs fp64 and vs the register-resident
    kernel (M + 1 rows force that one)."""
    from glue_factory_amd import lib as L_
    from glue_factory_amd.ops import _p, _stream
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M + 1, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    th = torch.rand(M + 1, 32, device="cuda", generator=g) * 6.28
    cs = torch.stack((torch.cos(th), torch.sin(th)), -1).flatten(-2).contiguous()      # [M+1, 64] (cos, sin) per pair
    lib = L_.load()

    def run(rows):
        y = torch.full((M + 1, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        L_.check(lib.gf_gemm(_p(x), None, _p(w), _p(bias), None, _p(y), _p(cs), rot_n, rows, N, K, 0, K, 0, K, 0, N, 1, _stream()),
                 "gf_gemm")
        return y

    lin = x.double() @ w.double().t() + bias.double()
    r = lin[:, :rot_n].reshape(M + 1, rot_n // 64, 32, 2)
    c, s_ = cs.double()[:, 0::2][:, None, :], cs.double()[:, 1::2][:, None, :]
    rot = torch.stack((r[..., 0] * c - r[..., 1] * s_, r[..., 1] * c + r[..., 0] * s_), -1).reshape(M + 1, rot_n)
    ref = torch.cat([rot, lin[:, rot_n:]], 1)
    y_st, y_ws = run(M), run(M + 1)
    torch.testing.assert_close(y_st[:M].double(), ref[:M], rtol=2e-2, atol=3e-2)
    assert (y_st[:M].float() - y_ws[:M].float()).abs().max().item() <= 0.0625


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,M,N,K", [(2, 64, 64, 256), (3, 100, 37, 50), (1, 1024, 1024, 256), (2, 5, 300, 7)])
def test_bgemm_strided_operands(dtype, B, M, N, K):
    """gf_bgemm: C = alpha A B for plain, transposed and sliced views (no copies); fp32 on the exact-fp32 MFMA."""
    g = torch.Generator().manual_seed(B + M + N + K)
    a = torch.randn(B, M, K, generator=g).to(DEV, dtype)
    bt = torch.randn(B, N, K + 3, generator=g).to(DEV, dtype)[:, :, 1:K + 1]       # B^T stored [N, K] inside a wider buffer
    at = torch.randn(B, K, M, generator=g).to(DEV, dtype)                           # A^T stored [K, M]
    tol = dict(rtol=1e-5, atol=1e-5 * K ** 0.5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2 * K ** 0.5)
    for A, Bm in ((a, bt.transpose(1, 2)), (at.transpose(1, 2), bt.transpose(1, 2))):
        out = ops.bgemm(A, Bm, alpha=0.5)
        ref = 0.5 * torch.matmul(A.double().cpu(), Bm.double().cpu())
        torch.testing.assert_close(out.double().cpu(), ref, **tol)
    # "NN" with a transposed left operand written into a strided output slice
    big = torch.zeros(B, M + 2, N + 5, device=DEV, dtype=dtype)
    ops.bgemm(at.transpose(1, 2), bt.transpose(1, 2), out=big[:, 1:M + 1, 2:N + 2])
    torch.testing.assert_close(big[:, 1:M + 1, 2:N + 2].double().cpu(), torch.matmul(at.transpose(1, 2).double().cpu(), bt.transpose(1, 2).double().cpu()), **tol)
    assert float(big[:, 0].abs().max()) == 0 and float(big[:, :, :2].abs().max()) == 0


@pytest.mark.parametrize("B,L0,L1,N0,N1", [(2, 12, 9, 40, 33), (1, 512, 512, 1200, 1100), (2, 1, 3, 8, 8)])
def test_line_head_kernels_vs_torch_autograd(B, L0, L1, N0, N1):
    """GlueStick line head (gluestick.py:336-376, :772-783): gather + endpoint scores + pairing max + bin-augmented
    double softmax through the HIP kernels vs the stock-torch formulation in fp64, forward and every gradient."""
    D = 256
    g = torch.Generator().manual_seed(L0 + L1)
    x0 = torch.randn(B, N0, D, generator=g, dtype=torch.float64) * 0.3
    x1 = torch.randn(B, N1, D, generator=g, dtype=torch.float64) * 0.3
    idx0 = torch.randint(0, min(N0, 2 * L0), (B, 2 * L0), generator=g)
    idx1 = torch.randint(0, min(N1, 2 * L1), (B, 2 * L1), generator=g)
    beta = torch.tensor(0.7, dtype=torch.float64)
    Gup = torch.randn(B, L0 + 1, L1 + 1, generator=g, dtype=torch.float64)

    def reference(x0, x1, beta):
        g0 = x0.gather(1, idx0[..., None].expand(-1, -1, D))
        g1 = x1.gather(1, idx1[..., None].expand(-1, -1, D))
        s = torch.bmm(g0, g1.transpose(1, 2)) / D ** 0.5
        s = s.reshape(B, L0, 2, L1, 2)
        raw = 0.5 * torch.maximum(s[:, :, 0, :, 0] + s[:, :, 1, :, 1], s[:, :, 0, :, 1] + s[:, :, 1, :, 0])
        r = torch.logsumexp(torch.cat([raw, beta.expand(B, L0, 1)], 2), 2)
        c = torch.logsumexp(torch.cat([raw, beta.expand(B, 1, L1)], 1), 1)
        out = raw.new_zeros(B, L0 + 1, L1 + 1)
        out[:, :L0, :L1] = raw - 0.5 * (r[:, :, None] + c[:, None, :])
        out[:, :L0, L1] = beta - r
        out[:, L0, :L1] = beta - c
        return raw, out

    xr0, xr1, br = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    raw_ref, out_ref = reference(xr0, xr1, br)
    (out_ref * Gup).sum().backward()

    xd0, xd1 = x0.float().to(DEV).requires_grad_(True), x1.float().to(DEV).requires_grad_(True)
    bd = beta.float().to(DEV).requires_grad_(True)
    i0, i1 = idx0.to(DEV), idx1.to(DEV)
    gr0, gr1 = ops.line_graph(i0, N0), ops.line_graph(i1, N1)
    raw = ops.line_pair_scores(ops.rows_gather(xd0, i0, *gr0), ops.rows_gather(xd1, i1, *gr1), D ** -0.5)
    out = ops.dense_log_double_softmax(raw, bd)
    (out * Gup.float().to(DEV)).sum().backward()
    torch.testing.assert_close(raw.detach().double().cpu(), raw_ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.detach().double().cpu(), out_ref.detach(), rtol=1e-5, atol=2e-5)
    for name, a, r in (("dx0", xd0.grad, xr0.grad), ("dx1", xd1.grad, xr1.grad), ("dbeta", bd.grad, br.grad)):
        sc = max(float(r.abs().max()), 1e-6)
        torch.testing.assert_close(a.double().cpu() / sc, r / sc, rtol=1e-4, atol=1e-4, msg=lambda m: f"{name}: {m}")


def test_precast_derived_weights_and_gradient_map():
    """ops.precast(derived=...): prepared weights built inside the per-step cast launch (gf_multi_cast_transpose entries with
    a row gather, a per-row scale, a scalar scale and row blocks of one stacked output; biases in fp32) and their way back
    (gf_weight_grad_map through ops._DerivedWeight) against the torch expressions they replace -- LightGlue's Wqkv row
    order + q scale (lightglue.py:97-128) and the cross block's stacked (to_qk * s | to_v) (:196-221)."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    wq = torch.nn.Parameter(torch.randn(96, 40, device="cuda", generator=g))
    bq = torch.nn.Parameter(torch.randn(96, device="cuda", generator=g))
    wa = torch.nn.Parameter(torch.randn(24, 40, device="cuda", generator=g))
    wb = torch.nn.Parameter(torch.randn(40, 40, 1, device="cuda", generator=g))          # a Conv1d(k=1) weight
    other = torch.nn.Parameter(torch.randn(8, 8, device="cuda", generator=g))
    perm = torch.randperm(96, device="cuda", generator=g)
    rs = torch.rand(96, device="cuda", generator=g) + 0.5
    cperm = torch.randperm(8, device="cuda", generator=g)
    specs = [("q.w", [(wq, perm, rs, 1.0)]), ("q.b", [(bq, perm, rs, 1.0)]),
             ("cat.w", [(wa, None, None, 0.75), (wb, None, None, 1.0)]),
             ("colperm.w", [(other, None, None, 1.0, cperm)])]          # column gather (SuperGlue's merge convolution)
    key = "test-derived"
    ops.precast([wq, bq, wa, wb, other], torch.bfloat16, key=key, derived=specs)
    w1 = ops.derived_weight(key, torch.bfloat16, "q.w", wq)
    b1 = ops.derived_weight(key, torch.bfloat16, "q.b", bq)
    w2 = ops.derived_weight(key, torch.bfloat16, "cat.w", wa, wb)
    assert ops.derived_weight(key, torch.float32, "q.w", wq) is None            # no fp32 precast of this key: caller falls back
    ref_w1 = wq.detach().index_select(0, perm) * rs[:, None]
    ref_b1 = bq.detach().index_select(0, perm) * rs
    ref_w2 = torch.cat([wa.detach() * 0.75, wb.detach().squeeze(-1)], 0)
    lp1, lp2 = ops._lp(w1, torch.bfloat16), ops._lp(w2, torch.bfloat16)
    assert lp1.dtype == torch.bfloat16 and torch.equal(lp1, ref_w1.to(torch.bfloat16))
    assert torch.equal(lp2, ref_w2.to(torch.bfloat16))
    assert torch.equal(ops._wt_t(lp1), ref_w1.to(torch.bfloat16).t().contiguous())
    assert torch.equal(ops._wt_t(lp2, 8, 24), ref_w2.to(torch.bfloat16)[:, 8:24].t().contiguous())
    assert b1.dtype == torch.float32 and torch.allclose(b1, ref_b1, rtol=1e-6, atol=0)
    # gradients: the handle receives the fp32 gradient of the prepared tensor, the sources get theirs
    g1 = torch.randn(96, 40, device="cuda", generator=g)
    gb = torch.randn(96, device="cuda", generator=g)
    g2 = torch.randn(64, 40, device="cuda", generator=g)
    torch.autograd.backward([w1, b1, w2], [g1, gb, g2])
    exp_wq = torch.zeros_like(wq).index_add_(0, perm, g1 * rs[:, None])
    exp_bq = torch.zeros_like(bq).index_add_(0, perm, gb * rs)
    torch.testing.assert_close(wq.grad, exp_wq, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(bq.grad, exp_bq, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(wa.grad, g2[:24] * 0.75, rtol=1e-6, atol=1e-7)
    assert wb.grad.shape == wb.shape and torch.equal(wb.grad.squeeze(-1), g2[24:])
    w3 = ops.derived_weight(key, torch.bfloat16, "colperm.w", other)
    assert torch.equal(ops._lp(w3, torch.bfloat16), other.detach().index_select(1, cperm).to(torch.bfloat16))
    g3 = torch.randn(8, 8, device="cuda", generator=g)
    w3.backward(g3)
    torch.testing.assert_close(other.grad, torch.zeros_like(other).index_add_(1, cperm, g3), rtol=1e-6, atol=1e-7)
    # the launch re-derives from the CURRENT parameter values (optimiser steps do not bump versions under a graph)
    with torch.no_grad():
        wq.mul_(2.0)
    ops.precast([wq, bq, wa, wb, other], torch.bfloat16, key=key, derived=specs)
    w1b = ops.derived_weight(key, torch.bfloat16, "q.w", wq)
    assert torch.equal(ops._lp(w1b, torch.bfloat16), (2.0 * ref_w1).to(torch.bfloat16))


@pytest.mark.parametrize("R,K,N,c0,perm,bias", [(512, 256, 256, 256, False, True), (512, 256, 256, 256, True, True),
                                                (96, 40, 72, 24, True, True), (130, 64, 64, 64, False, False)])
def test_folded_linear_weights_and_gradients(R, K, N, c0, perm, bias):
    """gf_fold_linear_fwd + the cast launch's column-block entries (ops.precast(derived=[(name, "fold", ...)])) and
    gf_fold_linear_bwd vs torch autograd of  W' = [W0[:, :c0] | W0[:, c0:] Wo[:, cperm]],  b' = b0 + W0[:, c0:] bo
    (lightglue.py:131-163 out_proj -> ffn.0; superglue.py:137-160 merge -> mlp.0)."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(R + K)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)                      # noqa: E731
    W0 = torch.nn.Parameter(rnd(R, c0 + K) * 0.1)
    Wo = torch.nn.Parameter(rnd(K, N) * 0.1)
    b0 = torch.nn.Parameter(rnd(R)) if bias else None
    bo = torch.nn.Parameter(rnd(K)) if bias else None
    cperm = torch.randperm(N, device="cuda", generator=g) if perm else None
    params = [p for p in (W0, Wo, b0, bo) if p is not None]
    ops.precast(params, torch.bfloat16, key="foldtest", derived=[("blk.ffn0", "fold", W0, b0, Wo, bo, c0, cperm)])
    out = ops.folded_linear("foldtest", torch.bfloat16, "blk.ffn0", W0, b0, Wo, bo)
    assert out is not None
    handle, bc = out
    wo_g = Wo if cperm is None else Wo[:, cperm]
    ref_w = torch.cat([W0[:, :c0], W0[:, c0:] @ wo_g], 1)
    v = ops._lp(handle, torch.bfloat16)
    assert v.dtype == torch.bfloat16 and v.shape == (R, c0 + N)
    torch.testing.assert_close(v.float(), ref_w.detach().bfloat16().float(), rtol=0, atol=2e-3 * float(ref_w.abs().max()))
    torch.testing.assert_close(ops._wt_t(v).float(), v.float().t())                  # the transposed copy of the same values
    if bias:
        ref_b = b0 + W0[:, c0:] @ bo
        torch.testing.assert_close(bc.detach(), ref_b.detach(), rtol=1e-5, atol=1e-5)
    gW, gb = rnd(R, c0 + N), rnd(R)
    loss = (handle * gW).sum() + ((bc * gb).sum() if bias else 0.0)
    loss.backward()
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    ((ref_w * gW).sum() + ((ref_b * gb).sum() if bias else 0.0)).backward()
    for p, a in zip(params, got):
        torch.testing.assert_close(a, p.grad, rtol=2e-5, atol=2e-5 * float(p.grad.abs().max()))


@pytest.mark.parametrize("B,H,Nq,Nk,scale", [(2, 4, 256, 320, None), (1, 4, 1000, 1024, None), (2, 2, 192, 192, "ln2")])
def test_attention_split_products_are_fp32_equivalent(B, H, Nq, Nk, scale):
    """GF_ATTN_SPLIT (gf_attn_fwd_ex / gf_attn_bwd_acc): P and dS enter the second products as hi + lo bf16 pairs, so on the
    same bf16 operands the kernels agree with an fp64 attention up to the bf16 rounding of the OUTPUTS -- the arithmetic the
    reference prescribes for GlueStick's attention under mixed precision (gluestick.py:524-529, fp32 on autocast inputs).
    Measured as the error before that last rounding would matter: relative L2 error of O, dq, dk, dv against fp64, split vs
    not split."""
    from glue_factory_amd import ops
    g = torch.Generator(device="cuda").manual_seed(Nq + Nk)
    mk = lambda n: (torch.randn(B, n, H, 64, device="cuda", generator=g) * 1.5).bfloat16()      # noqa: E731
    q, k, v = mk(Nq), mk(Nk), mk(Nk)
    do = mk(Nq)
    sc = 64 ** -0.5 if scale is None else ops.LN2
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = _attn_ref(qd, kd, vd, sc)
    (ref * do.double()).sum().backward()
    errs = {}
    for split in (False, True):
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = ops.attention(qq, kk, vv, scale=sc, split=split)
        (out.float() * do.float()).sum().backward()
        rel = lambda a, r: float((a.double() - r).norm() / r.norm())                                # noqa: E731
        errs[split] = (rel(out, ref.detach()), rel(qq.grad, qd.grad), rel(kk.grad, kd.grad), rel(vv.grad, vd.grad))
    print(f"attention {B}x{H}x{Nq}x{Nk}: relative L2 error vs fp64 (O, dq, dk, dv): bf16 products {errs[False]}, split {errs[True]}")
    exact = (ref.detach(), qd.grad, kd.grad, vd.grad)
    rounding = [float((t.bfloat16().double() - t).norm() / t.norm()) for t in exact]      # the fp64 result rounded to bf16
    print(f"   rounding the exact results to bf16 alone: {rounding}")
    for e_split, e_plain, e0 in zip(errs[True], errs[False], rounding):
        assert e_split <= 1.1 * e0                          # nothing left but the rounding of the result itself
        assert e_split <= e_plain * 1.02


@pytest.mark.parametrize("B,N,H,scale", [(2, 128, 4, None), (1, 448, 4, "ln2"), (3, 192, 2, None), (2, 2048, 4, "ln2")])
def test_cross_attention_fused_backward(B, N, H, scale):
    """gf_attn_cross_bwd (csrc/attention_xbwd.hip: both directions of LightGlue's cross attention from ONE score tile per
    image side, lightglue.py:203-216) against the fp64 reference and against the two gf_attn_bwd_acc calls it replaces."""
    D = 64
    g = torch.Generator().manual_seed(N + H)
    p = (torch.randn(2 * B, N, 2, H, D, generator=g) * (1.0 if scale is None else 0.6)).to(DEV, torch.bfloat16)
    dm = torch.randn(2 * B, N, H, D, generator=g).to(DEV, torch.bfloat16)
    sc = D ** -0.5 if scale is None else ops.LN2
    grads = {}
    for fused in (True, False):
        ops.XBWD_ENABLED = fused
        try:
            ps = p.clone().requires_grad_(True)
            m = ops.cross_attention_stacked(ps, scale=sc)
            (m * dm).sum().backward()
            grads[fused] = ps.grad.clone()
        finally:
            ops.XBWD_ENABLED = True
    x = p.detach().cpu().double().requires_grad_(True)
    if N <= 512:
        m0, _ = _attn_ref(x[:B, :, 0], x[B:, :, 0], x[B:, :, 1], sc)
        m1, _ = _attn_ref(x[B:, :, 0], x[:B, :, 0], x[:B, :, 1], sc)
        (torch.cat([m0, m1], 0) * dm.cpu().double()).sum().backward()
        ref = x.grad
        s_ = ref.abs().max().item()
        tol = _tols(torch.bfloat16)
        torch.testing.assert_close(grads[True].cpu().double() / s_, ref / s_, **tol)
        e_f = float((grads[True].cpu().double() - ref).norm() / ref.norm())
        e_o = float((grads[False].cpu().double() - ref).norm() / ref.norm())
        print(f"cross bwd {B}x{N}x{H}: relative L2 error vs fp64: fused {e_f:.2e}, two launches {e_o:.2e}")
        assert e_f <= 1.25 * e_o + 1e-4
    d = float((grads[True].float() - grads[False].float()).norm() / grads[False].float().norm())
    print(f"cross bwd {B}x{N}x{H}: fused vs two launches relative L2 difference {d:.2e}")
    assert d < 8e-3


@pytest.mark.parametrize("M,O,K", [(131072, 32, 2), (65536, 32, 3), (70000, 32, 5), (4097, 16, 8), (100, 1, 1)])
def test_small_linear_forward_and_weight_gradient(M, O, K):
    """ops.small_linear (gf_small_dw): the K <= 8 column input linears of the positional / keypoint / line-endpoint encoders
    (lightglue.py:52-65, superglue.py:82-91, gluestick.py:489-521) vs an fp64 product -- no library GEMM for K = 5 either."""
    g = torch.Generator().manual_seed(M + O + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(O, K, generator=g)
    dy = torch.randn(M, O, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    before = dict(ops.LIBRARY_GEMMS)
    y = ops.small_linear(xd, wd)
    (y * dy.to(DEV)).sum().backward()
    assert dict(ops.LIBRARY_GEMMS) == before
    torch.testing.assert_close(y.detach().cpu().double(), x.double() @ w.double().t(), rtol=1e-5, atol=1e-5)
    ref_dw = dy.double().t() @ x.double()
    sc = float(ref_dw.abs().max())
    torch.testing.assert_close(wd.grad.cpu().double() / sc, ref_dw / sc, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu().double(), dy.double() @ w.double(), rtol=1e-5, atol=1e-5)
