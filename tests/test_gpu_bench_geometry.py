"""Parity AT THE BENCHMARKED GEOMETRY (bench.py: B=32 pairs, N=2048, L=9 / 18 GNN layers / 100 Sinkhorn iterations):
the code paths that only a large batch takes.

  * Sinkhorn: the multi-chunk path (`batch_chunk()` splits B=32, N=2048 into 2 chunks of 16 with per-chunk pointer /
    history-stride arithmetic, csrc/sinkhorn.hip) and the generic `sk_*` kernels (N+1 > 2304) against the fp64 oracle
    of superglue.py:186-214 (oracle/sinkhorn_oracle.py), forward and backward; every pair of the big batch additionally
    against a single-chunk launch of the same pair;
  * batch consistency of the whole train step at B=32 (the 64-image x 4-head x 2048^2 launch geometry of the attention /
    GEMM / loss kernels): LightGlue, SuperGlue and GlueStick, fp32 and bf16 -- the outputs of pairs 0 / 15 / 16 / 31
    equal the B=1 runs of those pairs (which tests/test_gpu_baseline_configs.py / test_gpu_configs45.py pin to the
    reference), and the batch gradient equals the mean of the 32 per-pair gradients.  Image pairs are independent in all
    three matchers except through train-mode BatchNorm statistics (SuperGlue / GlueStick), so those modules run with
    their BatchNorm layers in eval mode here (running statistics; everything else in train mode).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 32
PAIRS = (0, 15, 16, 31)


# ------------------------------------------------------------------------------------------------ Sinkhorn
def _oracle_pair(scores_b, alpha, T, G_b):
    """fp64 oracle of ONE pair: (out, dZ) for the upstream gradient G_b."""
    from oracle import sinkhorn_oracle as so
    M, N = scores_b.shape
    Zc = so.couplings(scores_b[None].double(), alpha.double()).requires_grad_(True)
    lmu, lnu, norm = so.marginals(M, N, Zc)
    ref, _, _ = so.sinkhorn(Zc, lmu, lnu, T)
    ref = ref - norm
    (ref * G_b[None].double()).sum().backward()
    return ref.detach()[0], Zc.grad[0]


@pytest.mark.parametrize("Bsz,M,N,T,pairs", [
    (20, 2048, 2048, 3, (0, 9, 10, 19)),          # 2 chunks of 10
    (32, 2048, 2048, 5, (0, 15, 16, 31)),         # 2 chunks of 16: the benchmarked geometry
    (3, 2400, 2400, 4, (0, 2)),                   # N + 1 > 2304: the generic sk_* kernels
    (2, 2310, 2500, 3, (1,)),                     # generic, ragged
])
def test_sinkhorn_multichunk_and_generic_paths_vs_oracle(Bsz, M, N, T, pairs):
    from glue_factory_amd import ops
    from oracle import sinkhorn_oracle as so
    g = torch.Generator().manual_seed(Bsz * 7 + M + T)
    scores = torch.randn(Bsz, M, N, generator=g) * 2.5
    alpha = torch.tensor(0.7)
    Gd = torch.randn(Bsz, M + 1, N + 1, generator=g)
    Z = so.couplings(scores, alpha).float()                       # [B, M+1, N+1]
    Zd = Z.cuda().requires_grad_(True)
    out = ops.sinkhorn(Zd, T)
    (out * Gd.cuda()).sum().backward()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    worst_o = worst_g = 0.0
    for b in pairs:
        ref, gref = _oracle_pair(scores[b], alpha, T, Gd[b])
        worst_o = max(worst_o, float((out[b].detach().cpu().double() - ref).abs().max()))
        sc = float(gref.abs().max())
        worst_g = max(worst_g, float((Zd.grad[b].cpu().double() - gref).abs().max()) / sc)
    print(f"sinkhorn B={Bsz} {M}x{N} T={T}: max|out - fp64| {worst_o:.2e}, max|dZ - fp64|/max|dZ| {worst_g:.2e} on pairs {pairs}")
    assert worst_o < 1e-4 and worst_g < 5e-4
    # every pair of the batch == the same pair launched alone (one chunk, chunk offset 0)
    # (the two launch geometries are not bit-identical: a pair's rows are cut into row blocks at different places)
    wo = wg = 0.0
    for b in range(Bsz):
        z1 = Z[b:b + 1].cuda().requires_grad_(True)
        o1 = ops.sinkhorn(z1, T)
        (o1 * Gd[b:b + 1].cuda()).sum().backward()
        wo = max(wo, float((o1.detach() - out[b:b + 1].detach()).abs().max()))
        wg = max(wg, float((z1.grad - Zd.grad[b:b + 1]).abs().max()) / float(z1.grad.abs().max()))
    print(f"   batch vs single-pair launches, all {Bsz} pairs: max |d out| {wo:.2e}, max |d dZ| / max|dZ| {wg:.2e}")
    # (single pairs run the streaming kernels, the batch the chip-resident sweeps: two summation orders of the column
    # partials; 1.5e-5 is ONE fp32 ulp of an output in [128, 256))
    assert wo < 4e-5 and wg < 1e-4      # (measured 1.5e-5 / 4.1e-5; each schedule is within 5e-4 of the fp64 oracle above)


# ------------------------------------------------------------------------------------------------ batch consistency
def _slice(data, i):
    if isinstance(data, dict):
        return {k: _slice(v, i) for k, v in data.items()}
    if torch.is_tensor(data) and data.dim() > 0 and data.shape[0] == B:
        return data[i:i + 1]
    return data


def _bn_eval(model):
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()


def _step(model, data, bf16, keys):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = model(data)
        losses, _ = model.loss(pred, {**pred, **data})
    losses["total"].mean().backward()
    outs = {k: pred[k].detach().clone() for k in keys}
    outs["loss.total"] = losses["total"].detach().clone()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return outs, grads


def _consistency(tag, model, data, bf16, keys, out_tol, grad_tol):
    outs_b, grads_b = _step(model, data, bf16, keys)
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in grads_b.items()}
    worst_out = {}
    for i in range(B):
        o1, g1 = _step(model, _slice(data, i), bf16, keys if i in PAIRS else ())
        assert set(g1) == set(grads_b)
        for k, v in g1.items():
            acc[k] += v.double()
        if i in PAIRS:
            for k, v in o1.items():
                a, r = outs_b[k][i:i + 1], v
                if a.dtype in (torch.int64, torch.int32):
                    assert torch.equal(a, r), (tag, k, i)
                else:
                    worst_out[k] = max(worst_out.get(k, 0.0), float((a.float() - r.float()).abs().max()))
    rels = {}
    for k, gb in grads_b.items():
        mean = acc[k] / B
        rels[k] = float((gb.double() - mean).norm() / mean.norm().clamp(min=1e-30))
    # gradients that are analytically zero hold rounding noise only (e.g. key biases the softmax cancels): judged by
    # their size relative to the layer's weight gradient
    # (SuperGlue / GlueStick key bias `attn.proj.1.bias`: softmax is invariant to a shift of all its logits of a row)
    sig = {k: v for k, v in rels.items()
           if not k.endswith("attn.proj.1.bias")
           and not (k.endswith(".bias") and k[:-5] + ".weight" in grads_b
                    and float(grads_b[k].norm()) < 1e-4 * float(grads_b[k[:-5] + ".weight"].norm()))}
    k_w = max(sig, key=sig.get)
    print(f"{tag} {'bf16' if bf16 else 'fp32'} B={B}: pairs {PAIRS} vs their B=1 runs: max |d| {worst_out}; batch gradient vs "
          f"mean of {B} per-pair gradients: worst relative error {sig[k_w]:.2e} ({k_w}), tensors {len(sig)}")
    for k, v in worst_out.items():
        assert v <= out_tol, (tag, k, v)
    assert sig[k_w] <= grad_tol, (tag, k_w, sig[k_w])


@pytest.mark.parametrize("bf16", [False, True])
def test_lightglue_b32_step_is_the_sum_of_its_pairs(bf16):
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import lightglue_oracle as lgo
    model = LightGlue({"n_layers": 9, "filter_threshold": 0.0})
    model.load_state_dict(lgo.init_params(9, 256, 4, seed=141), strict=True)
    model = model.cuda().train()
    data = to_device(make_pairs(B, 2048, dim=256, size=(1024, 1024), seed=142), "cuda")
    _consistency("lightglue", model, data, bf16, ("log_assignment", "matches0", "matches1", "matching_scores0"),
                 out_tol=1e-5 if not bf16 else 1e-5, grad_tol=2e-5 if not bf16 else 2e-4)


@pytest.mark.parametrize("bf16", [False, True])
def test_superglue_b32_step_is_the_sum_of_its_pairs(bf16):
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import superglue_oracle as sgo
    model = SuperGlue({"num_sinkhorn_iterations": 100})
    model.load_state_dict(sgo.init_params(256, gnn_layers=18, seed=143), strict=True)
    model = model.cuda().train()
    _bn_eval(model)
    data = to_device(make_pairs(B, 2048, dim=256, size=(1024, 1024), seed=144), "cuda")
    _consistency("superglue", model, data, bf16, ("log_assignment", "matches0", "matching_scores0"),
                 out_tol=2e-5, grad_tol=5e-5 if not bf16 else 5e-4)


@pytest.mark.parametrize("bf16", [False, True])
def test_gluestick_b32_step_is_the_sum_of_its_pairs(bf16):
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs, to_device
    from oracle import gluestick_oracle as gso
    model = GlueStick({})
    model.load_state_dict(gso.init_params(256, gnn_layers=18, inter=None, seed=145), strict=True)
    model = model.cuda().train()
    _bn_eval(model)
    data = to_device(make_point_line_pairs(B, 2048, 512, dim=256, size=(1024, 1024), seed=146), "cuda")
    _consistency("gluestick", model, data, bf16,
                 ("log_assignment", "line_log_assignment", "matches0", "line_matches0", "raw_line_scores"),
                 out_tol=2e-5, grad_tol=5e-5 if not bf16 else 5e-4)
