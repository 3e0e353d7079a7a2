"""glue_factory_amd.optim.FusedAdam (gf_multi_adam: every parameter tensor in one table-driven launch per 80 tensors)
against torch.optim.Adam -- the optimiser of the reference's training configs (gluefactory/train.py:513)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(g, n_extra=0):
    shapes = [(768, 256), (768,), (256, 256), (1,), (32, 2), (513, 7), (4099,)] + [(5,)] * n_extra
    return [(torch.randn(*s, device="cuda", generator=g) * 0.1) for s in shapes]


@pytest.mark.parametrize("wd", [0.0, 0.01])
@pytest.mark.parametrize("n_extra", [0, 180])           # 180 more tensors: three launches of <= 80 table entries
def test_fused_adam_equals_torch_adam(wd, n_extra):
    from glue_factory_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(3)
    base = _params(g, n_extra)
    a = [torch.nn.Parameter(t.clone()) for t in base]
    b = [torch.nn.Parameter(t.clone()) for t in base]
    ref = torch.optim.Adam(a, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
    ours = FusedAdam(b, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
    for it in range(6):
        if it == 3:                                      # a scheduler step
            for opt in (ref, ours):
                opt.param_groups[0]["lr"] = 1e-3
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, device="cuda", generator=g)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        ref.step()
        ours.step()
        for i, (pa, pb) in enumerate(zip(a, b)):
            torch.testing.assert_close(pb, pa, rtol=2e-6, atol=1e-8, msg=lambda m: f"step {it} tensor {i}: {m}")
    sa, sb = ref.state[a[0]], ours.state[b[0]]
    torch.testing.assert_close(sb["exp_avg"], sa["exp_avg"], rtol=2e-6, atol=3e-7)          # (fma contraction: one rounding fewer)
    torch.testing.assert_close(sb["exp_avg_sq"], sa["exp_avg_sq"], rtol=2e-6, atol=3e-7)
    assert float(sb["step"]) == 6.0


def test_fused_adam_found_inf_skips_update_and_step_count():
    from glue_factory_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(4)
    ps = [torch.nn.Parameter(t) for t in _params(g)]
    opt = FusedAdam(ps, lr=1e-2)
    for p in ps:
        p.grad = torch.randn(p.shape, device="cuda", generator=g)
    opt.step()
    before = [p.detach().clone() for p in ps]
    m_before = opt.state[ps[0]]["exp_avg"].clone()
    opt.found_inf = torch.ones((), device="cuda")
    opt.grad_scale = torch.ones((), device="cuda")
    opt.step()
    del opt.found_inf, opt.grad_scale
    assert all(torch.equal(p, q) for p, q in zip(ps, before))
    assert torch.equal(opt.state[ps[0]]["exp_avg"], m_before) and float(opt.state[ps[0]]["step"]) == 1.0
    opt.found_inf = torch.zeros((), device="cuda")
    opt.grad_scale = torch.full((), 2.0, device="cuda")          # gradients arrive scaled by 2
    opt.step()
    del opt.found_inf, opt.grad_scale
    assert float(opt.state[ps[0]]["step"]) == 2.0 and not torch.equal(ps[0], before[0])


def test_train_step_graph_with_fused_adam_matches_eager():
    """TrainStep(graph=True) with FusedAdam: the captured step (only kernel nodes: the tensor table rides in the kernel
    arguments) equals the eager one over several replays."""
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    data = to_device(make_pairs(2, 128, dim=256, seed=5), "cuda")
    outs = []
    for graph in (False, True):
        torch.manual_seed(0)
        model = LightGlue({"n_layers": 2}).cuda().train()
        step = TrainStep(model, FusedAdam(model.parameters(), lr=1e-3), amp_dtype=torch.bfloat16, device_ids=[0], graph=graph)
        losses = [float(step(data)["total"].mean()) for _ in range(6)]
        outs.append((losses, [p.detach().clone() for p in model.parameters()]))
    for la, lb in zip(*[o[0] for o in outs]):
        assert abs(la - lb) < 5e-3 * max(1.0, abs(la))
    for pa, pb in zip(outs[0][1], outs[1][1]):
        torch.testing.assert_close(pa, pb, rtol=5e-3, atol=5e-4)


def test_fused_adam_on_unaligned_flat_gradient_views():
    """Gradients that are views of one flat buffer at arbitrary element offsets (what train_step.GradBuckets leaves in
    p.grad on a multi-rank run): the 16-byte path is not usable, the scalar one gives the same numbers."""
    from glue_factory_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(9)
    base = _params(g)
    a = [torch.nn.Parameter(t.clone()) for t in base]
    b = [torch.nn.Parameter(t.clone()) for t in base]
    ref, ours = torch.optim.Adam(a, lr=2e-3), FusedAdam(b, lr=2e-3)
    flat = torch.empty(sum(t.numel() + 3 for t in base) + 1, device="cuda")
    for it in range(3):
        off = 1
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, device="cuda", generator=g)
            view = flat[off:off + gr.numel()].view(gr.shape)
            view.copy_(gr)
            pa.grad, pb.grad = gr.clone(), view
            off += gr.numel() + 3
        assert any(p.grad.data_ptr() % 16 for p in b)
        ref.step()
        ours.step()
        for i, (pa, pb) in enumerate(zip(a, b)):
            torch.testing.assert_close(pb, pa, rtol=2e-6, atol=1e-8, msg=lambda m: f"step {it} tensor {i}: {m}")
