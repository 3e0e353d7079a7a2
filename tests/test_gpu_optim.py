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


@pytest.mark.parametrize("source", ["fused", "torch"])
def test_fused_adam_resumes_from_a_cpu_mapped_checkpoint(source, tmp_path):
    """The reference's resume flow (train.py:229/256 torch.load(map_location='cpu'), :380 optimizer.load_state_dict): a
    FusedAdam restored from its own or from a torch.optim.Adam checkpoint continues exactly like torch.optim.Adam."""
    from glue_factory_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(11)
    base = _params(g)
    a = [torch.nn.Parameter(t.clone()) for t in base]
    b = [torch.nn.Parameter(t.clone()) for t in base]
    ref = torch.optim.Adam(a, lr=2e-3, weight_decay=0.01)
    first = FusedAdam(b, lr=2e-3, weight_decay=0.01) if source == "fused" else torch.optim.Adam(b, lr=2e-3, weight_decay=0.01)
    grads = [[torch.randn(t.shape, device="cuda", generator=g) for t in base] for _ in range(5)]

    def run(opt, ps, its):
        for it in its:
            for p_, gr in zip(ps, grads[it]):
                p_.grad = gr.clone()
            opt.step()

    run(ref, a, range(5))
    run(first, b, range(2))
    torch.save(first.state_dict(), tmp_path / "opt.tar")
    sd = torch.load(tmp_path / "opt.tar", map_location="cpu")
    assert not any(k.startswith("_") for k in sd["param_groups"][0])
    ours = FusedAdam(b, lr=1.0)                       # hyper-parameters come from the checkpoint
    ours.load_state_dict(sd)
    run(ours, b, range(2, 5))
    for i, (pa, pb) in enumerate(zip(a, b)):
        torch.testing.assert_close(pb, pa, rtol=4e-6, atol=1e-8, msg=lambda m: f"tensor {i}: {m}")
    assert float(ours.state[b[0]]["step"]) == 5.0 and ours.state[b[0]]["step"].is_cuda


def test_train_step_graph_follows_a_learning_rate_schedule():
    """the reference steps its scheduler every iteration (train.py:517); a REPLAYED step must see the new
    learning rate (device scalar refreshed by TrainStep in front of the replay), like the eager one."""
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    data = to_device(make_pairs(2, 128, dim=256, seed=5), "cuda")
    outs = []
    for graph in (False, True):
        torch.manual_seed(0)
        model = LightGlue({"n_layers": 2}).cuda().train()
        opt = FusedAdam(model.parameters(), lr=1e-3)
        step = TrainStep(model, opt, amp_dtype=torch.bfloat16, device_ids=[0], graph=graph)
        for it in range(7):
            if it >= 4:                             # replays (graph_warmup = 2, capture at call 3): the schedule moves
                opt.param_groups[0]["lr"] = 1e-3 * 0.5 ** (it - 3)
            step(data)
        outs.append([p.detach().clone() for p in model.parameters()])
    frozen = []
    torch.manual_seed(0)
    model = LightGlue({"n_layers": 2}).cuda().train()
    step = TrainStep(model, FusedAdam(model.parameters(), lr=1e-3), amp_dtype=torch.bfloat16, device_ids=[0], graph=False)
    for it in range(7):
        step(data)
    frozen = [p.detach().clone() for p in model.parameters()]
    err = max(float((pa - pb).abs().max()) for pa, pb in zip(*outs))
    gap = max(float((pa - pf).abs().max()) for pa, pf in zip(outs[0], frozen))
    assert gap > 5e-4, "the schedule must matter for this test to say anything"
    assert err < 0.1 * gap, (err, gap)


def test_load_state_dict_after_the_step_was_captured_keeps_the_graph_valid():
    """FusedAdam.load_state_dict used to drop its device scalars; a hipGraph captured earlier kept pointing at
    the old step / lr scalars (and at the old moment tensors).  A checkpoint restored AFTER the capture must continue
    exactly like an eager optimiser restored from the same checkpoint."""
    import copy
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import make_pairs, to_device
    from glue_factory_amd.train_step import TrainStep
    data = to_device(make_pairs(2, 128, dim=256, seed=5), "cuda")
    outs = []
    for graph in (False, True):
        torch.manual_seed(0)
        model = LightGlue({"n_layers": 2}).cuda().train()
        opt = FusedAdam(model.parameters(), lr=1e-3)
        step = TrainStep(model, opt, amp_dtype=torch.bfloat16, device_ids=[0], graph=graph)
        for _ in range(2):
            step(data)
        ckpt = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))       # after 2 steps
        for _ in range(3):                                                                # capture at call 3 + replays
            step(data)
        model.load_state_dict(ckpt[0])
        sd = copy.deepcopy(ckpt[1])
        sd["param_groups"][0]["lr"] = 5e-4                                                # the checkpoint's own lr
        opt.load_state_dict(sd)
        for _ in range(3):
            step(data)
        outs.append(([p.detach().clone() for p in model.parameters()], float(opt.state[next(model.parameters())]["step"])))
    (pa, sa), (pb, sb) = outs
    assert sa == sb == 5.0
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-5)       # (bf16 step, graph vs eager: as the other replay tests)
