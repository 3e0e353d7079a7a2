"""Host-side contract of glue_factory_amd.optim.FusedAdam (no GPU): the optimiser protocol TrainStep relies on."""
import pytest
import torch


def test_fused_adam_declares_the_fused_capturable_protocol_and_rejects_cpu_tensors():
    from glue_factory_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4, 3))
    opt = FusedAdam([p], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01)
    g = opt.param_groups[0]
    assert g["fused"] and g["capturable"] and g["betas"] == (0.9, 0.95) and g["weight_decay"] == 0.01
    assert getattr(opt, "_step_supports_amp_scaling", False)          # found_inf / grad_scale (train_step.TrainStep)
    opt.step()                                                        # no gradients yet: nothing to do, no library needed
    p.grad = torch.ones_like(p)
    with pytest.raises(RuntimeError, match="HIP device"):             # the product path has no CPU fallback
        opt.step()
    with pytest.raises(ValueError):
        FusedAdam([p], lr=-1.0)
    with pytest.raises(ValueError):
        FusedAdam([p], betas=(1.0, 0.9))


def test_fused_adam_state_dict_carries_no_device_scalars_and_loads_torch_adam_checkpoints():
    """the launch's device scalars must not leak into param_groups (a checkpoint restored with
    map_location='cpu' -- the reference's resume flow, train.py:229/256/380 -- would hand host pointers to the kernel)."""
    import copy
    import io
    from glue_factory_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4, 3))
    opt = FusedAdam([p], lr=1e-3)
    sd = opt.state_dict()
    assert set(sd["param_groups"][0]) == {"lr", "betas", "eps", "weight_decay", "fused", "capturable", "params"}
    # a torch.optim.Adam checkpoint (host tensors, non-fused flags) loads and the groups stay fused + capturable
    q = torch.nn.Parameter(torch.zeros(4, 3))
    ref = torch.optim.Adam([q], lr=5e-4, betas=(0.8, 0.9))
    q.grad = torch.ones_like(q)
    ref.step()
    ref.step()
    buf = io.BytesIO()
    torch.save(ref.state_dict(), buf)
    buf.seek(0)
    opt.load_state_dict(torch.load(buf, map_location="cpu"))
    g = opt.param_groups[0]
    assert g["fused"] and g["capturable"] and g["lr"] == 5e-4 and g["betas"] == (0.8, 0.9)
    assert not any(k in g for k in ("amsgrad", "maximize", "foreach", "differentiable")) and not any(k.startswith("_") for k in g)
    assert float(opt.state[p]["step"]) == 2.0
    torch.testing.assert_close(opt.state[p]["exp_avg"], ref.state[q]["exp_avg"])
    bad = copy.deepcopy(ref.state_dict())
    bad["param_groups"][0]["amsgrad"] = True
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)
    opt.sync_lr()           # nothing on a device yet: a no-op, not an error
