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
