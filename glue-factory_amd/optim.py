"""Optimiser step of the train loop (gluefactory/train.py:513 ``optimizer.step()`` with torch.optim.Adam, the
optimiser of every shipped training config) as ONE table-driven HIP launch per <= 80 parameter tensors.

torch's fused Adam walks its tensor lists in 7 launches of at most 320 blocks for LightGlue's 12 M parameters: 0.5 ms
for 336 MB of traffic.  ``FusedAdam`` hands the (parameter, gradient, exp_avg, exp_avg_sq) pointers of every tensor to
``gf_multi_adam`` by value in the kernel arguments -- no device-side table to refresh when autograd re-allocates the
gradients, only kernel nodes in a captured step -- and every 4096-element chunk is a block.

Same numbers as ``torch.optim.Adam(amsgrad=False, maximize=False)`` (tests/test_gpu_optim.py), same ``state_dict``
layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), and the GradScaler protocol ``TrainStep`` uses for its
device-side skip: a ``found_inf`` attribute > 0 leaves parameters, moments and the step count untouched.
"""
import struct

import torch

from . import lib as _lib


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True          # step() honours self.found_inf / self.grad_scale (device scalars)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # `fused` / `capturable`: what TrainStep(graph=True) asks of an optimiser (no host synchronisation, device-side
        # step count and learning rate)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True, capturable=True))

    def _init_group(self, group):
        ps = [p for p in group["params"] if p.grad is not None]
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise RuntimeError("FusedAdam: fp32 parameters and dense fp32 gradients on the HIP device only")
        if not ps:
            return ps
        dev = ps[0].device
        if "_step_dev" not in group:            # ONE step count / learning rate per group, on the device
            group["_step_dev"] = torch.zeros((), dtype=torch.float32, device=dev)
            group["_lr_dev"] = torch.full((), float(group["lr"]), dtype=torch.float32, device=dev)
            group["_lr_host"] = float(group["lr"])
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = group["_step_dev"]             # shared tensor: torch keeps one (equal) count per parameter
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            elif st["step"] is not group["_step_dev"]:      # state loaded from a torch.optim.Adam checkpoint
                group["_step_dev"].copy_(torch.as_tensor(st["step"], dtype=torch.float32))
                st["step"] = group["_step_dev"]
        return ps

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        found_inf = getattr(self, "found_inf", None)
        grad_scale = getattr(self, "grad_scale", None)
        for group in self.param_groups:
            ps = self._init_group(group)
            if not ps:
                continue
            if float(group["lr"]) != group["_lr_host"]:     # a scheduler moved it (host side; outside a capture)
                group["_lr_host"] = float(group["lr"])
                group["_lr_dev"].fill_(group["_lr_host"])
            rec = b"".join(
                struct.pack("<QQQQqii", p.data_ptr(), (p.grad if p.grad.is_contiguous() else p.grad.contiguous()).data_ptr(),
                            self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel(), 0, 0)
                for p in ps)
            for p in ps:
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdam: contiguous parameters and gradients only")
            assert len(rec) == len(ps) * L.gf_adam_entry_bytes()
            b1, b2 = group["betas"]
            st = torch.cuda.current_stream(ps[0].device).cuda_stream
            _lib.check(L.gf_multi_adam(rec, len(ps), group["_lr_dev"].data_ptr(), group["_step_dev"].data_ptr(),
                                       0 if found_inf is None else found_inf.data_ptr(),
                                       0 if grad_scale is None else grad_scale.data_ptr(),
                                       float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), st),
                       "gf_multi_adam")
        return loss
