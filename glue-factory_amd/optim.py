"""Optimiser step of the train loop (gluefactory/train.py:513 ``optimizer.step()`` with torch.optim.Adam, the
optimiser of every shipped training config) as ONE table-driven HIP launch per <= 80 parameter tensors.

torch's fused Adam walks its tensor lists in 7 launches of at most 320 blocks for LightGlue's 12 M parameters: 0.5 ms
for 336 MB of traffic.  ``FusedAdam`` hands the (parameter, gradient, exp_avg, exp_avg_sq) pointers of every tensor to
``gf_multi_adam`` by value in the kernel arguments -- no device-side table to refresh when autograd re-allocates the
gradients, only kernel nodes in a captured step -- and every 4096-element chunk is a block.

Same numbers as ``torch.optim.Adam(amsgrad=False, maximize=False)`` (tests/test_gpu_optim.py), same ``state_dict``
layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter; the device scalars of the launch live outside
``param_groups``), checkpoints of either optimiser load into the other (resume flow of train.py:229-256/380), and the GradScaler protocol ``TrainStep`` uses for its
device-side skip: a ``found_inf`` attribute > 0 leaves parameters, moments and the step count untouched.
"""
import struct

import torch

from . import lib as _lib


class FusedAdam(torch.optim.Optimizer):
    """Restriction (documented, checked nowhere else): ONE step count per parameter group -- a parameter that receives
    its first gradient later than its group's first update is bias-corrected with the group's count, where
    torch.optim.Adam counts per parameter.  Every parameter of the matchers gets a gradient in every step."""
    _step_supports_amp_scaling = True          # step() honours self.found_inf / self.grad_scale (device scalars)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # `fused` / `capturable`: what TrainStep(graph=True) asks of an optimiser (no host synchronisation, device-side
        # step count and learning rate)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True, capturable=True))
        # device scalars (step count, learning rate) per group.  NOT in param_groups: state_dict() would serialise them and
        # load_state_dict(torch.load(map_location="cpu")) -- the reference's resume flow, train.py:229/256/380 -- would hand
        # host pointers to the kernel.  Only state[p]["step"] travels in a checkpoint, as in torch.optim.Adam.
        self._dev = {}

    def _scalars(self, gi, group, dev):
        d = self._dev.get(gi)
        if d is None or d["step"].device != dev:
            d = self._dev[gi] = {"step": torch.zeros((), dtype=torch.float32, device=dev),
                                 "lr": torch.full((), float(group["lr"]), dtype=torch.float32, device=dev),
                                 "lr_host": float(group["lr"])}
        return d

    def _init_group(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None and p.numel() > 0]
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise RuntimeError("FusedAdam: fp32 parameters and dense fp32 gradients on the HIP device only")
        if not ps:
            return ps, None
        dev = ps[0].device
        d = self._scalars(gi, group, dev)
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = d["step"]                      # shared tensor: torch keeps one (equal) count per parameter
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            else:
                if st.get("step") is not d["step"]:
                    # state that came through load_state_dict (ours or a torch.optim.Adam checkpoint): a python number, a
                    # host tensor or a per-parameter device copy -- the group's device scalar takes its value
                    d["step"].copy_(torch.as_tensor(st.get("step", 0.0), dtype=torch.float32).reshape(()))
                    st["step"] = d["step"]
                for k in ("exp_avg", "exp_avg_sq"):         # a checkpoint loaded with map_location="cpu" and cast by hand
                    if st[k].device != dev or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                        st[k] = st[k].to(device=dev, dtype=torch.float32).contiguous()
        return ps, d

    def load_state_dict(self, state_dict):
        """Accepts our own checkpoints and torch.optim.Adam's (whatever its fused / capturable / foreach flags were and
        wherever torch.load mapped the tensors): moments go to the parameters' device, the step count into the group's
        device scalar at the next step(), and the groups stay fused + capturable.  Safe after TrainStep(graph=True) has
        captured the step: device scalars and moment tensors that already exist keep their addresses and are overwritten in place."""
        for g in state_dict["param_groups"]:
            if g.get("amsgrad") or g.get("maximize"):
                raise ValueError("FusedAdam: amsgrad / maximize checkpoints are not supported")
        live = {p_: (st["exp_avg"], st["exp_avg_sq"]) for p_, st in self.state.items() if "exp_avg" in st}
        super().load_state_dict(state_dict)
        for p_, olds in live.items():           # moments a captured step already points at take the loaded values IN PLACE
            st = self.state.get(p_)
            if st is not None and "exp_avg" in st:
                for k, old in zip(("exp_avg", "exp_avg_sq"), olds):
                    if old.shape == st[k].shape:
                        old.copy_(st[k])
                        st[k] = old
        foreign = ("amsgrad", "maximize", "foreach", "differentiable", "decoupled_weight_decay")
        for g in self.param_groups:
            g["fused"], g["capturable"] = True, True
            for k in foreign + tuple(k for k in g if k.startswith("_")):      # (older checkpoints of ours leaked _step_dev ...)
                g.pop(k, None)
        # the device scalars stay where they are -- a captured step (TrainStep(graph=True)) holds their addresses -- and
        # take the loaded values IN PLACE: the step count at the next _init_group (state[p]["step"] is no longer the
        # shared tensor), the learning rate here
        for gi, g in enumerate(self.param_groups):
            d = self._dev.get(gi)
            if d is not None:
                d["lr_host"] = float(g["lr"])
                d["lr"].fill_(d["lr_host"])
                steps = [self.state[p_].get("step") for p_ in g["params"] if p_ in self.state]
                if steps:
                    d["step"].copy_(torch.as_tensor(steps[0], dtype=torch.float32).reshape(()))
                for p_ in g["params"]:
                    if p_ in self.state:
                        self.state[p_]["step"] = d["step"]

    def sync_lr(self):
        """Copy each group's (host-side, scheduler-owned) learning rate into its device scalar when it moved.  step() does
        it; a hipGraph REPLAY of step() does not run this python, so TrainStep calls it in front of every replay."""
        for gi, group in enumerate(self.param_groups):
            d = self._dev.get(gi)
            if d is not None and float(group["lr"]) != d["lr_host"]:
                d["lr_host"] = float(group["lr"])
                d["lr"].fill_(d["lr_host"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        found_inf = getattr(self, "found_inf", None)
        grad_scale = getattr(self, "grad_scale", None)
        for gi, group in enumerate(self.param_groups):
            ps, d = self._init_group(gi, group)
            if not ps:
                continue
            if float(group["lr"]) != d["lr_host"]:          # a scheduler moved it (host side; outside a capture)
                d["lr_host"] = float(group["lr"])
                d["lr"].fill_(d["lr_host"])
            for p in ps:
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdam: contiguous parameters and gradients only")
            rec = b"".join(
                struct.pack("<QQQQqii", p.data_ptr(), p.grad.data_ptr(),
                            self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel(), 0, 0)
                for p in ps)
            assert len(rec) == len(ps) * L.gf_adam_entry_bytes()
            b1, b2 = group["betas"]
            st = torch.cuda.current_stream(ps[0].device).cuda_stream
            _lib.check(L.gf_multi_adam(rec, len(ps), d["lr"].data_ptr(), d["step"].data_ptr(),
                                       0 if found_inf is None else found_inf.data_ptr(),
                                       0 if grad_scale is None else grad_scale.data_ptr(),
                                       float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), st),
                       "gf_multi_adam")
        return loss
