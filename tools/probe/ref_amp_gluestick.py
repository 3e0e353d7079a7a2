"""Build-container only (needs /root/reference): what does the REFERENCE's own mixed-precision arithmetic cost GlueStick in
gradient accuracy?  Runs the unmodified reference GlueStick at BASELINE configs[4] (2048 keypoints + 512 lines, B=1) on the
CPU under torch.autocast(bfloat16) with its attention forced to fp32 exactly as `@AMP_CUSTOM_FWD_F32` does on CUDA
(gluestick.py:18-22, 524-529: inputs cast to fp32, autocast disabled inside), and measures the per-tensor gradient error
against the reference's fp32 run (the golden) with the same digest the GPU probe (tools/probe/gs_precision.py) uses.

    python tools/probe/ref_amp_gluestick.py            # prints median / p90 / worst per-tensor gradient error
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.append("/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
from config_golden import grad_digest_errors, gs_config_inputs, la_digest_error, significant_grads  # noqa: E402

import gluefactory.models.matchers.gluestick as ref_gs  # noqa: E402

_attention = ref_gs.attention.__wrapped__ if hasattr(ref_gs.attention, "__wrapped__") else ref_gs.attention


def attention_fp32(query, key, value):
    """custom_fwd(cast_inputs=torch.float32): floating inputs cast up, the body runs with autocast disabled."""
    with torch.autocast("cpu", enabled=False):
        return _attention(query.float(), key.float(), value.float())


# The reference's LineLayer does not run under autocast at all (torch 2.10): `update0.scatter_reduce_(src=lupdate0)` scatters
# the bf16 MLP output into an fp32 zeros_like(ldesc0) -> "scatter(): Expected self.dtype to be equal to src.dtype"
# (gluestick.py:672).  The smallest change that lets it run -- the source cast up to the destination's dtype -- is applied
# HERE, from outside, so the measurement below can exist; the reference file is untouched.
_scatter_reduce_ = torch.Tensor.scatter_reduce_


def _scatter_reduce_cast(self, dim, index, src, reduce, *, include_self=True):
    return _scatter_reduce_(self, dim, index, src.to(self.dtype), reduce, include_self=include_self)


def run(mode):
    z, params, data, nl = gs_config_inputs()
    model = ref_gs.GlueStick({"weights": None})
    model.load_state_dict(params, strict=True)
    model.train()
    ref_gs.attention = attention_fp32 if mode == "amp_attention_fp32" else _attention
    torch.Tensor.scatter_reduce_ = _scatter_reduce_ if mode == "fp32" else _scatter_reduce_cast
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=mode != "fp32"):
        pred = model(data)
        losses, _ = model.loss(pred, {**pred, **data})
    losses["total"].float().mean().backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    big = significant_grads(grad_digest_errors(z, grads))
    rels = sorted(e[1] for e in big.values())
    worst = max((e[1], k) for k, e in big.items())
    print(f"{mode}: log-assignment (max, p99, mean) {la_digest_error(z, pred['log_assignment'], int(z['meta'][5]))}; "
          f"per-tensor gradient error vs the fp32 golden: median {rels[len(rels) // 2]:.4f} p90 {rels[int(0.9 * len(rels))]:.4f} "
          f"worst {worst[0]:.4f} ({worst[1]}), {len(rels)} tensors", flush=True)
    by = {}
    for k, e in big.items():
        grp = "gnn.layers" if "gnn.layers" in k else "gnn.line" if "line_layers" in k else k.split(".")[0]
        by.setdefault(grp, []).append(e[1])
    print("   by group median:", {g: round(float(np.median(v)), 4) for g, v in by.items()}, flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for m in sys.argv[1:] or ["fp32", "amp_attention_fp32", "amp_all_bf16"]:
        run(m)
