"""GlueStick config 5 (2048 keypoints + 512 lines, B=1): per-tensor gradient error of the bf16 step against the reference's
fp32 golden, with the attention (a) on the bf16 kernels, (b) in fp32 (the reference's AMP semantics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from config_golden import grad_digest_errors, gs_config_inputs, la_digest_error, significant_grads
from glue_factory_amd.matchers.gluestick import GlueStick
from glue_factory_amd.matchers.superglue import AttentionalPropagation
from glue_factory_amd.synthetic import to_device
z, params, data, nl = gs_config_inputs()
cdata = to_device(data, "cuda")
for mode in sys.argv[1:] or ["bf16", "reference"]:
    model = GlueStick({"attention_precision": "reference" if mode == "reference" else "bf16"})
    model.load_state_dict(params, strict=True)
    model = model.cuda().train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(cdata)
        losses, _ = model.loss(pred, {**pred, **cdata})
    losses["total"].mean().backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    big = significant_grads(grad_digest_errors(z, grads))
    rels = sorted(e[1] for e in big.values())
    worst = max((e[1], k) for k, e in big.items())
    print(mode, "la", la_digest_error(z, pred["log_assignment"], int(z["meta"][5])), "grad median", rels[len(rels) // 2], "p90", rels[int(0.9 * len(rels))], "worst", worst, flush=True)
    by = {}
    for k, e in big.items():
        grp = "gnn.layers" if "gnn.layers" in k else "gnn.line" if "line_layers" in k else k.split(".")[0]
        by.setdefault(grp, []).append(e[1])
    print("   by group median:", {g: round(float(np.median(v)), 4) for g, v in by.items()})
