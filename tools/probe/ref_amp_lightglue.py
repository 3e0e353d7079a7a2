"""Build-container only (needs /root/reference): accuracy of the REFERENCE's own mixed precision on the path's headline
models, as a yardstick for the stated bf16 bounds of tests/test_gpu_baseline_configs.py / test_gpu_configs45.py.
The unmodified reference LightGlue (BASELINE configs[0] and the configs[1] matcher, N=2048, L=9) and SuperGlue
(configs[3]) run on the CPU under torch.autocast(bfloat16) -- the arithmetic `--mp bfloat16` selects (train.py:468-476) --
and are compared with their own fp32 runs (the goldens) through the same digests the GPU tests use.

    python tools/probe/ref_amp_lightglue.py [lightglue_config1 lightglue_n2048_l9 superglue_config4]
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
from config_golden import config_inputs, sg_config_inputs  # noqa: E402


def step(name, amp):
    if name.startswith("lightglue"):
        from gluefactory.models.matchers.lightglue import LightGlue
        z, params, data, L = config_inputs(name)
        model = LightGlue({"n_layers": L, "descriptor_dim": 256, "input_dim": 256, "num_heads": 4, "weights": None,
                           "flash": False, "checkpointed": False, "filter_threshold": 0.0})
    else:
        from gluefactory_nonfree.superglue import SuperGlue
        z, params, data, nl, iters = sg_config_inputs(name)
        data["view0"]["image"] = torch.zeros(data["keypoints0"].shape[0], 1, 1024, 1024)      # (shape only, as in gen_golden.py)
        data["view1"]["image"] = torch.zeros(data["keypoints0"].shape[0], 1, 1024, 1024)
        model = SuperGlue({"weights": None, "num_sinkhorn_iterations": iters})
    res = model.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.train()
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
        pred = model(data)
        losses = model.loss(pred, {**pred, **data})
        losses = losses[0] if isinstance(losses, tuple) else losses
    losses["total"].float().mean().backward()
    grads = {k: p.grad.detach().double() for k, p in model.named_parameters() if p.grad is not None}
    la = pred["log_assignment"].detach().float()
    return la, {k: v.detach().float() for k, v in losses.items() if torch.is_tensor(v)}, grads, pred["matches0"].numpy(), z


def run(name):
    la0, loss0, g0, m0, z = step(name, False)
    np.testing.assert_array_equal(m0, z["train.matches0"])            # the fp32 run IS the golden
    la1, loss1, g1, m1, _ = step(name, True)
    d = (la1 - la0).abs().flatten()
    fin = torch.isfinite(d)
    d = d[fin]
    rels = {k: float((g1[k] - g0[k]).norm() / g0[k].norm().clamp(min=1e-30)) for k in g0}
    # analytically-zero gradients (rounding noise in the reference too) are judged as in tests/config_golden.py
    sig = {k: v for k, v in rels.items() if not (k.endswith(".bias") and k[:-5] + ".weight" in g0
                                                and float(g0[k].norm()) < 1e-4 * float(g0[k[:-5] + ".weight"].norm()))}
    srt = sorted(sig.values())
    kw = max(sig, key=sig.get)
    lrel = max(float((loss1[k].mean() - loss0[k].mean()).abs() / loss0[k].mean().abs().clamp(min=1e-12)) for k in loss0)
    print(f"{name}: reference autocast(bf16) vs reference fp32: |d log_assignment| max {float(d.max()):.4f} p99 "
          f"{float(d.quantile(0.99)) if d.numel() < 2 ** 24 else float(d[::8].quantile(0.99)):.4f} mean {float(d.mean()):.4f}; "
          f"worst loss entry {lrel:.2e} relative; per-tensor gradient error median {srt[len(srt) // 2]:.4f} p90 "
          f"{srt[int(0.9 * len(srt))]:.4f} worst {sig[kw]:.4f} ({kw}), {len(srt)} tensors; matches0 equal on "
          f"{(m0 == m1).mean() * 100:.2f} % of the rows", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for n in sys.argv[1:] or ["lightglue_config1", "lightglue_n2048_l9", "superglue_config4"]:
        run(n)
