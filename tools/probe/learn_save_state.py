"""Train the HIP SuperGlue / GlueStick of tests/learning_cases.py for lc.STEPS steps (TrainStep: fused Adam, one hipGraph; fp32 or
bf16 autocast) and save its state_dict (parameters + BatchNorm buffers) for offline comparison with the reference's
(tools/probe/learn_anchor_report.py):  python tools/probe/learn_save_state.py superglue|gluestick out.pt [fp32|bf16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import learning_cases as lc  # noqa: E402
import test_gpu_zz_learning as tl  # noqa: E402


def main():
    kind, out = sys.argv[1], sys.argv[2]
    bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import to_device
    from glue_factory_amd.train_step import TrainStep
    torch.set_num_threads(8)
    model = tl._model(kind)
    print(kind, "bf16" if bf16 else "fp32", "held-out before:", [round(v, 4) for v in tl._evaluate(kind, model, bf16)], flush=True)
    model.train()
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16 if bf16 else None, graph=True,
                     graph_warmup=2)
    trace = []
    for i in range(int(os.environ.get("GF_STEPS", lc.STEPS))):
        res = step(to_device(lc.batch(kind, 1000 + i), "cuda"))
        if i % 50 == 49:
            trace.append(round(float(res["total"].mean()), 3))
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out)
    print(kind, "bf16" if bf16 else "fp32", "train loss every 50 steps:", trace, "skipped", step.skipped)
    print(kind, "bf16" if bf16 else "fp32", "held-out after (eval mode):", [round(v, 4) for v in tl._evaluate(kind, model, bf16)], flush=True)
    step.close()


if __name__ == "__main__":
    main()
