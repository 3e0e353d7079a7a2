"""Train the HIP SuperGlue of tests/learning_cases.py for lc.STEPS steps (fp32, graph) and save its state_dict (parameters +
BatchNorm buffers) for offline comparison with the reference's: python tools/probe/learn_save_state.py superglue out.pt"""
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
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.synthetic import to_device
    from glue_factory_amd.train_step import TrainStep
    model = tl._model(kind)
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=None, graph=True, graph_warmup=2)
    for i in range(lc.STEPS):
        step(to_device(lc.batch(kind, 1000 + i), "cuda"))
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out)
    print("saved", out, "skipped", step.skipped)


if __name__ == "__main__":
    main()
