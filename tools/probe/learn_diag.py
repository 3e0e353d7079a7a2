"""Diagnosis of tests/test_gpu_zz_learning.py for the BatchNorm matchers: (1) the reference's TRAINED state (saved by
GF_SAVE_FINAL=... tools/probe/ref_learning_curve.py) loaded into the HIP module -> held-out eval must equal the reference's;
(2) the HIP module trained here (graph and eager) -> its BatchNorm buffers against the reference's, layer by layer.
python tools/probe/learn_diag.py superglue tools/probe/build/ref_sg_final.pt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import learning_cases as lc  # noqa: E402
import test_gpu_zz_learning as tl  # noqa: E402


def main():
    kind, path = sys.argv[1], sys.argv[2]
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.train_step import TrainStep
    ref_state = torch.load(path, map_location="cpu")
    model = tl._model(kind)
    model.load_state_dict(ref_state, strict=True)
    print("reference-trained state in the HIP module, held-out eval (fp32):", [round(v, 4) for v in tl._evaluate(kind, model, False)])
    print("                                                        (bf16):", [round(v, 4) for v in tl._evaluate(kind, model, True)])
    for graph in (True,):
        model = tl._model(kind)
        step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=None, graph=graph, graph_warmup=2)
        for i in range(lc.STEPS):
            step(tl._batch(kind, 1000 + i))
        print(f"trained here (fp32, graph={graph}): held-out eval", [round(v, 4) for v in tl._evaluate(kind, model, False)])
        sd = model.state_dict()
        worst = []
        for k, v in sd.items():
            r = ref_state[k].float()
            d = float((v.float().cpu() - r).norm() / r.norm().clamp(min=1e-12))
            worst.append((d, k))
        worst.sort(reverse=True)
        print("  largest relative differences to the reference's trained state:")
        for d, k in worst[:24]:
            print(f"    {k:50s} {d:.3e}")
        nb = [k for k in sd if k.endswith("num_batches_tracked")]
        print("  num_batches_tracked:", {k: (int(sd[k]), int(ref_state[k])) for k in nb[:4]})
        step.close()


if __name__ == "__main__":
    main()
