"""Build-container only: what separates a HIP-trained SuperGlue state from the reference's after the 300 steps of
tests/test_gpu_zz_learning.py?  Inputs: a HIP-trained state (tools/probe/learn_save_state.py on the GPU box) and two reference
states (GF_THREADS=3 / 5 GF_SAVE_FINAL=... tools/probe/ref_learning_curve.py superglue), all evaluated with the REFERENCE
module on the CPU.  Prints (1) held-out loss in eval / train mode and with re-estimated BatchNorm statistics, (2) parameter /
buffer hybrids, (3) drift-from-initialisation correlations HIP-vs-reference against reference-vs-reference.
python tools/probe/learn_offline.py ours.pt ref_a.pt ref_b.pt        (results: profiles/r05e_learning_curve_reference_cpu.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.append("/root/reference")
import torch  # noqa: E402

import learning_cases as lc  # noqa: E402


def main():
    from gluefactory.models.utils.metrics import matcher_metrics
    from gluefactory_nonfree.superglue import SuperGlue
    torch.set_num_threads(4)
    ours, ref, ref_b = (torch.load(p, map_location="cpu") for p in sys.argv[1:4])
    init = lc.initial_params("superglue")
    held = [lc.batch("superglue", s) for s in lc.HELD_OUT]
    isbuf = lambda k: "running_" in k or "num_batches" in k          # noqa: E731

    def evaluate(state, mode="eval", reestimate=0):
        model = SuperGlue({**lc.conf("superglue"), "weights": None})
        model.load_state_dict(state, strict=True)
        if reestimate:            # population statistics: cumulative average over fresh batches
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
                    m.momentum = None
            model.train()
            with torch.no_grad():
                for s in range(reestimate):
                    model(lc.batch("superglue", 5000 + s))
        model.train() if mode == "train" else model.eval()
        rows = []
        with torch.no_grad():
            for data in held:
                pred = model(data)
                losses = model.loss(pred, {**pred, **data})
                losses = losses[0] if isinstance(losses, tuple) else losses
                m = matcher_metrics(pred, {**pred, **data})
                rows.append((float(losses["total"].mean()), float(m["match_precision"].mean()), float(m["match_recall"].mean())))
        return [round(sum(v) / len(v), 4) for v in zip(*rows)]

    mix = lambda p, b: {k: (b[k] if isbuf(k) else p[k]) for k in p}   # noqa: E731
    print("reference state, eval mode          :", evaluate(ref))
    print("HIP-trained state, eval mode        :", evaluate(ours))
    print("HIP parameters + reference buffers  :", evaluate(mix(ours, ref)))
    print("reference parameters + HIP buffers  :", evaluate(mix(ref, ours)))
    print("reference state, TRAIN-mode held-out:", evaluate(ref, "train"))
    print("HIP-trained state, TRAIN-mode       :", evaluate(ours, "train"))
    print("reference parameters, re-estimated statistics:", evaluate(ref, reestimate=20))
    print("HIP parameters, re-estimated statistics      :", evaluate(ours, reestimate=20))
    for g in ["kenc.", "gnn.layers.0.", "gnn.layers.1.", "gnn.layers.2.", "gnn.layers.3.", "final_proj."]:
        a = {k: (ours[k] if k.startswith(g) else ref[k]) for k in ref}
        b = {k: (ref[k] if k.startswith(g) else ours[k]) for k in ref}
        print(f"reference with the HIP run's {g:15s}: {evaluate(a)[0]:7.3f}   HIP run with the reference's {g:15s}: {evaluate(b)[0]:7.3f}")

    def drift(a, b):
        cs, rel = [], []
        for k in init:
            if k.endswith(".weight") and init[k].ndim >= 2:
                da, db = (a[k] - init[k]).float().flatten(), (b[k] - init[k]).float().flatten()
                cs.append(float(torch.corrcoef(torch.stack([da, db]))[0, 1]))
                rel.append(float((a[k] - b[k]).float().norm() / b[k].float().norm()))
        return f"drift correlation min {min(cs):.3f} mean {sum(cs) / len(cs):.3f}; relative weight difference max {max(rel):.4f} mean {sum(rel) / len(rel):.4f}"

    print("HIP vs reference A        :", drift(ours, ref))
    print("HIP vs reference B        :", drift(ours, ref_b))
    print("reference A vs reference B:", drift(ref, ref_b))


if __name__ == "__main__":
    main()
