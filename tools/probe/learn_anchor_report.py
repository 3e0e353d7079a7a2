"""Build-container only: where do the trained SuperGlue (or GlueStick) states of DIFFERENT ARITHMETICS land after the lc.STEPS
steps of tests/test_gpu_zz_learning.py -- same initial state, same batches, same optimiser settings?

    python tools/probe/learn_anchor_report.py superglue ANCHOR.pt name=state.pt [name=state.pt ...]

ANCHOR.pt is the fp64 run of the unmodified reference module (GF_DTYPE=float64 tools/probe/ref_learning_curve.py); the other
states come from the reference in fp32 at several thread counts (summation orders), tools/probe/learn_third_arithmetic.py
(stock PyTorch-ROCm ops on the MI355X, fp32 / fp64) and tools/probe/learn_save_state.py (the HIP path, fp32 / bf16).
Every state is evaluated by the SAME evaluator: the reference module on the CPU in fp32 --
  * held-out loss / precision / recall in eval mode (running statistics) and in TRAINING mode (per-image batch statistics),
  * the same with the state's BatchNorm buffers replaced by the anchor's,
  * ||theta - theta_anchor|| / ||theta_anchor - theta_0|| over the weight matrices (distance to the anchor in units of the
    anchor's own drift from the initial state) and the correlation of the two drifts.
Results: profiles/r06_learning_anchor.txt."""
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
    kind = sys.argv[1]
    if kind == "superglue":
        from gluefactory_nonfree.superglue import SuperGlue as Model
    else:
        from gluefactory.models.matchers.gluestick import GlueStick as Model
    torch.set_num_threads(int(os.environ.get("GF_THREADS", "4")))
    anchor = torch.load(sys.argv[2], map_location="cpu")
    anchor = {k: (v.float() if v.is_floating_point() else v) for k, v in anchor.items()}
    states = {"anchor(fp64)": anchor}
    for a in sys.argv[3:]:
        name, path = a.split("=", 1)
        st = torch.load(path, map_location="cpu")
        states[name] = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
    init = lc.initial_params(kind)
    held = [lc.batch(kind, s) for s in lc.HELD_OUT]
    isbuf = lambda k: "running_" in k or "num_batches" in k          # noqa: E731

    def evaluate(state, mode):
        model = Model({**lc.conf(kind), "weights": None})
        model.load_state_dict(state, strict=True)
        model.train() if mode == "train" else model.eval()
        rows = []
        with torch.no_grad():
            for data in held:
                pred = model(data)
                losses = model.loss(pred, {**pred, **data})
                losses = losses[0] if isinstance(losses, tuple) else losses
                m = matcher_metrics(pred, {**pred, **data})
                rows.append((float(losses["total"].mean()), float(m["match_precision"].mean()), float(m["match_recall"].mean())))
        return [round(sum(v) / len(v), 3) for v in zip(*rows)]

    wkeys = [k for k in init if k.endswith(".weight") and init[k].ndim >= 2]

    def dist(a):
        num = sum(float((a[k] - anchor[k]).double().pow(2).sum()) for k in wkeys) ** 0.5
        den = sum(float((anchor[k] - init[k]).double().pow(2).sum()) for k in wkeys) ** 0.5
        da = torch.cat([(a[k] - init[k]).flatten() for k in wkeys]).double()
        db = torch.cat([(anchor[k] - init[k]).flatten() for k in wkeys]).double()
        return num / den, float(torch.corrcoef(torch.stack([da, db]))[0, 1])

    print(f"{'state':24s} {'eval loss/prec/rec':26s} {'train-mode loss/prec/rec':26s} {'eval, anchor buffers':26s} |d|/|drift|  corr")
    for name, st in states.items():
        mixed = {k: (anchor[k] if isbuf(k) else st[k]) for k in st}
        d, c = dist(st)
        print(f"{name:24s} {str(evaluate(st, 'eval')):26s} {str(evaluate(st, 'train')):26s} {str(evaluate(mixed, 'eval')):26s} {d:9.4f} {c:7.4f}",
              flush=True)
    names = list(states)
    print("pairwise |theta_a - theta_b| / |anchor drift| over the weight matrices:")
    den = sum(float((anchor[k] - init[k]).double().pow(2).sum()) for k in wkeys) ** 0.5
    for i, a in enumerate(names):
        row = []
        for b in names:
            row.append(sum(float((states[a][k] - states[b][k]).double().pow(2).sum()) for k in wkeys) ** 0.5 / den)
        print(f"  {a:24s}", " ".join(f"{v:7.4f}" for v in row))


if __name__ == "__main__":
    main()
