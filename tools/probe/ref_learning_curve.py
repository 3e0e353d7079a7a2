"""Build-container only: the UNMODIFIED reference matchers trained on the CPU (fp32, torch.optim.Adam, lr per tests/learning_cases.py) on the very
batches tests/test_gpu_zz_learning.py feeds the HIP modules (tests/learning_cases.py) -- the yardstick for that test.
python tools/probe/ref_learning_curve.py lightglue|superglue|gluestick [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.append("/root/reference")
import torch  # noqa: E402

import learning_cases as lc  # noqa: E402


def main():
    from gluefactory.models.utils.metrics import matcher_metrics
    kind = sys.argv[1] if len(sys.argv) > 1 else "lightglue"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else lc.STEPS
    if kind == "lightglue":
        from gluefactory.models.matchers.lightglue import LightGlue as Model
        extra = {"weights": None, "flash": False}
    elif kind == "superglue":
        from gluefactory_nonfree.superglue import SuperGlue as Model
        extra = {"weights": None}
    else:
        from gluefactory.models.matchers.gluestick import GlueStick as Model
        extra = {"weights": None}
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("GF_THREADS", "16")))
    model = Model({**lc.conf(kind), **extra})
    params = lc.initial_params(kind)
    if params is not None:
        model.load_state_dict(params, strict=True)
    dtype = getattr(torch, os.environ.get("GF_DTYPE", "float32"))      # float64: the anchor run of tools/probe/learn_anchor_report.py
    model = model.to(dtype)
    opt = torch.optim.Adam(model.parameters(), lr=lc.LR[kind])
    make_batch = lc.batch

    def cast(o):
        if isinstance(o, dict):
            return {k: cast(v) for k, v in o.items()}
        return o.to(dtype) if torch.is_tensor(o) and o.is_floating_point() else o

    lc_batch = lambda kind, seed: cast(make_batch(kind, seed))   # noqa: E731

    def evaluate():
        model.eval()
        out = []
        with torch.no_grad():
            for s in lc.HELD_OUT:
                data = lc_batch(kind, s)
                pred = model(data)
                losses = model.loss(pred, {**pred, **data})
                losses = losses[0] if isinstance(losses, tuple) else losses
                m = matcher_metrics(pred, {**pred, **data})
                row = [float(losses["total"].mean()), float(m["match_precision"].mean()), float(m["match_recall"].mean())]
                if kind == "gluestick":
                    ml = matcher_metrics(pred, {**pred, **data}, prefix="line_", prefix_gt="line_")
                    row += [float(ml["line_match_precision"].mean()), float(ml["line_match_recall"].mean())]
                out.append(row)
        return [round(sum(v) / len(v), 4) for v in zip(*out)]

    print(kind, "held-out before:", evaluate(), flush=True)
    t0 = time.time()
    trace = []
    for i in range(steps):
        model.train()
        data = lc_batch(kind, 1000 + i)
        opt.zero_grad()
        pred = model(data)
        losses = model.loss(pred, {**pred, **data})
        losses = losses[0] if isinstance(losses, tuple) else losses
        losses["total"].mean().backward()
        opt.step()
        if i % 50 == 49:
            trace.append(round(float(losses["total"].detach().mean()), 3))
            print(i + 1, trace[-1], f"{time.time() - t0:.0f} s", flush=True)
    print(kind, "trace:", trace)
    print(kind, "held-out after:", evaluate())
    if os.environ.get("GF_SAVE_FINAL"):            # diagnosis: the trained reference state (parameters + BatchNorm buffers)
        torch.save(model.state_dict(), os.environ["GF_SAVE_FINAL"])


if __name__ == "__main__":
    main()
