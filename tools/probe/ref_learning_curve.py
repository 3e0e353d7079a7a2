"""Build-container only: the UNMODIFIED reference LightGlue trained on the CPU (fp32, torch.optim.Adam lr 1e-3) on the very
batches tests/test_gpu_learning.py feeds the HIP module -- the yardstick for that test's thresholds.
python tools/probe/ref_learning_curve.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.append("/root/reference")
import torch  # noqa: E402

from glue_factory_amd.synthetic import make_pairs  # noqa: E402


def main():
    from gluefactory.models.matchers.lightglue import LightGlue
    from gluefactory.models.utils.metrics import matcher_metrics
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    torch.manual_seed(0)
    torch.set_num_threads(16)
    model = LightGlue({"n_layers": 3, "filter_threshold": 0.1, "weights": None, "flash": False})
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def evaluate():
        model.eval()
        out = []
        with torch.no_grad():
            for s in (9001, 9002, 9003):
                data = make_pairs(8, 256, dim=256, size=(640, 480), seed=s)
                pred = model(data)
                losses, _ = model.loss(pred, {**pred, **data})
                m = matcher_metrics(pred, {**pred, **data})
                out.append((float(losses["total"].mean()), float(m["match_precision"].mean()), float(m["match_recall"].mean())))
        return [sum(v) / len(v) for v in zip(*out)]

    print("held-out before:", evaluate())
    t0 = time.time()
    for i in range(steps):
        model.train()
        data = make_pairs(8, 256, dim=256, size=(640, 480), seed=1000 + i)
        opt.zero_grad()
        pred = model(data)
        losses, _ = model.loss(pred, {**pred, **data})
        losses["total"].mean().backward()
        opt.step()
        if i % 50 == 49:
            print(i + 1, round(float(losses["total"].mean()), 3), f"{time.time() - t0:.0f} s", flush=True)
    print("held-out after:", evaluate())


if __name__ == "__main__":
    main()
