"""Soak of the one-graph scope-P step: N replays back to back (a training run replays the captured step thousands of
times; bench.py only 28), loss read every 50 steps.  The batch is fixed, so the loss must fall monotonically-ish and stay
finite; a dropped graph node or a stale buffer shows up as a jump.  python tools/probe/soak.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    from glue_factory_amd import lib
    torch.cuda.set_device(0)
    lib.load()
    step, _, stepper = bench.make_pipeline_step(args, 0, 0, graph=True)
    for _ in range(bench.PRIME_STEPS):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trace = []
    for i in range(steps):
        loss = step()
        if i % 50 == 49:
            trace.append(float(loss))        # (one host read per 50 steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{steps} replays in {dt:.1f} s = {1e3 * dt / steps:.2f} ms/step, skipped updates: {stepper.skipped}")
    print("loss every 50 steps:", " ".join(f"{v:.3f}" for v in trace))
    assert all(v == v and 0.0 <= v < 100.0 for v in trace), "non-finite / implausible loss"
    worst_rise = max((b - a for a, b in zip(trace, trace[1:])), default=0.0)
    print(f"largest rise between two readings: {worst_rise:.3f}")
    assert stepper.skipped == 0 and trace[-1] < trace[0] and worst_rise < 0.5


if __name__ == "__main__":
    main()
