"""Same-process A/B of a python-level switch of the matcher step (box-to-box spread of the pool is +-5 %, so only
same-process numbers decide).  Usage: python tools/probe/ab_matcher.py [--model lightglue|superglue|gluestick] [--switch FOLD_ENABLED]
Builds one captured TrainStep per setting and alternates timed rounds A B A B ..."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from glue_factory_amd import ops  # noqa: E402
from glue_factory_amd.synthetic import to_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="lightglue")
ap.add_argument("--switch", default="FOLD_ENABLED")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
args = bench.parse()
steppers = {}
for val in (True, False):
    setattr(ops, a.switch, val)
    model, cpu = bench.build_matcher(args, 0, a.model)
    st = bench.make_stepper(args, model, 0)
    data = to_device(cpu, "cuda")
    for _ in range(4):                      # eager warm-up + capture under this setting
        st(data)
    torch.cuda.synchronize()
    steppers[val] = (st, data)
res = {True: [], False: []}
for r in range(a.rounds):
    for val in (True, False):
        st, data = steppers[val]
        setattr(ops, a.switch, val)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st(data)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / a.steps * 1e3)
for val in (True, False):
    print(f"{a.model} {a.switch}={val}: ms/step per round {[round(x, 3) for x in res[val]]}  best {min(res[val]):.3f}")
