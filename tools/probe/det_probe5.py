"""One process = the first steps of the SuperGlue bf16 learning run exactly as tests/test_gpu_zz_learning.py drives it (eval first,
8 CPU threads, a fresh CPU batch per step); prints every step's loss.  Run in several processes and diff."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.train_step import TrainStep
bf16, steps = sys.argv[1] == "bf16", int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "superglue"
torch.set_num_threads(8)
model = tl._model(kind)
if not os.environ.get("GF_NO_EVAL"):
    print("before", tl._evaluate(kind, model, bf16))
step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16 if bf16 else None, graph=True, graph_warmup=2)
for i in range(steps):
    out = step(tl._batch(kind, 1000 + i))
    print(i, repr(float(out["total"].mean())), flush=True)
