"""Run-to-run determinism of the SuperGlue learning run (tests/learning_cases.py) across PROCESSES: per-step checksums of the
loss and of every parameter after the update, for the first steps; run twice and diff.
python tools/probe/det_probe2.py fp32|bf16 graph|eager steps > out.txt"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
from glue_factory_amd.matchers.superglue import SuperGlue
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
bf16, graph, steps = sys.argv[1] == "bf16", sys.argv[2] == "graph", int(sys.argv[3])
junk = torch.full((int(os.environ.get("GF_JUNK_MB", "0")) * 262144 + 1,), float(os.environ.get("GF_JUNK_VAL", "nan")), device="cuda")
del junk                      # poison the allocator's free blocks: an uninitialised read shows up as NaN / a different number
torch.manual_seed(0)
model = SuperGlue(lc.conf("superglue")); model.load_state_dict(lc.initial_params("superglue"), strict=True); model = model.cuda()
if os.environ.get("GF_EVAL_FIRST"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_zz_learning as tl
    print("eval before:", tl._evaluate("superglue", model, bf16), flush=True)
    model.train()
step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR["superglue"]), amp_dtype=torch.bfloat16 if bf16 else None, graph=graph, graph_warmup=2)
for i in range(steps):
    out = step(to_device(lc.batch("superglue", 1000 + i), "cuda"))
    torch.cuda.synchronize()
    cs = sum(float(p.detach().double().abs().sum()) for p in model.parameters())
    bs = sum(float(b.detach().double().abs().sum()) for b in model.buffers())
    print(i, repr(float(out["total"].mean())), repr(cs), repr(bs), flush=True)
