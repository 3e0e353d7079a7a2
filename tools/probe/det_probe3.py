"""In ONE process: the first steps of the SuperGlue learning run repeated many times from scratch (fresh module, fresh TrainStep,
same batches); every repetition must reproduce the first one bit for bit.  On a mismatch: the step, and which parameters differ.
python tools/probe/det_probe3.py fp32|bf16 graph|eager steps reps [kind]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
bf16, graph, steps, reps = sys.argv[1] == "bf16", sys.argv[2] == "graph", int(sys.argv[3]), int(sys.argv[4])
kind = sys.argv[5] if len(sys.argv) > 5 else "superglue"
torch.set_num_threads(8)
batches = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
ref = None
nbad = 0
for rep in range(reps):
    model = tl._model(kind)
    if os.environ.get("GF_EVAL_FIRST"):
        tl._evaluate(kind, model, bf16); model.train()
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16 if bf16 else None, graph=graph, graph_warmup=2)
    hist = []
    for i in range(steps):
        out = step(batches[i])
        torch.cuda.synchronize()
        hist.append((float(out["total"].mean()), {k: v.detach().clone() for k, v in model.state_dict().items()}))
    step.close()
    if ref is None:
        ref = hist
        continue
    for i, (a, b) in enumerate(zip(ref, hist)):
        diff = [k for k in a[1] if not torch.equal(a[1][k], b[1][k])]
        if a[0] != b[0] or diff:
            nbad += 1
            print(f"rep {rep}: first mismatch at step {i}: loss {a[0]!r} vs {b[0]!r}; {len(diff)} of {len(a[1])} state entries differ; first few: {diff[:6]}", flush=True)
            break
print(f"{kind} {'bf16' if bf16 else 'fp32'} {'graph' if graph else 'eager'}: {reps} repetitions of {steps} steps, {nbad} diverged")
