"""All gradients of every step of the graph-replayed SuperGlue run, cloned on the device (no host synchronisation in the loop), for
several runs: the first step at which two runs differ, and every gradient that differs there (count of entries, max |d|)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
kind, steps, nruns = "superglue", int(sys.argv[1]), int(sys.argv[2])
noisy = os.environ.get("GF_SUB") == "h2d"
det = os.environ.get("GF_DET") == "1"          # TrainStep(deterministic_replay=True): wait for every replay
torch.set_num_threads(8)
dev = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
pre_cpu = torch.randn(8, 256, 256); pre_dev = torch.zeros(8, 256, 256, device="cuda")
def run(noise):
    model = tl._model(kind)
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=True, graph_warmup=2,
                     deterministic_replay=det)
    rec = []
    for i in range(steps):
        if noise: pre_dev.copy_(pre_cpu)
        out = step(dev[i])
        rec.append({k: p.grad.clone() for k, p in model.named_parameters()} | {"~loss": out["total"].clone()})
    torch.cuda.synchronize(); step.close()
    return rec
base = run(False)
for r in range(1, nruns):
    cur = run(noisy)
    for i in range(steps):
        diff = [(k, int((base[i][k] != cur[i][k]).sum()), float((base[i][k].float() - cur[i][k].float()).abs().max()), float(base[i][k].float().abs().max()))
                for k in base[i] if not torch.equal(base[i][k], cur[i][k])]
        if diff:
            print(f"run {r} ({'noisy' if noisy else 'quiet'}): first difference at step {i}: {len(diff)} tensors")
            for k, n, d, m in diff[:60]:
                print(f"    {k:44s} {n:8d} entries, max |d| {d:.3e} (max |g| {m:.3e})")
            break
    else:
        print(f"run {r}: identical to run 0 over {steps} steps")
