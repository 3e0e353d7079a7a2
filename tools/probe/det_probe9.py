"""Per-step, device-side checksums (no host synchronisation inside the loop) of every parameter, BatchNorm buffer and loss of the
graph-replayed SuperGlue run, quiet vs with a blocking pageable H2D copy in front of every call: the first step and the first
tensors that differ."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
kind, steps = "superglue", int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.set_num_threads(8)
dev = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
pre_cpu = torch.randn(8, 256, 256); pre_dev = torch.zeros(8, 256, 256, device="cuda")
def run(noisy):
    model = tl._model(kind)
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
    names = [k for k, v in model.state_dict().items() if v.is_floating_point()]
    rec = torch.zeros(steps, len(names) + 1, device="cuda", dtype=torch.float64)
    for i in range(steps):
        if noisy: pre_dev.copy_(pre_cpu)
        out = step(dev[i])
        sd = model.state_dict()
        rec[i, :-1] = torch.stack([sd[k].detach().double().sum() for k in names])
        rec[i, -1] = out["total"].double().sum()
    torch.cuda.synchronize()
    step.close()
    return names + ["loss.total"], rec.cpu()
names, a = run(False)
_, b = run(True)
_, c = run(False)
for tag, x, y in (("quiet vs quiet", a, c), ("quiet vs noisy", a, b)):
    d = (x != y)
    if not d.any():
        print(tag, ": identical over", steps, "steps"); continue
    i = int(d.any(1).nonzero()[0])
    ks = [names[j] for j in d[i].nonzero().flatten().tolist()]
    print(f"{tag}: first difference at step {i}: {len(ks)} of {len(names)} entries differ: {ks[:30]}")
    i2 = min(i + 1, steps - 1)
    print(f"   step {i2}: {int(d[i2].sum())} entries differ")
