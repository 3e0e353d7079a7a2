"""kenc.encoder.0.bias -- the first tensor to differ between two runs of the graph-replayed SuperGlue steps: its gradient (the
column sum of the gradient in front of a train-mode BatchNorm: analytically zero, pure rounding residue) and its value per step,
recorded on the device, for several quiet runs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
kind, steps = "superglue", 12
torch.set_num_threads(8)
dev = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
graph = os.environ.get("GF_EAGER") != "1"
def run():
    model = tl._model(kind)
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=graph, graph_warmup=2)
    b = model.kenc.encoder[0].bias
    w = model.kenc.encoder[0].weight
    rec = torch.zeros(steps, 3, 32, device="cuda")
    for i in range(steps):
        step(dev[i])
        rec[i, 0] = b.grad
        rec[i, 1] = b.detach()
        rec[i, 2] = w.grad.reshape(32, -1)[:, 0]
    torch.cuda.synchronize(); step.close()
    return rec.cpu()
runs = [run() for _ in range(4)]
for r in range(1, 4):
    for i in range(steps):
        for j, nm in enumerate(("bias.grad", "bias", "weight.grad[:,0]")):
            if not torch.equal(runs[0][i, j], runs[r][i, j]):
                d = (runs[0][i, j] - runs[r][i, j]).abs()
                print(f"run 0 vs run {r}: step {i} {nm}: {int((d > 0).sum())} of 32 entries differ, max |d| {float(d.max()):.3e}; |value| max {float(runs[0][i, j].abs().max()):.3e}")
                break
        else:
            continue
        break
    else:
        print(f"run 0 vs run {r}: identical")
print("bias.grad of run 0 at steps 0..3:", runs[0][:4, 0, :6])
