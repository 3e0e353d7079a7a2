"""Does the captured step SEE the batch that was copied into its static inputs?  _step is wrapped so that, inside the capture, every
input tensor is cloned first thing (a kernel node at the head of the graph); after each replay the clones are compared ON THE DEVICE
with the batch that was passed in.  With GF_SUB=h2d a blocking pageable host-to-device copy is issued before every call."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
kind, steps = "superglue", int(sys.argv[1]) if len(sys.argv) > 1 else 60
sub = os.environ.get("GF_SUB", "")
torch.set_num_threads(8)
dev = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
pre_cpu = torch.randn(8, 256, 256); pre_dev = torch.zeros(8, 256, 256, device="cuda")
def flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict): out.update(flat(v, prefix + k + "."))
        elif torch.is_tensor(v): out[prefix + k] = v
    return out
class Probe(TrainStep):
    def _step(self, data):
        self.snap = {k: v.clone() for k, v in flat(data).items()}
        return super()._step(data)
model = tl._model(kind)
step = Probe(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
mism = None
for i in range(steps):
    if sub == "h2d": pre_dev.copy_(pre_cpu)
    out = step(dev[i])
    if i >= 3:
        cur = flat(dev[i])
        if mism is None: mism = {k: torch.zeros((), device="cuda", dtype=torch.int64) for k in cur}
        for k, v in cur.items():
            mism[k] += (step.snap[k] != v).any().long()
torch.cuda.synchronize()
print(f"sub={sub or 'none'}: {steps - 3} replays; inputs the graph saw differently from the batch passed in:",
      {k: int(v) for k, v in mism.items() if int(v)} or "none", "| final loss", repr(float(out['total'].mean())))
