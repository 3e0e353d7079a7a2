import sys, types, torch
sys.path.insert(0, ".")
import glue_factory_amd
import bench
from glue_factory_amd.synthetic import to_device
mode = sys.argv[1]
args = types.SimpleNamespace(batch=32, kpts=2048, layers=9, dtype="bf16", no_graph=(mode == "eager"), model="lightglue", lines=512,
                             sinkhorn_iters=100)
model, cpu_data = bench.build_matcher(args, 0, "lightglue")
stepper = bench.make_stepper(args, model, 0)
data = to_device(cpu_data, "cuda")
pipeline_step, extract = bench.make_pipeline_step(args, stepper, 0)
losses = []
for i in range(40):
    losses.append(pipeline_step())
    if mode.endswith("sync"): torch.cuda.synchronize()
torch.cuda.synchronize()
print(mode, [round(float(l), 3) for l in losses], "skipped", stepper.skipped, flush=True)
pn = torch.stack([p.detach().float().norm() for p in model.parameters()]).max()
print("max param norm", float(pn))
