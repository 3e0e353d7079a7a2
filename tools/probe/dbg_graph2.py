"""Which part of the bench's stepper re-use breaks under hipGraph replay?"""
import sys, types, torch
sys.path.insert(0, ".")
import glue_factory_amd
import bench
mode = sys.argv[1]
args = types.SimpleNamespace(batch=32, kpts=2048, layers=9, dtype="bf16", no_graph=False, model="lightglue", lines=512,
                             sinkhorn_iters=100)
model, cpu_data = bench.build_matcher(args, 0, "lightglue")
stepper = bench.make_stepper(args, model, 0)
from glue_factory_amd.synthetic import to_device
data = to_device(cpu_data, "cuda")
if mode == "both":           # the bench's sequence: matcher data first (captured), then the pipeline re-captures
    for i in range(4):
        print("matcher", i, float(stepper(data)["total"].mean()), flush=True)
pipeline_step, extract = bench.make_pipeline_step(args, stepper, 0)
if mode == "warm":           # eager steps on the pipeline signature first
    stepper._calls = 0
for i in range(6):
    print("pipeline", i, float(pipeline_step()), flush=True)
losses = [pipeline_step() for _ in range(8)]
torch.cuda.synchronize()
print("nosync", [float(l) for l in losses], flush=True)
losses = []
for _ in range(8):
    losses.append(pipeline_step()); torch.cuda.synchronize()
print("sync", [float(l) for l in losses], flush=True)
