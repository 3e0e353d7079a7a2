import sys, types, torch
sys.path.insert(0, ".")
import glue_factory_amd
import bench
from glue_factory_amd.synthetic import to_device
mode = sys.argv[1]
args = types.SimpleNamespace(batch=32, kpts=2048, layers=9, dtype="bf16", no_graph=False, model="lightglue", lines=512,
                             sinkhorn_iters=100)
model, cpu_data = bench.build_matcher(args, 0, "lightglue")
stepper = bench.make_stepper(args, model, 0)
data = to_device(cpu_data, "cuda")
stepper.max_inflight = int(sys.argv[2])
pipeline_step, extract = bench.make_pipeline_step(args, stepper, 0)
import os
from glue_factory_amd.extractors.superpoint_open import SuperPoint
if os.environ.get("NO_CONV64"):
    SuperPoint._conv64_block = lambda self, name, blk, x, params, pool: self._fused_block(name, blk, x, params, pool=pool)
if os.environ.get("NO_FUSED"):
    SuperPoint._use_fused = lambda self, image: False
losses, mem = [], []
if mode == "matcher":
    step = lambda: stepper(data)["total"].mean()
elif mode == "extract_then_matcher":       # extractor in the loop but its output unused
    def step():
        extract()
        return stepper(data)["total"].mean()
else:
    step = pipeline_step
for i in range(40):
    losses.append(step())
    mem.append((torch.cuda.memory_reserved() >> 30, torch.cuda.memory_allocated() >> 30))
torch.cuda.synchronize()
print(mode, [round(float(l), 3) for l in losses], "skipped", stepper.skipped, flush=True)
print("retries", torch.cuda.memory_stats()["num_alloc_retries"], "ooms", torch.cuda.memory_stats()["num_ooms"])
