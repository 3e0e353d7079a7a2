"""Node types of the captured train step (hipGraph debug dump): kernels vs memcpy / memset nodes."""
import sys, types, re, collections, torch
sys.path.insert(0, ".")
import glue_factory_amd
import bench
from glue_factory_amd.synthetic import to_device
from glue_factory_amd import train_step as ts
name = sys.argv[1]
args = types.SimpleNamespace(batch=4, kpts=512, layers=9 if name == "lightglue" else 9, dtype="bf16", no_graph=False, model=name,
                             lines=128, sinkhorn_iters=20)
model, cpu_data = bench.build_matcher(args, 0, name)
stepper = bench.make_stepper(args, model, 0)
data = to_device(cpu_data, "cuda")
orig = torch.cuda.CUDAGraph
class G(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls, *a, **k)
        return g
    def capture_begin(self, *a, **k):
        self.enable_debug_mode()
        return super().capture_begin(*a, **k)
torch.cuda.CUDAGraph = G
for i in range(4):
    stepper(data)
torch.cuda.synchronize()
import os; os.makedirs("gpurun_out", exist_ok=True); stepper._g[1].debug_dump(os.path.abspath(f"gpurun_out/graph_{name}.dot"))
txt = open(f"gpurun_out/graph_{name}.dot").read()
kinds = collections.Counter(re.findall(r'label="?\s*([A-Za-z_]+)', txt))
print(name, "nodes:", len(re.findall(r"\blabel=", txt)), kinds.most_common(12))
for m in re.findall(r'label="[^"]*(?:MEMSET|MEMCPY|Memset|Memcpy|memset|memcpy)[^"]*"', txt)[:10]:
    print(m[:200])
