"""Shape of the captured SuperGlue learning step's hipGraph: nodes, edges, roots, leaves, max fan-out / fan-in.  A capture from one
stream must be a chain (every node one predecessor and one successor)."""
import os, re, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.train_step import TrainStep
kind = sys.argv[1] if len(sys.argv) > 1 else "superglue"
orig = torch.cuda.CUDAGraph
class G(orig):
    def capture_begin(self, *a, **k):
        self.enable_debug_mode()
        return super().capture_begin(*a, **k)
torch.cuda.CUDAGraph = G
import glue_factory_amd.train_step as ts
model = tl._model(kind)
step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
for i in range(4):
    step(tl._batch(kind, 1000 + i))
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", f"graph_{kind}_learning.dot")
step._g[1].debug_dump(path)
txt = open(path).read()
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
nodes = set(re.findall(r'^\s*"?(\w+)"?\s*\[', txt, flags=re.M)) | {a for a, _ in edges} | {b for _, b in edges}
out_deg, in_deg = collections.Counter(a for a, _ in edges), collections.Counter(b for _, b in edges)
roots = [n for n in nodes if in_deg[n] == 0]
leaves = [n for n in nodes if out_deg[n] == 0]
print(f"{kind}: nodes {len(nodes)}, edges {len(edges)}, roots {len(roots)}, leaves {len(leaves)}, max fan-out {max(out_deg.values(), default=0)}, "
      f"max fan-in {max(in_deg.values(), default=0)}, nodes with fan-out > 1: {sum(1 for v in out_deg.values() if v > 1)}")
print(txt[:600])
