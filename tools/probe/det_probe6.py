"""Does the captured step read memory that the caching allocator regards as FREE?  After the capture, grab the allocator's cached
free blocks (many allocations of assorted sizes), fill them with NaN (GF_POISON=nan) or leave them untouched (GF_POISON=keep),
hold them, and keep replaying on pre-generated device batches: a graph that reads freed memory turns NaN / changes its numbers."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
kind, steps = "superglue", 30
torch.set_num_threads(8)
dev = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
model = tl._model(kind)
step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
poison = os.environ.get("GF_POISON", "")
hold = []
for i in range(steps):
    out = step(dev[i])
    if i == 4 and poison:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        print("before grab: reserved", st["reserved_bytes.all.current"] >> 20, "MB, allocated", st["allocated_bytes.all.current"] >> 20, "MB, inactive split", st["inactive_split_bytes.all.current"] >> 20, "MB")
        for sz in [1 << k for k in range(9, 27)]:
            for _ in range(40 if sz < (1 << 20) else 8):
                t = torch.empty(sz // 4, device="cuda")
                if poison == "nan":
                    t.fill_(float("nan"))
                hold.append(t)
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        print("after grab: reserved", st["reserved_bytes.all.current"] >> 20, "MB, allocated", st["allocated_bytes.all.current"] >> 20, "MB")
    if i % 5 == 4:
        print(i, repr(float(out["total"].mean())), "skipped", step.skipped, flush=True)
