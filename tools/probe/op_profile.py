"""torch.profiler view of one LightGlue train step: aten ops by launch count / shape (finds the small-kernel overhead)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from glue_factory_amd.matchers.lightglue import LightGlue
from glue_factory_amd.synthetic import make_pairs, to_device
from glue_factory_amd.train_step import TrainStep
torch.manual_seed(0)
model = LightGlue({"n_layers": 9}).cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
step = TrainStep(model, opt, amp_dtype=torch.bfloat16, device_ids=[0])
data = to_device(make_pairs(32, 2048, dim=256, seed=100), "cuda")
for _ in range(3): step(data)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(data)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="count", row_limit=70, max_name_column_width=40, max_shapes_column_width=70))
