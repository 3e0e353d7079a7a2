"""torch.profiler view of one matcher train step: aten ops by launch count / shape (finds the small-kernel overhead).
python tools/probe/op_profile.py [lightglue|superglue|gluestick]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from glue_factory_amd.synthetic import make_pairs, to_device
from glue_factory_amd.train_step import TrainStep
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "lightglue"
if which == "lightglue":
    from glue_factory_amd.matchers.lightglue import LightGlue
    model = LightGlue({"n_layers": 9}).cuda().train()
    data = to_device(make_pairs(32, 2048, dim=256, seed=100), "cuda")
else:       # the bench's own builders (BASELINE configs[3] / configs[4])
    import argparse
    import bench
    model, cpu_data = bench.build_matcher(argparse.Namespace(batch=32, kpts=2048, lines=512, layers=9, sinkhorn_iters=100), 0, which)
    data = to_device(cpu_data, "cuda")
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
step = TrainStep(model, opt, amp_dtype=torch.bfloat16, device_ids=[0])
for _ in range(3): step(data)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(data)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="count", row_limit=70, max_name_column_width=40, max_shapes_column_width=70))
# second view: aten operators that launch device work, by device time (what the "elementwise sweep" has to remove)
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print("\n== aten operators with device time, one train step ==")
tot = 0.0
for e in rows[:60]:
    tot += e.self_device_time_total
    print(f"{e.self_device_time_total / 1e3:8.3f} ms  x{e.count:4d}  {e.key:28s} {str(e.input_shapes)[:110]}")
print(f"total aten device time {sum(e.self_device_time_total for e in rows) / 1e3:.3f} ms in {sum(e.count for e in rows)} calls")
