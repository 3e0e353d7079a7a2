"""SuperPoint-open forward timing under library-selection knobs (one knob per process):
   python sp_variants.py [benchmark]      env: MIOPEN_FIND_MODE, MIOPEN_FIND_ENFORCE ..."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
if "benchmark" in sys.argv:
    torch.backends.cudnn.benchmark = True
from glue_factory_amd.extractors.superpoint_open import SuperPoint
sp = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(7)
img = torch.rand(64, 1, 1024, 1024, device="cuda", generator=g)
def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return sp({"image": img})
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); first = time.perf_counter() - t0
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
print(sys.argv[1:], {k: v for k, v in os.environ.items() if k.startswith("MIOPEN")}, "first call %.1f s, steady %.2f ms" % (first, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
