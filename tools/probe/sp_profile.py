"""SuperPoint-open forward on 64 random 1024x1024 images under bf16 autocast (scope P extractor)."""
import os, sys, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from glue_factory_amd.extractors.superpoint_open import SuperPoint
sp = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(7)
img = torch.rand(64, 1, 1024, 1024, device="cuda", generator=g)
def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return sp({"image": img})
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize(); print("ms per call", (time.perf_counter() - t0) / 3 * 1e3)
