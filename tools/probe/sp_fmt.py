import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch.nn.functional as F
from glue_factory_amd.extractors import superpoint_open as spo
orig = F.conv2d
def traced(x, w, b=None, *a, **k):
    y = orig(x, w, b, *a, **k)
    print("conv in", tuple(x.shape), x.stride(), "cl" if x.is_contiguous(memory_format=torch.channels_last) else "--",
          "w", tuple(w.shape), "->", "cl" if y.is_contiguous(memory_format=torch.channels_last) else "NCHW", y.dtype)
    return y
spo.F.conv2d = traced
sp = spo.SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}).cuda().eval()
img = torch.rand(2, 1, 256, 256, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    sp({"image": img})
