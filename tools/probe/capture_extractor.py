"""Does the frozen SuperPoint forward (and the homography ground truth) capture into a hipGraph?  Stage by stage, each
checked against the eager result.  python tools/probe/capture_extractor.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from glue_factory_amd.extractors.superpoint_open import SuperPoint
from glue_factory_amd.gt import gt_matches_from_homography_fused
B, IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 1024
sp = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(7)
img0 = torch.rand(B, 1, IMG, IMG, device="cuda", generator=g)
images = torch.cat([img0, img0.roll(8, -1)], 0)
Hm = torch.tensor([[1.0, 0, 8], [0, 1, 0], [0, 0, 1]], device="cuda")[None].repeat(B, 1, 1)
def extract():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return sp({"image": images})
def gt(f):
    return gt_matches_from_homography_fused(f["keypoints"][:B], f["keypoints"][B:], Hm, 3.0, 3.0)
for _ in range(3):
    ref = extract(); ref_gt = gt(ref)
torch.cuda.synchronize()
print("eager ok", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): extract()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = extract()
print("extractor captured", flush=True)
gr.replay(); torch.cuda.synchronize()
print("extractor replayed; keypoint scores equal:", torch.equal(out["keypoint_scores"], ref["keypoint_scores"]), flush=True)
gr2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr2):
    out2 = extract(); g2 = gt(out2)
print("extractor + gt captured", flush=True)
gr2.replay(); torch.cuda.synchronize()
print("extractor + gt replayed; matches equal:", torch.equal(g2["matches0"], ref_gt["matches0"]), flush=True)
