import sys, torch
sys.path.insert(0, ".")
import glue_factory_amd
from glue_factory_amd.extractors.superpoint_open import SuperPoint
torch.manual_seed(0)
m = SuperPoint({"max_num_keypoints": 512, "force_num_keypoints": True, "detection_threshold": 0.0}).cuda().eval()
for B, H, W in [(2, 256, 256), (8, 1024, 1024), (64, 1024, 1024)]:
    img = torch.rand(B, 1, H, W, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        det, desc = m._fused_features(img)
        orig = m._conv64_block
        calls = []
        def off(name, blk, x, params, pool):
            calls.append(name)
            return m._fused_block(name, blk, x, params, pool=pool)
        m._conv64_block = off
        det2, desc2 = m._fused_features(img)
        m._conv64_block = orig
    print(B, H, W, calls, "det", (det.float() - det2.float()).abs().max().item(), det2.float().abs().max().item(),
          "desc", (desc.float() - desc2.float()).abs().max().item(), desc2.float().abs().max().item(),
          torch.isfinite(det.float()).all().item(), torch.isfinite(desc.float()).all().item(), flush=True)
