import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from glue_factory_amd.extractors import superpoint_open as spo
torch.manual_seed(0)
model = spo.SuperPoint({"max_num_keypoints": 512, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}).cuda().eval()
img = torch.rand(4, 1, 256, 512, device="cuda")
with torch.no_grad():
    ref = model._dense_unfused(img)                                  # stock fp32 modules
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = {}
        for v in (True, False):
            spo.WIDE64 = v
            outs[v] = [t.float() for t in model._fused_features(img)]
for i, nm in enumerate(("detector", "descriptor")):
    a, b = outs[True][i], outs[False][i]
    r = ref[i].float()
    if nm == "detector":     # fused path returns the score map; compare the two bf16 settings with each other only
        print(nm, "own kernel vs library path: max |d|", float((a - b).abs().max()), "mean |d|", float((a - b).abs().mean()), "scale", float(b.abs().mean()))
    else:
        print(nm, "own vs library: rel", float((a - b).norm() / b.norm()), "| own vs fp32 stock: rel", float((a - r).norm() / r.norm()), "| library vs fp32 stock: rel", float((b - r).norm() / r.norm()))
from superpoint_nonfree_check import check
for v in (True, False):
    spo.WIDE64 = v
    print("WIDE64", v, "non-free golden under bf16 autocast: miss", check("cuda", autocast=True))
