"""Same-process A/B of the frozen SuperPoint-open forward (64 images of 1024^2, bf16 autocast) under a module-level switch of
extractors/superpoint_open.py (default: WIDE64): python tools/probe/ab_extractor.py [SWITCH]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_amd.extractors import superpoint_open as spo
sw = sys.argv[1] if len(sys.argv) > 1 else "WIDE64"
torch.manual_seed(0)
model = spo.SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3}).cuda().eval()
img = torch.rand(64, 1, 1024, 1024, device="cuda")
res = {True: [], False: []}
outs = {}
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for r in range(4):
        for val in (True, False):
            setattr(spo, sw, val)
            for _ in range(2): o = model({"image": img})
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): o = model({"image": img})
            torch.cuda.synchronize(); res[val].append((time.perf_counter() - t0) / 5 * 1e3)
            outs[val] = o
for val in (True, False):
    print(f"{sw}={val}: ms per forward {[round(x, 3) for x in res[val]]}  best {min(res[val]):.3f}")
ka, kb = outs[True]["keypoints"], outs[False]["keypoints"]
same = sum(len({tuple(k) for k in ka[i].round().long().tolist()} & {tuple(k) for k in kb[i].round().long().tolist()}) for i in range(ka.shape[0]))
print(f"keypoints shared between the two settings: {same} of {ka.shape[0] * ka.shape[1]}")
