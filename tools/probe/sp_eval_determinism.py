"""Is the frozen SuperPoint forward bit-reproducible from call to call?  (tests/test_gpu_extractor.py::
test_nonfree_superpoint_randomized_keypoints_in_training_mode compares two eval calls exactly and failed once in ~11 runs.)
Repeats the eval forward of the test's model and reports which outputs / stages differ from the first call."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from glue_factory_amd.extractors.superpoint import SuperPoint
z = load_golden("superpoint_nonfree")
image = torch.from_numpy(z["image"]).cuda()
conf = {"force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3, "max_num_keypoints": 64, "dense_outputs": True}
torch.manual_seed(int(z["seed"]))
model = SuperPoint(conf)
model.convPb.weight.data.mul_(40.0)
model = model.cuda().eval()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ref = None
bad = {}
with torch.no_grad():
    for it in range(N):
        if it % 3 == 1:            # (the test interleaves train-mode calls with multinomial draws: some other work in between)
            model.train(); model({"image": image}); model.eval()
        det, desc = model._fused_features(image)
        out = model({"image": image})
        cur = {"det_map": det.float(), "desc_map": desc.float(), **{k: v.float() for k, v in out.items() if torch.is_tensor(v)}}
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
            continue
        for k, v in cur.items():
            if v.shape == ref[k].shape and not torch.equal(v, ref[k]):
                d = (v - ref[k]).abs()
                bad.setdefault(k, []).append((it, int((d > 0).sum()), float(d.max())))
for k, v in bad.items():
    print(f"{k}: differs from call 0 in {len(v)} of {N - 1} calls; first: call {v[0][0]}, {v[0][1]} entries, max |d| {v[0][2]:.3e}")
print("outputs compared:", sorted(ref), "| differing:", sorted(bad) or "none")
