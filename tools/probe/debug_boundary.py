"""Debug: loss entries of the reference-driven pipeline vs ours, both orders, repeated."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import build_ref
assert build_ref.import_reference()
import test_gpu_reference_boundary as t
from omegaconf import OmegaConf
from gluefactory.models.two_view_pipeline import TwoViewPipeline as RefPipeline
from glue_factory_amd.pipeline import TwoViewPipeline as OurPipeline
from glue_factory_amd.synthetic import to_device
torch.manual_seed(0)
rp = RefPipeline(OmegaConf.create(t.CONF)).cuda().train()
ours = {**t.CONF, "matcher": {**t.CONF["matcher"], "name": "matchers.lightglue"}, "ground_truth": {**t.CONF["ground_truth"], "name": "matchers.homography_matcher"}}
op = OurPipeline(ours).cuda().train()
op.load_state_dict(rp.state_dict(), strict=True)
print("conf ref", dict(rp.matcher.conf.items()) if hasattr(rp.matcher.conf, "items") else rp.matcher.conf)
print("conf our", dict(op.matcher.conf.items()) if hasattr(op.matcher.conf, "items") else op.matcher.conf)
data, _ = t._views(3, 256, seed=7)
data = to_device(data, "cuda")
for rep in range(3):
    for name, pipe in (("ref", rp), ("our", op), ("ref", rp)):
        pred, losses, grads = t._train(pipe, data, False)
        print(rep, name, {k: [round(float(x), 6) for x in v.flatten()] for k, v in losses.items() if torch.is_tensor(v)})
        print("   gt pos", (pred["gt_matches0"] >= 0).sum(1).tolist(), "col0" , "gt_assignment_col0" in pred, sorted(k for k in pred if k.startswith("gt_")))
print("---- bf16")
for rep in range(2):
    for name, pipe in (("ref", rp), ("our", op), ("ref", rp), ("our", op)):
        pred, losses, grads = t._train(pipe, data, True)
        print(rep, name, {k: [round(float(x), 6) for x in v.detach().flatten()] for k, v in losses.items() if torch.is_tensor(v)})
