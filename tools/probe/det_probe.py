import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import learning_cases as lc
from glue_factory_amd.matchers.superglue import SuperGlue
from glue_factory_amd.synthetic import to_device
kind = "superglue"
def run(bf16, env=None):
    if env: os.environ.update(env)
    torch.manual_seed(0)
    model = SuperGlue(lc.conf(kind)); model.load_state_dict(lc.initial_params(kind), strict=True); model = model.cuda().train()
    data = to_device(lc.batch(kind, 1000), "cuda")
    outs = []
    for rep in range(3):
        model.zero_grad(set_to_none=True)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            pred = model(data); losses, _ = model.loss(pred, {**pred, **data})
        losses["total"].mean().backward()
        model.load_state_dict(sd)       # undo the BatchNorm buffer updates
        outs.append((pred["log_assignment"].detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    la_same = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    bad = [k for k in outs[0][1] if not all(torch.equal(outs[0][1][k], o[1][k]) for o in outs[1:])]
    print("bf16" if bf16 else "fp32", env or "", "log_assignment identical:", la_same, "| gradients that differ between repeats:", len(bad), bad[:8])
run(False); run(True); run(False, {"GF_SINKHORN_RESIDENT": "0"})
