"""Graph-replayed SuperGlue steps on pre-generated device batches, quiet vs with ONE blocking pageable H2D copy issued right after
each step call (overlapping the replay in flight): per step, loss + checksum of every parameter / buffer; prints the first step at
which the two runs differ and WHICH state entries differ first."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
kind, steps = "superglue", 14
torch.set_num_threads(8)
dev = [to_device(lc.batch(kind, 1000 + i), "cuda") for i in range(steps)]
pre_cpu = torch.randn(8, 256, 256); pre_dev = torch.zeros(8, 256, 256, device="cuda")
def run(noisy):
    model = tl._model(kind)
    opt = FusedAdam(model.parameters(), lr=lc.LR[kind])
    step = TrainStep(model, opt, amp_dtype=torch.bfloat16, graph=True, graph_warmup=2)
    hist = []
    for i in range(steps):
        out = step(dev[i])
        if noisy and i >= 3:
            pre_dev.copy_(pre_cpu)                       # blocking, pageable: overlaps the replay in flight
        torch.cuda.synchronize()
        st = {k: v.detach().clone() for k, v in model.state_dict().items()}
        st.update({"grad." + k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
        st.update({"loss." + k: v.detach().clone() for k, v in out.items()})
        hist.append(st)
    step.close()
    return hist
a, b = run(False), run(True)
c = run(False)
for name, x, y in (("quiet vs quiet", a, c), ("quiet vs noisy", a, b)):
    for i, (s, t) in enumerate(zip(x, y)):
        diff = [k for k in s if not torch.equal(s[k], t[k])]
        if diff:
            g = [k for k in diff if k.startswith("grad.")]
            l = [k for k in diff if k.startswith("loss.")]
            print(f"{name}: first difference at step {i}: {len(diff)} entries; losses {l}; {len(g)} gradients differ")
            for k in (g[:40] or diff[:10]):
                print(f"     {k:60s} max|d| {float((s[k].float() - t[k].float()).abs().max()):.3e}  (max|x| {float(s[k].float().abs().max()):.3e})")
            break
    else:
        print(f"{name}: identical over {steps} steps")
