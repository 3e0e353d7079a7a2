"""Like det_probe3, but the batches reach the step the way tests/test_gpu_zz_learning.py feeds them, with switches:
  mode A: generated on the CPU and copied inside the loop, device tensors dropped after the call   (the test)
  mode B: as A, but every device batch is kept alive in a list                                        (no address reuse)
  mode C: as A, with torch.cuda.synchronize() between the copy and the step
  mode D: CPU batches pre-generated, copied inside the loop, dropped after the call
python tools/probe/det_probe4.py bf16|fp32 MODE steps reps"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import learning_cases as lc
import test_gpu_zz_learning as tl
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import to_device
from glue_factory_amd.train_step import TrainStep
bf16, mode, steps, reps = sys.argv[1] == "bf16", sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
kind = os.environ.get("GF_KIND", "superglue")
import random, time
PERTURB = os.environ.get("GF_PERTURB", "")
random.seed(1)
if PERTURB == "hold" or PERTURB.startswith("dirty"):
    import conftest
    probe = conftest.test_probe()
    side = torch.cuda.Stream()
torch.set_num_threads(8)
cpu = [lc.batch(kind, 1000 + i) for i in range(steps)] if mode in ("D", "F", "G") else None
dev = [to_device(c, "cuda") for c in cpu] if mode in ("F", "G") else None      # F: device batches pre-generated: no allocation / copy in the loop
junk = []
pre_cpu = torch.randn(8, 256, 256); pre_pin = pre_cpu.pin_memory(); pre_dev = torch.zeros(8, 256, 256, device="cuda"); pre_dev2 = torch.zeros(8, 256, 256, device="cuda")
ref, nbad = None, 0
for rep in range(reps):
    model = tl._model(kind)
    step = TrainStep(model, FusedAdam(model.parameters(), lr=lc.LR[kind]), amp_dtype=torch.bfloat16 if bf16 else None, graph=os.environ.get('GF_EAGER') != '1', graph_warmup=2)
    hist, keep = [], []
    for i in range(steps):
        if mode == "G":        # G: pre-generated device batches + an unrelated allocation and host-to-device copy per step
            junk.append(torch.randn(8, 256, 256).to("cuda"))
        sub = os.environ.get("GF_SUB", "")
        if sub == "alloc": junk.append(torch.empty(8 * 256 * 256, device="cuda"))
        if sub == "h2d": pre_dev.copy_(pre_cpu)
        if sub == "h2d_pinned": pre_dev.copy_(pre_pin, non_blocking=True)
        if sub == "d2d": pre_dev2.copy_(pre_dev)
        if sub == "kernel": pre_dev.add_(1.0)
        if sub == "sync": torch.cuda.current_stream().synchronize()
        if sub == "hostsleep": time.sleep(0.004)
        d = dev[i] if dev else to_device(cpu[i] if cpu else lc.batch(kind, 1000 + i), "cuda")
        if mode == "B": keep.append(d)
        if mode == "C": torch.cuda.synchronize()
        if PERTURB == "sleep" and rep:
            time.sleep(random.choice([0, 0.002, 0.01]))
        if PERTURB.startswith("dirty"):
            pat = {"dirtynan": 0x7fc00000, "dirtyzero": 0, "dirtybig": 0x7f000000}[PERTURB]
            assert probe.gf_test_dirty_lds(256, pat, torch.cuda.current_stream().cuda_stream) == 0
        if PERTURB == "hold" and rep:
            probe.gf_test_hold_cus(random.choice([16, 64, 128, 224]), random.choice([1, 3]), side.cuda_stream)
        out = step(d)
        if mode == "E": torch.cuda.synchronize()
        if PERTURB == "sleep2" and rep:
            time.sleep(random.choice([0, 0.002, 0.01]))
        del d
        if i % 10 == 9 or i == steps - 1:
            hist.append(float(out["total"].mean()))
    torch.cuda.synchronize()
    hist.append(sum(float(p.detach().double().abs().sum()) for p in model.parameters()))
    step.close()
    if rep == 0: print("rep 0 readings:", [repr(h) for h in hist[:5]], flush=True)
    if ref is None: ref = hist
    elif hist != ref:
        nbad += 1
        j = next(k for k in range(len(ref)) if ref[k] != hist[k])
        print(f"rep {rep}: diverged by reading {j} ({ref[j]!r} vs {hist[j]!r})", flush=True)
print(f"mode {mode} {'bf16' if bf16 else 'fp32'}: {reps} repetitions of {steps} steps, {nbad} diverged")
