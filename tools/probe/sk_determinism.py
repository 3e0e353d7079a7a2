"""Bitwise repeatability of gf_sinkhorn_fwd / _bwd at a given geometry over many launches (same input): a rare stale read in the
resident kernel's inter-workgroup hand-off would show up as an occasional mismatch.
python tools/probe/sk_determinism.py B N T reps [schedule]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_amd import ops
B, N, T, reps = (int(v) for v in sys.argv[1:5])
sched = ops.sinkhorn_schedule(int(sys.argv[5])) if len(sys.argv) > 5 else ops.sinkhorn_schedule()
g = torch.Generator(device="cuda").manual_seed(1)
Z = (torch.randn(B, N + 1, N + 1, device="cuda", generator=g) * 2)
G = torch.randn(B, N + 1, N + 1, device="cuda", generator=g)
ref = None
bad_f = bad_b = 0
worst = 0.0
for r in range(reps):
    z = Z.clone().requires_grad_(True)
    out = ops.sinkhorn(z, T, schedule=sched)
    (out * G).sum().backward()
    cur = (out.detach().clone(), z.grad.clone())
    if ref is None:
        ref = cur
    else:
        if not torch.equal(cur[0], ref[0]):
            bad_f += 1; worst = max(worst, float((cur[0] - ref[0]).abs().max()))
        if not torch.equal(cur[1], ref[1]):
            bad_b += 1
print(f"B={B} N={N} T={T} schedule={sched}: {reps} launches, forward mismatches {bad_f} (max |d| {worst:.2e}), backward mismatches {bad_b}")
