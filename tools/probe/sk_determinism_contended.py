"""sk_determinism.py with another stream's kernel holding a varying number of CUs under every launch (tests/libgf_test_probe.so)."""
import ctypes, os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from glue_factory_amd import ops
import conftest
probe = conftest.test_probe()
B, N, T, reps = (int(v) for v in sys.argv[1:5])
g = torch.Generator(device="cuda").manual_seed(1)
Z = (torch.randn(B, N + 1, N + 1, device="cuda", generator=g) * 2)
G = torch.randn(B, N + 1, N + 1, device="cuda", generator=g)
side = torch.cuda.Stream()
random.seed(0)
ref = None
bad_f = bad_b = nan = 0
for r in range(reps):
    if r:
        n_cus, ms = random.choice([8, 32, 64, 128, 200]), random.choice([1, 2, 5])
        assert probe.gf_test_hold_cus(n_cus, ms, side.cuda_stream) == 0
    z = Z.clone().requires_grad_(True)
    out = ops.sinkhorn(z, T)
    (out * G).sum().backward()
    cur = (out.detach().clone(), z.grad.clone())
    if ref is None:
        torch.cuda.synchronize(); ref = cur
    else:
        if torch.isnan(cur[0]).any(): nan += 1
        elif not torch.equal(cur[0], ref[0]): bad_f += 1
        if not torch.isnan(cur[1]).any() and not torch.equal(cur[1], ref[1]): bad_b += 1
torch.cuda.synchronize()
print(f"contended: B={B} N={N} T={T}: {reps} launches, forward mismatches {bad_f}, backward mismatches {bad_b}, NaN (expired waits) {nan}")
