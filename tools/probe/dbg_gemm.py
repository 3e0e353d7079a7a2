import sys, torch
sys.path.insert(0, ".")
import glue_factory_amd
from glue_factory_amd import lib as L_
from glue_factory_amd.ops import _p, _stream
lib = L_.load()
for (M, N, K) in [(64, 256, 256), (4096, 768, 256), (4096, 256, 256)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / 16).bfloat16()
    y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    L_.check(lib.gf_gemm(_p(x), None, _p(w), None, None, _p(y), None, 0, M, N, K, 0, K, 0, K, 0, N, 1, _stream()), "g")
    ref = (x.float() @ w.float().t())
    bad = ~((y.float() - ref).abs() < 0.05)
    print(M, N, K, "bad", int(bad.sum()), "nan", int(torch.isnan(y.float()).sum()))
    if bad.any():
        idx = bad.nonzero()
        rows = idx[:, 0] % 64; cols = idx[:, 1]
        print(" rows%64 hist", torch.bincount(rows, minlength=64).tolist())
        print(" col//8 %32 hist", torch.bincount((cols // 8) % 32, minlength=32).tolist())
        print(" col//256 hist", torch.bincount(cols // 256).tolist(), "tile hist", torch.bincount(idx[:, 0] // 64)[:8].tolist())
