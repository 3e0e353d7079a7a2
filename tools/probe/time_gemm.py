"""gf_linear_fwd vs the library GEMM (torch F.linear, TunableOp table if PYTORCH_TUNABLEOP_* is set) at the step's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from glue_factory_amd import lib as L_
from glue_factory_amd.ops import _p, _stream
M = 131072
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=20):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best
lib = L_.load()
for N, K in ((768, 256), (256, 256), (512, 256), (256, 512)):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / 16).bfloat16()
    b32 = torch.randn(N, device="cuda", generator=g); b16 = b32.bfloat16()
    res = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_hip = timeit(lambda: lib.gf_linear_fwd(_p(x), _p(w), _p(b32), None, _p(y), M, N, K, K, K, 0, N, 1, _stream()))
    t_hipr = timeit(lambda: lib.gf_linear_fwd(_p(x), _p(w), _p(b32), _p(res), _p(y), M, N, K, K, K, N, N, 1, _stream()))
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b16))
    t_libr = timeit(lambda: torch.nn.functional.linear(x, w, b16).add_(res))
    print(f"N={N} K={K}: hip {t_hip*1e3:.1f} us  hip+res {t_hipr*1e3:.1f}  lib {t_lib*1e3:.1f}  lib+add {t_libr*1e3:.1f}", flush=True)
