"""gf_gemm + gf_bn_stats against gf_gemm_bnstats (the BatchNorm sums in the GEMM's epilogue), same process:
python tools/probe/time_gemm_bnstats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glue_factory_amd import lib as L_
from glue_factory_amd.ops import _p, _stream
lib = L_.load()
def timeit(fn, iters=20):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best * 1e3
for M in (131072, 196608):
    N = K = 512
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / 22).bfloat16()
    bias = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    sets = 2
    nblk = lib.gf_gemm_bnstats_nblk(M, N, K, sets); part = torch.empty(sets, nblk, 2, N, device="cuda")
    nb2 = lib.gf_bn_nblk(M // sets); part2 = torch.empty(nb2, 2, N, device="cuda")
    x0, x1 = x[:, :256], x[:, 256:]
    def plain(): L_.check(lib.gf_gemm(_p(x0), _p(x1), _p(w), _p(bias), None, _p(y), None, 0, M, N, 256, 256, K, K, K, 0, N, 1, _stream()), "g")
    def fused(): L_.check(lib.gf_gemm_bnstats(_p(x0), _p(x1), _p(w), _p(bias), _p(y), _p(part), sets, M, N, 256, 256, K, K, K, N, 1, _stream()), "gs")
    def stats():
        for h in range(sets):
            L_.check(lib.gf_bn_stats(_p(y[h * (M // sets):]), _p(part2), M // sets, N, 1, _stream()), "s")
    print(f"M={M}: gf_gemm {timeit(plain):.1f} us   gf_gemm_bnstats {timeit(fused):.1f} us   gf_bn_stats x{sets} {timeit(stats):.1f} us", flush=True)
