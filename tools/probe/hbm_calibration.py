"""Practical HBM rates of this box for the access mixes the HBM-bound kernels have (calibration for the `hbm` roofline
fractions): device copy (1 read : 1 write, the mix of gemm_st at K = N), fill (write only), reduction (read only), at
the step's tensor size (M = 131072 rows x 256 bf16 = 67 MB) and at 1 GB.  python tools/probe/hbm_calibration.py"""
import torch


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e-3)
    return best


for rows in (131072, 2 * 1024 * 1024):
    x = torch.randn(rows, 256, device="cuda").bfloat16()
    y = torch.empty_like(x)
    nb = x.numel() * 2
    tc = t(lambda: y.copy_(x))
    tf = t(lambda: y.zero_())
    tr = t(lambda: x.view(torch.int16).max())
    print(f"{nb / 1e6:7.0f} MB tensors: copy {2 * nb / tc / 1e9:6.0f} GB/s ({tc * 1e6:6.1f} us)   fill {nb / tf / 1e9:6.0f} GB/s   "
          f"read (max-reduce) {nb / tr / 1e9:6.0f} GB/s")
