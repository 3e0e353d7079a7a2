"""Time gf_bgemm on the shapes the matcher steps use (dense-gradient backward of the assignment heads, line head)
against torch.bmm on the same operands.  python tools/probe/time_bgemm.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gf = importlib.import_module("glue-factory_amd")
ops = importlib.import_module("glue-factory_amd.ops")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dt in (torch.bfloat16, torch.float32):
    for (B, M, N, K, ta) in ((32, 2048, 256, 2048, False), (32, 2048, 256, 2048, True), (32, 3072, 256, 3072, True),
                             (32, 512, 512, 256, False)):
        a = torch.randn(B, K, M, device="cuda", dtype=dt).transpose(1, 2) if ta else torch.randn(B, M, K, device="cuda", dtype=dt)
        b = torch.randn(B, K, N, device="cuda", dtype=dt)
        ref = torch.bmm(a.double(), b.double())
        out = ops.bgemm(a, b)
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        us = t(lambda: ops.bgemm(a, b))
        us_t = t(lambda: torch.bmm(a, b))
        print(f"{str(dt)[6:]:9s} B{B} M{M} N{N} K{K} transA={ta}: gf_bgemm {us:8.1f} us ({2*B*M*N*K/us/1e6:6.1f} TF)  torch.bmm {us_t:8.1f} us  rel err {err:.2e}")
