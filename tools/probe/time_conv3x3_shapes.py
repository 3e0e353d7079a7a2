"""gf_conv3x3_c64 at EQUAL work (4.9 TFLOP, 8.6 GB in) in different geometries: is the 1024^2 layer slower per FLOP than the
512^2 layers because of its geometry (row stride, tiles per image) or because of the duration (clock under power)?"""
import sys, torch
sys.path.insert(0, ".")
import glue_factory_amd  # noqa
from glue_factory_amd import lib as L_
def t(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / n)
    return best
import ctypes
lib = L_.load()
if len(sys.argv) > 1:
    lib = ctypes.CDLL(sys.argv[1]); lib.gf_conv3x3_c64.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    print(sys.argv[1])
import os
SH = [(64, 1024, 1024, 1), (64, 1024, 1056, 1), (64, 512, 512, 0), (64, 512, 512, 1), (64, 256, 256, 0)]
for (B, H, W, pool) in SH:
    x = torch.randn(B, 64, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).to(torch.bfloat16)
    taps = w.permute(2, 3, 0, 1).contiguous()
    bias, scale, shift = (torch.randn(64, device="cuda") for _ in range(3))
    out = torch.empty((B, 64, H // 2, W // 2) if pool else (B, 64, H, W), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream
    def mine():
        L_.check(lib.gf_conv3x3_c64(x.data_ptr(), taps.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                    out.data_ptr(), B, H, W, 1, pool, 1, st), "c")
    tm = t(mine)
    fl = 2 * B * H * W * 64 * 576
    print(f"B{B:5d} {H}x{W} pool{pool}: {tm:.3f} ms  {fl/tm/1e9:.0f} TFLOP/s   ({tm * 1024 / W if H == 1024 else tm * (512 if H == 512 else 256) / W:.3f} ms scaled to the power-of-two width)", flush=True)
    del x, out
