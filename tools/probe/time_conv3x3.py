"""Time gf_conv3x3_c64 at the three SuperPoint shapes of the headline batch (64 images of 1024^2) against the library
convolution + tail pass it replaces."""
import sys, torch
sys.path.insert(0, ".")
import glue_factory_amd  # noqa
from glue_factory_amd import lib as L_
import torch.nn.functional as F

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

lib = L_.load()
for (B, H, W, pool) in [(64, 1024, 1024, 1), (64, 512, 512, 0), (64, 512, 512, 1)]:
    x = torch.randn(B, 64, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).to(torch.bfloat16)
    wl = w.contiguous(memory_format=torch.channels_last)
    taps = w.permute(2, 3, 0, 1).contiguous()
    bias, scale, shift = (torch.randn(64, device="cuda") for _ in range(3))
    out = torch.empty((B, 64, H // 2, W // 2) if pool else (B, 64, H, W), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream
    def mine():
        L_.check(lib.gf_conv3x3_c64(x.data_ptr(), taps.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                    out.data_ptr(), B, H, W, 1, pool, 1, st), "c")
    def libconv():
        y = F.conv2d(x, wl, None, 1, 1)
        L_.check(lib.gf_bias_act_bn_nhwc(y.data_ptr(), out.data_ptr() if pool else y.data_ptr(), bias.data_ptr(), scale.data_ptr(),
                                         shift.data_ptr(), B, H, W, 64, 1, pool, 1, st), "t")
    tm, tl = t(mine), t(libconv)
    fl = 2 * B * H * W * 64 * 576
    by = B * H * W * 128 * (1.25 if pool else 2)
    print(f"B{B} {H}x{W} pool{pool}: mine {tm:.3f} ms ({fl/tm/1e9:.0f} TFLOP/s, {by/tm/1e6:.0f} GB/s)  library+tail {tl:.3f} ms", flush=True)
