"""Are the stock fp32 / bf16 convolutions (MIOpen through F.conv2d) bit-reproducible from call to call at the SuperPoint shapes?"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import glue_factory_amd  # noqa: F401  (sets MIOpen's find mode like the product)
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for (b, cin, cout, hw, k) in [(4, 1, 64, 240, 3), (4, 64, 64, 240, 3), (4, 64, 128, 60, 3), (4, 128, 128, 60, 3), (4, 128, 256, 30, 3), (4, 256, 65, 30, 1), (4, 256, 256, 30, 1)]:
        x = torch.randn(b, cin, hw, hw, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
        outs = [F.conv2d(x, w, None, 1, k // 2) for _ in range(40)]
        torch.cuda.synchronize()
        nd0 = sum(not torch.equal(o, outs[0]) for o in outs[1:])
        nd1 = sum(not torch.equal(o, outs[1]) for o in outs[2:])
        md = max(float((o.float() - outs[1].float()).abs().max()) for o in outs[2:])
        print(f"{str(dtype):15s} {cin:3d}->{cout:3d} k{k} {hw}x{hw}: calls differing from call 0: {nd0}/39, from call 1: {nd1}/38, max |d| vs call 1: {md:.2e}")
