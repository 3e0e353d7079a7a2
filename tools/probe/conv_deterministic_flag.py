"""torch.backends.cudnn.deterministic for the stock (MIOpen) convolutions of the frozen SuperPoint forward: time and call-to-call
reproducibility of the 128 / 256-channel blocks at the benchmark's shapes (64 images of 1024^2 -> 256^2 / 128^2 maps), bf16."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import glue_factory_amd  # noqa: F401
shapes = [(64, 64, 128, 256, 3), (64, 128, 128, 256, 3), (64, 128, 128, 128, 3), (64, 128, 256, 128, 3), (64, 256, 65, 128, 1)]
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    tot = 0.0
    for (b, cin, cout, hw, k) in shapes:
        x = torch.randn(b, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        try:
            outs = [F.conv2d(x, w, None, 1, k // 2) for _ in range(6)]
        except RuntimeError as e:
            print(f"deterministic={det} {cin}->{cout} k{k} {hw}^2: {str(e)[:100]}"); continue
        torch.cuda.synchronize()
        same = all(torch.equal(o, outs[1]) for o in outs[2:])
        t0 = time.perf_counter()
        for _ in range(20): F.conv2d(x, w, None, 1, k // 2)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        tot += ms
        print(f"deterministic={det} {cin:3d}->{cout:3d} k{k} {hw}^2: {ms:.3f} ms, calls identical: {same}", flush=True)
    print(f"deterministic={det}: sum {tot:.3f} ms")
