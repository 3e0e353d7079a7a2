"""A/B of gf_conv3x3_c64 builds in one process: python tools/probe/time_conv3x3_variants.py name ..."""
import ctypes, sys, torch
names = sys.argv[1:]
libs = {n: ctypes.CDLL(f"tools/probe/libv_{n}.so") for n in names}
P, I = ctypes.c_void_p, ctypes.c_int
for l in libs.values():
    l.gf_conv3x3_c64.argtypes = [P, P, P, P, P, P, I, I, I, I, I, I, P]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (B, H, W, pool) in [(64, 1024, 1024, 1), (64, 512, 512, 0)]:
    x = torch.randn(B, 64, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    taps = (torch.randn(9, 64, 64, device="cuda") * 0.05).to(torch.bfloat16)
    bias, scale, shift = (torch.randn(64, device="cuda") for _ in range(3))
    out = torch.empty((B, 64, H // 2, W // 2) if pool else (B, 64, H, W), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for rep in range(2):
        for n, l in libs.items():
            def f():
                assert l.gf_conv3x3_c64(x.data_ptr(), taps.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                        out.data_ptr(), B, H, W, 1, pool, 1, st) == 0
            res.append(f"{n} {t(f):.3f}")
    print(f"B{B} {H}x{W} pool{pool}: " + "  ".join(res), flush=True)
