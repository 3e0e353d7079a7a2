"""gf_conv1_bias_act_bn (backbone.0.0 + tail) at the benchmark's shape, 64 x 1024^2 -> 8.6 GB of bf16: python tools/probe/time_conv1.py [lib.so ...]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = sys.argv[1:] or [os.path.join(ROOT, "glue-factory_amd", "libgf_amd.so")]
B, H, W = 64, 1024, 1024
img = torch.rand(B, H, W, device="cuda").bfloat16()
w = (torch.randn(64, 9, device="cuda") * 0.3).bfloat16()
bias, scale, shift = (torch.randn(64, device="cuda") for _ in range(3))
out = torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
ref = None
for path in libs:
    lib = ctypes.CDLL(path)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.gf_conv1_bias_act_bn.argtypes = [P, P, P, P, P, P, I, I, I, I, I, I, P]
    def run():
        assert lib.gf_conv1_bias_act_bn(img.data_ptr(), w.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), B, H, W, 64, 1, 1, st) == 0
    for _ in range(3): run()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 10)
    cur = out[:2].float().clone()
    if ref is None: ref = cur
    print(f"{os.path.basename(path)}: {best:.3f} ms = {B*H*W*64*2/best/1e9:.2f} TB/s written; max |d| vs first lib {float((cur-ref).abs().max()):.3e}, differing {int((cur!=ref).sum())} of {cur.numel()}")
