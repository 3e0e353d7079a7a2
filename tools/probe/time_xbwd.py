"""time gf_attn_cross_bwd of probe builds (tools/probe/build_xbwd_variants.sh) at the benchmark geometry, same process:
   python tools/probe/time_xbwd.py libx_a.so libx_b.so ...   (+ the two gf_attn_bwd_acc launches it replaces, from libgf_amd.so)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_amd import ops
B2, H, N, D = 64, 4, 2048, 64
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
S = ctypes.POINTER(ctypes.c_int64)
def st(t): return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))
g = torch.Generator(device="cuda").manual_seed(0)
p = (torch.randn(B2, N, 2, H, D, device="cuda", generator=g) * 0.6).bfloat16()
dm = torch.randn(B2, N, H, D, device="cuda", generator=g).bfloat16()
m = ops.cross_attention_stacked(p.clone().requires_grad_(True), scale=ops.LN2)
lse = torch.randn(B2, H, N, device="cuda") * 0.1 + 11.0
stat = torch.empty(2, B2, H, N, device="cuda")
d = torch.empty_like(p)
stream = torch.cuda.current_stream().cuda_stream
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
    return best
ps = p.clone().requires_grad_(True)
mm = ops.cross_attention_stacked(ps, scale=ops.LN2)
def old():
    ops.XBWD_ENABLED = False
    ps.grad = None
    mm.backward(dm, retain_graph=True)
    ops.XBWD_ENABLED = True
print(f"two gf_attn_bwd_acc launches (autograd op): {timeit(old)*1e3:.1f} us", flush=True)
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.gf_attn_cross_bwd.argtypes = [P] * 8 + [I] * 5 + [S] * 6 + [F, I, P]
    def run():
        rc = lib.gf_attn_cross_bwd(p[:, :, 0].data_ptr(), p[:, :, 1].data_ptr(), m.data_ptr(), dm.data_ptr(), lse.data_ptr(),
                                   stat.data_ptr(), d[:, :, 0].data_ptr(), d[:, :, 1].data_ptr(), B2, B2 // 2, H, N, D,
                                   st(p[:, :, 0]), st(p[:, :, 1]), st(m), st(dm), st(d[:, :, 0]), st(d[:, :, 1]),
                                   0.6931471805599453, 1, stream)
        assert rc == 0, rc
    print(f"{os.path.basename(path)}: {timeit(run)*1e3:.1f} us", flush=True)
