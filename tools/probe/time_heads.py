"""A/B timing of the assignment-head kernels at the train step's shape: python tools/probe/time_heads.py libA.so libB.so"""
import ctypes, sys, torch
B, N, D = 32, 2048, 256
P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best
a = (torch.randn(B, N, D, device="cuda", generator=g) * 0.3).bfloat16()
b = (torch.randn(B, N, D, device="cuda", generator=g) * 0.3).bfloat16()
z = torch.randn(B, N, device="cuda", generator=g)
r = torch.empty(B, N, device="cuda"); c = torch.empty(B, N, device="cuda")
gr = torch.randn(B, N, device="cuda", generator=g); gc = torch.randn(B, N, device="cuda", generator=g)
vmax = torch.empty(B, N, device="cuda"); arg = torch.empty(B, N, dtype=torch.int64, device="cuda")
dS = torch.empty(B, N, N, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
rows = {}
for path in sys.argv[1:] * 2:
    lib = ctypes.CDLL(path)
    lib.gf_rows_lse.argtypes = [P, P, P, P, I, I, I, I, I, P]
    lib.gf_rows_lse_argmax.argtypes = [P, P, P, P, F, P, P, P, I, I, I, I, I, P]
    lib.gf_dual_softmax_bwd.argtypes = [P, P, P, P, P, P, P, L, F, P, I, I, I, I, I, P]
    lib.gf_rows_lse(a.data_ptr(), b.data_ptr(), None, r.data_ptr(), B, N, N, D, 1, st)
    lib.gf_rows_lse(b.data_ptr(), a.data_ptr(), None, c.data_ptr(), B, N, N, D, 1, st)
    t = {
        "rows_lse": timeit(lambda: lib.gf_rows_lse(a.data_ptr(), b.data_ptr(), None, r.data_ptr(), B, N, N, D, 1, st)),
        "lse_argmax": timeit(lambda: lib.gf_rows_lse_argmax(a.data_ptr(), b.data_ptr(), z.data_ptr(), c.data_ptr(), 2.0, r.data_ptr(), vmax.data_ptr(), arg.data_ptr(), B, N, N, D, 1, st)),
        "argmax": timeit(lambda: lib.gf_rows_lse_argmax(a.data_ptr(), b.data_ptr(), z.data_ptr(), c.data_ptr(), 2.0, None, vmax.data_ptr(), arg.data_ptr(), B, N, N, D, 1, st)),
        "dual_bwd": timeit(lambda: lib.gf_dual_softmax_bwd(a.data_ptr(), b.data_ptr(), r.data_ptr(), c.data_ptr(), gr.data_ptr(), gc.data_ptr(), None, 0, 0.0, dS.data_ptr(), B, N, N, D, 1, st)),
    }
    print(path, " ".join(f"{k}={v*1e3:.1f}us" for k, v in t.items()), flush=True)
