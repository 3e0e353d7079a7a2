"""A/B timing of the BatchNorm statistics / apply kernels at the SuperGlue / GlueStick step's shapes (one image set:
M = 65536 or 98304 rows, C = 512, bf16): python tools/probe/time_bn.py libA.so libB.so"""
import ctypes, sys, torch
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
def timeit(fn, iters=50):
    fn(); best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best * 1e3
st = torch.cuda.current_stream().cuda_stream
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.gf_bn_stats.argtypes = [P, P, I, I, I, P]
    lib.gf_bn_bwd_stats.argtypes = [P, P, P, P, P, P, P, I, I, I, I, P]
    lib.gf_bn_act_fwd.argtypes = [P, P, P, P, P, P, I, I, I, I, P]
    lib.gf_bn_bwd_dx.argtypes = [P, P, P, P, P, P, P, P, P, I, I, I, I, P]
    for M in (65536, 98304):
        C = 512
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, C, device="cuda", generator=g).bfloat16(); dy = torch.randn(M, C, device="cuda", generator=g).bfloat16()
        y = torch.empty_like(x)
        part = torch.empty(512, 2, C, device="cuda")
        mean = torch.zeros(C, device="cuda"); rstd = torch.ones(C, device="cuda"); ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda")
        t1 = timeit(lambda: lib.gf_bn_stats(x.data_ptr(), part.data_ptr(), M, C, 1, st))
        t2 = timeit(lambda: lib.gf_bn_bwd_stats(x.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ga.data_ptr(), be.data_ptr(), part.data_ptr(), M, C, 1, 1, st))
        t3 = timeit(lambda: lib.gf_bn_act_fwd(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ga.data_ptr(), be.data_ptr(), y.data_ptr(), M, C, 1, 1, st))
        t4 = timeit(lambda: lib.gf_bn_bwd_dx(x.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ga.data_ptr(), be.data_ptr(), mean.data_ptr(), mean.data_ptr(), y.data_ptr(), M, C, 1, 1, st))
        byt = M * C * 2 / 1e6
        print(f"{path} M={M}: stats fwd {t1:.1f} us ({byt / t1:.2f} TB/s)  stats bwd {t2:.1f} us ({2 * byt / t2:.2f})  apply fwd {t3:.1f} us ({2 * byt / t3:.2f})  dx {t4:.1f} us ({3 * byt / t4:.2f})", flush=True)
