"""gf_gemm of probe builds vs the library GEMM at the LightGlue step's shapes (M = 131072 tokens, bf16), ONE process:
python tools/probe/time_gemm_ws.py ../../glue-factory_amd/libgf_amd.so ./libv_x.so ..."""
import ctypes, sys, torch
M = 131072
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
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
    return best * 1e3
shapes = [(768, 256, False), (256, 256, False), (512, 512, True), (256, 512, False), (512, 256, False)]
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.gf_gemm.argtypes = [P] * 7 + [I] * 5 + [L] * 5 + [I, P]
    libs.append((path, lib))
for N, K, two in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bb = bias.bfloat16()
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w, bb))
    byt = M * (K + N) * 2
    line = f"N={N:4d} K={K:4d}: library {t_lib:6.1f} us ({byt / t_lib / 1e6:5.2f} TB/s)"
    ref = torch.nn.functional.linear(x, w, bb).float()
    for path, lib in libs:
        if two:
            x0, x1 = x[:, :K // 2], x[:, K // 2:]
            fn = lambda: lib.gf_gemm(x0.data_ptr(), x1.data_ptr(), w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), None, 0,
                                     M, N, K // 2, K // 2, x.stride(0), x.stride(0), K, 0, N, 1, st)
        else:
            fn = lambda: lib.gf_gemm(x.data_ptr(), None, w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), None, 0,
                                     M, N, K, 0, K, 0, K, 0, N, 1, st)
        rc = fn(); assert rc == 0, rc
        t = timeit(fn)
        err = (y.float() - ref).abs().max().item()
        line += f" | {path.split('/')[-1]} {t:6.1f} us ({byt / t / 1e6:5.2f} TB/s, err {err:.3f})"
    print(line, flush=True)
