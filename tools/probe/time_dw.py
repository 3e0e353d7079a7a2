"""A/B timing of gf_linear_dw (bf16) at the train step's shapes: python tools/probe/time_dw.py libA.so libB.so"""
import ctypes, sys, torch
M = 131072
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
g = torch.Generator(device="cuda").manual_seed(0)
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
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.gf_linear_dw_ws_bytes.restype = L
    lib.gf_linear_dw_ws_bytes.argtypes = [I, I, I]
    lib.gf_linear_dw.argtypes = [P, P, P, P, P, I, I, I, I, P]
    libs.append((path, lib))
for nout, k in ((768, 256), (256, 256), (512, 256), (256, 512), (512, 512)):
    dy = torch.randn(M, nout, device="cuda", dtype=torch.bfloat16, generator=g)
    x = torch.randn(M, k, device="cuda", dtype=torch.bfloat16, generator=g)
    dw, db = torch.empty(nout, k, device="cuda"), torch.empty(nout, device="cuda")
    row = []
    for path, lib in libs + libs:
        ws = torch.empty(int(lib.gf_linear_dw_ws_bytes(M, nout, k)), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        t = timeit(lambda: lib.gf_linear_dw(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, nout, k, 1, st))
        row.append(f"{t*1e3:7.1f}")
    print(f"{nout}x{k}: " + " ".join(row), flush=True)
