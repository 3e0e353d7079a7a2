"""gf_ln_gelu_fwd / gf_ln_gelu_bwd of probe builds at the FFN's shape ([131072, 512] bf16), one process:
python tools/probe/time_ln.py tools/probe/libv_a.so tools/probe/libv_b.so"""
import ctypes, sys, torch
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
R, C = 131072, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(R, C, device="cuda", generator=g).bfloat16()
dy = torch.randn(R, C, device="cuda", generator=g).bfloat16()
gamma, beta = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g)
y, dx = torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best * 1e3
ref = None
for path in sys.argv[1:] * 2:
    lib = ctypes.CDLL(path)
    lib.gf_ln_gelu_fwd.argtypes = [P, P, P, P, P, P, I, I, F, I, P]
    lib.gf_ln_gelu_bwd.argtypes = [P, P, P, P, P, P, P, P, P, I, I, I, P]
    nb = lib.gf_ln_gelu_nblk(R)
    dg, db = torch.empty(nb, C, device="cuda"), torch.empty(nb, C, device="cuda")
    fw = lambda: lib.gf_ln_gelu_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, C, 1e-5, 1, st)
    bw = lambda: lib.gf_ln_gelu_bwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), R, C, 1, st)
    assert fw() == 0 and bw() == 0
    torch.cuda.synchronize()
    out = (y.clone(), dx.clone())
    if ref is None: ref = out
    print(f"{path}: fwd {t(fw):.1f} us  bwd {t(bw):.1f} us  same as first: {torch.equal(out[0], ref[0])} {torch.equal(out[1], ref[1])}", flush=True)
