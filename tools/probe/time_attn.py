"""time gf_attn_fwd / gf_attn_bwd of probe builds: python tools/probe/time_attn.py libv_a.so libv_b.so ..."""
import ctypes, sys, torch
B2, H, N, D = 64, 4, 2048, 64
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
S = ctypes.POINTER(ctypes.c_int64)
def st(t): return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B2, N, 3, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
do = torch.randn(B2, N, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
o = torch.empty(B2, N, H, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B2, H, N, device="cuda"); delta = torch.empty(2, B2, H, N, device="cuda")
dqkv = torch.empty_like(qkv)
stream = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=20):
    for _ in range(10): fn()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best
# "+zero" after the mode (lib.so@ln2+zero) times the same launches on all-zero operands: same instruction stream, no
# data toggling -- separates issue limits from the power limit
# an argument "lib.so@ln2" times the pre-multiplied-operand instantiations (scale = ln 2: no multiply per score)
for arg in sys.argv[1:]:
    path, _, mode = arg.partition("@")
    mode, _, data = mode.partition("+")
    if data == "zero":
        qkv.zero_(); do.zero_()
    elif data == "small":
        qkv.mul_(1e-3); do.mul_(1e-3)
    SC = 0.6931471805599453 if mode == "ln2" else D ** -0.5
    lib = ctypes.CDLL(path)
    lib.gf_attn_fwd.argtypes = [P, P, P, P, P, I, I, I, I, I, S, S, S, S, F, I, P]
    lib.gf_attn_bwd.argtypes = [P] * 10 + [I] * 5 + [S] * 8 + [F, I, P]
    def fwd():
        rc = lib.gf_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B2, H, N, N, D,
                             st(q), st(k), st(v), st(o), SC, 1, stream)
        assert rc == 0, rc
    def bwd():
        rc = lib.gf_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                             delta.data_ptr(), dqkv[:, :, 0].data_ptr(), dqkv[:, :, 1].data_ptr(), dqkv[:, :, 2].data_ptr(),
                             B2, H, N, N, D, st(q), st(k), st(v), st(o), st(do), st(q), st(k), st(v), SC, 1, stream)
        assert rc == 0, rc
    fwd()
    print(f"{arg}: fwd {timeit(fwd)*1e3:.1f} us   bwd {timeit(bwd)*1e3:.1f} us", flush=True)
ga = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); gb = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
gc = torch.empty_like(ga)
t = timeit(lambda: torch.matmul(ga, gb, out=gc), iters=10)
print(f"hipBLASLt 8192^3 bf16 random: {t*1e3:.0f} us = {2*8192**3/t/1e9:.0f} TFLOP/s")
ga.zero_(); gb.zero_()
t = timeit(lambda: torch.matmul(ga, gb, out=gc), iters=10)
print(f"hipBLASLt 8192^3 bf16 zeros : {t*1e3:.0f} us = {2*8192**3/t/1e9:.0f} TFLOP/s")
