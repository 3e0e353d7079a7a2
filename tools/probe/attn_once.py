"""A few launches of gf_attn_fwd / gf_attn_bwd of ONE probe build at the roofline shape (rocprofv3 --pmc target):
python tools/probe/attn_once.py tools/probe/libv_x.so [fwd|bwd|both]"""
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
lib = ctypes.CDLL(sys.argv[1])
what = sys.argv[2] if len(sys.argv) > 2 else "both"
lib.gf_attn_fwd.argtypes = [P, P, P, P, P, I, I, I, I, I, S, S, S, S, F, I, P]
lib.gf_attn_bwd.argtypes = [P] * 10 + [I] * 5 + [S] * 8 + [F, I, P]
for _ in range(4):
    if what in ("fwd", "both") or _ == 0:
        assert lib.gf_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B2, H, N, N, D,
                               st(q), st(k), st(v), st(o), D ** -0.5, 1, stream) == 0
    if what in ("bwd", "both"):
        assert lib.gf_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                               delta.data_ptr(), dqkv[:, :, 0].data_ptr(), dqkv[:, :, 1].data_ptr(), dqkv[:, :, 2].data_ptr(),
                               B2, H, N, N, D, st(q), st(k), st(v), st(o), st(do), st(q), st(k), st(v), D ** -0.5, 1, stream) == 0
torch.cuda.synchronize()
