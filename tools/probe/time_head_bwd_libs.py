"""gf_head_bwd of probe builds in one process: python tools/probe/time_head_bwd_libs.py liba.so libb.so"""
import ctypes, sys, torch
P, I = ctypes.c_void_p, ctypes.c_int
B, N, D = 32, 2048, 256
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.randn(B, N, D, device="cuda", generator=g) * 0.25).bfloat16()
b = (torch.randn(B, N, D, device="cuda", generator=g) * 0.25).bfloat16()
S = torch.bmm(a.float(), b.float().transpose(1, 2))
r, c = S.logsumexp(2).contiguous(), S.logsumexp(1).contiguous()
del S
gr, gc = torch.randn(B, N, device="cuda", generator=g), torch.randn(B, N, device="cuda", generator=g)
da, db = torch.empty_like(a), torch.empty_like(b)
st = torch.cuda.current_stream().cuda_stream
for path in sys.argv[1:] * 2:
    lib = ctypes.CDLL(path)
    lib.gf_head_bwd.argtypes = [P] * 8 + [I] * 5 + [P]
    fn = lambda: lib.gf_head_bwd(a.data_ptr(), b.data_ptr(), r.data_ptr(), c.data_ptr(), gr.data_ptr(), gc.data_ptr(), da.data_ptr(), db.data_ptr(), B, N, N, D, 1, st)
    assert fn() == 0
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{path}: head_bwd {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
