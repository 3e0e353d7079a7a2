"""Fused assignment-head backward (gf_head_bwd) vs the dS tensor + two library products it replaced, same process."""
import sys, torch
sys.path.insert(0, ".")
import glue_factory_amd
from glue_factory_amd import lib as L_, ops
lib = L_.load()
B, N, D = 32, 2048, 256
g = torch.Generator(device="cuda").manual_seed(0)
md = (torch.randn(2 * B, N, D, device="cuda", generator=g) * 0.25).to(torch.bfloat16)
a, b = md[:B], md[B:]
r, c = ops.rows_lse(a, b), ops.rows_lse(b, a)
gr, gc = torch.randn(B, N, device="cuda", generator=g), torch.randn(B, N, device="cuda", generator=g)
d = torch.empty_like(md)
st = torch.cuda.current_stream().cuda_stream
def fused():
    L_.check(lib.gf_head_bwd(a.data_ptr(), b.data_ptr(), r.data_ptr(), c.data_ptr(), gr.data_ptr(), gc.data_ptr(),
                             d[:B].data_ptr(), d[B:].data_ptr(), B, N, N, D, 1, st), "x")
dS = torch.empty((B, N, N), dtype=torch.bfloat16, device="cuda")
def old():
    L_.check(lib.gf_dual_softmax_bwd(a.data_ptr(), b.data_ptr(), r.data_ptr(), c.data_ptr(), gr.data_ptr(), gc.data_ptr(), None, 0,
                                     0.0, dS.data_ptr(), B, N, N, D, 1, st), "y")
    torch.bmm(dS, b, out=d[:B]); torch.bmm(dS.transpose(1, 2), a, out=d[B:])
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(2):
    print(f"fused {t(fused):.1f} us   dS + 2 bmm {t(old):.1f} us", flush=True)
