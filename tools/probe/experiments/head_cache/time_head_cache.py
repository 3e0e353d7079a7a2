"""The three dependent passes of a LightGlue layer head (B = 32, N = 2048, D = 256, bf16): recomputing kernels (csrc/assignment.hip)
against the storing first pass + the two streaming passes of csrc/head_cache.hip, kernel by kernel in one process."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_amd import lib as _lib, ops
L = _lib.load()
B, N, D = 32, 2048, 256
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.randn(B, N, D, device="cuda", generator=g) * 0.35).to(torch.bfloat16)
b = (torch.randn(B, N, D, device="cuda", generator=g) * 0.35).to(torch.bfloat16)
z0 = torch.randn(B, N, device="cuda", generator=g); z1 = torch.randn(B, N, device="cuda", generator=g)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
c = torch.empty(B, N, device="cuda"); r = torch.empty(B, N, device="cuda")
v = torch.empty(B, N, device="cuda"); ai = torch.empty(B, N, dtype=torch.int64, device="cuda")
s16 = torch.empty(B, N, N, dtype=torch.float16, device="cuda")
ws = torch.empty(int(L.gf_cached_cols_ws_bytes(B, N, N)), dtype=torch.uint8, device="cuda")
def timeit(fn, iters=20):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3
k = {
 "pass 1 rows_lse (recompute)": lambda: L.gf_rows_lse(p(b), p(a), None, p(c), B, N, N, D, 1, st),
 "pass 1 rows_lse_cache (+ fp16 store)": lambda: L.gf_rows_lse_cache(p(b), p(a), p(c), p(s16), B, N, N, D, 1, st),
 "pass 2 rows_lse_argmax (recompute)": lambda: L.gf_rows_lse_argmax(p(a), p(b), p(z1), p(c), 2.0, p(r), p(v), p(ai), B, N, N, D, 1, st),
 "pass 2 cached_rows_lse_argmax": lambda: L.gf_cached_rows_lse_argmax(p(s16), p(z1), p(c), 2.0, p(r), p(v), p(ai), B, N, N, st),
 "pass 3 rows_lse_argmax cols (recompute)": lambda: L.gf_rows_lse_argmax(p(b), p(a), p(z0), p(r), 2.0, None, p(v), p(ai), B, N, N, D, 1, st),
 "pass 3 cached_cols_argmax (+ merge)": lambda: L.gf_cached_cols_argmax(p(s16), p(z0), p(r), 2.0, p(v), p(ai), p(ws), B, N, N, st),
}
for name, fn in k.items():
    assert fn() == 0, name
    print(f"{name:42s} {timeit(fn):8.1f} us", flush=True)
x = torch.empty(B * N * N, dtype=torch.float16, device="cuda")
print(f"{'torch copy of the 268 MB cache (r + w)':42s} {timeit(lambda: x.copy_(s16.view(-1))):8.1f} us")
