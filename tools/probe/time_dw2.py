"""gf_linear_dw2 (one launch over [x1 | x2]) against two gf_linear_dw launches + the cat, at the train step's shape
(M = 131072, 512 <- 256 + 256, bf16), same process: python tools/probe/time_dw2.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from glue_factory_amd import lib as L_
L = L_.load()
M, N, K1, K2 = 131072, 512, 256, 256
g = torch.Generator(device="cuda").manual_seed(0)
dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
x1 = torch.randn(M, K1, device="cuda", generator=g).bfloat16()
x2 = torch.randn(M, K2, device="cuda", generator=g).bfloat16()
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(int(L.gf_linear_dw_ws_bytes(M, N, K1 + K2)), dtype=torch.uint8, device="cuda")
dw = torch.empty(N, K1 + K2, device="cuda"); db = torch.empty(N, device="cuda")
dwa = torch.empty(N, K1, device="cuda"); dwb = torch.empty(N, K2, device="cuda")
def fused():
    assert L.gf_linear_dw2(dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), K1, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, K1 + K2, 1, st) == 0
def split():
    assert L.gf_linear_dw(dy.data_ptr(), x1.data_ptr(), dwa.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, K1, 1, st) == 0
    assert L.gf_linear_dw(dy.data_ptr(), x2.data_ptr(), dwb.data_ptr(), 0, ws.data_ptr(), M, N, K2, 1, st) == 0
    return torch.cat([dwa, dwb], 1)
def timeit(fn, iters=20):
    fn(); best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best
for _ in range(2):
    print(f"two launches + cat {timeit(split) * 1e3:.1f} us   one launch {timeit(fused) * 1e3:.1f} us", flush=True)
