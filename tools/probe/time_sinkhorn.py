"""time gf_sinkhorn_fwd / gf_sinkhorn_bwd of probe builds (B=32, N=2048, 100 iterations) in ONE process and check each
against the first: python tools/probe/time_sinkhorn.py libv_a.so libv_b.so ...
An argument of the form  path@0 / path@1  passes schedule 0 / 1 to the calls of that run (streaming vs chip-resident sweeps;
ABI >= 14: the schedule is an argument of gf_sinkhorn_fwd / _bwd);
GF_PROBE_B / GF_PROBE_T override the batch and the iteration count."""
import ctypes, os, sys, torch
B, N, T = int(os.environ.get("GF_PROBE_B", 32)), 2048, int(os.environ.get("GF_PROBE_T", 100))
P, I = ctypes.c_void_p, ctypes.c_int
g = torch.Generator(device="cuda").manual_seed(0)
Z = torch.randn(B, N + 1, N + 1, device="cuda", generator=g) * 2
G = torch.randn(B, N + 1, N + 1, device="cuda", generator=g)
gr, gc = G.sum(2).contiguous(), G.sum(1).contiguous()
out, gZ = torch.empty_like(Z), torch.empty_like(Z)
uh = torch.empty(T, B, N + 1, device="cuda"); vh = torch.empty(T, B, N + 1, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=3):
    fn(); best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best
ref = None
for arg in sys.argv[1:]:
    path, _, mode = arg.partition("@")
    lib = ctypes.CDLL(path)
    sched = int(mode) if mode else 1
    lib.gf_sinkhorn_ws_bytes.restype = ctypes.c_int64
    lib.gf_sinkhorn_ws_bytes.argtypes = [I, I, I, I]
    lib.gf_sinkhorn_fwd.argtypes = [P, P, P, P, P, I, I, I, I, I, P]
    lib.gf_sinkhorn_bwd.argtypes = [P] * 8 + [I, I, I, I, I, P]
    ws = torch.empty(int(lib.gf_sinkhorn_ws_bytes(B, N, N, T)), dtype=torch.uint8, device="cuda")
    def fwd():
        assert lib.gf_sinkhorn_fwd(Z.data_ptr(), out.data_ptr(), uh.data_ptr(), vh.data_ptr(), ws.data_ptr(), B, N, N, T, sched, st) == 0
    def bwd():
        assert lib.gf_sinkhorn_bwd(Z.data_ptr(), G.data_ptr(), gr.data_ptr(), gc.data_ptr(), uh.data_ptr(), vh.data_ptr(),
                                   gZ.data_ptr(), ws.data_ptr(), B, N, N, T, sched, st) == 0
    tf, tb = timeit(fwd), timeit(bwd)
    cur = (out.clone(), gZ.clone())
    if ref is None:
        ref = cur
    d = [float((a - b).abs().max()) for a, b in zip(cur, ref)]
    print(f"{arg}: fwd {tf:.2f} ms  bwd {tb:.2f} ms   max|d out| {d[0]:.2e} max|d gZ| {d[1]:.2e} vs first", flush=True)
