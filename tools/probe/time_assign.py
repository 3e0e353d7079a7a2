"""gf_assign_write at the benchmark size, same process, for builds with different GF_WRITE_SPLIT: python time_assign.py lib1.so lib2.so"""
import ctypes, sys, torch
B, N, D = 32, 2048, 256
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(B, N, D, device="cuda", generator=g).bfloat16(); b = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
rb = torch.randn(B, N, device="cuda"); cb = torch.randn(B, N, device="cuda"); bc = torch.randn(B, N, device="cuda"); br = torch.randn(B, N, device="cuda")
out = torch.empty(B, N + 1, N + 64, device="cuda"); es = torch.empty(B, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ref = None
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.gf_assign_write.argtypes = [P] * 6 + [F, F, P, P, I, I, I, I, I, P]
    def run():
        rc = lib.gf_assign_write(a.data_ptr(), b.data_ptr(), rb.data_ptr(), cb.data_ptr(), bc.data_ptr(), br.data_ptr(), 2.0, 0.0,
                                 out.data_ptr(), es.data_ptr(), B, N, N, D, 1, st)
        assert rc == 0, rc
    for _ in range(5): run()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 20)
    if ref is None: ref = out.clone(); eref = es.clone()
    print(f"{path}: {best*1e3:.1f} us = {B*(N+1)**2*4/best/1e6:.0f} GB/s   equal to the first build: {torch.equal(out, ref)}  expsum close: {torch.allclose(es, eref, rtol=1e-5)}", flush=True)
