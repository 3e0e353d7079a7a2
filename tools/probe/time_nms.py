"""gf_nms_scores of probe builds on 64 x 1024^2 score maps (radius 3), one process: python time_nms.py lib..."""
import ctypes, sys, torch
P, I = ctypes.c_void_p, ctypes.c_int
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.rand(64, 1024, 1024, device="cuda", generator=g)
st = torch.cuda.current_stream().cuda_stream
ref = None
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.gf_nms_scores.argtypes = [P, P, I, I, I, I, I, P]
    out = torch.empty_like(s)
    fn = lambda: lib.gf_nms_scores(s.data_ptr(), out.data_ptr(), 64, 1024, 1024, 3, 4, st)
    assert fn() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    print(f"{path}: {a.elapsed_time(b) / 5:.3f} ms  equal to first: {torch.equal(out, ref)}", flush=True)
