"""per-workgroup phase timestamps of gf_gemm (probe build with -DGW_TRACE=1): python trace_gemm.py ./libv_gwtrace.so"""
import ctypes, sys, torch, numpy as np
N, K, M = 256, 256, 131072
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib = ctypes.CDLL(sys.argv[1])
lib.gf_gemm.argtypes = [P] * 7 + [I] * 5 + [L] * 5 + [I, P]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) / 16).bfloat16()
bias = torch.randn(N, device="cuda", generator=g)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.gf_gemm(x.data_ptr(), None, w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), None, 0, M, N, K, 0, K, 0, K, 0, N, 1, st)
torch.cuda.synchronize()
nwg = M // 128
buf = (ctypes.c_ulonglong * (16 * nwg))()
lib.gf_gemm_trace.argtypes = [P, I]
assert lib.gf_gemm_trace(buf, 16 * nwg) == 0
t = np.array(buf, dtype=np.float64).reshape(nwg, 16)
t0 = t[:, 0].min()
names = ["start", "x loaded", "barrier0", "mfma pair0", "flush0", "slice0 end", "bar1", "slice1 end", "bar2", "slice2 end", "bar3", "slice3 end", "end"]
rel = t - t0
for i, n in enumerate(names):
    col = rel[:, i]
    print(f"{n:12s} min {col.min():10.0f} median {np.median(col):10.0f} max {col.max():10.0f}")
d = np.diff(t[:, :13], axis=1)
for i, n in enumerate(names[1:]):
    print(f"d {n:12s} median {np.median(d[:, i]):9.0f} p90 {np.percentile(d[:, i], 90):9.0f}")
print("start-time histogram:", np.histogram(rel[:, 0], bins=8)[0].tolist(), np.histogram(rel[:, 0], bins=8)[1].round().tolist())
