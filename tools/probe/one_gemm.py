"""one gf_gemm shape in a loop (rocprofv3 target): python one_gemm.py N K [lib]"""
import ctypes, sys, torch
N, K = int(sys.argv[1]), int(sys.argv[2])
path = sys.argv[3] if len(sys.argv) > 3 else "../../glue-factory_amd/libgf_amd.so"
M = 131072
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib = ctypes.CDLL(path)
lib.gf_gemm.argtypes = [P] * 7 + [I] * 5 + [L] * 5 + [I, P]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda", generator=g)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    assert lib.gf_gemm(x.data_ptr(), None, w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), None, 0, M, N, K, 0, K, 0, K, 0, N, 1, st) == 0
    torch.nn.functional.linear(x, w, bias.bfloat16())
torch.cuda.synchronize()
