"""Minimal question, no kernel of this repository involved: does a replayed graph always see the value an eager copy wrote into its
input buffer just before the replay -- also when a blocking host-to-device copy from pageable memory was issued in between steps?
graph: out = static_in * 1 (+ a chain of elementwise ops to give it some duration).  Per step: [optional pageable H2D copy],
static_in.copy_(batch_i) (or a kernel copy), g.replay(), then ON THE DEVICE mismatch += any(out != batch_i).  One host read at the end."""
import sys, torch
mode = sys.argv[1] if len(sys.argv) > 1 else "h2d"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = "cuda"
g_ = torch.Generator(device=dev).manual_seed(0)
batches = [torch.randn(8, 256, 256, device=dev, generator=g_) for _ in range(16)]
static_in = torch.zeros(8, 256, 256, device=dev)
pre_cpu = torch.randn(8, 256, 256); pre_dev = torch.zeros(8, 256, 256, device=dev)
def body():
    x = static_in * 1.0
    y = x
    for _ in range(30):
        y = y * 1.0001 + 0.5
    return x, y
for _ in range(3): body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, tail = body()
mismatch = torch.zeros((), device=dev, dtype=torch.int64)
for i in range(steps):
    b = batches[i % 16]
    if mode == "h2d": pre_dev.copy_(pre_cpu)                 # blocking, pageable
    if mode == "h2d_fresh": junk = torch.randn(8, 256, 256).to(dev)
    static_in.copy_(b)
    g.replay()
    mismatch += (out != b).any().long()
print(f"mode {mode}: {steps} replays, device-side mismatch count {int(mismatch)}")
