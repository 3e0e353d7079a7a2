"""Which op of the step changes its numbers when a BLOCKING host-to-device copy from pageable memory runs while its captured
graph is being replayed?  Each op (forward + backward) is captured alone and replayed many times, a background thread issuing
pageable H2D copies the whole time; every replay's outputs are compared bit for bit with a quiet replay's."""
import os, sys, threading, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_amd import ops
torch.manual_seed(0)
dev = "cuda"
B, N, D = 8, 256, 256
M = B * N


def case_bn():
    bn = torch.nn.BatchNorm1d(512).to(dev).train()
    x = torch.randn(2, M, 512, device=dev, dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(2, M, 512, device=dev, dtype=torch.bfloat16)
    def f():
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        x.grad = None; bn.weight.grad = None; bn.bias.grad = None
        y = ops.batch_norm_act_sets(x, bn, relu=True, replay=True)
        y.backward(dy)
        return [y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var]
    return f


def case_linear():
    w = torch.randn(512, 512, device=dev, requires_grad=True) * 0.05
    w = w.detach().requires_grad_(True)
    b = torch.zeros(512, device=dev, requires_grad=True)
    x = torch.randn(2 * M, 512, device=dev, dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(2 * M, 512, device=dev, dtype=torch.bfloat16)
    def f():
        x.grad = None; w.grad = None; b.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ops.linear(x, w, b)
        y.backward(dy)
        return [y.detach(), x.grad, w.grad, b.grad]
    return f


def case_attention():
    qkv = torch.randn(2 * B, N, 3, 4, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(2 * B, N, 4, 64, device=dev, dtype=torch.bfloat16)
    def f():
        qkv.grad = None
        o = ops.attention_qkv(qkv, cross=True)
        o.backward(do)
        return [o.detach(), qkv.grad]
    return f


def case_sinkhorn():
    Z = (torch.randn(B, N + 1, N + 1, device=dev) * 2).requires_grad_(True)
    G = torch.randn(B, N + 1, N + 1, device=dev)
    def f():
        Z.grad = None
        out = ops.sinkhorn(Z, 20)
        (out * G).sum().backward()
        return [out.detach(), Z.grad]
    return f


CASES = {"bn": case_bn, "linear": case_linear, "attention": case_attention, "sinkhorn": case_sinkhorn}
stop = False


def copier():
    src = torch.randn(8, 256, 256)
    dst = torch.zeros(8, 256, 256, device=dev)
    side = torch.cuda.Stream()
    while not stop:
        dst.copy_(src)                      # blocking copy from pageable memory
        time.sleep(0.0005)


for name in (sys.argv[1:] or list(CASES)):
    f = CASES[name]()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = f()
    g.replay(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    quiet = 0
    for _ in range(100):
        g.replay(); torch.cuda.synchronize()
        quiet += int(not all(torch.equal(a, b) for a, b in zip(outs, ref)))
    stop = False
    th = threading.Thread(target=copier); th.start()
    noisy, worst = 0, 0.0
    for _ in range(400):
        g.replay(); torch.cuda.synchronize()
        bad = [i for i, (a, b) in enumerate(zip(outs, ref)) if not torch.equal(a, b)]
        if bad:
            noisy += 1
            worst = max(worst, max(float((outs[i].float() - ref[i].float()).abs().max()) for i in bad))
    stop = True; th.join()
    print(f"{name:10s}: quiet replays that differ {quiet}/100; replays under concurrent pageable H2D copies that differ {noisy}/400 (max |d| {worst:.3e})", flush=True)
