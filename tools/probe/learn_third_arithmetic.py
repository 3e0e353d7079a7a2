"""A THIRD arithmetic for the learning-curve question of tests/test_gpu_zz_learning.py (round-5 review, item 1b/1c): the SuperGlue
of tests/learning_cases.py trained for lc.STEPS steps on the same batches from the same initial state with STOCK torch ops -- no
kernel of libgf_amd.so -- on whatever device / dtype is asked for:

    python tools/probe/learn_third_arithmetic.py superglue cuda float32 out.pt     # PyTorch-ROCm's own GEMMs / reductions on the MI355X
    python tools/probe/learn_third_arithmetic.py superglue cuda float64 out.pt     # an fp64 anchor computed on the GPU
    python tools/probe/learn_third_arithmetic.py superglue cpu  float64 out.pt     # the same anchor on the CPU (build container)

The model code is oracle/superglue_oracle.py (the functional restatement that tests/test_oracle_golden.py and
tests/test_reference_oracle_sweep.py hold to the reference at 1e-4); BatchNorm running statistics are updated the way the
reference module updates them, INCLUDING the second update of the GNN layers that the reference's activation checkpointing
causes (superglue.py:160-169: the block's forward is re-run in training mode during backward).  torch.optim.Adam with the
reference loop's settings.  TEST INFRASTRUCTURE: runs the oracle, never part of the product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import learning_cases as lc  # noqa: E402


def main():
    kind, device, dtype, out = sys.argv[1], sys.argv[2], getattr(torch, sys.argv[3]), sys.argv[4]
    assert kind == "superglue"
    from oracle import superglue_oracle as sgo
    torch.set_num_threads(int(os.environ.get("GF_THREADS", "8")))
    if device == "cuda":           # full-precision products: no TF32-style shortcuts in the "stock" run
        torch.backends.cuda.matmul.allow_tf32 = False
        torch.backends.cudnn.allow_tf32 = False
    conf = lc.conf(kind)
    names, iters = conf["GNN_layers"], conf["num_sinkhorn_iterations"]
    state = {k: v.to(device=device, dtype=dtype if v.is_floating_point() else v.dtype) for k, v in lc.initial_params(kind).items()}
    train = sgo.trainable_names(state)
    for k in train:
        state[k].requires_grad_(True)
    opt = torch.optim.Adam([state[k] for k in train], lr=lc.LR[kind])

    def bn_with_buffers(p, name, x, training):
        b, n, c = x.shape
        rm, rv = p[name + ".running_mean"], p[name + ".running_var"]
        y = F.batch_norm(x.reshape(b * n, c), rm, rv, p[name + ".weight"], p[name + ".bias"], training=training, momentum=0.1, eps=1e-5)
        if training:
            p[name + ".num_batches_tracked"] += 1
            if name.startswith("gnn."):        # the checkpointed block's forward runs a second time during the reference's backward
                with torch.no_grad():
                    F.batch_norm(x.detach().reshape(b * n, c), rm, rv, None, None, training=True, momentum=0.1, eps=1e-5)
                p[name + ".num_batches_tracked"] += 1
        return y.reshape(b, n, c)

    sgo._bn = bn_with_buffers

    def batch(seed):
        d = lc.batch(kind, seed)
        d = {k: v for k, v in d.items() if torch.is_tensor(v)} | {"image_size0": d["view0"]["image_size"], "image_size1": d["view1"]["image_size"]}
        return {k: v.to(device=device, dtype=dtype if v.is_floating_point() else v.dtype) for k, v in d.items()}

    trace, t0 = [], time.time()
    for i in range(int(os.environ.get("GF_STEPS", lc.STEPS))):
        data = batch(1000 + i)
        opt.zero_grad(set_to_none=True)
        pred = sgo.forward(state, data, names, iters, training=True)
        loss = sgo.loss(state, pred, data)["total"].mean()
        loss.backward()
        opt.step()
        if i % 50 == 49:
            trace.append(round(float(loss), 3))
            print(i + 1, trace[-1], f"{time.time() - t0:.0f} s", flush=True)
    print(f"{kind} {device} {sys.argv[3]} train loss every 50 steps: {trace}")
    torch.save({k: v.detach().to("cpu", torch.float32 if v.is_floating_point() else v.dtype) for k, v in state.items()}, out)


if __name__ == "__main__":
    main()
