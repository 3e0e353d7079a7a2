"""The N > 1 path over the REAL RCCL transport, as far as a single-GPU box can take it: a process group of ONE rank
(backend nccl = RCCL).  Every collective is then the identity, but it is issued through RCCL's API on the GPU: communicator
set-up under HSA_ENABLE_IPC_MODE_LEGACY=0, the bucketed gradient all-reduces from autograd hooks (async_op + wait), the packed
SyncBatchNorm exchange, the device-side skip flag riding in the last bucket, and -- what gloo cannot do -- all of it CAPTURED
in the step's hipGraph and replayed.  The 2-rank tests (tests/test_gpu_ddp.py) cover the arithmetic of sharding over gloo;
this one covers the transport calls the driver's multi-GPU run will make."""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["GF_ROOT"])
import torch.distributed as dist
from glue_factory_amd.matchers.superglue import SuperGlue
from glue_factory_amd.optim import FusedAdam
from glue_factory_amd.synthetic import make_pairs, to_device
from glue_factory_amd.train_step import TrainStep, init_distributed
mode, out = sys.argv[1], sys.argv[2]
if mode == "rccl":
    rank, world, local = init_distributed()          # GF_FORCE_DIST=1: a one-rank nccl group
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
torch.cuda.set_device(0)
torch.manual_seed(11)
model = SuperGlue({"GNN_layers": ["self", "cross"], "num_sinkhorn_iterations": 5}).cuda().train()
data = to_device(make_pairs(4, 128, dim=256, size=(640, 480), seed=12), "cuda")
step = TrainStep(model, FusedAdam(model.parameters(), lr=1e-3), amp_dtype=None, graph=True, graph_warmup=2,
                 force_distributed=(mode == "rccl"))
if mode == "rccl":
    assert step.distributed and step.buckets is not None and step.graph, (step.distributed, step.graph)
    assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in step.model.modules())
losses = [float(step(data)["total"].mean()) for _ in range(5)]       # 2 eager, capture, 2 replays
torch.cuda.synchronize()
assert step.skipped == 0
sd = {k: v.detach().cpu() for k, v in step.model.state_dict().items()}
torch.save({"state": sd, "losses": losses, "collectives": step.last_collectives,
            "captured": step._g is not None}, out)
if mode == "rccl":
    dist.barrier(device_ids=[0]); dist.destroy_process_group()
print("worker done", mode, losses)
'''


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return str(s_.getsockname()[1])


def _run(mode, out):
    env = dict(os.environ, GF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "rccl":
        env["GF_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, "-c", WORKER, mode, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out)


def test_one_rank_rccl_step_with_captured_collectives_equals_the_plain_step():
    with tempfile.TemporaryDirectory() as d:
        got = _run("rccl", os.path.join(d, "a.pt"))
        ref = _run("plain", os.path.join(d, "b.pt"))
    assert got["captured"] and ref["captured"]
    n_bn = sum(1 for k in ref["state"] if k.endswith("running_mean"))
    assert got["collectives"] == {"gradient_buckets": got["collectives"]["gradient_buckets"], "syncbn": 2 * n_bn}, got["collectives"]
    assert got["collectives"]["gradient_buckets"] >= 1
    for a, b in zip(got["losses"], ref["losses"]):
        assert abs(a - b) < 1e-4 * max(1.0, abs(b)), (got["losses"], ref["losses"])
    worst = 0.0
    for k, v in ref["state"].items():
        if not v.dtype.is_floating_point:
            assert int(got["state"][k]) == int(v), k
            continue
        sc = float(v.abs().max()) + 1e-6
        worst = max(worst, float((got["state"][k] - v).abs().max()) / sc)
    print(f"one-rank RCCL run vs plain run after 5 Adam steps: worst relative state difference {worst:.2e}")
    assert worst < 2e-3          # (the packed SyncBatchNorm kernels sum in another order; 5 Adam steps at lr 1e-3)


def test_bench_multi_rank_code_path_over_rccl_one_rank():
    """`bench.py --gpus N`'s N > 1 branch -- eager measurement first, the captured step under the watchdog, the data_parallel
    report with its measured all-reduce -- in a one-rank RCCL group (GF_FORCE_DIST=1)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               GF_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--matcher-only", "--batch", "4", "--kpts", "512", "--layers", "2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    dp = line["data_parallel"]
    assert dp["backend"] == "nccl" and dp["ranks_seen"] == [0] and dp["reducer"] == "buckets"
    assert dp["collectives_per_step"]["gradient_buckets"] == dp["buckets"] and dp["collectives_per_step"]["syncbn"] == 0
    assert dp["step_launch_modes"]["graph"]["status"] == "ok", dp["step_launch_modes"]
    assert dp["allreduce_16MB_ms"] > 0
    print("bench one-rank RCCL:", json.dumps(dp))
