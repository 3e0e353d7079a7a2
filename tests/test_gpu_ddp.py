"""The real HIP LightGlue data-parallel: two processes share cuda:0 and talk over gloo (one MI355X is all the test
box has; RCCL refuses two ranks on one device), with the reference's file:// rendezvous.  Everything of the N>1 path
except the RCCL transport itself runs here, for BOTH gradient reducers: the capturable bucket reducer (GradBuckets:
post-accumulate hooks on our custom autograd nodes, flat fp32 buckets, the skip flag in the last bucket) and stock
DistributedDataParallel; the per-step precast cache, the do_backward agreement, the fused loss.
After one SGD step on the sharded batch the weights must equal a single-process step on the whole batch."""
import os
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

L, N, B = 2, 192, 4


def _model_and_data():
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    from oracle import lightglue_oracle as lgo
    model = LightGlue({"n_layers": L}).cuda().train()
    model.load_state_dict(lgo.init_params(L, 256, 4, seed=5))
    data = to_device(make_pairs(B, N, dim=256, size=(640, 480), seed=6), "cuda")
    return model, data


def _worker(rank, world, lock, out, reducer):
    from glue_factory_amd.train_step import TrainStep, init_distributed, reduce_losses, shard_batch
    torch.cuda.set_device(0)
    init_distributed("gloo", init_method="file://" + lock, rank=rank, world_size=world)
    model, data = _model_and_data()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    step = TrainStep(model, opt, amp_dtype=None, device_ids=[0], reducer=reducer, bucket_cap_mb=4)
    assert step.distributed and (step.buckets is not None) == reducer.startswith("buckets")
    if reducer.startswith("buckets"):
        assert len(step.buckets.buckets) >= 3 and not step.graph      # (gloo is not capturable; nccl would be)
    losses = step(shard_batch(data, rank, world))
    red = reduce_losses(losses)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "red": red}, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("reducer", ["buckets", "buckets_bound", "ddp"])
def test_two_rank_ddp_step_equals_single_process(reducer):
    from glue_factory_amd.train_step import TrainStep, reduce_losses
    with tempfile.TemporaryDirectory() as d:
        lock, out = os.path.join(d, "distributed_lock"), os.path.join(d, "out.pt")
        mp.spawn(_worker, args=(2, lock, out, reducer), nprocs=2, join=True)
        got = torch.load(out)
    model, data = _model_and_data()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    step = TrainStep(model, opt, amp_dtype=None)
    losses = step(data)
    ref = reduce_losses(losses)
    assert abs(got["red"]["total"] - ref["total"]) < 1e-4 * max(1.0, abs(ref["total"]))
    moved = 0
    init = _model_and_data()[0].state_dict()
    for k, v in model.state_dict().items():
        a, b = got["params"][k], v.detach().cpu()
        sc = max((b - init[k].cpu()).abs().max().item(), 1e-8) if b.dtype.is_floating_point else 1.0
        if b.dtype.is_floating_point:
            torch.testing.assert_close((a - init[k].cpu()) / sc, (b - init[k].cpu()) / sc, rtol=2e-3, atol=2e-3,
                                       msg=lambda m: f"{k}: {m}")
            moved += int((b - init[k].cpu()).abs().max() > 0)
    assert moved > 40


def _sg_model_and_data():
    from glue_factory_amd.matchers.superglue import SuperGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    torch.manual_seed(7)
    model = SuperGlue({"GNN_layers": ["self", "cross"], "num_sinkhorn_iterations": 5}).cuda().train()
    data = to_device(make_pairs(B, 128, dim=256, size=(640, 480), seed=8), "cuda")
    return model, data


def _sg_worker(rank, world, lock, out):
    from glue_factory_amd.train_step import TrainStep, init_distributed, shard_batch
    torch.cuda.set_device(0)
    init_distributed("gloo", init_method="file://" + lock, rank=rank, world_size=world)
    model, data = _sg_model_and_data()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    step = TrainStep(model, opt, amp_dtype=None, device_ids=[0])
    assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in step.model.modules())
    step(shard_batch(data, rank, world))
    torch.cuda.synchronize()
    if rank == 0:
        sd = {k: v.detach().cpu() for k, v in step.model.state_dict().items()}
        sd["__collectives__"] = dict(step.last_collectives)
        sd["__bn_modules__"] = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in step.model.modules())
        torch.save(sd, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _check_syncbn_run(got, model, data):
    """`got`: rank 0's state after ONE data-parallel step of two ranks; against a single-process step on the whole batch --
    parameters, BatchNorm running statistics (incl. the second update under the reference's activation checkpointing) and
    num_batches_tracked -- plus the collective budget: ONE exchange per BatchNorm module and direction for both views."""
    from glue_factory_amd.train_step import TrainStep
    coll, n_bn = got.pop("__collectives__"), got.pop("__bn_modules__")
    assert coll["syncbn"] == 2 * n_bn, (coll, n_bn)            # (the per-call form issued 2 sets x 2 directions per module)
    assert coll["gradient_buckets"] is not None and coll["gradient_buckets"] <= 4
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    step = TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.05), amp_dtype=None)
    step(data)
    checked = buffers = 0
    for k, v in model.state_dict().items():
        if not v.dtype.is_floating_point:
            assert int(got[k]) == int(v), (k, int(got[k]), int(v))         # num_batches_tracked (2 sets, + 2 replayed)
            continue
        b, a = v.detach().cpu(), got[k]
        sc = (b - init[k]).abs().max().item()
        if sc < 1e-6:      # e.g. conv biases in front of a train-mode BatchNorm: the true gradient is exactly 0
            continue
        err = ((a - b) / sc).abs()
        # the shards sum their fp32 partials in another order than the whole batch; isolated ReLU-boundary flips allowed
        assert err.max() < 5e-2 and (err > 5e-3).float().mean() < 1e-2, (k, err.max().item())
        checked += 1
        buffers += int("running_" in k)
    return checked, buffers


def _gs_model_and_data():
    from glue_factory_amd.matchers.gluestick import GlueStick
    from glue_factory_amd.synthetic import make_point_line_pairs, to_device
    torch.manual_seed(9)
    model = GlueStick({"GNN_layers": ["self", "cross"], "checkpointed": True}).cuda().train()
    data = to_device(make_point_line_pairs(B, 96, 16, dim=256, size=(640, 480), seed=10), "cuda")
    return model, data


def _gs_worker(rank, world, lock, out):
    from glue_factory_amd.train_step import TrainStep, init_distributed, shard_batch
    torch.cuda.set_device(0)
    init_distributed("gloo", init_method="file://" + lock, rank=rank, world_size=world)
    model, data = _gs_model_and_data()
    step = TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.05), amp_dtype=None, device_ids=[0])
    step(shard_batch(data, rank, world))
    torch.cuda.synchronize()
    if rank == 0:
        sd = {k: v.detach().cpu() for k, v in step.model.state_dict().items()}
        sd["__collectives__"] = dict(step.last_collectives)
        sd["__bn_modules__"] = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in step.model.modules())
        torch.save(sd, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gluestick_syncbn_equals_single_process():
    """BASELINE configs[4]'s path (GlueStick, DDP + SyncBatchNorm, train.py:338) on two ranks, with `checkpointed` so that the
    running statistics also take the REPLAYED update (gluestick.py:724-757) from the global counts: parameters, buffers and
    num_batches_tracked equal the single-process step; collectives per step <= 2 x BatchNorm modules + gradient buckets."""
    with tempfile.TemporaryDirectory() as d:
        lock, out = os.path.join(d, "distributed_lock"), os.path.join(d, "out.pt")
        mp.spawn(_gs_worker, args=(2, lock, out), nprocs=2, join=True)
        got = torch.load(out)
    model, data = _gs_model_and_data()
    checked, buffers = _check_syncbn_run(got, model, data)
    assert checked > 30 and buffers >= 10


def test_two_rank_superglue_syncbn_equals_single_process():
    """BatchNorm statistics must be those of the GLOBAL batch (train.py:338 SyncBatchNorm): the fused
    BatchNorm+ReLU op all-reduces its sums across ranks."""
    from glue_factory_amd.train_step import TrainStep
    with tempfile.TemporaryDirectory() as d:
        lock, out = os.path.join(d, "distributed_lock"), os.path.join(d, "out.pt")
        mp.spawn(_sg_worker, args=(2, lock, out), nprocs=2, join=True)
        got = torch.load(out)
    model, data = _sg_model_and_data()
    checked, buffers = _check_syncbn_run(got, model, data)
    assert checked > 20 and buffers >= 10


def _uneven_worker(rank, world, lock, out):
    """Ranks with DIFFERENT row counts (3 pairs / 1 pair): the packed SyncBatchNorm exchange reduces the counts with the sums."""
    from glue_factory_amd import ops
    from glue_factory_amd.train_step import init_distributed
    torch.cuda.set_device(0)
    init_distributed("gloo", init_method="file://" + lock, rank=rank, world_size=world)
    torch.manual_seed(13)
    bn = torch.nn.SyncBatchNorm(256).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    g = torch.Generator(device="cuda").manual_seed(14)
    x = torch.randn(2, 4 * 96, 256, device="cuda", generator=g) * 2 + 0.5        # [views, 4 pairs x 96 rows, C]
    dy = torch.randn(2, 4 * 96, 256, device="cuda", generator=g)
    lo, hi = (0, 3 * 96) if rank == 0 else (3 * 96, 4 * 96)
    xs = x[:, lo:hi].contiguous().requires_grad_(True)
    y = ops.batch_norm_act_sets(xs, bn, relu=True, replay=True)
    y.backward(dy[:, lo:hi].contiguous())
    torch.cuda.synchronize()
    torch.save({"y": y.detach().cpu(), "dx": xs.grad.cpu(), "rm": bn.running_mean.cpu(), "rv": bn.running_var.cpu(),
                "nbt": int(bn.num_batches_tracked), "dg": bn.weight.grad.cpu(), "db": bn.bias.grad.cpu()}, out + str(rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_syncbn_exchange_with_unequal_row_counts_per_rank():
    """Statistics, running buffers (incl. the replayed update) and the input gradient of a two-rank call with 288 / 96 rows per
    view equal torch's BatchNorm1d on the 384 rows in one process; dgamma / dbeta are the LOCAL sums (the reducer averages them)."""
    with tempfile.TemporaryDirectory() as d:
        lock, out = os.path.join(d, "distributed_lock"), os.path.join(d, "out.pt")
        mp.spawn(_uneven_worker, args=(2, lock, out), nprocs=2, join=True)
        got = [torch.load(out + str(r)) for r in range(2)]
    torch.manual_seed(13)
    ref = torch.nn.BatchNorm1d(256).cuda().train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
    g = torch.Generator(device="cuda").manual_seed(14)
    x = (torch.randn(2, 4 * 96, 256, device="cuda", generator=g) * 2 + 0.5).requires_grad_(True)
    dy = torch.randn(2, 4 * 96, 256, device="cuda", generator=g)
    ys = []
    for h in range(2):                     # one module call per view, as the reference does -- twice (checkpoint recompute)
        ys.append(torch.relu(ref(x[h])))
    torch.stack(ys).backward(dy)
    with torch.no_grad():
        for h in range(2):
            ref(x[h].detach())
    y_ref, dx_ref = torch.stack(ys).detach().cpu(), x.grad.cpu()
    y = torch.cat([got[0]["y"], got[1]["y"]], 1)
    dx = torch.cat([got[0]["dx"], got[1]["dx"]], 1)
    torch.testing.assert_close(y, y_ref, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(dx, dx_ref, rtol=1e-4, atol=2e-5)
    for r in range(2):
        torch.testing.assert_close(got[r]["rm"], ref.running_mean.cpu(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got[r]["rv"], ref.running_var.cpu(), rtol=1e-5, atol=1e-6)
        assert got[r]["nbt"] == int(ref.num_batches_tracked) == 4
    torch.testing.assert_close(got[0]["dg"] + got[1]["dg"], ref.weight.grad.cpu(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[0]["db"] + got[1]["db"], ref.bias.grad.cpu(), rtol=1e-4, atol=1e-4)
