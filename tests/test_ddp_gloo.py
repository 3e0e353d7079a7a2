"""N>1 path on CPU: 2 gloo processes with the reference's file:// rendezvous run the shipped
TrainStep on a stand-in module with the BaseModel forward/loss interface (the HIP matchers
need a GPU); the averaged gradients / updated weights must equal a single process run on the
concatenated batch, the do_backward flag must be agreed across ranks, losses reduce to rank 0."""
import os
import tempfile

import torch
import torch.multiprocessing as mp

from glue_factory_amd.train_step import TrainStep, init_distributed, reduce_losses, shard_batch


class Toy(torch.nn.Module):
    """Pairs are independent; per-sample 'total' loss, like the matchers."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(6, 8)
        self.head = torch.nn.Linear(8, 1)

    def forward(self, data):
        return {"y": torch.tanh(self.a(data["x"]))}

    def loss(self, pred, data):
        total = ((self.head(pred["y"]).squeeze(-1) - data["t"]) ** 2).mean(-1)
        if data.get("freeze", False):
            total = total.detach()
        return {"total": total, "aux": total.detach() * 2}, {}


def _batch(n=8):
    g = torch.Generator().manual_seed(1)
    return {"x": torch.randn(n, 5, 6, generator=g), "t": torch.randn(n, 5, generator=g),
            "view0": {"image_size": torch.ones(n, 2)}}


def _worker(rank, world, lock, out, reducer):
    torch.set_num_threads(1)
    init_distributed("gloo", init_method="file://" + lock, rank=rank, world_size=world)
    model = Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    # tiny buckets: the two linears land in different buckets, the skip flag rides in the last one
    step = TrainStep(model, opt, reducer=reducer, bucket_cap_mb=1e-4)
    assert step.distributed and (step.buckets is not None) == reducer.startswith("buckets")
    if reducer.startswith("buckets"):
        assert len(step.buckets.buckets) >= 2
    data = shard_batch(_batch(), rank, world)
    assert data["x"].shape[0] == 4 and data["view0"]["image_size"].shape[0] == 4
    losses = step(data)
    red = reduce_losses(losses)
    # one rank without a differentiable loss -> every rank skips the update (train.py:482-488)
    before = [p.detach().clone() for p in model.parameters()]
    step(dict(data, freeze=(rank == 1)))
    assert step.skipped == 1
    assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    if rank == 0:
        torch.save({"params": [p.detach() for p in model.parameters()], "red": red}, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("reducer", ["buckets", "buckets_bound", "ddp"])
def test_two_rank_gloo_equals_single_process(reducer):
    """Both gradient reducers -- the capturable bucket reducer (default) and stock DistributedDataParallel -- give the
    single-process result on the concatenated batch."""
    with tempfile.TemporaryDirectory() as d:
        lock, out = os.path.join(d, "distributed_lock"), os.path.join(d, "out.pt")
        mp.spawn(_worker, args=(2, lock, out, reducer), nprocs=2, join=True)
        got = torch.load(out)
    model = Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    step = TrainStep(model, opt)
    assert not step.distributed
    losses = step(_batch())
    for a, b in zip(got["params"], model.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-6, atol=1e-7)
    ref = reduce_losses(losses)
    assert abs(got["red"]["total"] - ref["total"]) < 1e-6 and abs(got["red"]["aux"] - ref["aux"]) < 1e-6


def test_nan_loss_skips_the_update_and_shard_batch_keeps_tables():
    """train.py:477-480: a NaN loss must not reach the weights; non-batched tensors are not sliced."""
    model = Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    step = TrainStep(model, opt)
    data = _batch()
    before = [p.detach().clone() for p in model.parameters()]
    bad = dict(data, t=data["t"].clone())
    bad["t"][3, 2] = float("nan")
    out = step(bad)
    assert torch.isnan(out["total"]).any() and step.skipped == 1
    assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    step(data)
    assert step.skipped == 1 and not all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
    d = dict(data, table=torch.arange(5.0), square=torch.ones(3, 3))
    sh = shard_batch(d, 1, 2)
    assert sh["x"].shape[0] == 4 and sh["table"].shape == (5,) and sh["square"].shape == (3, 3)
