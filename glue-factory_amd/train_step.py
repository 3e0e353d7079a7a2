"""Data-parallel train step around a BaseModel-style module (one process per GPU).

Mirrors the semantics of the reference hot loop (gluefactory/train.py:465-517): zero_grad,
autocast forward, ``loss_fn(pred, data)``, mean of ``losses["total"]``, cross-rank agreement on
whether the loss is differentiable (all_reduce PRODUCT, train.py:482-488), backward (bucketed gradient all-reduce
over RCCL/xGMI overlapped with the backward), optional gradient clipping, optimizer step (optionally the whole step
as ONE hipGraph replay: ``graph=True``); a NaN / non-finite loss (train.py:477-480) or gradient norm skips the
update.  On the GPU path the skip decision never touches the host: the cross-rank flag is a device value handed to
the fused optimiser as ``found_inf``.  Image pairs are independent, so the batch shards across ranks with no
data-path collective; the only collectives are the gradient buckets (the skip flag rides in the last one) and the
logging reduce (train.py:525-531).

Gradient reduction (``reducer="buckets"``, the default for world size > 1): ``GradBuckets`` below -- plain
``dist.all_reduce`` calls on slices of ONE flat fp32 buffer, issued from autograd's post-accumulate hooks the moment a
bucket's last gradient exists (buckets follow the backward: assignment / confidence heads first, so their reduce
runs under the transformer backward), on the process group's own stream.  Unlike DistributedDataParallel's C++ reducer
this is a fixed sequence of kernels and collectives with no host decisions, i.e. it can be CAPTURED: with the
``nccl`` (= RCCL) backend ``graph=True`` replays the multi-rank step -- collectives included -- as one hipGraph.
``reducer="ddp"`` keeps the stock wrapper (never captured) for comparison; ``reducer="buckets_bound"`` pre-binds ``p.grad`` to the
flat buffer (no per-bucket copy; see GradBuckets).

The same code runs on CPU with the ``gloo`` backend for tests (any module with the
forward/loss interface), and on GPUs with ``nccl`` (= RCCL on ROCm).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, init_method=None, rank=None, world_size=None):
    """Join the process group.  Defaults come from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT); ``init_method='file://...'`` reproduces the reference's
    single-node file rendezvous (train.py:276-281)."""
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world_size == 1 and os.environ.get("GF_FORCE_DIST") != "1":      # GF_FORCE_DIST=1: a process group of ONE rank (smoke runs
        return 0, 1, local                                              # of the RCCL path on a single-GPU box)
    if backend is None:         # GF_DIST_BACKEND=gloo: several ranks on ONE GPU (smoke runs of the multi-rank path; RCCL refuses that)
        backend = os.environ.get("GF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if torch.cuda.is_available():           # every launcher enqueues on the CURRENT device's stream (ops._chk)
        torch.cuda.set_device(local % torch.cuda.device_count())
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local)
    if init_method is not None:
        dist.init_process_group(backend, init_method=init_method, rank=rank, world_size=world_size, **kw)
    else:
        dist.init_process_group(backend, **kw)
    return rank, world_size, local


def _first_batch_dim(data):
    if isinstance(data, dict):
        for v in data.values():
            b = _first_batch_dim(v)
            if b is not None:
                return b
    elif torch.is_tensor(data) and data.dim() > 0:
        return data.shape[0]
    return None


def shard_batch(data, rank, world_size, batch_size=None):
    """Slice the batched tensors of a (nested) batch dict: global batch -> this rank's pairs (what
    DistributedSampler + batch_size // n_gpus do in train.py:285-288).  Only tensors whose leading dimension IS
    the global batch size are sliced (``batch_size``; default: the leading dimension of the first tensor found);
    anything else (tables, per-dataset constants) is passed through whole."""
    if batch_size is None:
        batch_size = _first_batch_dim(data)
    if isinstance(data, dict):
        return {k: shard_batch(v, rank, world_size, batch_size) for k, v in data.items()}
    if torch.is_tensor(data) and data.dim() > 0 and data.shape[0] == batch_size:
        assert batch_size % world_size == 0, f"global batch {batch_size} not divisible by world size {world_size}"
        per = batch_size // world_size
        return data[rank * per:(rank + 1) * per]
    return data


class GradBuckets:
    """Bucketed data-parallel gradient averaging without DistributedDataParallel.

    * one flat fp32 buffer holds every gradient (+ ``extra`` scalar slots at its end: the step's skip flag);
    * parameters are assigned to buckets of <= ``cap_mb`` in REVERSE registration order (~ the order the backward
      produces them: heads and confidence classifiers first, the first transformer layer and the input projection
      last);
    * a post-accumulate-grad hook per parameter counts arrivals; when a bucket is complete its gradients are gathered
      into the flat slice by one multi-tensor copy and ``dist.all_reduce(slice, async_op=True)`` is issued -- always in
      bucket order, so every rank issues the same sequence of collectives whatever order its autograd engine ran in;
    * ``finish()`` issues whatever is left (parameters that received no gradient contribute zeros), waits for the
      collectives (a stream dependency on NCCL/RCCL, not a host sync) and re-points ``p.grad`` at the flat views.
    The loss is pre-divided by the world size, so SUM yields DDP's average.

    ``bind_grads`` (reducer="buckets_bound"): ``p.grad`` IS the flat view from ``start()`` on -- autograd accumulates into
    the (zeroed) buffer in place and the per-bucket copy disappears.  Same numbers (tests/test_ddp_gloo.py,
    tests/test_gpu_ddp.py run both).  NOT the default: the copy variant moves 2 x 47 MB per LightGlue step (one read of the
    freshly produced gradients, one write of the flat buffer, ~25 us at 4 TB/s), the bound variant 4 x (zero fill + read
    view + read gradient + write view), because autograd's AccumulateGrad adds into a defined ``.grad``."""

    def __init__(self, params, cap_mb=16, extra=1, group=None, bind_grads=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.bind_grads = bool(bind_grads)
        dev = self.params[0].device
        order = list(reversed(self.params))
        total = sum(p.numel() for p in order)
        self.flat = torch.zeros(total + extra, dtype=torch.float32, device=dev)
        self.extra = self.flat[total:]
        cap = max(1, int(cap_mb * (1 << 20)) // 4)
        self.buckets, self.views, self.bucket_of = [], {}, {}
        lo = off = 0
        cur = []
        for p in order:
            if cur and off - lo + p.numel() > cap:
                self.buckets.append((lo, off, cur))
                lo, cur = off, []
            self.views[p] = self.flat[off:off + p.numel()].view_as(p)
            self.bucket_of[p] = len(self.buckets)
            cur.append(p)
            off += p.numel()
        self.buckets.append((lo, off + extra, cur))           # the last bucket carries the extra slots
        self._hooks = [p.register_post_accumulate_grad_hook(self._arrived) for p in self.params]
        self._count, self._next, self._work = [0] * len(self.buckets), 0, []
        self._active = False

    def start(self):
        self._count, self._next, self._work = [0] * len(self.buckets), 0, []
        self._active = True
        if self.bind_grads:
            self.flat[:self.flat.numel() - self.extra.numel()].zero_()
            for p in self.params:
                p.grad = self.views[p]

    def close(self):
        """Detach from the parameters (hooks removed): a second reducer may now own them."""
        for h in self._hooks:
            h.remove()
        self._hooks, self._active = [], False

    def _arrived(self, p):
        if not self._active:
            return
        b = self.bucket_of[p]
        self._count[b] += 1
        while self._next < len(self.buckets) and self._count[self._next] == len(self.buckets[self._next][2]):
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        lo, hi, ps = self.buckets[b]
        have = [p for p in ps if p.grad is not None]
        if len(have) != len(ps) and not self.bind_grads:
            self.flat[lo:hi - (self.extra.numel() if b == len(self.buckets) - 1 else 0)].zero_()
        have = [p for p in have if p.grad.data_ptr() != self.views[p].data_ptr()]      # (bound gradients are already in place)
        if have:
            torch._foreach_copy_([self.views[p] for p in have], [p.grad for p in have])
        self._work.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        while self._next < len(self.buckets):      # parameters without a gradient this step: reduced as zeros
            self._launch(self._next)
            self._next += 1
        for w in self._work:
            w.wait()
        for p in self.params:
            p.grad = self.views[p]
        self._active = False


class TrainStep:
    """step(data) -> dict of detached per-sample losses; one optimiser update per call."""

    def __init__(self, model, optimizer, amp_dtype=None, clip_grad=None, device_ids=None,
                 bucket_cap_mb=16, find_unused_parameters=False, check_grads=True, graph=False, graph_warmup=2,
                 reducer="buckets", force_distributed=False, deterministic_replay=False):
        """deterministic_replay: wait for every hipGraph replay before returning (the host then does nothing else while a
        replay runs).  An option for loops that want that guarantee; it costs the overlap of host work with the step, the
        benchmark and training default leave it off.  Background (profiles/r06_graph_replay_determinism.txt): the run-to-run
        noise that prompted it turned out to be an asynchronous copy from PAGEABLE host memory racing with the loop that refilled
        the batch (_async_ok below: asynchronous only from pinned memory); on the final tree replayed SuperGlue / GlueStick runs
        are bit-reproducible over 300 steps under host-to-device copies, device copies, unrelated kernels, held CUs and dirtied
        LDS alike, with and without this flag.
        force_distributed: take the multi-rank path (SyncBatchNorm conversion, gradient reducer, collectives) in a process
        group of ONE rank too -- how the RCCL calls of that path are exercised on a single-GPU box (tests/test_gpu_rccl_one_rank.py);
        every collective is then the identity, so the step must equal the plain single-process one."""
        self.model = model
        self.optimizer = optimizer
        self.amp_dtype = amp_dtype
        self.clip_grad = clip_grad
        self.distributed = (dist.is_available() and dist.is_initialized()
                            and (dist.get_world_size() > 1 or bool(force_distributed)))
        self.fwd_model = model
        self.buckets = None
        if force_distributed and self.distributed:
            from . import ops as _ops
            _ops.FORCE_SYNC_BN = True
        if self.distributed:
            # train.py:338: BatchNorm layers (SuperGlue / GlueStick MLPs) use global-batch statistics;
            # ops.batch_norm_act all-reduces its sums when it sees a SyncBatchNorm module.
            if any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules()):
                self.model = model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
            # ~47 MB of fp32 gradients for LightGlue: 16 MB buckets let the all-reduce of the
            # assignment/confidence heads (first gradients of the backward) start while the
            # transformer backward is still running; loss() is called on the bare module (as the
            # reference binds loss_fn before wrapping, train.py:334-339).
            if reducer == "ddp":
                from torch.nn.parallel import DistributedDataParallel as DDP
                self.fwd_model = DDP(model, device_ids=device_ids, bucket_cap_mb=bucket_cap_mb,
                                     gradient_as_bucket_view=True,
                                     find_unused_parameters=find_unused_parameters)
            elif reducer in ("buckets", "buckets_bound"):
                self.buckets = GradBuckets(model.parameters(), cap_mb=bucket_cap_mb, extra=1,
                                           bind_grads=reducer == "buckets_bound")
            else:
                raise ValueError(f"TrainStep: unknown reducer {reducer!r} (buckets, buckets_bound, ddp)")
        p = next(model.parameters())
        self.device_type = p.device.type
        self.device = p.device
        self.check_grads = check_grads
        self._skipped_host = 0
        self._skipped_dev = None
        self.last_collectives = None
        self.deterministic_replay = bool(deterministic_replay)
        # hipGraph replay of the whole step (forward, loss, backward, optimiser): ~1200 launches per LightGlue step
        # leave a few per cent of the GPU idle between kernels when they are issued one by one.  Needs: single
        # process (no DDP hooks inside a capture), a fused + capturable optimiser, no host synchronisation in the
        # step (the fused skip path above), fixed shapes, and only KERNEL nodes in the capture: memset / memcpy nodes
        # were observed to be dropped from some replays on ROCm 7.2 when eager work is queued between replays (see
        # csrc/gf_common.h gf_zero_f32), so the launchers never call hipMemsetAsync / hipMemcpyAsync.  The first ``graph_warmup`` calls run eagerly, the next
        # one is captured and replayed; a batch of different shapes is run eagerly and re-captured.
        # Multi-rank: only with the bucket reducer on NCCL/RCCL (its collectives are stream-ordered and capturable; DDP's
        # reducer and gloo are not).
        capturable_dp = (not self.distributed) or (self.buckets is not None and dist.get_backend() == "nccl")
        self.graph = bool(graph) and self.device_type == "cuda" and capturable_dp
        self.graph_warmup = graph_warmup
        self._calls = 0
        self._g = None            # (shape signature, CUDAGraph, static inputs, static outputs)

    def close(self):
        """Release what the step holds on the model (gradient hooks of the bucket reducer, the captured graph)."""
        if self.buckets is not None:
            self.buckets.close()
        self._g = None

    def _device_skip_supported(self):
        """Fused CUDA optimisers take a device-side ``found_inf`` flag (the GradScaler protocol): the update is
        skipped inside the fused kernel, so no host synchronisation is needed to decide it."""
        opt = self.optimizer
        return (self.device_type == "cuda" and getattr(opt, "_step_supports_amp_scaling", False)
                and all(g.get("fused") for g in opt.param_groups))

    @property
    def skipped(self):
        """Number of steps whose update was skipped (non-differentiable or non-finite loss / gradient norm on any
        rank).  Reading it synchronises; the step itself never does on the fused path."""
        return self._skipped_host + (int(self._skipped_dev.item()) if self._skipped_dev is not None else 0)

    def __call__(self, data):
        """One train step.  ``data`` may live on the model's device or in (pinned) HOST memory, as a DataLoader hands it
        over (the reference moves it first: train.py:462-469 batch_to_device(non_blocking=True)): host tensors are copied to
        the device in stream order -- in replay mode straight into the captured graph's own input buffers, one
        host-to-device copy per tensor and no device-side copy behind it."""
        self._calls += 1
        if not self.graph or self._calls <= self.graph_warmup:
            return self._step(_to_device_tree(data, self.device))
        sig = _signature(data)
        if self._g is None or self._g[0] != sig:
            self._capture(_to_device_tree(data, self.device), sig)
        _, g, static_in, static_out = self._g
        _copy_into(static_in, data)
        sync_lr = getattr(self.optimizer, "sync_lr", None)
        if sync_lr is not None:     # the reference steps its lr scheduler every iteration (train.py:517): the captured Adam
            sync_lr()               # reads a DEVICE scalar, refreshed here (a fill kernel in front of the replay, only on change)
        g.replay()
        if self.deterministic_replay:       # see __init__: the host does nothing else while the replay runs
            torch.cuda.current_stream().synchronize()
        from . import ops as _ops       # parameters changed behind the precast cache's back (no version bump)
        _ops.invalidate_precast()
        return static_out

    def static_inputs(self):
        """The captured graph's own input tensors (None before the capture).  A caller that writes its batch INTO them and
        passes this very tree to ``__call__`` skips the per-step copy of the inputs (268 MB for 2 x 32 images of 1024^2)."""
        return None if self._g is None else self._g[2]

    def _capture(self, data, sig):
        if not (self._device_skip_supported() and all(g_.get("capturable") for g_ in self.optimizer.param_groups)):
            raise RuntimeError("TrainStep(graph=True) needs a fused, capturable optimiser "
                               "(e.g. torch.optim.Adam(..., fused=True, capturable=True))")
        static_in = _clone_tree(data)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # several ranks: RCCL's watchdog thread polls events while this thread captures -- only THIS thread's calls belong to
        # (and may invalidate) the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local" if self.distributed else "global"):
            static_out = self._step(static_in)
        self._g = (sig, g, static_in, static_out)

    def _step(self, data):
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        from . import ops
        bn0 = ops.COLLECTIVES["syncbn"]
        with torch.autocast(self.device_type, dtype=self.amp_dtype or torch.bfloat16,
                            enabled=self.amp_dtype is not None):
            pred = self.fwd_model(data)
            losses, _ = self.model.loss(pred, {**pred, **data})
            loss = torch.mean(losses["total"])
        out = {k: v.detach() for k, v in losses.items() if torch.is_tensor(v)}
        # train.py:477-488: skip the iteration on a NaN loss, and agree across ranks on whether the loss is
        # differentiable (all_reduce PRODUCT of the flag).  Both conditions are folded into ONE device-side
        # "bad" flag (MAX over ranks = the reference's PRODUCT of "good"); every rank always runs its backward
        # (DDP's bucketed all-reduce needs all of them), and the flag gates the parameter update.
        bad = (~torch.isfinite(loss.detach())).float().reshape(())
        if not loss.requires_grad:
            bad = bad + 1.0
            # keep the autograd graph (and the reduction hooks) alive with an exactly-zero contribution
            loss = loss.detach() + sum((p.sum() * 0.0 for p in self.model.parameters() if p.requires_grad))
        # ops.REPLAY_GATE: the BatchNorm replays of the backward (the reference's checkpoint recompute) do nothing on a step the
        # reference would have skipped BEFORE its backward (train.py:477-488).  Bucket reducer: this rank's own flag -- the other
        # ranks' arrive with the last bucket, after the backward; otherwise the flag is all-reduced first, like the reference's.
        try:
            if self.buckets is not None:
                # the flag rides in the last gradient bucket (SUM over ranks > 0 <=> bad on some rank): no collective of its
                # own and nothing on the critical path in front of the backward
                self.buckets.start()
                self.buckets.extra.copy_(bad.reshape(1))
                ops.REPLAY_GATE = bad
                (loss / dist.get_world_size()).backward()
                self.buckets.finish()
                bad = (self.buckets.extra[0] > 0).float()
                n_grad = len(self.buckets.buckets)
            else:
                if self.distributed:
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                ops.REPLAY_GATE = bad
                loss.backward()
                n_grad = None                      # (DistributedDataParallel issues its own bucket reductions + the flag's)
        finally:
            ops.REPLAY_GATE = None
        ops.SharedGradSum.check_all()           # a backward that skipped consumers of a shared tensor raises instead of losing gradient
        # collectives this step issued (eager call or capture): gradient buckets, SyncBatchNorm exchanges (ops.COLLECTIVES)
        self.last_collectives = {"gradient_buckets": n_grad, "syncbn": ops.COLLECTIVES["syncbn"] - bn0}
        if self.clip_grad is not None:
            gn = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad)   # no host sync
            bad = torch.maximum(bad, (~torch.isfinite(gn)).float().reshape(()).to(bad.device))
        elif self.check_grads:
            # the reference's error_if_nonfinite try/except (train.py:498-510) for the un-clipped case
            gsum = torch.stack(torch._foreach_norm([p.grad for p in self.model.parameters() if p.grad is not None])).sum()
            bad = torch.maximum(bad, (~torch.isfinite(gsum)).float().reshape(()))
        if self._device_skip_supported():
            opt = self.optimizer
            opt.found_inf = (bad > 0).float()
            opt.grad_scale = torch.ones((), device=bad.device)
            try:
                opt.step()                     # fused kernel: no-op (and no step count) where found_inf != 0
            finally:
                del opt.found_inf, opt.grad_scale
            if self._skipped_dev is None:
                self._skipped_dev = torch.zeros((), dtype=torch.long, device=bad.device)
            self._skipped_dev.add_((bad > 0).long())      # in place: also accumulates under graph replay
        elif bool(bad.item() > 0):             # CPU / non-fused optimisers: one host read
            self._skipped_host += 1
        else:
            self.optimizer.step()
        return out


def _signature(data):
    if isinstance(data, dict):
        return tuple((k, _signature(v)) for k, v in sorted(data.items()))
    if torch.is_tensor(data):
        return (tuple(data.shape), data.dtype)         # (not the device: a host batch replays the same graph)
    return None


def _to_device_tree(data, device):
    """`data` with every tensor that is not on `device` moved there (stream-ordered; pinned host memory copies
    asynchronously); tensors already there are passed through, not copied."""
    if isinstance(data, dict):
        return {k: _to_device_tree(v, device) for k, v in data.items()}
    if torch.is_tensor(data) and data.device != device:
        return data.to(device, non_blocking=_async_ok(data))
    return data


def _async_ok(src):
    """May a copy FROM `src` be issued without waiting for it?  Device tensors: yes (stream-ordered).  Host tensors: only
    PINNED ones -- torch's pinned allocator holds a pinned block until the copies recorded on it have run, whereas an
    asynchronous copy from PAGEABLE memory on ROCm reads the source after the call returned: a caller that frees or refills
    the batch afterwards (any Python loop that builds the next batch) corrupts the one in flight.  Found in round 6 as
    run-to-run noise of the learning-curve test (tests/test_gpu_zz_learning.py fed pageable batches with non_blocking=True)."""
    return src.device.type != "cpu" or src.is_pinned()


def _clone_tree(data):
    if isinstance(data, dict):
        return {k: _clone_tree(v) for k, v in data.items()}
    if torch.is_tensor(data):
        return data.detach().clone()
    return data


def _copy_into(static, data):
    for k, v in data.items():
        if isinstance(v, dict):
            _copy_into(static[k], v)
        elif torch.is_tensor(v) and static[k] is not v:
            static[k].copy_(v, non_blocking=_async_ok(v))


def reduce_losses(losses, dst=0):
    """Mean over samples, then mean over ranks on ``dst`` (logging path, train.py:525-531)."""
    out = {}
    for k in sorted(losses):
        v = losses[k].float().mean()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            v = v.clone()
            dist.reduce(v, dst=dst, op=dist.ReduceOp.SUM)
            v = v / dist.get_world_size()
        out[k] = float(v.item())
    return out
