"""Data parallelism over one 8xMI355X node: one process per GPU, weights resident, per-GPU BatchNorm
statistics, and ONE all-reduce of the flat fp32 gradient arena over RCCL/xGMI per step.

Replaces the reference's single-process nn.DataParallel (scripts/train_multiframe_detection.py:107-110),
which re-broadcasts all 931 parameters + 927 buffers every forward and reduces gradients to GPU 0.
BatchNorm stays per replica exactly as there (SyncBatchNorm without a process group is plain BN, hrnet.py:31).
The arena is reduced in `bucket_mb` slices so RCCL can pipeline them over the 7 xGMI links of each GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


import os

LOSS_SUMS = 26      # acc[0:26] of mfc_loss_partial are the batch sums; acc[26:29] are derived by mfc_loss_finalize


def _active(group=None):
    """collectives are issued when a process group exists and has more than one rank -- or, for rehearsing the RCCL code path on a
    one-GPU box, also with a single rank when MFC_DIST_FORCE=1 (bench.py --force-dist)"""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("MFC_DIST_FORCE") == "1")


def allreduce_loss_sums(acc: torch.Tensor, group=None):
    """SUM the loss partial sums (weighted NLL numerator / denominator, per-class I, sum p, sum [t==c]) over the ranks:
    the reference computes both loss terms on the gathered batch (src/engine.py:64-66, src/loss.py:57-58)."""
    if _active(group):
        dist.all_reduce(acc[:LOSS_SUMS], op=dist.ReduceOp.SUM, group=group)
    return acc


def _core(model):
    """the model behind a DataParallel-style wrapper (`.module`): the flat arenas are attributes of the model itself"""
    return getattr(model, "module", model)


def allreduce_grads(model, world_size=None, bucket_mb: int = 64, group=None, average=True):
    """Reduce `model._G` (every parameter's .grad is a view of it) across ranks, in place.
    average=True: mean over ranks (per-rank losses).  average=False: sum (loss evaluated over the global batch with
    mfc_loss(global_batch=True), whose logit gradients already carry the global normalisers)."""
    if not dist.is_initialized():
        return
    model = _core(model)
    world_size = world_size or dist.get_world_size(group)
    if world_size == 1 and not _active(group):
        return
    g = model._G
    n = g.numel()
    step = max(1, bucket_mb * (1 << 20) // 4)
    avg = average and dist.get_backend(group) == "nccl"
    works = []
    for a in range(0, n, step):
        sl = g[a:min(n, a + step)]
        works.append(dist.all_reduce(sl, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if average and not avg:
        g.div_(world_size)


def broadcast_params(model, src=0, group=None):
    """Make every rank start from rank `src`'s weights and BatchNorm buffers (one message per arena)."""
    if not dist.is_initialized():
        return
    model = _core(model)
    for t in (model._P, model._RS, model._NBT):
        dist.broadcast(t, src=src, group=group)


def shard_bounds(n: int, world: int, align: int = 4):
    """[lo, hi) of every rank's shard of a flat arena of n elements: equal shards of ceil(n / world) rounded up to `align` elements
    (the Adam kernel works on float4), the last one shorter (possibly empty)."""
    per = -(-n // world)
    per = -(-per // align) * align
    return [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]


class ShardedStep:
    """Alternative to the all-reduce exchange ("prefer reduce-scatter + all-gather that drives all 7 links at once"):
        reduce-scatter   every rank receives the SUM of ONE shard of the flat gradient arena (1/world of the wire bytes of an all-reduce
                         on the critical path before the optimizer),
        sharded Adam     every rank updates only its own shard of the parameter arena (the 28 B/parameter optimizer sweep is no longer
                         replicated: 1.8 GB per step for W48 becomes 1.8 / world),
        all-gather       the updated parameter shards go back to every rank.
    Same arithmetic as all-reduce + replicated Adam: the sum of the ranks' gradients (two ranks: the same bits; more ranks: RCCL's
    reduction order of the collective), the same element-wise update.  Moments stay full-size arenas of which a rank touches its shard only.

        sh = ShardedStep(model)                      # after broadcast_params
        loss.backward(); sh.step(optimizer)          # optimizer: FlatAdam (step(shard=...)) or any callable update(lo, hi)
    """

    def __init__(self, model, group=None, average=False):
        self.model, self.group, self.average = _core(model), group, average
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bounds = shard_bounds(self.model._G.numel(), self.world)
        self.lo, self.hi = self.bounds[self.rank]
        self._equal = len({hi - lo for lo, hi in self.bounds}) == 1

    def _native(self):
        return dist.is_initialized() and dist.get_backend(self.group) == "nccl" and self._equal

    def reduce_scatter(self):
        """after this, model._G[lo:hi] holds the reduced gradients of this rank's shard (the rest of the arena is stale)"""
        if self.world == 1 or not dist.is_initialized():
            return
        g = self.model._G
        if self._native():
            dist.reduce_scatter_tensor(g[self.lo:self.hi], g, op=dist.ReduceOp.SUM, group=self.group)
        else:            # gloo has no reduce-scatter, and uneven shards have no tensor form: one reduce per shard, to its owner
            works = [dist.reduce(g[lo:hi], dst=dist.get_global_rank(self.group, r) if self.group is not None else r, op=dist.ReduceOp.SUM,
                                 group=self.group, async_op=True) for r, (lo, hi) in enumerate(self.bounds) if hi > lo]
            for w in works:
                w.wait()
        if self.average:
            g[self.lo:self.hi].div_(self.world)

    def all_gather(self):
        """every rank's updated parameter shard -> every rank's arena"""
        if self.world == 1 or not dist.is_initialized():
            return
        p = self.model._P
        if self._native():
            dist.all_gather_into_tensor(p, p[self.lo:self.hi].clone(), group=self.group)
        else:
            works = [dist.broadcast(p[lo:hi], src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group,
                                    async_op=True) for r, (lo, hi) in enumerate(self.bounds) if hi > lo]
            for w in works:
                w.wait()

    def step(self, optimizer, **kw):
        """reduce-scatter -> update of the own shard -> all-gather.  `optimizer`: FlatAdam (its step takes shard=(lo, hi)) or a callable
        update(lo, hi) that updates model._P[lo:hi] from model._G[lo:hi]."""
        self.reduce_scatter()
        if self.hi > self.lo:
            if hasattr(optimizer, "step"):
                optimizer.step(shard=(self.lo, self.hi), **kw)
            else:
                optimizer(self.lo, self.hi)
        elif hasattr(optimizer, "step_count"):
            optimizer.step_count += 1          # (an empty shard still counts the step: bias correction stays in lock-step)
        self.all_gather()


class DataParallel(torch.nn.Module):
    """Drop-in for the reference's `model = nn.DataParallel(model)` (scripts/train_multiframe_detection.py:107-110,
    infer_multiframe_endovis15.py:330-333) in a one-process-per-GPU launch: exposes `.module` (the scripts reach `model.module.base_model`,
    `model.module.multiframe_net`, and `save_model` strips the `module.` prefix, utils/model_utils.py:6-12), forwards calls unchanged, and --
    when a process group with more than one rank exists -- starts every rank from rank 0's weights and overlaps the RCCL all-reduce of the
    gradient buckets with the backward pass (GradBucketReducer).  `finish()` (or `mfcnet_amd.train_step`) completes the exchange before
    the optimizer step.  `device_ids` / `output_device` / `dim` are accepted and ignored: the process owns one device."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0, group=None, average=False, broadcast=True):
        super().__init__()
        self.module = module
        self.group = group
        self.reducer = None
        if _active(group):
            if broadcast:
                broadcast_params(module, 0, group)
            self.reducer = GradBucketReducer(module, average=average, group=group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def finish(self):
        """wait for the bucket all-reduces the last backward started (no-op with one rank)"""
        return self.reducer.finish() if self.reducer is not None else []


class GradBucketReducer:
    """Overlap the gradient all-reduce with the backward pass: the backward program is cut into segments that
    finalise the flat gradient arena bucket by bucket (plan.py::_build_grad_buckets); after each segment has been enqueued,
    this hook starts the (asynchronous) RCCL all-reduce of that bucket, which then runs next to the remaining backward
    kernels.  `finish()` waits for all of them (call it before the optimizer step).

        red = GradBucketReducer(model, average=False)      # installs model.grad_bucket_hook
        loss.backward(); red.finish(); opt.step()
    """

    def __init__(self, model, average=True, group=None, wire_dtype=None):
        """wire_dtype = torch.bfloat16 / torch.float16: the buckets travel in 16 bits (half the xGMI bytes: 82 MB instead of 164 MB per step
        for W32) and are accumulated back into the fp32 arena on arrival -- every rank rounds its own contribution once, the sum itself is
        RCCL's 16-bit ring sum, so this trades gradient precision for link time; off (fp32 buckets) by default."""
        model = _core(model)
        self.model, self.average, self.group = model, average, group
        self.wire_dtype = wire_dtype
        self.works, self.ranges = [], []
        self._comm = None
        model.grad_bucket_hook = self._on_bucket
        model._bucket_reducer = self            # engine.train_step finishes these all-reduces instead of reducing the arena again

    def _on_bucket(self, lo, hi):
        self.ranges.append((lo, hi))
        if not _active(self.group):
            return
        sl = self.model._G[lo:hi]
        avg = self.average and dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if sl.is_cuda:
            # the bucket's weight gradients are summed by an unpack launch on the interpreter's detached stream, which the backward
            # chain does not wait for: order a side stream after (1) the chain so far and (2) that detached work, and issue the
            # collective from it (the process group orders its own stream after the stream the call is made on)
            from . import _lib as L
            import ctypes as C
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            self._comm.wait_stream(torch.cuda.current_stream())
            ctx = getattr(self.model, "_ctx", None)          # the model's interpreter context (its detached stream)
            if ctx is not None:
                L.check(L.lib.mfc_wait_detached_ctx(ctx.handle, C.c_void_p(self._comm.cuda_stream)), "mfc_wait_detached_ctx")
            else:
                L.check(L.lib.mfc_wait_detached(C.c_void_p(self._comm.cuda_stream)), "mfc_wait_detached")
            with torch.cuda.stream(self._comm):
                if self.wire_dtype is not None:
                    wire = sl.to(self.wire_dtype)
                    work = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
                    self.works.append((work, sl, self.average and not avg, wire))
                    return
                work = dist.all_reduce(sl, op=op, group=self.group, async_op=True)
        else:
            if self.wire_dtype is not None:
                wire = sl.to(self.wire_dtype)
                work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.works.append((work, sl, self.average, wire))
                return
            work = dist.all_reduce(sl, op=op, group=self.group, async_op=True)
        self.works.append((work, sl, self.average and not avg, None))

    def finish(self):
        ws = dist.get_world_size(self.group) if dist.is_initialized() else 1
        for w, sl, div, wire in self.works:
            w.wait()
            if wire is not None:
                sl.copy_(wire)                      # (fp32 arena <- the 16-bit sum)
            if div:
                sl.div_(ws)
        self.works = []
        r, self.ranges = self.ranges, []
        return r

    def remove(self):
        self.finish()
        self.model.grad_bucket_hook = None
        self.model._bucket_reducer = None

