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


def allreduce_grads(model, world_size=None, bucket_mb: int = 64, group=None):
    """Average `model._G` (every parameter's .grad is a view of it) across ranks, in place."""
    if not dist.is_initialized():
        return
    world_size = world_size or dist.get_world_size(group)
    if world_size == 1:
        return
    g = model._G
    n = g.numel()
    step = max(1, bucket_mb * (1 << 20) // 4)
    avg = dist.get_backend(group) == "nccl"
    works = []
    for a in range(0, n, step):
        sl = g[a:min(n, a + step)]
        works.append(dist.all_reduce(sl, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if not avg:
        g.div_(world_size)


def broadcast_params(model, src=0, group=None):
    """Make every rank start from rank `src`'s weights and BatchNorm buffers (one message per arena)."""
    if not dist.is_initialized():
        return
    for t in (model._P, model._RS, model._NBT):
        dist.broadcast(t, src=src, group=group)
