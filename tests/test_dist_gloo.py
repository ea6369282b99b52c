"""N > 1 path on CPU: two gloo ranks average the flat gradient arena (one bucketed all-reduce) and start from
rank 0's weights -- the RCCL path of bench.py / dist.py with the backend swapped (world_size 2)."""
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mfcnet_amd.dist import allreduce_grads, broadcast_params
    n = 1_000_003
    model = SimpleNamespace(_G=torch.full((n,), float(rank + 1)), _P=torch.full((n,), float(rank)),
                            _RS=torch.full((7,), float(rank)), _NBT=torch.full((3,), rank, dtype=torch.int64))
    model._G[:5] = torch.arange(5.0) * (rank + 1)
    broadcast_params(model, src=0)
    allreduce_grads(model, world, bucket_mb=1)                 # several buckets
    ok = bool((model._P == 0).all() and (model._RS == 0).all() and (model._NBT == 0).all())
    ok = ok and torch.allclose(model._G[5:], torch.full((n - 5,), 1.5)) and torch.allclose(model._G[:5], torch.arange(5.0) * 1.5)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
