"""N > 1 path on CPU: two gloo ranks average the flat gradient arena (one bucketed all-reduce) and start from
rank 0's weights -- the RCCL path of bench.py / dist.py with the backend swapped (world_size 2)."""
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


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
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _loss_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mfcnet_amd.dist import allreduce_grads, allreduce_loss_sums, LOSS_SUMS
    from oracle import mfcnet_oracle as O
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(4, 5, 12, 16, generator=g)
    target = torch.randint(0, 5, (4, 12, 16), generator=g)
    full, parts = O.total_loss(logits, target)                      # the reference's loss on the gathered batch
    lo, hi = rank * 2, rank * 2 + 2                                 # this rank's clips
    acc = O.loss_partial_sums(logits[lo:hi], target[lo:hi]).float()
    acc[LOSS_SUMS:] = -1.0                                          # the derived slots must not be reduced
    allreduce_loss_sums(acc)
    nll, jac, tot = O.loss_from_sums(acc.double())
    ok = abs(float(tot) - float(full)) < 1e-5 and abs(float(nll) - float(parts["loss_nll"])) < 1e-5
    ok = ok and abs(float(jac) - float(parts["loss_soft_jaccard"])) < 1e-5 and bool((acc[LOSS_SUMS:] == -1.0).all())
    # a loss normalised over the global batch wants the ranks' parameter gradients SUMMED
    model = SimpleNamespace(_G=torch.full((1000,), float(rank + 1)))
    allreduce_grads(model, world, average=False)
    ok = ok and torch.allclose(model._G, torch.full((1000,), 3.0))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_global_batch_loss_sums():
    """Splitting a batch over two ranks and all-reducing the 26 loss sums reproduces the loss of the whole batch
    (the reference evaluates NLL and soft-Jaccard after DataParallel's gather, src/engine.py:64-66)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mfcnet_amd.dist import GradBucketReducer
    model = SimpleNamespace(_G=torch.full((1000,), float(rank + 1)))
    red = GradBucketReducer(model, average=False)
    for lo, hi in ((600, 1000), (250, 600), (0, 250)):          # buckets become final from the end of the arena, as in the backward pass
        model.grad_bucket_hook(lo, hi)
    ranges = red.finish()
    ok = ranges == [(600, 1000), (250, 600), (0, 250)] and torch.allclose(model._G, torch.full((1000,), 3.0))
    model._G.fill_(float(rank + 1))
    red2 = GradBucketReducer(model, average=True)
    model.grad_bucket_hook(0, 1000)
    red2.finish()
    ok = ok and torch.allclose(model._G, torch.full((1000,), 1.5))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_overlap_reducer():
    """dist.GradBucketReducer: the per-bucket asynchronous all-reduces started from the backward pass's bucket hook."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _wire_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mfcnet_amd.dist import DataParallel, GradBucketReducer
    n = 4096
    g = torch.Generator().manual_seed(11 + rank)
    model = SimpleNamespace(_G=torch.randn(n, generator=g), _P=torch.full((n,), float(rank)), _RS=torch.zeros(3), _NBT=torch.zeros(1, dtype=torch.int64))
    mine = model._G.clone()
    both = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(both, mine)
    # 16-bit wire format: every rank rounds its contribution to bf16, the sum comes back into the fp32 arena
    red = GradBucketReducer(model, average=False, wire_dtype=torch.bfloat16)
    model.grad_bucket_hook(0, n // 2)
    model.grad_bucket_hook(n // 2, n)
    ranges = red.finish()
    want = sum(b.to(torch.bfloat16).float() for b in both)
    ok = ranges == [(0, n // 2), (n // 2, n)] and torch.allclose(model._G, want, rtol=2e-2, atol=2e-2) and model._G.dtype == torch.float32
    # the nn.DataParallel stand-in: .module, rank 0's weights everywhere, a reducer installed
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._G, self._P, self._RS, self._NBT = torch.zeros(8), torch.full((8,), float(rank + 3)), torch.zeros(2), torch.zeros(1, dtype=torch.int64)
        def forward(self, x, k=1):
            return x * k
    dp = DataParallel(M(), device_ids=[0])
    ok = ok and dp.module._P.eq(3.0).all().item() and dp.reducer is not None and float(dp(torch.ones(1), k=2)) == 2.0
    ok = ok and list(dp.state_dict().keys()) == [] and dp.finish() == []
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sixteen_bit_gradient_buckets_and_dataparallel_stand_in():
    """dist.GradBucketReducer(wire_dtype=bf16) and mfcnet_amd.DataParallel (the replacement of scripts/train_multiframe_detection.py:107-110)
    over two gloo ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mfcnet_amd.dist import ShardedStep, allreduce_grads, shard_bounds
    n = 10_007                                  # not a multiple of world * 4: uneven last shard
    gen = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=gen)
    grads = [torch.randn(n, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]

    def adam_like(P, G, lo, hi):                 # any element-wise update rule: the test is about the exchange
        P[lo:hi] -= 0.01 * G[lo:hi] / (G[lo:hi].abs().sqrt() + 1e-3)

    # path A: all-reduce (SUM) of the whole arena, replicated update
    a = SimpleNamespace(_G=grads[rank].clone(), _P=p0.clone())
    allreduce_grads(a, world, average=False)
    adam_like(a._P, a._G, 0, n)
    # path B: reduce-scatter -> update of the own shard -> all-gather
    b = SimpleNamespace(_G=grads[rank].clone(), _P=p0.clone())
    sh = ShardedStep(b)
    sh.step(lambda lo, hi: adam_like(b._P, b._G, lo, hi))
    bounds = shard_bounds(n, world)
    ok = bounds[0][0] == 0 and bounds[-1][1] == n and all(lo % 4 == 0 for lo, _ in bounds) and (sh.lo, sh.hi) == bounds[rank]
    ok = ok and torch.equal(a._P, b._P)          # the same bits on every rank (two ranks: a + b is commutative)
    ok = ok and torch.equal(a._G[sh.lo:sh.hi], b._G[sh.lo:sh.hi])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_scatter_sharded_update_all_gather_equals_allreduce():
    """dist.ShardedStep (reduce-scatter of the gradient arena, every rank updates its own shard, all-gather of the parameters) against the
    all-reduce + replicated update, bit for bit, over two gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_shard_bounds_tile_the_arena_for_every_world_size():
    """dist.shard_bounds (ShardedStep: reduce-scatter + sharded Adam + all-gather): for 1 ... 8 ranks and arena sizes that are not multiples of anything,
    the shards are contiguous, cover [0, n) exactly once, every interior boundary is a multiple of 4 elements (the Adam kernel's float4 granules never
    straddle two ranks), all but the last non-empty shard have the same size (what reduce_scatter / all_gather of equal chunks need), and the real arenas
    (W32: 41.1 M, W48: 66.1 M parameters) split evenly to within one granule."""
    import random
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    from mfcnet_amd.dist import shard_bounds
    rnd = random.Random(7)
    sizes = [1, 3, 4, 5, 17, 1023, 41_139_221, 66_104_869] + [rnd.randrange(1, 10_000_000) for _ in range(50)]
    for n in sizes:
        for world in range(1, 9):
            b = shard_bounds(n, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            for r in range(world):
                lo, hi = b[r]
                assert 0 <= lo <= hi <= n
                if r:
                    assert lo == b[r - 1][1]
                if hi < n:
                    assert hi % 4 == 0
            full = [hi - lo for lo, hi in b if hi < n]
            assert len(set(full)) <= 1
            if n > 64 * world:
                per = -(-n // world)
                assert max(hi - lo for lo, hi in b) - per < 4
