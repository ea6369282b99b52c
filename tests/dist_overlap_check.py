"""Run by tests/test_gpu_bench.py under torch.distributed.run (2 ranks, gloo, sharing the GPU): the gradients a step obtains with the
bucketed all-reduce started from inside the backward (segments, detached unpacks, mfc_wait_detached ordering) equal those of
the plain order -- whole backward, then one all-reduce of the arena."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))


def main():
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    import mfcnet_amd as mfc
    from mfcnet_amd.dist import GradBucketReducer, allreduce_grads, broadcast_params
    torch.manual_seed(7)
    # training-mode BatchNorm, as in a real step.  Nothing in a step depends on the order work finishes in (fp64 statistic / loss cells,
    # gradients as fixed-order slice sums), and a two-rank sum is commutative: the two orders must give the SAME BITS -- a bucket reduced
    # before its gradients were final would be off by O(1)
    model = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=16, compute_dtype="fp32").cuda().train()
    broadcast_params(model)
    g = torch.Generator().manual_seed(100 + rank)
    frames = [torch.randn(2, 3, 64, 96, generator=g).cuda() for _ in range(3)]
    mask = torch.randint(0, 5, (2, 64, 96), generator=g).cuda()

    def backward():
        for p in model.parameters():
            p.grad = None
        loss, _ = mfc.mfc_loss(model(frames), mask, global_batch=True)
        loss.backward()
        return float(loss)

    red = GradBucketReducer(model, average=False)
    l1 = backward()
    ranges = red.finish()
    torch.cuda.synchronize()
    g1 = model._G.detach().clone()
    red.remove()
    l2 = backward()
    allreduce_grads(model, average=False)
    torch.cuda.synchronize()
    g2 = model._G.detach().clone()
    rel = float((g1 - g2).double().norm() / g2.double().norm())
    # per bucket as well: a bucket reduced too early would be off by O(1) on its own range only
    worst = max(float((g1[lo:hi] - g2[lo:hi]).double().norm() / (g2[lo:hi].double().norm() + 1e-30)) for lo, hi in ranges)
    ok = l1 == l2 and torch.equal(g1, g2) and rel == 0.0 and worst == 0.0 and float(g2.abs().max()) > 0 and len(ranges) >= 2
    # train_step with the reducer installed: it must FINISH the bucket all-reduces, not reduce the arena a second time
    red2 = GradBucketReducer(model, average=False)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    mfc.train_step(model, opt, frames, mask, world_size=2)
    torch.cuda.synchronize()
    g3 = model._G.detach().clone()
    red2.remove()
    rel3 = float((g3 - g2).double().norm() / g2.double().norm())
    ok = ok and torch.equal(g3, g2) and red2.works == []
    # the same through the nn.DataParallel stand-in: train_step must find the reducer / arenas of the model INSIDE the wrapper
    from mfcnet_amd.dist import DataParallel, ShardedStep
    dp = DataParallel(model, device_ids=[0], broadcast=False)
    mfc.train_step(dp, opt, frames, mask, world_size=2)
    torch.cuda.synchronize()
    g4 = model._G.detach().clone()
    ok_dp = torch.equal(g4, g2) and dp.reducer.works == []
    dp.reducer.remove()
    ok = ok and ok_dp
    # reduce-scatter + sharded FlatAdam + all-gather against all-reduce + replicated FlatAdam: the same parameter bits on every rank
    p_start = model._P.detach().clone()
    oa = mfc.FlatAdam(model, lr=1e-3)
    backward(); allreduce_grads(model, average=False); oa.step()
    torch.cuda.synchronize()
    p_a = model._P.detach().clone()
    model._P.copy_(p_start)
    ob = mfc.FlatAdam(model, lr=1e-3)
    sh = ShardedStep(model)
    backward(); sh.step(ob)
    torch.cuda.synchronize()
    p_b = model._P.detach().clone()
    ok_sh = torch.equal(p_a, p_b) and float((p_a - p_start).abs().max()) > 0
    ok = ok and ok_sh
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"OVERLAP_CHECK ok={int(flag.item())} rel={rel:.3e} worst_bucket={worst:.3e} loss={l1:.6f}/{l2:.6f} buckets={len(ranges)} train_step_rel={rel3:.3e} wrapper={int(ok_dp)} sharded={int(ok_sh)}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
