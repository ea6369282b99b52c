"""Host-side training flow of the reference (scripts/train_multiframe_detection.py:128-165, src/engine.py:54-71) on the
CPU, without any kernel: gradient exposure / accumulation semantics of the flat arena, the frozen-base (head-only) mode,
FlatAdam as a torch Optimizer under an lr scheduler, and the bucket reducer's hand-shake with train_step.
A stub plan stands in for the HIP programs: it writes a known pattern into the gradient arena."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mfcnet_amd as mfc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 8


class StubPlan:
    """run_backward of plan.Plan: zeroes the arena (the MEMSET record), then writes this pass's gradient."""

    def __init__(self, model, frozen):
        self.m, self.frozen, self.calls = model, frozen, 0

    def run_backward(self, gout):
        self.calls += 1
        self.m._G.zero_()
        lo = self.m._n_base if self.frozen else 0
        self.m._G[lo:] = 1.0


def make(frozen):
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=W)
    if frozen:                                    # train_multiframe_detection.py:159-165
        m.base_model.eval()
        for p in m.base_model.parameters():
            p.requires_grad = False
        m.multiframe_net.train()
    return m


def test_head_only_mode_never_accumulates_across_zero_grad():
    m = make(frozen=True)
    plan = StubPlan(m, True)
    opt = torch.optim.Adam(m.multiframe_net.parameters(), lr=1e-4)      # the reference's optimizer in this mode
    head_w = m.multiframe_net.multiframe_net[9].weight
    for it in range(4):
        opt.zero_grad()
        m._run_backward(plan, None)
        assert float(head_w.grad.mean()) == 1.0, it                    # (used to read 1, 2, 3, ...)
        assert all(p.grad is None for p in m.base_model.parameters())
    # without zero_grad the head accumulates, as autograd would
    m._run_backward(plan, None)
    assert float(head_w.grad.mean()) == 2.0
    assert all(p.grad is None for p in m.base_model.parameters())


def test_full_mode_accumulation_and_mixed_state():
    m = make(frozen=False)
    plan = StubPlan(m, False)
    m._run_backward(plan, None)
    w0, wh = m.base_model.conv1.weight, m.multiframe_net.multiframe_net[9].weight
    assert float(w0.grad.mean()) == 1.0 and float(wh.grad.mean()) == 1.0
    m._run_backward(plan, None)                                         # no zero_grad: both accumulate
    assert float(w0.grad.mean()) == 2.0 and float(wh.grad.mean()) == 2.0
    w0.grad = None                                                      # mixed: only the head still holds a gradient
    m._run_backward(plan, None)
    assert float(w0.grad.mean()) == 1.0 and float(wh.grad.mean()) == 3.0


def test_frozen_base_selects_a_plan_without_base_backward():
    """_get_plan keys on the freeze state; Plan(base_frozen=True) is exercised on the GPU (tests/test_gpu_model.py)."""
    import inspect
    from mfcnet_amd import plan as P
    assert "base_frozen" in inspect.signature(P.Plan.__init__).parameters
    src = inspect.getsource(mfc.HRNetMultiLarge._get_plan)
    assert "requires_grad" in src and "base_frozen=frozen" in src


def test_flat_adam_is_an_optimizer_with_param_groups():
    m = make(frozen=False)
    opt = mfc.FlatAdam(m, lr=1e-4)
    assert isinstance(opt, torch.optim.Optimizer)
    assert [g["name"] for g in opt.param_groups] == ["base_model", "multiframe_net"]
    assert abs(opt.param_groups[0]["lr"] - 1e-4 / 3) < 1e-12 and opt.param_groups[1]["lr"] == 1e-4   # lr / T, lr (:137-151)
    n = sum(len(g["params"]) for g in opt.param_groups)
    assert n == len(list(m.parameters()))
    # the reference builds a StepLR even for 'Constant' (train_multiframe_detection.py:152-157)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sched.step()
    assert abs(opt.param_groups[1]["lr"] - 1e-5) < 1e-12 and abs(opt.lrs["base_model"] - 1e-5 / 3) < 1e-12
    opt100 = mfc.FlatAdam(m, lr=1e-4, load_wts_base_model=True)
    assert abs(opt100.param_groups[0]["lr"] - 1e-4 / 300) < 1e-15                                       # lr / (100 T) (:131-135)
    # frozen base: one group, the per-frame network is never updated
    mf = make(frozen=True)
    of = mfc.FlatAdam(mf, lr=1e-4)
    assert [g["name"] for g in of.param_groups] == ["multiframe_net"]
    assert of.param_groups[0]["segment"] == (mf._n_base, mf._np)
    sd = of.state_dict()
    assert sd["layout"] == "flat-arena-v1" and set(sd["lrs"]) == {"multiframe_net"}


# ---------------------------------------------------------------- bucket reducer <-> train_step (2 gloo ranks, CPU)
def _reducer_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from mfcnet_amd import engine
    from mfcnet_amd.dist import GradBucketReducer
    n = 1000
    model = SimpleNamespace(_G=torch.zeros(n), grad_bucket_hook=None)

    def backward():                 # what plan.run_backward does with a hook: finalise the arena bucket by bucket
        model._G[:] = float(rank + 1)
        hook = model.grad_bucket_hook
        if hook is not None:
            for lo, hi in ((600, 1000), (250, 600), (0, 250)):
                hook(lo, hi)

    red = GradBucketReducer(model, average=False)
    ok = model._bucket_reducer is red
    backward()
    engine.reduce_gradients(model, world)                       # the reduce step of train_step
    ok = ok and torch.allclose(model._G, torch.full((n,), 3.0))       # 1 + 2, ONCE (used to be reduced twice: 6)
    ok = ok and red.works == [] and red.ranges == []
    red.remove()
    ok = ok and model._bucket_reducer is None and model.grad_bucket_hook is None
    backward()
    engine.reduce_gradients(model, world)                       # no reducer: one all-reduce of the arena
    ok = ok and torch.allclose(model._G, torch.full((n,), 3.0))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_finishes_an_installed_bucket_reducer():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


# ------------------------------------------------------------------ round 4
def test_flat_adam_with_an_external_base_model():
    """base_model="resunet_vb": the per-frame network's parameters are ordinary nn.Parameters outside the flat arena; FlatAdam used to
    raise KeyError on the first of them.  It now covers the arena (the temporal head) and steps a torch Adam for the rest."""
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, base_model="resunet_vb")
    opt = mfc.FlatAdam(m, lr=1e-3)
    assert [g["name"] for g in opt.param_groups] == ["multiframe_net"]
    ext = [p for n, p in m.named_parameters() if n not in m._poff]
    assert opt.external is not None and len(opt.external.param_groups[0]["params"]) == len(ext) > 0
    assert abs(opt.external.param_groups[0]["lr"] - 1e-3 / 3) < 1e-12          # the reference's base-group rate lr / T
    for p in ext:
        p.grad = torch.ones_like(p)
    opt.zero_grad()
    assert all(p.grad is None for p in ext)
    sd = opt.state_dict()
    assert "external" in sd and sd["layout"] == "flat-arena-v1"
    opt.load_state_dict(sd)


def test_flat_adam_state_carries_skip_counter_and_loss_scaler():
    m = make(frozen=False)
    m.loss_scaler = mfc.LossScaler(4096.0)
    m.loss_scaler._steps, m.loss_scaler._clean, m.loss_scaler._seen = 50, 7, 2
    opt = mfc.FlatAdam(m, lr=1e-3)
    sd = opt.state_dict()
    assert sd["skipped_steps"] == 0 and sd["loss_scaler"]["scale"] == 4096.0 and sd["loss_scaler"]["seen"] == 2
    assert all("_runs" not in g for g in opt.param_groups)          # (the run cache no longer lives in the param_groups)
    m2 = make(frozen=False)
    o2 = mfc.FlatAdam(m2, lr=1e-3)
    sd["skipped_steps"] = 3
    o2.load_state_dict(sd)
    assert m2.loss_scaler.scale == 4096.0 and m2.loss_scaler._clean == 7 and o2.skipped_steps() == 3


def test_step_helpers_unwrap_the_dataparallel_stand_in():
    """train_step / reduce_gradients / loss_scale_for read the reducer, the compute dtype and the arenas from the model inside a
    mfcnet_amd.DataParallel wrapper (they used to look at the wrapper and silently lose the fp16 loss scale)."""
    from mfcnet_amd import _lib as L
    from mfcnet_amd.engine import loss_scale_for, unwrap
    m = make(frozen=False)
    dp = mfc.DataParallel(m)
    assert unwrap(dp) is m and unwrap(m) is m
    m.compute_dtype = L.F16
    out = torch.empty(8, 5, 480, 640, device="meta")
    assert loss_scale_for(dp, out) == loss_scale_for(m, out) == 2.0 ** 17
    opt = mfc.FlatAdam(dp, lr=1e-3)                        # the optimizer accepts the wrapper as well
    assert opt.model is m
