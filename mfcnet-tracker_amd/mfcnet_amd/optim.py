"""Fused Adam over the model's flat parameter arena.

Same update rule and the same learning-rate groups as the reference's
`torch.optim.Adam([{base_model params, lr/T or lr/(100T)}, {multiframe_net params, lr}])`
(scripts/train_multiframe_detection.py:128-151; default betas/eps, no weight decay), but one kernel
launch per group over contiguous fp32 segments instead of 931 per-tensor updates.

FlatAdam IS a `torch.optim.Optimizer`: `param_groups` holds one group per arena segment that has trainable
parameters (name, params, lr, betas, eps), so the reference's `StepLR(optimizer, ...)`
(train_multiframe_detection.py:152-157) and every other `lr_scheduler` drive it unchanged -- `step()` reads each
group's current `lr`.  A segment whose parameters all have `requires_grad=False` gets no group and is never
updated: with the per-frame network frozen (the reference's default mode, :159-165: `optim.Adam(
model.multiframe_net.parameters())`) only the temporal head moves.  Frozen parameters inside a trainable segment are
SKIPPED, like torch.optim.Adam skips parameters whose `.grad` is None: a segment is updated as its maximal runs of
consecutive trainable parameters (one launch per run), so a parameter frozen after some steps keeps its value and its
moments instead of drifting on stale momentum.

Checkpoints: `state_dict()` stores the two moment arenas as flat tensors (layout "flat-arena-v1"), not torch
Adam's per-parameter dict, so the 'optimizer' entry of a file written with FlatAdam loads into FlatAdam only (and
the reference's into `torch.optim.Adam`, which also works on this model: parameters are views of the arena).
"""
from __future__ import annotations

import torch

from . import _lib as L


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, load_wts_base_model=False, betas=(0.9, 0.999), eps=1e-8):
        model = getattr(model, "module", model)          # (a DataParallel-style wrapper: the arenas belong to the model inside)
        self.model = model
        T = model.num_frames
        base_lr = lr / (100.0 * T) if load_wts_base_model else lr / T
        lrs = {"base_model": base_lr, "multiframe_net": lr}
        # only parameters that live in the flat arena (model._poff); with base_model="resunet_vb" the per-frame network's parameters are
        # ordinary nn.Parameters outside it (the arena's base segment is then empty): they get a torch.optim.Adam of their own, stepped
        # from step(), with the reference's base learning rate (scripts/train_multiframe_detection.py:128-151)
        named = [(n, p) for n, p in model.named_parameters() if n in model._poff]
        external = [p for n, p in model.named_parameters() if n not in model._poff and p.requires_grad]
        groups = []
        for name, (a, b) in model.flat_segments().items():
            if b <= a:
                continue
            ps = [p for n, p in named if a <= model._poff[n] < b]
            if any(p.requires_grad for p in ps):
                groups.append({"params": ps, "lr": lrs[name], "name": name, "segment": (a, b)})
        if not groups and not external:
            raise ValueError("FlatAdam: no parameter of the model requires a gradient")
        if not groups:
            raise ValueError("FlatAdam: no parameter of the flat arena requires a gradient (use torch.optim.Adam for an external base model alone)")
        super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.external = torch.optim.Adam(external, lr=base_lr, betas=tuple(betas), eps=eps) if external else None
        self._run_cache = {}         # group name -> (requires_grad pattern, runs): kept off the param_groups so that nothing serialises it
        self.step_count = 0
        self.m = torch.zeros_like(model._P)
        self.v = torch.zeros_like(model._P)
        self._arena = model._P
        self._flag = None            # device int32[2] of the overflow guard: [this step's gradients not finite, number of such steps]

    @property
    def lrs(self):
        return {g["name"]: g["lr"] for g in self.param_groups}

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, guard=None, shard=None):
        """`grad_scale` multiplies the gradients inside the kernel (1 / loss scale of an fp16 step).  `guard` (default: on whenever
        grad_scale != 1): check the whole gradient arena for Inf / NaN on the device first and make the update a no-op when there is
        one -- an overflowed half-precision gradient must not reach the moments.  No host synchronisation; `skipped_steps()` reads
        the counter.  `shard` = (lo, hi): update only that range of the arena (dist.ShardedStep: every rank updates its own shard after
        a reduce-scatter of the gradients and the shards are all-gathered afterwards); the overflow guard then looks at the shard only,
        so a data-parallel fp16 run should keep the all-reduce exchange, where every rank sees every gradient."""
        loss = closure() if closure is not None else None
        if self.external is not None:
            if grad_scale != 1.0:
                for p in self.external.param_groups[0]["params"]:
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
            self.external.step()
        mdl = self.model
        if mdl._P is not self._arena:
            raise L.MfcError("model was moved after the optimizer was built; rebuild FlatAdam")
        self.step_count += 1
        st = L.stream_ptr()
        if guard is None:
            guard = grad_scale != 1.0
        if guard:
            if self._flag is None:
                self._flag = torch.zeros(2, dtype=torch.int32, device=mdl._P.device)
            glo, ghi = (0, mdl._G.numel()) if shard is None else shard
            L.check(L.lib.mfc_grad_check(mdl._G.data_ptr() + 4 * glo, ghi - glo, self._flag.data_ptr(), st), "mfc_grad_check")
        for g in self.param_groups:
            for a, b in self._runs(g):
                if shard is not None:
                    a, b = max(a, shard[0]), min(b, shard[1])
                    if b <= a:
                        continue
                args = (mdl._P.data_ptr() + 4 * a, mdl._G.data_ptr() + 4 * a, self.m.data_ptr() + 4 * a, self.v.data_ptr() + 4 * a, b - a,
                        float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], self.step_count, grad_scale)
                if guard:
                    # (bias correction from the number of steps actually APPLIED: step_count minus the skipped ones, read on the device)
                    L.check(L.lib.mfc_adam_step_guarded(*args, self._flag.data_ptr(), st), "mfc_adam_step_guarded")
                else:
                    L.check(L.lib.mfc_adam_step(*args, st), "mfc_adam_step")
        return loss

    def _runs(self, g):
        """maximal runs [a, b) of consecutive trainable parameters of a group's segment (the whole segment when nothing in it is frozen)"""
        key = tuple(p.requires_grad for p in g["params"])
        cache = self._run_cache.setdefault(g["name"], {})
        if key not in cache:
            base = self.model._P.data_ptr()
            spans = sorted(((p.data_ptr() - base) // 4, (p.numel() + 3) // 4 * 4, p.requires_grad) for p in g["params"])      # (16-byte padded, as the arena lays them out)
            runs = []
            for off, n, tr in spans:
                if not tr:
                    continue
                if runs and runs[-1][1] == off:
                    runs[-1][1] = off + n
                else:
                    runs.append([off, off + n])
            if all(t for _, _, t in spans):
                runs = [list(g["segment"])]                  # (alignment padding between parameters included: one launch)
            cache.clear()
            cache[key] = [tuple(r) for r in runs]
        return cache[key]

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none)
        if self.external is not None:
            self.external.zero_grad(set_to_none)

    def skipped_steps(self) -> int:
        """guarded steps whose gradients were not finite (one host read)"""
        return 0 if self._flag is None else int(self._flag[1])

    # ---- checkpointing (utils/model_utils.py:6-12 stores optimizer.state_dict() next to the model's) ----
    def state_dict(self):
        sd = {"step": self.step_count, "exp_avg": self.m.detach().cpu().clone(), "exp_avg_sq": self.v.detach().cpu().clone(),
              "lrs": dict(self.lrs), "betas": tuple(self.param_groups[0]["betas"]), "eps": self.param_groups[0]["eps"],
              "layout": "flat-arena-v1",
              # guarded (loss-scaled) steps: the device counter of skipped steps enters the bias correction, and the adaptive loss scale is
              # part of the optimisation state -- an interrupted run resumes with both
              "skipped_steps": self.skipped_steps()}
        sc = getattr(self.model, "loss_scaler", None)
        if sc is not None:
            sd["loss_scaler"] = sc.state_dict()
        if self.external is not None:
            sd["external"] = self.external.state_dict()
        return sd

    def load_state_dict(self, sd):
        if sd.get("layout") != "flat-arena-v1" or sd["exp_avg"].numel() != self.m.numel():
            raise L.MfcError("optimizer state does not belong to a FlatAdam of this model (torch.optim.Adam state dicts load "
                             "into torch.optim.Adam, which also works on this model)")
        self.step_count = int(sd["step"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        for g in self.param_groups:
            if g["name"] in sd["lrs"]:
                g["lr"] = sd["lrs"][g["name"]]
            g["betas"], g["eps"] = tuple(sd["betas"]), float(sd["eps"])
        nskip = int(sd.get("skipped_steps", 0))
        if nskip or self._flag is not None:
            if self._flag is None:
                self._flag = torch.zeros(2, dtype=torch.int32, device=self.model._P.device)
            self._flag[0] = 0
            self._flag[1] = nskip
        if "loss_scaler" in sd:
            from .engine import LossScaler
            sc = getattr(self.model, "loss_scaler", None)
            if sc is None:
                sc = self.model.loss_scaler = LossScaler(1.0)
            sc.load_state_dict(sd["loss_scaler"])
        if self.external is not None and "external" in sd:
            self.external.load_state_dict(sd["external"])
