"""The step bodies of src/engine.py on device, without host synchronisation.

train_step  = engine.py:54-71   zero_grad -> model(input[, optflow][, depth]) -> log_softmax + 0.7*NLL + 0.3*soft-Jaccard
                                -> backward -> (gradient all-reduce) -> optimizer.step
eval_step   = engine.py:132-139 model.eval() forward under no_grad -> the same loss -> get_metrics
The reference calls `.item()` three times per training step and copies every validation batch to the host; here the loss
scalars stay in a 32-float device block and only the B*nc*nc confusion counts travel (metrics.py).
"""
from __future__ import annotations

import torch

from ._lib import MfcError
from .dist import allreduce_grads
from .loss import DEFAULT_CLASS_WEIGHTS, mfc_loss
from .metrics import confusion_counts, metrics_from_confusion


def _forward(model, input, optflow, depth):
    if optflow is not None and depth is not None:
        return model(input, optflow=optflow, depth=depth)
    if optflow is not None:
        return model(input, optflow=optflow)
    if depth is not None:
        return model(input, depth=depth)
    return model(input)


def unwrap(model):
    """the model behind a `mfcnet_amd.DataParallel` / `nn.DataParallel`-style wrapper (`.module`), or the model itself: the flat arenas,
    the compute dtype, the loss scale and the bucket reducer are attributes of the model, not of its wrapper"""
    return getattr(model, "module", model)


def reduce_gradients(model, world_size, group=None):
    """The data-parallel exchange of a step: finish the per-bucket all-reduces a `dist.GradBucketReducer` started inside the
    backward pass, or -- without one -- reduce the whole arena now.  SUM either way: `mfc_loss(global_batch=True)` already
    normalised the logit gradients over the global batch."""
    if world_size <= 1:
        return
    model = unwrap(model)
    red = getattr(model, "_bucket_reducer", None)
    if red is not None:
        red.finish()
    else:
        allreduce_grads(model, world_size, group=group, average=False)


class LossScaler:
    """Loss scale of fp16 training that adapts without a host round trip per step (torch.cuda.amp.GradScaler's policy: halve after a step
    whose gradients overflowed, double after `growth_interval` clean steps).  The overflow check and the skip of the update happen on
    the device (mfc_grad_check / mfc_adam_step_guarded); this object only reads the 4-byte skipped-step counter every `check_every`
    steps -- the scale therefore reacts with a delay of at most that many (skipped, hence harmless) steps."""

    def __init__(self, init_scale, growth_interval=2000, check_every=25, min_scale=1.0, max_scale=2.0 ** 24):
        self.scale, self.growth_interval, self.check_every = float(init_scale), int(growth_interval), int(check_every)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        self._steps, self._clean, self._seen = 0, 0, 0

    def state_dict(self):
        return {"scale": self.scale, "growth_interval": self.growth_interval, "check_every": self.check_every, "min_scale": self.min_scale,
                "max_scale": self.max_scale, "steps": self._steps, "clean": self._clean, "seen": self._seen}

    def load_state_dict(self, sd):
        self.scale, self.growth_interval, self.check_every = float(sd["scale"]), int(sd["growth_interval"]), int(sd["check_every"])
        self.min_scale, self.max_scale = float(sd["min_scale"]), float(sd["max_scale"])
        self._steps, self._clean, self._seen = int(sd["steps"]), int(sd["clean"]), int(sd["seen"])

    def update(self, optimizer):
        self._steps += 1
        self._clean += 1
        if self._steps % self.check_every:
            return
        skipped = optimizer.skipped_steps() if hasattr(optimizer, "skipped_steps") else 0
        if skipped > self._seen:                       # overflow(s) since the last look: back off
            self._seen = skipped
            self.scale = max(self.scale * 0.5, self.min_scale)
            self._clean = 0
        elif self._clean >= self.growth_interval:
            self.scale = min(self.scale * 2.0, self.max_scale)
            self._clean = 0


def loss_scale_for(model, output, world_size=1) -> float:
    """Loss scale of a step.  1 for fp32 / bf16 storage.  fp16 storage (BASELINE configs[4]) keeps gradients in IEEE half, whose
    smallest normal is 6.1e-5 while a logit gradient of the mean-reduced loss is about 1 / (B*H*W) (4e-7 at B=8, 480x640): the loss is
    multiplied by a power of two near B*H*W / 16 (largest logit gradient, class weight 1000, stays below ~64) and FlatAdam divides it
    out in fp32.  `model.loss_scale` (a number) overrides the rule; `model.loss_scaler` (a LossScaler, created by train_step on first use
    unless `model.loss_scale` pins the scale) adapts it.  B is the GLOBAL batch (world_size x the local one): the loss is normalised over it."""
    from . import _lib as L
    model = unwrap(model)
    if getattr(model, "compute_dtype", None) != L.F16:
        return 1.0
    s = getattr(model, "loss_scale", None)
    if s:
        return float(s)
    import math
    sc = getattr(model, "loss_scaler", None)
    if sc is not None:
        return sc.scale
    B, _, H, W = output.shape
    return float(2 ** max(0, round(math.log2(max(B * max(int(world_size), 1) * H * W / 16.0, 1.0)))))


def train_step(model, optimizer, input, mask, optflow=None, depth=None, loss_wts=(0.7, 0.3),
               class_weights=DEFAULT_CLASS_WEIGHTS, world_size=1, group=None):
    """One optimisation step; returns (output logits, acc) with acc[26:29] = (nll, soft_jaccard, total) on the device.
    With world_size > 1 the loss is evaluated over the global batch (26 all-reduced sums) and the ranks' gradients are
    summed, i.e. the arithmetic of the reference's DataParallel step."""
    core = unwrap(model)              # (a DataParallel wrapper forwards the call; scaler / reducer / arenas belong to the model inside)
    if world_size > 1 and getattr(core, "base_kind", "hrnet") != "hrnet" and any(p.requires_grad for p in core.base_model.parameters()):
        raise MfcError("data-parallel training with a trainable external base model (base_model='resunet_vb') is not supported: its parameters "
                       "live outside the flat arenas, so neither the initial broadcast nor the gradient all-reduce covers them")
    optimizer.zero_grad()
    output = _forward(model, input, optflow, depth)
    loss, acc = mfc_loss(output, mask, class_weights, loss_wts[0], loss_wts[1], global_batch=world_size > 1, group=group)
    scale = loss_scale_for(core, output, world_size)
    if scale != 1.0 and getattr(core, "loss_scaler", None) is None and not getattr(core, "loss_scale", None):
        core.loss_scaler = LossScaler(scale)           # fp16: starts at the static rule's value, then adapts
    (loss if scale == 1.0 else loss * scale).backward()
    reduce_gradients(core, world_size, group)
    if scale == 1.0:
        optimizer.step()
    else:
        optimizer.step(grad_scale=1.0 / scale)
        if getattr(core, "loss_scaler", None) is not None:
            core.loss_scaler.update(optimizer)
    return output.detach(), acc


@torch.no_grad()
def eval_step(model, input, mask, metric_fns=("iou", "dice"), num_classes=None, optflow=None, depth=None,
              loss_wts=(0.7, 0.3), class_weights=DEFAULT_CLASS_WEIGHTS):
    """Validation batch: returns (output logits, acc, metric values per class, metric dict).  `model` must be in eval mode
    (engine.py:101); the T frames are batched through the base model in one pass (running statistics make that legal)."""
    output = _forward(model, input, optflow, depth)
    _, acc = mfc_loss(output, mask, class_weights, loss_wts[0], loss_wts[1])
    nc = num_classes if num_classes is not None else output.shape[1]
    conf = confusion_counts(output, mask, nc)              # argmax(log_softmax(x)) == argmax(x)
    vals, md = metrics_from_confusion(conf.cpu().numpy(), list(metric_fns))
    return output, acc, vals, md
