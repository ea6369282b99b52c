"""The step bodies of src/engine.py on device, without host synchronisation.

train_step  = engine.py:54-71   zero_grad -> model(input[, optflow][, depth]) -> log_softmax + 0.7*NLL + 0.3*soft-Jaccard
                                -> backward -> (gradient all-reduce) -> optimizer.step
eval_step   = engine.py:132-139 model.eval() forward under no_grad -> the same loss -> get_metrics
The reference calls `.item()` three times per training step and copies every validation batch to the host; here the loss
scalars stay in a 32-float device block and only the B*nc*nc confusion counts travel (metrics.py).
"""
from __future__ import annotations

import torch

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


def reduce_gradients(model, world_size, group=None):
    """The data-parallel exchange of a step: finish the per-bucket all-reduces a `dist.GradBucketReducer` started inside the
    backward pass, or -- without one -- reduce the whole arena now.  SUM either way: `mfc_loss(global_batch=True)` already
    normalised the logit gradients over the global batch."""
    if world_size <= 1:
        return
    red = getattr(model, "_bucket_reducer", None)
    if red is not None:
        red.finish()
    else:
        allreduce_grads(model, world_size, group=group, average=False)


def loss_scale_for(model, output) -> float:
    """Static loss scale of a step.  1 for fp32 / bf16 storage.  fp16 storage (BASELINE configs[4]) keeps gradients in IEEE half, whose
    smallest normal is 6.1e-5 while a logit gradient of the mean-reduced loss is about 1 / (B*H*W) (4e-7 at B=8, 480x640): the loss is
    multiplied by a power of two near B*H*W / 16 (largest logit gradient, class weight 1000, stays below ~64) and FlatAdam divides it
    out in fp32.  `model.loss_scale` (a number) overrides the rule."""
    from . import _lib as L
    if getattr(model, "compute_dtype", None) != L.F16:
        return 1.0
    s = getattr(model, "loss_scale", None)
    if s:
        return float(s)
    import math
    B, _, H, W = output.shape
    return float(2 ** max(0, round(math.log2(max(B * H * W / 16.0, 1.0)))))


def train_step(model, optimizer, input, mask, optflow=None, depth=None, loss_wts=(0.7, 0.3),
               class_weights=DEFAULT_CLASS_WEIGHTS, world_size=1, group=None):
    """One optimisation step; returns (output logits, acc) with acc[26:29] = (nll, soft_jaccard, total) on the device.
    With world_size > 1 the loss is evaluated over the global batch (26 all-reduced sums) and the ranks' gradients are
    summed, i.e. the arithmetic of the reference's DataParallel step."""
    optimizer.zero_grad()
    output = _forward(model, input, optflow, depth)
    loss, acc = mfc_loss(output, mask, class_weights, loss_wts[0], loss_wts[1], global_batch=world_size > 1, group=group)
    scale = loss_scale_for(model, output)
    (loss if scale == 1.0 else loss * scale).backward()
    reduce_gradients(model, world_size, group)
    if scale == 1.0:
        optimizer.step()
    else:
        optimizer.step(grad_scale=1.0 / scale)
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
