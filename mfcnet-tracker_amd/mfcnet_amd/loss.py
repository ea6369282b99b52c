"""Fused log_softmax + class-weighted NLL + soft-Jaccard on device (src/engine.py:65-66, src/loss.py:6-63).

`mfc_loss(logits, target, ...)` returns (total_loss_tensor, acc) where `acc[26:29]` holds
(nll, soft_jaccard, total) as device scalars -- no `.item()` host sync is needed inside the step
(the reference syncs three times per step).

Data parallel (`global_batch=True` with an initialised process group): the 26 partial sums are all-reduced between the
partial and the finalize kernels, so both terms are evaluated over the WHOLE batch exactly as the reference does after
DataParallel's gather (engine.py:64-66; the Jaccard I/U sums span the batch, loss.py:57-58).  The logit gradients then
carry the global normalisers, and the parameter gradients of the ranks must be SUMMED (dist.allreduce_grads(average=False)).
"""
from __future__ import annotations

import torch

from . import _lib as L

DEFAULT_CLASS_WEIGHTS = (1.0, 1000.0, 1000.0, 1000.0, 1000.0)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, class_w, w_nll, w_jac, group):
        logits = logits.contiguous().float()
        target = target.contiguous()
        B, nc, H, W = logits.shape
        acc = torch.empty(L.LOSS_ACC_FLOATS, dtype=torch.float32, device=logits.device)[:32]     # (the view keeps the whole block alive; [32:] is kernel scratch)
        d = L.LossDesc(logits.data_ptr(), target.data_ptr(), class_w.data_ptr(), acc.data_ptr(), 0, B, nc, H, W, w_nll, w_jac, 1.0)
        if group is None:
            L.call(L.lib.mfc_loss_fwd, d)
        else:
            from .dist import allreduce_loss_sums
            L.call(L.lib.mfc_loss_partial, d)
            allreduce_loss_sums(acc, group if group is not True else None)
            L.call(L.lib.mfc_loss_finalize, d)
        ctx.save_for_backward(logits, target, class_w, acc)
        ctx.w = (w_nll, w_jac)
        ctx.mark_non_differentiable(acc)
        return acc[28].clone(), acc

    @staticmethod
    def backward(ctx, gtot, _gacc):
        logits, target, class_w, acc = ctx.saved_tensors
        B, nc, H, W = logits.shape
        dl = torch.empty_like(logits)
        d = L.LossDesc(logits.data_ptr(), target.data_ptr(), class_w.data_ptr(), acc.data_ptr(), dl.data_ptr(), B, nc, H, W,
                       ctx.w[0], ctx.w[1], 1.0)
        L.call(L.lib.mfc_loss_bwd, d)
        return dl * gtot, None, None, None, None, None


def mfc_loss(logits, target, class_weights=DEFAULT_CLASS_WEIGHTS, w_nll=0.7, w_jac=0.3, global_batch=False, group=None,
             check_labels=False):
    """`global_batch=True`: evaluate the loss over all ranks' clips (see module docstring); `group` = process group.
    Targets outside [0, num_classes) are ignored by the kernels and counted in acc[29] (nn.NLLLoss raises on them, loss.py:31-43);
    `check_labels=True` reads that counter back (one host sync) and raises IndexError like the reference would."""
    if not logits.is_cuda:
        raise L.MfcError("mfc_loss runs on the GPU only")
    cw = torch.as_tensor(class_weights, dtype=torch.float32, device=logits.device)
    if target.dtype != torch.int64:
        target = target.long()
    g = None
    if global_batch:
        import torch.distributed as dist
        from .dist import _active
        if _active(group):
            g = group if group is not None else True
    loss, acc = _LossFn.apply(logits, target, cw, float(w_nll), float(w_jac), g)
    if check_labels and float(acc[29]) > 0:
        raise IndexError(f"Target out of bounds: {int(acc[29])} label(s) outside [0, {logits.shape[1]})")
    return loss, acc
