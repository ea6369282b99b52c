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


_CW_CACHE = {}


def _class_weights(class_weights, device):
    """the class-weight vector on the device, uploaded ONCE per (values, device): a host-to-device copy from pageable memory makes the host wait
    for everything already enqueued on the stream (HIP stages it through the host), i.e. a hidden synchronisation between the forward and the
    backward of every step -- the GPU then idles while Python enqueues the loss and the first backward records (1.5 ms in the event timeline of
    round 4, `profiles/r04_timeline.txt`)."""
    if torch.is_tensor(class_weights):
        if class_weights.device == device and class_weights.dtype == torch.float32:
            return class_weights
        class_weights = tuple(float(x) for x in class_weights.detach().cpu().tolist())
    key = (tuple(float(x) for x in class_weights), str(device))
    t = _CW_CACHE.get(key)
    if t is None:
        t = _CW_CACHE[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
    return t


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
        # the upstream gradient (1, or the loss scale of an fp16 step) is a device scalar: the kernel multiplies it in (no host read, no
        # second pass over the 49 MB logit gradient)
        gt = gtot.detach().to(device=logits.device, dtype=torch.float32).contiguous()
        d = L.LossDesc(logits.data_ptr(), target.data_ptr(), class_w.data_ptr(), acc.data_ptr(), dl.data_ptr(), B, nc, H, W,
                       ctx.w[0], ctx.w[1], 1.0, 0, gt.data_ptr())
        L.call(L.lib.mfc_loss_bwd, d)
        return dl, None, None, None, None, None


def mfc_loss(logits, target, class_weights=DEFAULT_CLASS_WEIGHTS, w_nll=0.7, w_jac=0.3, global_batch=False, group=None,
             check_labels=False):
    """`global_batch=True`: evaluate the loss over all ranks' clips (see module docstring); `group` = process group.
    Targets outside [0, num_classes) are ignored by the kernels and counted in acc[29] (nn.NLLLoss raises on them, loss.py:31-43);
    `check_labels=True` reads that counter back (one host sync) and raises IndexError like the reference would."""
    if not logits.is_cuda:
        raise L.MfcError("mfc_loss runs on the GPU only")
    cw = _class_weights(class_weights, logits.device)
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


class _LazyLossDict(dict):
    """`loss_dict` of get_loss: the reference fills it with `loss.item()` floats (three host synchronisations per step, src/loss.py:19-20);
    here the values stay on the device until somebody READS them -- indexing returns the Python float (one sync, at the reader)."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return float(v) if torch.is_tensor(v) else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


def get_loss(outputs, targets, loss_fns, loss_wts, args):
    """Drop-in for the reference's `get_loss(outputs, targets, loss_fns, loss_wts, args)` (src/loss.py:6-21), called as in
    src/engine.py:65-66 with `outputs = F.log_softmax(model(...), dim=1)`:
        total_loss, loss_dict = get_loss(output, mask, args.loss_fns, args.loss_wts, args)
    'nll' (class weights `args.class_weights`, an ndarray or None, src/loss.py:31-43) and 'soft_jaccard' (:45-63), in any order and with
    any weights, are evaluated by ONE fused kernel pair on the device (forward sums, gradient w.r.t. `outputs`); 'mse' is torch's
    `F.mse_loss` (not on this path's hot loop); any other name raises the reference's ValueError.  `outputs` may be log-probabilities
    (what the reference passes) or raw logits: the kernels normalise with log_softmax, which is idempotent -- log_softmax(log_softmax(x)) =
    log_softmax(x), and its Jacobian J = I - 1 p^T satisfies J J = J, so the gradient that reaches the logits through torch's log_softmax
    node is the reference's.  `loss_dict` holds 'loss_<fn>' and 'loss_total' like the reference's; its values are read from the device
    when accessed (no `.item()` inside the step)."""
    if len(loss_fns) != len(loss_wts):
        loss_fns, loss_wts = loss_fns[:min(len(loss_fns), len(loss_wts))], loss_wts[:min(len(loss_fns), len(loss_wts))]      # (zip() semantics)
    for fn in loss_fns:
        if fn not in ("mse", "nll", "soft_jaccard"):
            raise ValueError(f"Loss function {fn} not implemented")
    w_nll = float(sum(w for f, w in zip(loss_fns, loss_wts) if f == "nll"))
    w_jac = float(sum(w for f, w in zip(loss_fns, loss_wts) if f == "soft_jaccard"))
    nc = int(getattr(args, "num_classes", outputs.shape[1]))
    if nc != outputs.shape[1]:
        raise ValueError(f"get_loss: args.num_classes = {nc} but the output has {outputs.shape[1]} channels")
    cw = getattr(args, "class_weights", None)
    cw = tuple(float(x) for x in cw) if cw is not None else (1.0,) * nc
    out = _LazyLossDict()
    total = 0.0
    if any(f in ("nll", "soft_jaccard") for f in loss_fns):
        total, acc = mfc_loss(outputs, targets, cw, w_nll, w_jac, global_batch=bool(getattr(args, "global_batch_loss", False)))
        if "nll" in loss_fns:
            dict.__setitem__(out, "loss_nll", acc[26])
        if "soft_jaccard" in loss_fns:
            dict.__setitem__(out, "loss_soft_jaccard", acc[27])
    for f, w in zip(loss_fns, loss_wts):
        if f == "mse":
            m = torch.nn.functional.mse_loss(outputs, targets)
            total = total + w * m
            dict.__setitem__(out, "loss_mse", m.detach())
    dict.__setitem__(out, "loss_total", total.detach() if torch.is_tensor(total) else total)
    return total, out
