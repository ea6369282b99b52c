"""Validation metrics with the argmax and the confusion counts on device (src/metrics.py).

The reference copies every validation batch's `[B, nc, H, W]` log-probabilities and masks to the host
(`outputs.data.cpu().numpy().argmax(axis=1)`, metrics.py:6-7; called from src/engine.py:139).  Here one kernel
(`mfc_confusion_counts`) leaves `B * nc * nc` int64 counts on the device and only those travel; the per-class ratios are
the reference's, quirks included: IoU is taken from the FIRST sample of the batch (`get_jaccard(...)[0]`, metrics.py:41-45),
Dice over the whole batch (metrics.py:47-48), background excluded from both.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def confusion_counts(outputs: torch.Tensor, targets: torch.Tensor, num_classes: int) -> torch.Tensor:
    """int64 [B, num_classes, num_classes] on the device: conf[b][truth][prediction]."""
    if not outputs.is_cuda:
        raise L.MfcError("confusion_counts runs on the GPU only")
    out = outputs.detach().contiguous().float()
    tgt = targets.detach().contiguous()
    if tgt.dtype != torch.int64:
        tgt = tgt.long()
    B, nc, H, W = out.shape
    if nc != num_classes or tuple(tgt.shape) != (B, H, W):
        raise L.MfcError(f"outputs {tuple(out.shape)} / targets {tuple(tgt.shape)} do not match num_classes={num_classes}")
    conf = torch.empty(B, nc, nc, dtype=torch.int64, device=out.device)
    L.check(L.lib.mfc_confusion_counts(out.data_ptr(), tgt.data_ptr(), conf.data_ptr(), B, nc, H, W, L.stream_ptr()), "mfc_confusion_counts")
    return conf


def metrics_from_confusion(conf, metric_fns):
    conf = np.asarray(conf, dtype=np.float64)
    nc = conf.shape[1]
    eps = 1e-15
    vals, md = [], {}
    for fn in metric_fns:
        if fn == "jaccard":
            raise NotImplementedError
        if fn not in ("iou", "dice"):
            raise ValueError(f"Metric function {fn} not implemented")
        c = conf[0] if fn == "iou" else conf.sum(axis=0)          # IoU: first sample only (metrics.py:45), Dice: whole batch
        per = []
        for k in range(1, nc):
            inter, true, pred = c[k, k], c[k, :].sum(), c[:, k].sum()
            per.append((inter + eps) / (true + pred - inter + eps) if fn == "iou" else (2 * inter + eps) / (true + pred + eps))
        vals.append(per)
        md["metric_" + fn] = float(np.mean(per))
    return vals, md


def get_metrics(outputs, targets, metric_fns, args):
    """Drop-in for src/metrics.py::get_metrics(outputs, targets, metric_fns, args) -> (metric_vals_per_class, metric_dict)."""
    conf = confusion_counts(outputs, targets, args.num_classes)
    return metrics_from_confusion(conf.cpu().numpy(), metric_fns)
