"""Checkpoint helpers with the reference's on-disk format (utils/model_utils.py:6-39).

A checkpoint is `torch.save({'model': state_dict, 'optimizer': optimizer.state_dict(), 'epoch': epoch})` named
`model_{epoch:03d}.pth`; loading strips the `module.` prefix nn.DataParallel adds and is strict.  The HIP model's
`state_dict()` has the reference's key names in the reference's order (tests/test_abi_and_host.py), so files written by
either side load into the other; `model.base_model.load_state_dict(ckpt['model'])` seeds the per-frame HRNet from a
single-frame checkpoint exactly as scripts/train_multiframe_detection.py:115-118 does.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch


def save_model(model, ckpt_dir, optimizer=None, epoch=None):
    """utils/model_utils.py:6-12 (like the reference, `optimizer` and `epoch` are needed despite their defaults)."""
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch},
               os.path.join(ckpt_dir, "model_{:03d}.pth".format(epoch)))


def strip_module_prefix(state):
    """`module.` prefix of nn.DataParallel checkpoints removed (utils/model_utils.py:33-37)."""
    out = OrderedDict()
    for key, value in state.items():
        out[key.replace("module.", "") if key.startswith("module.") else key] = value
    return out


def load_model_weights(model, loadpath, model_type):
    """utils/model_utils.py:14-39: returns (model, epoch, 1) after a strict load, or (model, 1, 0) without a path;
    NameError if the file is missing.  (`model_type` only selects the non-strict DeepLab/FCN branch in the reference;
    the HRNetMulti types load strictly.)"""
    if loadpath is None:
        return model, 1, 0
    if not os.path.exists(loadpath):
        raise NameError("File {} does not exist".format(loadpath))
    state = torch.load(loadpath, map_location="cpu")
    model.load_state_dict(strip_module_prefix(state["model"]))
    return model, state["epoch"], 1


def load_base_model_weights(model, loadpath):
    """scripts/train_multiframe_detection.py:115-118: a single-frame HRNet checkpoint seeds `model.base_model`."""
    state = torch.load(loadpath, map_location="cpu")
    model.base_model.load_state_dict(strip_module_prefix(state["model"]))
    return model
