"""Checkpoint format of the reference (utils/model_utils.py:6-39; scripts/train_multiframe_detection.py:115-121): host logic only."""
import os
from collections import OrderedDict

import pytest
import torch

import mfcnet_amd as mfc
from oracle import mfcnet_oracle as O

W = 8          # a narrow HRNet keeps the files small; the key set does not depend on the width


def make(T=3):
    return mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False, width=W)


def hashed(T=3):
    return O.hashed_state(O.mfcnet_table("HRNetMulti-Large", W, 5, T, False, False))


def test_save_and_load_roundtrip_with_dataparallel_prefix(tmp_path):
    m = make()
    m.load_state_dict(hashed())
    opt = mfc.FlatAdam(m, lr=1e-4)
    opt.step_count = 3
    mfc.save_model(m, str(tmp_path), optimizer=opt, epoch=7)
    path = os.path.join(str(tmp_path), "model_007.pth")
    assert os.path.exists(path)                                            # "model_{:03d}.pth" (model_utils.py:11)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch"} and ck["epoch"] == 7
    # a checkpoint written under nn.DataParallel carries the `module.` prefix: rewrite the file that way
    ck["model"] = OrderedDict(("module." + k, v) for k, v in ck["model"].items())
    torch.save(ck, path)
    m2 = make()
    m2, epoch, flag = mfc.load_model_weights(m2, path, "HRNetMulti-Large")
    assert (epoch, flag) == (7, 1)
    a, b = m.state_dict(), m2.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    opt2 = mfc.FlatAdam(m2, lr=1e-4)
    opt2.load_state_dict(ck["optimizer"])
    assert opt2.step_count == 3 and opt2.lrs == opt.lrs


def test_no_path_and_missing_file(tmp_path):
    m = make()
    assert mfc.load_model_weights(m, None, "HRNetMulti-Large")[1:] == (1, 0)        # model_utils.py:38-39
    with pytest.raises(NameError):
        mfc.load_model_weights(m, os.path.join(str(tmp_path), "nope.pth"), "HRNetMulti-Large")


def test_single_frame_checkpoint_seeds_base_model(tmp_path):
    """train_multiframe_detection.py:115-118: model.base_model.load_state_dict(state['model']) with HRNet keys (no prefix)."""
    sd = hashed()
    base = OrderedDict((k[len("base_model."):], v) for k, v in sd.items() if k.startswith("base_model."))
    path = os.path.join(str(tmp_path), "hrnet_single.pth")
    torch.save({"model": base, "optimizer": {}, "epoch": 11}, path)
    m = make()
    before_head = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("multiframe_net.")}
    mfc.load_base_model_weights(m, path)
    after = m.state_dict()
    assert all(torch.equal(after["base_model." + k], v) for k, v in base.items())
    assert all(torch.equal(after[k], v) for k, v in before_head.items())              # the temporal head is untouched
    # the flat arena the kernels read holds the loaded values (parameters are views of it)
    k0 = "base_model.conv1.weight"
    off = m._poff[k0]
    assert torch.equal(m._P[off:off + base["conv1.weight"].numel()].view_as(base["conv1.weight"]), base["conv1.weight"])
    del base["conv1.weight"]
    torch.save({"model": base, "optimizer": {}, "epoch": 11}, path)
    with pytest.raises(RuntimeError):                                                  # strict, as nn.Module.load_state_dict
        mfc.load_base_model_weights(make(), path)
