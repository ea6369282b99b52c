"""Single-frame 'HRNet' model type (models/__init__.py:38-46, scripts/train_toolpose_segmentation.py:162-163): the oracle
restatement and the HIP model against fixtures captured from the imported reference (tests/golden/make_golden.py::
run_single_case), and the hand-over of a single-frame state_dict to the multi-frame model's base_model."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import mfcnet_oracle as O
from golden_util import compare_logits, load_case, sample16

ATOL, GRAD_RTOL = 1e-3, 6e-2


def _inputs(cfg):
    frames, _, _, mask = O.synthetic_clip(cfg["name"], cfg["B"], 1, cfg["H"], cfg["W"], False, False)
    return frames[0], mask


def _state():
    return O.hashed_state(O.hrnet_table(48, 5, ""))


@pytest.mark.parametrize("name", ["hrnet_single_train", "hrnet_single_eval"])
def test_oracle_single_frame_matches_reference(name):
    cfg, z = load_case(name)
    net = O.SingleNet(_state(), 48, 5)
    x, mask = _inputs(cfg)
    if cfg["mode"] == "eval":
        net.eval()
        with torch.no_grad():
            compare_logits(z, net(x).numpy(), 2e-4)
        return
    net.train()
    opt = torch.optim.Adam(net.params(""), lr=1e-4)
    y = net(x)
    compare_logits(z, y.detach().numpy(), 2e-4)
    loss, parts = O.total_loss(y, mask, 5)
    for k in ("loss_nll", "loss_soft_jaccard", "loss_total"):
        assert abs(float(parts[k]) - float(z[k])) < 2e-5, k
    loss.backward()
    for key in [f for f in z.files if f.startswith("gradnorm/")]:
        p = key.split("/", 1)[1]
        ref = float(z[key])
        assert abs(float(net.sd[p].grad.double().norm()) - ref) <= 2e-3 * ref + 1e-7, p
    opt.step()
    for key in [f for f in z.files if f.startswith("paramsample/")]:
        p = key.split("/", 1)[1]
        np.testing.assert_allclose(sample16(net.sd[p]), z[key], rtol=0, atol=2.1e-4)
    for key in [f for f in z.files if f.startswith("bn_mean/")]:
        b = key.split("/", 1)[1]
        np.testing.assert_allclose(net.sd[b + ".running_mean"].numpy(), z[key], atol=1e-5)
        assert int(net.sd[b + ".num_batches_tracked"]) == int(z["bn_count/" + b])


def test_factory_and_state_dict_layout():
    import mfcnet_amd as mfc
    m = mfc.get_tooltip_segmentation_model(SimpleNamespace(model_type="HRNet", num_classes=5, pretrained=False), width=8)
    assert list(m.state_dict().keys()) == [t[0] for t in O.hrnet_table(8, 5, "")]      # HRNet keys, no prefix, reference order
    with pytest.raises(ValueError):
        mfc.get_tooltip_segmentation_model(SimpleNamespace(model_type="TernausNet11", num_classes=5, pretrained=False))
    # a single-frame state_dict seeds the multi-frame model's base_model (train_multiframe_detection.py:115-118)
    m.load_state_dict(O.hashed_state(O.hrnet_table(8, 5, "")))
    multi = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=8)
    multi.base_model.load_state_dict(m.state_dict())
    a, b = m.state_dict(), multi.state_dict()
    assert all(torch.equal(a[k], b["base_model." + k]) for k in a)
    with pytest.raises(mfc.MfcError):
        m(torch.zeros(1, 3, 64, 96))                                                     # CPU tensors: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hrnet_single_train", "hrnet_single_eval"])
def test_hip_single_frame_matches_reference(name):
    import mfcnet_amd as mfc
    cfg, z = load_case(name)
    m = mfc.HighResolutionNetHIP(num_classes=5, width=48, compute_dtype="fp32")
    m.load_state_dict(_state(), strict=True)
    m = m.cuda()
    x, mask = _inputs(cfg)
    if cfg["mode"] == "eval":
        m.eval()
        with torch.no_grad():
            compare_logits(z, m(x.cuda()).cpu().numpy(), ATOL)
        return
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    opt.zero_grad()
    y = m(x.cuda())
    compare_logits(z, y.detach().cpu().numpy(), ATOL)
    loss, acc = mfc.mfc_loss(y, mask.cuda())
    assert abs(float(loss) - float(z["loss_total"])) < 1e-4 and abs(float(acc.cpu()[26]) - float(z["loss_nll"])) < 1e-4
    loss.backward()
    named = dict(m.named_parameters())
    for key in [f for f in z.files if f.startswith("gradnorm/")]:
        p = key.split("/", 1)[1]
        ref = float(z[key])
        got = float(named[p].grad.double().norm())
        assert abs(got - ref) <= 2e-2 * ref + 1e-6, (p, got, ref)
    opt.step()
    for key in [f for f in z.files if f.startswith("paramsample/")]:
        p = key.split("/", 1)[1]
        np.testing.assert_allclose(sample16(named[p]), z[key], rtol=0, atol=2.1e-4, err_msg=p)
    st = m.state_dict()
    for key in [f for f in z.files if f.startswith("bn_mean/")]:
        b = key.split("/", 1)[1]
        np.testing.assert_allclose(st[b + ".running_mean"].cpu().numpy(), z[key], atol=5e-5, err_msg=b)
        assert int(st[b + ".num_batches_tracked"]) == int(z["bn_count/" + b]), b
