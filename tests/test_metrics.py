"""Validation metrics (src/metrics.py): the oracle restatement against fixtures produced by the imported reference
(tests/golden/make_golden.py::run_metric_case), and the device kernel against both."""
import glob
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mfcnet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "metrics_*.npz")))


def load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    B, nc, H, W, seed = (int(v) for v in z["case"])
    logits, mask = O.metric_case_inputs(B, nc, H, W, seed, float(z["scale"]))
    return z, nc, F.log_softmax(logits, dim=1), mask


def test_fixtures_present():
    assert len(CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_oracle_metrics_match_reference(name):
    z, nc, out, mask = load(name)
    vals, md = O.get_metrics(out, mask, ["iou", "dice"], nc)
    assert np.allclose(vals[0], z["iou"], rtol=1e-12, atol=0) and np.allclose(vals[1], z["dice"], rtol=1e-12, atol=0)
    assert abs(md["metric_iou"] - float(z["metric_iou"])) < 1e-12 and abs(md["metric_dice"] - float(z["metric_dice"])) < 1e-12
    conf = O.confusion_per_sample(out.numpy().argmax(axis=1), mask.numpy(), nc)
    assert (conf.sum(axis=0) == z["confusion"]).all()          # metrics.py:60-67 sums over the batch
    with pytest.raises(NotImplementedError):
        O.metrics_from_confusion(conf, ["jaccard"])
    with pytest.raises(ValueError):
        O.metrics_from_confusion(conf, ["f1"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_metrics_match_reference(name):
    import mfcnet_amd as mfc
    z, nc, out, mask = load(name)
    conf = mfc.confusion_counts(out.cuda(), mask.cuda(), nc).cpu().numpy()
    ref = O.confusion_per_sample(out.numpy().argmax(axis=1), mask.numpy(), nc)
    assert (conf == ref).all()                                   # integer counts: bit-exact, ties resolved like numpy.argmax
    assert (conf.sum(axis=0) == z["confusion"]).all()
    vals, md = mfc.get_metrics(out.cuda(), mask.cuda(), ["iou", "dice"], SimpleNamespace(num_classes=nc))
    assert np.allclose(vals[0], z["iou"], rtol=1e-12, atol=0) and np.allclose(vals[1], z["dice"], rtol=1e-12, atol=0)
    assert abs(md["metric_iou"] - float(z["metric_iou"])) < 1e-12
    with pytest.raises(ValueError):
        mfc.get_metrics(out.cuda(), mask.cuda(), ["f1"], SimpleNamespace(num_classes=nc))


@pytest.mark.gpu
def test_device_metrics_full_size():
    """480x640, B=8: counts add up to the pixel count and agree with the oracle."""
    import mfcnet_amd as mfc
    logits, mask = O.metric_case_inputs(8, 5, 480, 640, 105, 2.0)
    conf = mfc.confusion_counts(logits.cuda(), mask.cuda(), 5).cpu().numpy()
    assert (conf.sum(axis=(1, 2)) == 480 * 640).all()
    assert (conf == O.confusion_per_sample(logits.numpy().argmax(axis=1), mask.numpy(), 5)).all()
