"""The CPU oracle must reproduce every golden vector captured from the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle ('Oracle')."""
import numpy as np
import pytest
import torch

from oracle import mfcnet_oracle as O
from golden_util import BIG_CASES, SMALL_CASES, case_inputs, case_state, compare_logits, load_case, sample16


def _run(name):
    cfg, z = load_case(name)
    net = O.Net(case_state(cfg), cfg["model_type"], 48, 5, cfg["T"], cfg["optflow"], cfg["depth"])
    frames, flows, depths, mask = case_inputs(cfg)
    if cfg["mode"] == "eval":
        net.eval()
        with torch.no_grad():
            y = net(frames, optflow=flows, depth=depths)
        compare_logits(z, y.numpy(), 2e-4)
        return
    net.train(base=(cfg["mode"] == "train"), head=True)
    opt = O.make_adam(net, 1e-4)
    opt.zero_grad()
    y = net(frames, optflow=flows, depth=depths)
    compare_logits(z, y.detach().numpy(), 2e-4)
    loss, parts = O.total_loss(y, mask, 5)
    for k in ("loss_nll", "loss_soft_jaccard", "loss_total"):
        assert abs(float(parts[k]) - float(z[k])) < 2e-5, k
    loss.backward()
    for key in [f for f in z.files if f.startswith("gradnorm/")]:
        p = key.split("/", 1)[1]
        g = net.sd[p].grad
        ref = float(z[key])
        assert abs(float(g.double().norm()) - ref) <= 2e-3 * ref + 1e-7, p
        np.testing.assert_allclose(sample16(g), z["gradsample/" + p], rtol=5e-3, atol=2e-3 * ref / np.sqrt(g.numel()) + 1e-8)
    opt.step()
    for key in [f for f in z.files if f.startswith("paramsample/")]:
        p = key.split("/", 1)[1]
        np.testing.assert_allclose(sample16(net.sd[p]), z[key], rtol=0, atol=2.1e-4 / cfg["T"] if p.startswith("base") else 2.1e-4)
    for key in [f for f in z.files if f.startswith("bn_mean/")]:
        b = key.split("/", 1)[1]
        np.testing.assert_allclose(net.sd[b + ".running_mean"].numpy(), z[key], atol=1e-5)
        np.testing.assert_allclose(net.sd[b + ".running_var"].numpy(), z["bn_var/" + b], rtol=1e-4, atol=1e-5)
        assert int(net.sd[b + ".num_batches_tracked"]) == int(z["bn_count/" + b])


@pytest.mark.parametrize("name", SMALL_CASES)
def test_oracle_matches_reference_small(name):
    _run(name)


@pytest.mark.parametrize("name", BIG_CASES)
def test_oracle_matches_reference_480x640(name):
    _run(name)


def test_state_table_counts():
    t = O.mfcnet_table("HRNetMulti-Large", 48, 5, 3, False, False)
    assert len(t) == 1858                                   # 931 params + 927 buffers
    n_par = sum(int(np.prod(s)) for _, s, k in t if k in ("conv_w", "conv_b", "bn_gamma", "bn_beta"))
    assert n_par == 65_883_065 or abs(n_par - 65.88e6) < 0.02e6
    assert sum(1 for _, _, k in t if k == "conv_w") == 311


def test_hash_generator_is_stable():
    v = O.hash_uniform("base_model.conv1.weight", 4)
    assert v.dtype == np.float32 and np.all((v >= 0) & (v < 1))
    np.testing.assert_array_equal(v, O.hash_uniform("base_model.conv1.weight", 8)[:4])
