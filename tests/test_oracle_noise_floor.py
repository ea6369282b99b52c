"""Measures the reference path's own fp32 gradient noise floor with the CPU oracle (no GPU needed):
a 1e-7 RELATIVE perturbation of the input frames -- smaller than one fp32 ulp -- moves train-mode
gradients by percents, because BatchNorm over few pixels and ReLU sign flips amplify rounding through
~300 layers.  This number is the basis of GRAD_RTOL in tests/test_gpu_model.py."""
import torch

from oracle import mfcnet_oracle as O
from golden_util import case_inputs, case_state, load_case


def _grads(cfg, frames, mask):
    net = O.Net(case_state(cfg), cfg["model_type"], 48, 5, cfg["T"])
    net.train()
    y = net(frames)
    loss, _ = O.total_loss(y, mask, 5)
    loss.backward()
    return y.detach(), net.sd["base_model.conv1.weight"].grad.clone(), net.sd["multiframe_net.multiframe_net.0.weight"].grad.clone()


def test_reference_gradient_noise_floor():
    cfg, _ = load_case("large_rgb_train")
    frames, _, _, mask = case_inputs(cfg)
    y0, g0, h0 = _grads(cfg, frames, mask)
    gen = torch.Generator().manual_seed(0)
    pert = [f * (1 + 1e-7 * torch.randn(f.shape, generator=gen)) for f in frames]
    y1, g1, h1 = _grads(cfg, pert, mask)
    stem = float((g1 - g0).norm() / g0.norm())
    head = float((h1 - h0).norm() / h0.norm())
    print(f"noise floor: logits {float((y1 - y0).abs().max()):.2e}, conv1 grad {stem:.2e}, head grad {head:.2e}")
    assert float((y1 - y0).abs().max()) < 1e-3          # the forward stays inside the 1e-3 heat-map budget
    assert 2e-3 < stem < 6e-2                           # ... while gradients sit at the percent level
