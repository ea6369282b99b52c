"""Fused Adam over the model's flat parameter arena.

Same update rule and the same two learning-rate groups as the reference's
`torch.optim.Adam([{base_model params, lr/T or lr/(100T)}, {multiframe_net params, lr}])`
(scripts/train_multiframe_detection.py:128-151; default betas/eps, no weight decay), but one kernel
launch per group over contiguous fp32 segments instead of 931 per-tensor updates.
`torch.optim.Adam(model.parameters())` also works on the model (parameters are views of the arena).
"""
from __future__ import annotations

import torch

from . import _lib as L


class FlatAdam:
    def __init__(self, model, lr=1e-4, load_wts_base_model=False, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        T = model.num_frames
        base_lr = lr / (100.0 * T) if load_wts_base_model else lr / T
        self.lrs = {"base_model": base_lr, "multiframe_net": lr}
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.m = torch.zeros_like(model._P)
        self.v = torch.zeros_like(model._P)
        self._arena = model._P

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def step(self, grad_scale: float = 1.0):
        mdl = self.model
        if mdl._P is not self._arena:
            raise L.MfcError("model was moved after the optimizer was built; rebuild FlatAdam")
        self.step_count += 1
        st = L.stream_ptr()
        for name, (a, b) in mdl.flat_segments().items():
            n = b - a
            L.check(L.lib.mfc_adam_step(mdl._P.data_ptr() + 4 * a, mdl._G.data_ptr() + 4 * a, self.m.data_ptr() + 4 * a,
                                        self.v.data_ptr() + 4 * a, n, self.lrs[name], self.betas[0], self.betas[1], self.eps,
                                        self.step_count, grad_scale, st), "mfc_adam_step")

    # ---- checkpointing (utils/model_utils.py:6-12 stores optimizer.state_dict() next to the model's) ----
    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.m.detach().cpu().clone(), "exp_avg_sq": self.v.detach().cpu().clone(),
                "lrs": dict(self.lrs), "betas": tuple(self.betas), "eps": self.eps, "layout": "flat-arena-v1"}

    def load_state_dict(self, sd):
        if sd.get("layout") != "flat-arena-v1" or sd["exp_avg"].numel() != self.m.numel():
            raise L.MfcError("optimizer state does not belong to a FlatAdam of this model")
        self.step_count = int(sd["step"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.lrs.update(sd["lrs"])
        self.betas, self.eps = tuple(sd["betas"]), float(sd["eps"])
