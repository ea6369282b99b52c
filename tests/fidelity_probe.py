"""Where does the 16-bit training mode lose gradient fidelity?  (CPU only, not collected by pytest.)

The CPU oracle (fp32 graph pinned by the reference's goldens) is run with storage rounding switched on separately for the FORWARD
tensors (what `Net(store_dtype=...)` emulates) and for the GRADIENT tensors (a straight-through hook at the same storage points:
every convolution input / output and every residual / fuse / transition sum -- the tensors the HIP plan materialises), and the
resulting parameter gradients are compared with the all-fp32 ones:

    python tests/fidelity_probe.py [--width 32] [--height 128] [--width-px 192] [--batch 2]

prints, per (forward type, gradient type), the cosine and relative L2 error of the whole gradient, of the temporal head's
gradient and of a few per-stage groups.  Reference step: src/engine.py:54-71.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mfcnet_oracle as O  # noqa: E402


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fdt, bdt):
        ctx.bdt = bdt
        return x if fdt is None else x.to(fdt).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.bdt is None else g.to(ctx.bdt).to(torch.float32)), None, None


class ProbeNet(O.Net):
    """O.Net whose storage points round the forward value to `fwd_dtype` and the gradient passing through them to `bwd_dtype`;
    weights are rounded in the forward only (weight gradients and the optimizer are fp32 in every mode of the product)."""

    def __init__(self, *a, fwd_dtype=None, bwd_dtype=None, skip_residual_bwd=False, **k):
        super().__init__(*a, **k)
        self.fdt, self.bdt = fwd_dtype, bwd_dtype
        self.store_dtype = fwd_dtype if fwd_dtype is not None else (torch.float32 if bwd_dtype is not None else None)
        self.skip_residual_bwd = skip_residual_bwd

    def _q(self, t):
        if self.fdt is None and self.bdt is None:
            return t
        if not t.requires_grad or t.is_leaf:            # parameters / inputs: forward rounding only
            return t if self.fdt is None else t.to(self.fdt).to(torch.float32)
        return _Round.apply(t, self.fdt, self.bdt)

    def _bn(self, x, name, training):
        """training-mode BatchNorm with the product's rounding points AND a differentiable mean / variance (O.Net's emulation branch is
        forward-only: it detaches the statistics): statistics from the fp32 convolution result, normalisation of the rounded tensor,
        the gradient w.r.t. the convolution output rounded once (what mfc_bnbwd_apply stores)."""
        if not training or (self.fdt is None and self.bdt is None):
            return O.Net._bn(self, x, name, training)
        s = self.sd
        s[name + ".num_batches_tracked"] += 1
        x = _Round.apply(x, None, self.bdt)                        # d(conv output): one rounding
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = x.numel() // x.shape[1]
            s[name + ".running_mean"].mul_(1 - O.BN_MOMENTUM).add_(O.BN_MOMENTUM * mean)
            s[name + ".running_var"].mul_(1 - O.BN_MOMENTUM).add_(O.BN_MOMENTUM * var * n / max(n - 1, 1))
        scale = s[name + ".weight"] / torch.sqrt(var + O.BN_EPS)
        shift = s[name + ".bias"] - mean * scale
        xq = x if self.fdt is None else x + (x.to(self.fdt).to(torch.float32) - x).detach()      # stored tensor, straight-through
        return xq * scale[None, :, None, None] + shift[None, :, None, None]


def grads(net, frames, mask, scale=1.0):
    net.zero_grad()
    out = net(frames)
    loss, _ = O.total_loss(out, mask, 5)
    (loss * scale).backward()
    return {n: net.sd[n].grad.detach().clone() / scale for n in net.param_names}, float(loss)


def compare(ref, got, names):
    a = torch.cat([ref[n].flatten() for n in names]).double()
    b = torch.cat([got[n].flatten() for n in names]).double()
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
    rel = float((a - b).norm() / (a.norm() + 1e-300))
    return cos, rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=32)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width-px", type=int, default=192)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--damp", type=float, default=1.0, help="scale of the LAST BatchNorm weight of every residual block (bn2 of BasicBlocks, bn3 of "
                    "Bottlenecks): 1 = the hashed weights as they are; < 1 makes the residual branches small against the skip path, as in a trained network")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    T, B, H, W = args.frames, args.batch, args.height, args.width_px
    sd = O.hashed_state(O.mfcnet_table("HRNetMulti-Large", args.width, 5, T, False, False))
    if args.damp != 1.0:
        for n in sd:
            if n.endswith((".bn2.weight", ".bn3.weight")) and (".branches." in n or ".layer1." in n):
                sd[n] = sd[n] * args.damp
    frames, _, _, mask = O.synthetic_clip("fidelity", B, T, H, W, False, False)
    bf, hf = torch.bfloat16, torch.float16
    S = float(2 ** round(__import__("math").log2(max(B * H * W / 16.0, 1.0))))
    modes = [("fp32", "fp32", None, None, 1.0), ("bf16", "fp32", bf, None, 1.0), ("fp32", "bf16", None, bf, 1.0), ("bf16", "bf16", bf, bf, 1.0),
             ("fp16", "fp16", hf, hf, S), ("fp16", "bf16", hf, bf, 1.0), ("fp16", "fp32", hf, None, 1.0), ("fp32", "fp16", None, hf, S)]
    ref = None
    groups = None
    print(f"W{args.width} B={B} T={T} {H}x{W}; loss scale of the fp16-gradient rows {S:g}")
    for fname, bname, fdt, bdt, scale in modes:
        net = ProbeNet(sd, "HRNetMulti-Large", args.width, 5, T, fwd_dtype=fdt, bwd_dtype=bdt).train()
        t0 = time.time()
        g, loss = grads(net, frames, mask, scale)
        if ref is None:
            ref = g
            names = list(g)
            groups = {"all": names, "head": [n for n in names if n.startswith("multiframe_net.")],
                      "last_layer": [n for n in names if ".last_layer." in n],
                      "stage4": [n for n in names if ".stage4." in n], "stage3": [n for n in names if ".stage3." in n],
                      "stage2": [n for n in names if ".stage2." in n], "layer1": [n for n in names if ".layer1." in n],
                      "stem": [n for n in names if n.endswith(("base_model.conv1.weight", "base_model.conv2.weight"))]}
        line = f"forward {fname:5s} gradients {bname:5s} loss {loss:.5f} ({time.time() - t0:4.1f}s)"
        for k, ns in groups.items():
            cos, rel = compare(ref, g, ns)
            line += f" | {k} cos {cos:.4f} rel {rel:.3f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
