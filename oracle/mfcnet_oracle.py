"""CPU oracle for the MFCNet multi-frame forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`mfcnet-tracker_amd/`) may import this file; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
there only as the checker / the CPU baseline -- never as the thing shipped.

What it is: a from-scratch *functional* restatement (plain torch CPU ops, fp32,
NCHW) of the reference path

  models/multiframe_model.py:408-471   HRNetMultiBasic / HRNetMultiLarge
  models/multiframe_model.py:14-35     MultiFrameNetBase channel rule
  models/multiframe_model.py:51-185    MultiFrameNetBasic (+ optical-flow warp)
  models/multiframe_model.py:187-205   MultiFrameNetLarge
  models/hrnet.py:45-476               HighResolutionNet (HRNet, hard-coded W48)
  src/loss.py:6-63                     log-prob -> weighted NLL + soft-Jaccard
  src/engine.py:54-71                  the training step body

The network is described by a parameter dict keyed with the reference's own
`state_dict` names (so a reference checkpoint loads 1:1) and evaluated by
straight-line functions -- there is no nn.Module tree here.  `width` is a
parameter (48 = the reference's hard-coded widths, models/hrnet.py:297-330;
32 = BASELINE.json's "w32" label, for which the reference has no model).

Parity status: PINNED.  `tests/golden/*.npz` were produced by importing the
reference's own `models/hrnet.py`, `models/multiframe_model.py` and
`src/loss.py` in the build container (`tests/golden/make_golden.py`) and this
file reproduces every one of them (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.SyncBatchNorm / nn.BatchNorm2d default (hrnet.py:31, multiframe_model.py:193)
BN_MOMENTUM = 0.1      # hrnet.py:34
WARP_GRID_HW = (576, 720)   # multiframe_model.py:179


# --------------------------------------------------------------------------
# Deterministic "weights by key hash" generator.
# The same generator runs on the GPU box, so only seeds travel, not weights.
# --------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(key: str, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """n float32 values in [lo, hi), a pure function of (key, index)."""
    seed = np.uint64(zlib.crc32(key.encode("utf-8"))) << np.uint64(32)
    with np.errstate(over="ignore"):
        bits = _splitmix64(seed + np.arange(n, dtype=np.uint64))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(np.float32)


def hash_normal(key: str, n: int) -> np.ndarray:
    """n float32 ~N(0,1) values (Box-Muller on two hashed uniforms)."""
    u1 = hash_uniform(key + "#a", n).astype(np.float64)
    u2 = hash_uniform(key + "#b", n).astype(np.float64)
    u1 = np.minimum(u1, 1.0 - 2.0 ** -24)          # (the float32 rounding of hash_uniform can return exactly 1.0: log(0))
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * math.pi * u2)).astype(np.float32)


# --------------------------------------------------------------------------
# Parameter / buffer tables with the reference's state_dict names and order.
# --------------------------------------------------------------------------
def _bn_entries(name: str, c: int):
    return [(name + ".weight", (c,), "bn_gamma"), (name + ".bias", (c,), "bn_beta"),
            (name + ".running_mean", (c,), "bn_mean"), (name + ".running_var", (c,), "bn_var"),
            (name + ".num_batches_tracked", (), "bn_count")]


def _conv_entry(name: str, cout: int, cin: int, k: int, bias: bool = False):
    out = [(name + ".weight", (cout, cin, k, k), "conv_w")]
    if bias:
        out.append((name + ".bias", (cout,), "conv_b"))
    return out


def hrnet_table(width: int = 48, num_classes: int = 5, prefix: str = "") -> List[Tuple[str, tuple, str]]:
    """state_dict layout of HighResolutionNet (hrnet.py:271-351), generalised over `width`."""
    W = [width, 2 * width, 4 * width, 8 * width]
    t: List[Tuple[str, tuple, str]] = []
    p = prefix
    # stem, hrnet.py:280-284
    t += _conv_entry(p + "conv1", 64, 3, 3) + _bn_entries(p + "bn1", 64)
    t += _conv_entry(p + "conv2", 64, 64, 3) + _bn_entries(p + "bn2", 64)
    # layer1: 4 Bottlenecks 64 -> 256, hrnet.py:288-293, 77-115, 391-405
    inpl = 64
    for b in range(4):
        q = f"{p}layer1.{b}."
        t += _conv_entry(q + "conv1", 64, inpl, 1) + _bn_entries(q + "bn1", 64)
        t += _conv_entry(q + "conv2", 64, 64, 3) + _bn_entries(q + "bn2", 64)
        t += _conv_entry(q + "conv3", 256, 64, 1) + _bn_entries(q + "bn3", 256)
        if b == 0:
            t += _conv_entry(q + "downsample.0", 256, inpl, 1) + _bn_entries(q + "downsample.1", 256)
        inpl = 256

    def transition(name, pre, cur):              # hrnet.py:353-389
        for i, c in enumerate(cur):
            if i < len(pre):
                if c != pre[i]:
                    t.extend(_conv_entry(f"{p}{name}.{i}.0", c, pre[i], 3) + _bn_entries(f"{p}{name}.{i}.1", c))
            else:
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = c if j == i - len(pre) else cin
                    t.extend(_conv_entry(f"{p}{name}.{i}.{j}.0", cout, cin, 3) + _bn_entries(f"{p}{name}.{i}.{j}.1", cout))

    def stage(name, n_modules, chans):           # hrnet.py:407-423, 118-233
        nb = len(chans)
        for m in range(n_modules):
            q = f"{p}{name}.{m}."
            for i in range(nb):
                for b in range(4):
                    r = f"{q}branches.{i}.{b}."
                    t.extend(_conv_entry(r + "conv1", chans[i], chans[i], 3) + _bn_entries(r + "bn1", chans[i]))
                    t.extend(_conv_entry(r + "conv2", chans[i], chans[i], 3) + _bn_entries(r + "bn2", chans[i]))
            for i in range(nb):
                for j in range(nb):
                    r = f"{q}fuse_layers.{i}.{j}."
                    if j > i:
                        t.extend(_conv_entry(r + "0", chans[i], chans[j], 1) + _bn_entries(r + "1", chans[i]))
                    elif j < i:
                        for k in range(i - j):
                            cout = chans[i] if k == i - j - 1 else chans[j]
                            t.extend(_conv_entry(f"{r}{k}.0", cout, chans[j], 3) + _bn_entries(f"{r}{k}.1", cout))

    transition("transition1", [256], W[:2])
    stage("stage2", 1, W[:2])
    transition("transition2", W[:2], W[:3])
    stage("stage3", 4, W[:3])
    transition("transition3", W[:3], W[:4])
    stage("stage4", 3, W[:4])
    last = sum(W)                                 # hrnet.py:333-351
    t += _conv_entry(p + "last_layer.0", last, last, 1, bias=True) + _bn_entries(p + "last_layer.1", last)
    t += _conv_entry(p + "last_layer.3", num_classes, last, 1, bias=True)
    return t


def head_in_channels(model_type: str, num_classes: int, T: int, optflow: bool, depth: bool) -> int:
    """multiframe_model.py:23-32 (Large) and :54-56 (Basic: flow channels are consumed by the warp)."""
    c = T * num_classes
    if "Large" in model_type and optflow:
        c += 2 * (T - 1)
    if depth:
        c += T
    return c


def head_table(model_type: str, num_classes: int, T: int, optflow: bool, depth: bool,
               prefix: str = "multiframe_net.") -> List[Tuple[str, tuple, str]]:
    """state_dict layout of MultiFrameNetLarge / MultiFrameNetBasic (multiframe_model.py:61-82,191-202)."""
    cin = head_in_channels(model_type, num_classes, T, optflow, depth)
    mid = T * num_classes
    q = prefix + "multiframe_net."
    t: List[Tuple[str, tuple, str]] = []
    if "Basic" in model_type:
        t.append((prefix + "grid", (1, 2) + WARP_GRID_HW, "grid"))   # registered before... after the Sequential
    t2 = _conv_entry(q + "0", mid, cin, 11) + _bn_entries(q + "1", mid)
    t2 += _conv_entry(q + "3", mid, mid, 3) + _bn_entries(q + "4", mid)
    t2 += _conv_entry(q + "6", mid, mid, 3) + _bn_entries(q + "7", mid)
    t2 += _conv_entry(q + "9", num_classes, mid, 1)
    # nn.Module.state_dict emits own buffers (grid) before child modules.
    return t + t2


def mfcnet_table(model_type: str = "HRNetMulti-Large", width: int = 48, num_classes: int = 5, T: int = 3,
                 optflow: bool = False, depth: bool = False):
    return hrnet_table(width, num_classes, "base_model.") + head_table(model_type, num_classes, T, optflow, depth)


def warp_grid() -> torch.Tensor:
    """multiframe_model.py:172-185: mesh grid normalised for 576x720, [1,2,576,720]."""
    H, W = WARP_GRID_HW
    y, x = torch.meshgrid(torch.arange(0, H), torch.arange(0, W), indexing="ij")
    gy = 2.0 * y / (H - 1) - 1.0
    gx = 2.0 * x / (W - 1) - 1.0
    return torch.stack((gx, gy), dim=0).float().unsqueeze(0)


def hashed_state(table, salt: str = "") -> "OrderedDict[str, torch.Tensor]":
    """Fill a state_dict table deterministically (SURVEY.md 8(c) 'weights-by-key-hash')."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape, kind in table:
        n = int(np.prod(shape)) if len(shape) else 1
        k = salt + name
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / math.sqrt(fan_in)
            v = hash_uniform(k, n, -b, b)
        elif kind == "conv_b":
            v = hash_uniform(k, n, -0.1, 0.1)
        elif kind == "bn_gamma":
            v = hash_uniform(k, n, 0.5, 1.5)
        elif kind == "bn_beta":
            v = hash_uniform(k, n, -0.2, 0.2)
        elif kind == "bn_mean":
            v = hash_uniform(k, n, -0.2, 0.2)
        elif kind == "bn_var":
            v = hash_uniform(k, n, 0.5, 1.5)
        elif kind == "bn_count":
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        elif kind == "grid":
            sd[name] = warp_grid()
            continue
        else:
            raise KeyError(kind)
        sd[name] = torch.from_numpy(v.reshape(shape).copy())
    return sd


def synthetic_clip(case: str, B: int, T: int, H: int, W: int, optflow: bool, depth: bool, num_classes: int = 5):
    """Hash-generated inputs: frames ~N(0,1), depth ~U[0,1), flow ~3*N(0,1) px, mask in 0..C-1."""
    frames = [torch.from_numpy(hash_normal(f"{case}/frame{i}", B * 3 * H * W).reshape(B, 3, H, W)) for i in range(T)]
    flows = None
    depths = None
    if optflow:
        flows = [torch.from_numpy(3.0 * hash_normal(f"{case}/flow{i}", B * 2 * H * W).reshape(B, 2, H, W))
                 for i in range(T - 1)]
    if depth:
        depths = [torch.from_numpy(hash_uniform(f"{case}/depth{i}", B * H * W).reshape(B, 1, H, W)) for i in range(T)]
    mask = torch.from_numpy((hash_uniform(f"{case}/mask", B * H * W) * num_classes).astype(np.int64)
                            .clip(0, num_classes - 1).reshape(B, H, W))
    return frames, flows, depths, mask


# --------------------------------------------------------------------------
# Functional network
# --------------------------------------------------------------------------
class Net:
    """Holds the state dict (params require grad) and evaluates the reference graph functionally."""

    def __init__(self, sd: Dict[str, torch.Tensor], model_type: str = "HRNetMulti-Large", width: int = 48,
                 num_classes: int = 5, T: int = 3, optflow: bool = False, depth: bool = False, store_dtype=None):
        """store_dtype=torch.bfloat16: FORWARD-ONLY emulation of the MI355X throughput mode's storage rounding on top of the
        pinned fp32 graph -- every tensor the HIP plan materialises in bf16 is rounded to bf16 where the plan rounds it (conv
        operands: input and weights; conv outputs, after the BatchNorm statistics were taken from the fp32 accumulators; the
        outputs of the residual / fuse-layer / transition / concat sums), everything else (accumulation, BatchNorm
        coefficients, bilinear weights, biases) stays fp32.  It is the checker for the bf16 kernels' arithmetic: against it
        only summation order differs, not the 2^-8 storage rounding itself."""
        self.store_dtype = store_dtype
        self.model_type, self.width, self.nc, self.T = model_type, width, num_classes, T
        self.optflow, self.depth = optflow, depth
        self.table = mfcnet_table(model_type, width, num_classes, T, optflow, depth)
        self.sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.param_names: List[str] = []
        for name, shape, kind in self.table:
            v = sd[name].detach().clone()
            assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
            if kind in ("conv_w", "conv_b", "bn_gamma", "bn_beta"):
                v.requires_grad_(True)
                self.param_names.append(name)
            self.sd[name] = v
        self.base_training = True
        self.head_training = True

    # mode toggles, engine.py:22-26
    def train(self, base: bool = True, head: bool = True):
        self.base_training, self.head_training = base, head
        return self

    def eval(self):
        return self.train(False, False)

    def params(self, prefix: str = "") -> List[torch.Tensor]:
        return [self.sd[n] for n in self.param_names if n.startswith(prefix)]

    def zero_grad(self):
        for n in self.param_names:
            self.sd[n].grad = None

    # -- primitive layers ---------------------------------------------------
    def _q(self, t):
        """storage rounding of the emulated throughput mode (identity in the pinned fp32 mode)"""
        sdt = getattr(self, "store_dtype", None)
        return t if sdt is None else t.to(sdt).to(torch.float32)

    def _conv(self, x, name, stride=1, pad=0, raw=False):
        y = F.conv2d(self._q(x), self._q(self.sd[name + ".weight"]), self.sd.get(name + ".bias"), stride=stride, padding=pad)
        return y if raw else self._q(y)

    def _bn(self, x, name, training):
        s = self.sd
        if training:
            s[name + ".num_batches_tracked"] += 1     # torch BatchNorm bookkeeping; momentum is fixed (0.1)
        if training and getattr(self, "store_dtype", None) is not None:
            # emulated throughput mode: the statistics come from the fp32 accumulators, the normalisation reads the rounded tensor
            with torch.no_grad():
                mean = x.mean(dim=(0, 2, 3))
                var = x.var(dim=(0, 2, 3), unbiased=False)
                n = x.numel() // x.shape[1]
                s[name + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
                s[name + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
            scale = s[name + ".weight"] / torch.sqrt(var + BN_EPS)
            shift = s[name + ".bias"] - mean * scale
            return self._q(x) * scale[None, :, None, None] + shift[None, :, None, None]
        return F.batch_norm(self._q(x), s[name + ".running_mean"], s[name + ".running_var"], s[name + ".weight"],
                            s[name + ".bias"], training, BN_MOMENTUM, BN_EPS)

    def _cbr(self, x, conv, bn, k, stride, relu, training):
        y = self._bn(self._conv(x, conv, stride, k // 2, raw=True), bn, training)
        return F.relu(y) if relu else y

    # -- HRNet, hrnet.py:425-476 ---------------------------------------------
    def hrnet(self, x, p="base_model."):
        tr = self.base_training
        W = [self.width * m for m in (1, 2, 4, 8)]
        x = self._cbr(x, p + "conv1", p + "bn1", 3, 2, True, tr)       # hrnet.py:426-428
        x = self._cbr(x, p + "conv2", p + "bn2", 3, 2, True, tr)       # :429-431
        for b in range(4):                                              # layer1 Bottlenecks, :95-115
            q = f"{p}layer1.{b}."
            res = x
            o = self._cbr(x, q + "conv1", q + "bn1", 1, 1, True, tr)
            o = self._cbr(o, q + "conv2", q + "bn2", 3, 1, True, tr)
            o = self._cbr(o, q + "conv3", q + "bn3", 1, 1, False, tr)
            if b == 0:
                res = self._cbr(x, q + "downsample.0", q + "downsample.1", 1, 1, False, tr)
            x = self._q(F.relu(o + res))

        def transition(name, ys, pre, cur):                            # :434-461, 353-389
            out = []
            for i, c in enumerate(cur):
                if i < len(pre):
                    if c != pre[i]:
                        out.append(self._q(self._cbr(ys[i], f"{p}{name}.{i}.0", f"{p}{name}.{i}.1", 3, 1, True, tr)))
                    else:
                        out.append(ys[i])
                else:
                    v = ys[-1]
                    for j in range(i + 1 - len(pre)):
                        v = self._cbr(v, f"{p}{name}.{i}.{j}.0", f"{p}{name}.{i}.{j}.1", 3, 2, True, tr)
                    out.append(self._q(v))
            return out

        def module(q, xs):                                              # HighResolutionModule.forward :238-262
            nb = len(xs)
            xs = list(xs)
            for i in range(nb):                                         # branches of 4 BasicBlocks :58-74
                v = xs[i]
                for b in range(4):
                    r = f"{q}branches.{i}.{b}."
                    o = self._cbr(v, r + "conv1", r + "bn1", 3, 1, True, tr)
                    o = self._cbr(o, r + "conv2", r + "bn2", 3, 1, False, tr)
                    v = self._q(F.relu(o + v))
                xs[i] = v
            outs = []
            for i in range(nb):
                y = None
                for j in range(nb):
                    r = f"{q}fuse_layers.{i}.{j}."
                    if j == i:
                        term = xs[j]
                    elif j > i:                                         # 1x1 + BN, bilinear up (align_corners=None==False) :251-257
                        term = self._cbr(xs[j], r + "0", r + "1", 1, 1, False, tr)
                        term = F.interpolate(term, size=xs[i].shape[-2:], mode="bilinear", align_corners=False)
                    else:                                               # chain of 3x3 s2 (+ReLU on all but last) :211-230
                        term = xs[j]
                        for k in range(i - j):
                            term = self._cbr(term, f"{r}{k}.0", f"{r}{k}.1", 3, 2, k != i - j - 1, tr)
                    y = term if y is None else y + term
                outs.append(self._q(F.relu(y)))
            return outs

        ys = transition("transition1", [x], [256], W[:2])
        ys = module(f"{p}stage2.0.", ys)
        ys = transition("transition2", ys, W[:2], W[:3])
        for m in range(4):
            ys = module(f"{p}stage3.{m}.", ys)
        ys = transition("transition3", ys, W[:3], W[:4])
        for m in range(3):
            ys = module(f"{p}stage4.{m}.", ys)
        h, w = ys[0].shape[-2:]                                         # :464-469
        cat = self._q(torch.cat([ys[0]] + [F.interpolate(v, size=(h, w), mode="bilinear", align_corners=False)
                                           for v in ys[1:]], 1))
        o = self._cbr(cat, p + "last_layer.0", p + "last_layer.1", 1, 1, True, tr)     # :470, 334-351
        o = self._conv(o, p + "last_layer.3")
        return F.interpolate(o, size=(o.shape[2] * 4, o.shape[3] * 4), mode="bilinear", align_corners=False)  # :473-474

    # -- temporal head --------------------------------------------------------
    def _warp_single(self, m, flow):
        """multiframe_model.py:141-170 (grid normalised for 576x720 and merely cropped -- reference quirk)."""
        _, _, H, W = m.shape
        grid = self.sd["multiframe_net.grid"][:, :, :H, :W]
        fx = flow[:, 0] / ((W - 1) / 2.0)
        fy = flow[:, 1] / ((H - 1) / 2.0)
        g = (grid + torch.stack((fx, fy), dim=1)).permute(0, 2, 3, 1)
        return F.grid_sample(m, g, mode="bilinear", padding_mode="zeros", align_corners=True)

    def _warp(self, x):
        """multiframe_model.py:89-139."""
        N, K = self.nc, self.T
        seg = x[:, :N * K]
        flow = x[:, N * K:N * K + 2 * K - 2]
        dep = x[:, N * K + 2 * K - 2:] if self.depth else None
        segs, deps = [seg[:, 0:N]], ([dep[:, 0:1]] if self.depth else [])
        for i in range(1, K):
            fl = flow[:, 2 * (i - 1):2 * i]
            for j in range(N):
                segs.append(self._warp_single(seg[:, i * N + j:i * N + j + 1], fl))
            if self.depth:
                deps.append(self._warp_single(dep[:, i:i + 1], fl))
        out = torch.cat(segs, 1)
        return torch.cat((out, torch.cat(deps, 1)), 1) if self.depth else out

    def head(self, x):
        tr = self.head_training
        if "Basic" in self.model_type and self.optflow:        # multiframe_model.py:84-87
            x = self._warp(x)
        q = "multiframe_net.multiframe_net."
        x = self._cbr(x, q + "0", q + "1", 11, 1, True, tr)
        x = self._cbr(x, q + "3", q + "4", 3, 1, True, tr)
        x = self._cbr(x, q + "6", q + "7", 3, 1, True, tr)
        return self._conv(x, q + "9")

    def forward(self, x: Sequence[torch.Tensor], optflow=None, depth=None) -> torch.Tensor:
        """multiframe_model.py:457-471: T sequential base_model calls (separate BN batches), cat, head."""
        ys = [self.hrnet(f) for f in x]
        if optflow is not None:
            ys += list(optflow)
        if depth is not None:
            ys += list(depth)
        return self.head(torch.cat(ys, dim=1))

    __call__ = forward


# --------------------------------------------------------------------------
# Loss (src/loss.py) and the training step (src/engine.py:54-71)
# --------------------------------------------------------------------------
DEFAULT_CLASS_WEIGHTS = (1.0, 1000.0, 1000.0, 1000.0, 1000.0)   # README.md:85-86


class SingleNet(Net):
    """Single-frame HighResolutionNet with the 'HRNet' model type's last_layer (models/__init__.py:38-46: Conv 1x1 C->C,
    BN(momentum 0.1), ReLU, Conv 1x1 C->num_classes -- the same structure hrnet.py:333-351 builds, so the layer table is
    hrnet_table without prefix).  forward(x) = hrnet.py:425-476 on ONE [B,3,H,W] tensor; the caller applies log_softmax
    (scripts/train_toolpose_segmentation.py:162-163)."""

    def __init__(self, sd, width: int = 48, num_classes: int = 5, store_dtype=None):
        self.store_dtype = store_dtype            # see Net.__init__
        self.model_type, self.width, self.nc, self.T = "HRNet", width, num_classes, 1
        self.optflow, self.depth = False, False
        self.table = hrnet_table(width, num_classes, "")
        self.sd = OrderedDict()
        self.param_names = []
        for name, shape, kind in self.table:
            v = sd[name].detach().clone()
            assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
            if kind in ("conv_w", "conv_b", "bn_gamma", "bn_beta"):
                v.requires_grad_(True)
                self.param_names.append(name)
            self.sd[name] = v
        self.base_training = True
        self.head_training = True

    def forward(self, x):
        return self._q(self.hrnet(x, p=""))       # (the x4 up-sampled map is the output tensor: stored, hence rounded, in the emulated mode)

    __call__ = forward


def loss_nll(logp, target, class_weights):
    """loss.py:31-43: nn.NLLLoss(weight) -> weighted mean."""
    w = torch.as_tensor(class_weights, dtype=torch.float32)
    return F.nll_loss(logp, target, weight=w)


def loss_soft_jaccard(logp, target, num_classes):
    """loss.py:45-63: classes 1..C-1, batch-global I/U, -log, sum / C."""
    eps = 1e-15
    loss = 0.0
    for c in range(1, num_classes):
        t = (target == c).float()
        o = logp[:, c].exp()
        inter = (o * t).sum()
        union = o.sum() + t.sum() - inter
        loss = loss + (-torch.log((inter + eps) / (union + eps)))
    return loss / num_classes


def total_loss(logits, target, num_classes=5, loss_fns=("nll", "soft_jaccard"), loss_wts=(0.7, 0.3),
               class_weights=DEFAULT_CLASS_WEIGHTS):
    """engine.py:65-66 + loss.py:6-21."""
    logp = F.log_softmax(logits, dim=1)
    parts = {}
    tot = 0.0
    for fn, wt in zip(loss_fns, loss_wts):
        if fn == "nll":
            l = loss_nll(logp, target, class_weights)
        elif fn == "soft_jaccard":
            l = loss_soft_jaccard(logp, target, num_classes)
        else:
            raise ValueError(fn)
        parts["loss_" + fn] = l
        tot = tot + wt * l
    parts["loss_total"] = tot
    return tot, parts


def loss_partial_sums(logits, target, class_weights=DEFAULT_CLASS_WEIGHTS):
    """The 26 batch sums both loss terms are made of (layout of the build's `acc` block, include/mfcnet_hip.h):
    [0] sum w_t*(-logp_t), [1] sum w_t (loss.py:31-43, weighted-mean NLL); [2+c] I_c = sum p_c*[t==c], [10+c] sum p_c,
    [18+c] sum [t==c] (loss.py:45-63).  Sums over disjoint parts of a batch add up to the sums of the whole batch, which is
    what a data-parallel run all-reduces (the reference evaluates the loss on the gathered batch, engine.py:64-66)."""
    logp = F.log_softmax(logits.double(), dim=1)
    nc = logits.shape[1]
    cw = torch.as_tensor(class_weights, dtype=torch.float64)
    acc = torch.zeros(32, dtype=torch.float64)
    wt = cw[target]
    acc[0] = -(wt * logp.gather(1, target[:, None]).squeeze(1)).sum()
    acc[1] = wt.sum()
    for c in range(1, nc):
        p = logp[:, c].exp()
        tc = (target == c).double()
        acc[2 + c], acc[10 + c], acc[18 + c] = (p * tc).sum(), p.sum(), tc.sum()
    return acc


def loss_from_sums(acc, num_classes=5, loss_wts=(0.7, 0.3)):
    """(nll, soft_jaccard, total) from the batch sums: loss.py:31-43 (weighted mean) and loss.py:45-63
    (-log((I+eps)/(U+eps)), U = sum p + sum [t==c] - I, summed over classes 1.. and divided by num_classes)."""
    eps = 1e-15
    nll = acc[0] / acc[1]
    jac = 0.0
    for c in range(1, num_classes):
        I = acc[2 + c]
        U = acc[10 + c] + acc[18 + c] - I
        jac = jac - torch.log((I + eps) / (U + eps))
    jac = jac / num_classes
    return nll, jac, loss_wts[0] * nll + loss_wts[1] * jac


# ------------------------------------------------------------------------------------------
# validation metrics (src/metrics.py), numpy restatement
# ------------------------------------------------------------------------------------------
def metric_case_inputs(B, nc, H, W, seed, scale):
    """Deterministic (logits fp32 [B,nc,H,W], mask int64 [B,H,W]) of a metrics fixture.  scale == 0 gives logits quantised
    to {0, 0.5, 1} (many exact ties); the case named `empty` (seed 104) uses only classes 0..2 in the mask."""
    g = torch.Generator().manual_seed(int(seed))
    if scale == 0:
        logits = torch.randint(0, 3, (B, nc, H, W), generator=g).float() * 0.5
    else:
        logits = torch.randn(B, nc, H, W, generator=g) * float(scale)
    hi = 3 if int(seed) == 104 else nc
    mask = torch.randint(0, hi, (B, H, W), generator=g)
    return logits, mask


def confusion_per_sample(pred_classes, target_classes, num_classes):
    """[B][truth][prediction] counts; summed over B it is metrics.py:60-67 (histogramdd over (ground_truth, prediction))."""
    import numpy as np
    p = np.asarray(pred_classes).reshape(len(pred_classes), -1)
    t = np.asarray(target_classes).reshape(len(target_classes), -1)
    out = np.zeros((p.shape[0], num_classes, num_classes), dtype=np.int64)
    for b in range(p.shape[0]):
        np.add.at(out[b], (t[b], p[b]), 1)
    return out


def metrics_from_confusion(conf, metric_fns=("iou", "dice")):
    """metrics.py:4-39 from the per-sample confusion counts, background (class 0) excluded.
    iou_c follows get_jaccard (metrics.py:41-45): the per-sample ratio ... `[0]`, i.e. of the FIRST sample of the batch only;
    dice_c follows get_dice (metrics.py:47-48): sums over the whole batch.  Returns (values per class, metric_dict)."""
    import numpy as np
    conf = np.asarray(conf, dtype=np.float64)
    nc = conf.shape[1]
    eps = 1e-15
    vals, md = [], {}
    for fn in metric_fns:
        per = []
        if fn == "iou":
            c0 = conf[0]
            for c in range(1, nc):
                inter, true, pred = c0[c, c], c0[c, :].sum(), c0[:, c].sum()
                per.append((inter + eps) / (true + pred - inter + eps))
        elif fn == "dice":
            call = conf.sum(axis=0)
            for c in range(1, nc):
                inter, true, pred = call[c, c], call[c, :].sum(), call[:, c].sum()
                per.append((2 * inter + eps) / (true + pred + eps))
        elif fn == "jaccard":
            raise NotImplementedError
        else:
            raise ValueError(f"Metric function {fn} not implemented")
        vals.append(per)
        md["metric_" + fn] = float(np.mean(per))
    return vals, md


def get_metrics(outputs, targets, metric_fns, num_classes):
    """metrics.py:4-39: argmax over the class axis (first maximum wins, numpy.argmax), then the per-class ratios."""
    pred = outputs.detach().cpu().numpy().argmax(axis=1)
    conf = confusion_per_sample(pred, targets.detach().cpu().numpy(), num_classes)
    return metrics_from_confusion(conf, metric_fns)


def make_adam(net: Net, lr: float = 1e-4, load_wts_base_model: bool = False):
    """scripts/train_multiframe_detection.py:128-151: two groups, base lr/T (or lr/(100T)), head lr."""
    base_lr = lr / (100.0 * net.T) if load_wts_base_model else lr / net.T
    return torch.optim.Adam([{"params": net.params("base_model."), "lr": base_lr},
                             {"params": net.params("multiframe_net."), "lr": lr}])


def train_step(net: Net, opt, frames, mask, optflow=None, depth=None, **loss_kw):
    """engine.py:54-71: zero_grad, forward, log_softmax, loss, backward, step."""
    opt.zero_grad(set_to_none=True)
    out = net(frames, optflow=optflow, depth=depth)
    loss, parts = total_loss(out, mask, net.nc, **loss_kw)
    loss.backward()
    opt.step()
    return out.detach(), {k: float(v) for k, v in parts.items()}


# --------------------------------------------------------------------------
# ResUnet_VB (models/resunet.py:97-180): weight-standardised 3x3 conv + GroupNorm + SiLU residual U-Net.
# Dead code in the reference (nothing constructs it) but named by SURVEY.md 8(f) rank 4 as an alternative base; restated here
# functionally (inference), keyed by the reference's state_dict names, and pinned by tests/golden/resunet_vb_eval.npz.
# --------------------------------------------------------------------------
def resunet_table(channels=3, dim=16, out_dim=5, dim_mults=(1, 2, 4, 8), init_dim=None):
    """(name, shape, kind) in the reference's registration order (resunet.py:107-155)."""
    init_dim = init_dim or dim
    dims = [init_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    t = []

    def conv(name, cout, cin, k):
        t.append((name + ".weight", (cout, cin, k, k), "conv_w"))
        t.append((name + ".bias", (cout,), "conv_b"))

    def block(name, cin, cout):                      # Block: proj (WS conv 3x3) + GroupNorm, resunet.py:62-76
        conv(name + ".proj", cout, cin, 3)
        t.append((name + ".norm.weight", (cout,), "bn_gamma"))
        t.append((name + ".norm.bias", (cout,), "bn_beta"))

    def resblock(name, cin, cout):                   # resunet.py:78-95
        block(name + ".block1", cin, cout)
        block(name + ".block2", cout, cout)
        if cin != cout:
            conv(name + ".res_conv", cout, cin, 1)

    conv("init_conv", init_dim, channels, 7)
    n = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        resblock(f"downs.{i}.0", ci, ci)
        if i < n - 1:
            conv(f"downs.{i}.1.1", co, ci * 4, 1)     # Downsample = Sequential(Rearrange, Conv2d 1x1), :40-44
        else:
            conv(f"downs.{i}.1", co, ci, 3)
    for i, (ci, co) in enumerate(reversed(in_out)):   # (self.ups is registered before self.mid_block: resunet.py:121-122, 135)
        resblock(f"ups.{i}.0", co + ci, co)
        if i < n - 1:
            conv(f"ups.{i}.1.1", ci, co, 3)           # Upsample = Sequential(nn.Upsample, Conv2d 3x3), :34-38
        else:
            conv(f"ups.{i}.1", ci, co, 3)
    resblock("mid_block", dims[-1], dims[-1])
    resblock("final_res_block", dim * 2, dim)
    conv("output_layer", out_dim, dim, 1)
    return t


def resunet_forward(sd, x, dim=16, dim_mults=(1, 2, 4, 8), groups=8, init_dim=None):
    """resunet.py:157-180 with the module tree flattened; sd = {name: tensor} as resunet_table lists them."""
    init_dim = init_dim or dim
    dims = [init_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))

    def ws_conv(x, name):                            # WeightStandardizedConv2d.forward, :50-60 (fp32: eps 1e-5)
        w = sd[name + ".weight"]
        mean = w.mean(dim=(1, 2, 3), keepdim=True)
        var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        return F.conv2d(x, (w - mean) * (var + 1e-5).rsqrt(), sd[name + ".bias"], padding=1)

    def block(x, name):
        y = ws_conv(x, name + ".proj")
        y = F.group_norm(y, groups, sd[name + ".norm.weight"], sd[name + ".norm.bias"], 1e-5)
        return F.silu(y)

    def resblock(x, name):
        h = block(block(x, name + ".block1"), name + ".block2")
        if name + ".res_conv.weight" in sd:
            return h + F.conv2d(x, sd[name + ".res_conv.weight"], sd[name + ".res_conv.bias"])
        return h + x

    def conv(x, name, pad):
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)

    x = conv(x, "init_conv", 3)
    r = x
    hs = []
    n = len(in_out)
    for i in range(n):
        x = resblock(x, f"downs.{i}.0")
        hs.append(x)
        if i < n - 1:                                # 'b c (h p1) (w p2) -> b (c p1 p2) h w', then 1x1
            B, Cc, H, W = x.shape
            x = x.view(B, Cc, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, Cc * 4, H // 2, W // 2)
            x = conv(x, f"downs.{i}.1.1", 0)
        else:
            x = conv(x, f"downs.{i}.1", 1)
    x = resblock(x, "mid_block")
    for i in range(n):
        x = torch.cat((x, hs.pop()), dim=1)
        x = resblock(x, f"ups.{i}.0")
        if i < n - 1:
            x = conv(F.interpolate(x, scale_factor=2, mode="nearest"), f"ups.{i}.1.1", 1)
        else:
            x = conv(x, f"ups.{i}.1", 1)
    x = torch.cat((x, r), dim=1)
    x = resblock(x, "final_res_block")
    return conv(x, "output_layer", 0)
