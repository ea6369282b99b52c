"""Parameter / buffer table of the MFCNet hot path, in the reference's state_dict names and order.

Mirrors (by construction rule, not by code) the module registration order of
  models/hrnet.py:271-351        HighResolutionNet   (prefix base_model.)
  models/multiframe_model.py:61-82, 191-202   MultiFrameNetBasic / MultiFrameNetLarge (prefix multiframe_net.)
so that `state_dict()` / strict `load_state_dict()` of the reference and of this package interchange
(utils/model_utils.py:6-39).  `width` generalises the reference's hard-coded 48 (hrnet.py:297-330).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

WARP_GRID_HW = (576, 720)        # multiframe_model.py:179


@dataclass
class Entry:
    name: str
    shape: Tuple[int, ...]
    kind: str            # conv_w | conv_b | bn_w | bn_b | bn_rm | bn_rv | bn_nbt | grid

    @property
    def is_param(self):
        return self.kind in ("conv_w", "conv_b", "bn_w", "bn_b")

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def _conv(out: List[Entry], name, cout, cin, k, bias=False):
    out.append(Entry(name + ".weight", (cout, cin, k, k), "conv_w"))
    if bias:
        out.append(Entry(name + ".bias", (cout,), "conv_b"))


def _bn(out: List[Entry], name, c):
    out += [Entry(name + ".weight", (c,), "bn_w"), Entry(name + ".bias", (c,), "bn_b"),
            Entry(name + ".running_mean", (c,), "bn_rm"), Entry(name + ".running_var", (c,), "bn_rv"),
            Entry(name + ".num_batches_tracked", (), "bn_nbt")]


def branch_widths(width):
    return [width, 2 * width, 4 * width, 8 * width]


def hrnet_entries(width=48, num_classes=5, p="base_model.") -> List[Entry]:
    W = branch_widths(width)
    t: List[Entry] = []
    _conv(t, p + "conv1", 64, 3, 3); _bn(t, p + "bn1", 64)
    _conv(t, p + "conv2", 64, 64, 3); _bn(t, p + "bn2", 64)
    cin = 64
    for b in range(4):
        q = f"{p}layer1.{b}."
        _conv(t, q + "conv1", 64, cin, 1); _bn(t, q + "bn1", 64)
        _conv(t, q + "conv2", 64, 64, 3); _bn(t, q + "bn2", 64)
        _conv(t, q + "conv3", 256, 64, 1); _bn(t, q + "bn3", 256)
        if b == 0:
            _conv(t, q + "downsample.0", 256, cin, 1); _bn(t, q + "downsample.1", 256)
        cin = 256

    def transition(name, pre, cur):
        for i, c in enumerate(cur):
            if i < len(pre):
                if c != pre[i]:
                    _conv(t, f"{p}{name}.{i}.0", c, pre[i], 3); _bn(t, f"{p}{name}.{i}.1", c)
            else:
                for j in range(i + 1 - len(pre)):
                    ci = pre[-1]
                    co = c if j == i - len(pre) else ci
                    _conv(t, f"{p}{name}.{i}.{j}.0", co, ci, 3); _bn(t, f"{p}{name}.{i}.{j}.1", co)

    def stage(name, n_modules, ch):
        nb = len(ch)
        for m in range(n_modules):
            q = f"{p}{name}.{m}."
            for i in range(nb):
                for b in range(4):
                    r = f"{q}branches.{i}.{b}."
                    _conv(t, r + "conv1", ch[i], ch[i], 3); _bn(t, r + "bn1", ch[i])
                    _conv(t, r + "conv2", ch[i], ch[i], 3); _bn(t, r + "bn2", ch[i])
            for i in range(nb):
                for j in range(nb):
                    r = f"{q}fuse_layers.{i}.{j}."
                    if j > i:
                        _conv(t, r + "0", ch[i], ch[j], 1); _bn(t, r + "1", ch[i])
                    elif j < i:
                        for k in range(i - j):
                            co = ch[i] if k == i - j - 1 else ch[j]
                            _conv(t, f"{r}{k}.0", co, ch[j], 3); _bn(t, f"{r}{k}.1", co)

    transition("transition1", [256], W[:2]); stage("stage2", 1, W[:2])
    transition("transition2", W[:2], W[:3]); stage("stage3", 4, W[:3])
    transition("transition3", W[:3], W[:4]); stage("stage4", 3, W[:4])
    last = sum(W)
    _conv(t, p + "last_layer.0", last, last, 1, bias=True); _bn(t, p + "last_layer.1", last)
    _conv(t, p + "last_layer.3", num_classes, last, 1, bias=True)
    return t


def head_in_channels(basic: bool, num_classes, T, optflow, depth) -> int:
    """multiframe_model.py:23-32 (Large) / :54-56 (Basic: the warp consumes the flow channels)."""
    c = T * num_classes
    if optflow and not basic:
        c += 2 * (T - 1)
    if depth:
        c += T
    return c


def head_entries(basic: bool, num_classes, T, optflow, depth, p="multiframe_net.") -> List[Entry]:
    cin = head_in_channels(basic, num_classes, T, optflow, depth)
    mid = T * num_classes
    t: List[Entry] = []
    if basic:
        t.append(Entry(p + "grid", (1, 2) + WARP_GRID_HW, "grid"))
    q = p + "multiframe_net."
    _conv(t, q + "0", mid, cin, 11); _bn(t, q + "1", mid)
    _conv(t, q + "3", mid, mid, 3); _bn(t, q + "4", mid)
    _conv(t, q + "6", mid, mid, 3); _bn(t, q + "7", mid)
    _conv(t, q + "9", num_classes, mid, 1)
    return t
