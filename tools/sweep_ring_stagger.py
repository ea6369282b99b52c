"""Stagger sweep of conv3x3_ring_kernel (mfc_set_flag(37, cycles)): python tools/sweep_ring_stagger.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mfcnet_amd import _lib as L
from sweep_conv2 import time_op
from bench_ring import build
for (N, Cc, H, W) in [(24, 32, 120, 160), (24, 64, 60, 80)]:
    for v in ("plain", "stats", "xf+stats", "bn2", "acc+src+bn3"):
        if Cc == 64 and "bn" in v:
            continue
        line = f"N{N} C{Cc} {H}x{W} {v:12s}"
        for cyc in (0, 1024, 2048, 3072, 4096, 6144):
            L.lib.mfc_set_flag(37, cyc)
            ts = []
            for rep in range(3):
                d, out, stats, lay, keep = build(N, Cc, H, W, v, 1)
                op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
                ts.append(time_op(op))
            line += f" | {cyc}: {min(ts):5.1f}"
        print(line, flush=True)
L.lib.mfc_set_flag(37, 0)
