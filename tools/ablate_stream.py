"""Ablation of conv3x3_stream_kernel (plain variant): python tools/ablate_stream.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mfcnet_amd import _lib as L
from sweep_conv2 import time_op
from bench_ring import build
for (N, Cc, H, W) in [(24, 128, 30, 40), (24, 256, 15, 20)]:
    for mt in (0,):
        L.lib.mfc_set_flag(36, mt)
        line = f"N{N} C{Cc} {H}x{W} plain MT{mt}"
        for m, name in ((0, "full"), (5, "-mfma-wdma"), (5 + 8, "..-fixup"), (5 + 8 + 32, "..-planes"), (5 + 8 + 32 + 16, "..-vmwait"), (5 + 8 + 32 + 16 + 64, "..-barrier"), (5 + 8 + 32 + 16 + 64 + 2, "..-store")):
            L.lib.mfc_set_flag(35, m)
            d, out, stats, lay, keep = build(N, Cc, H, W, "plain", 1)
            op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
            line += f" | {name} {time_op(op):5.1f}"
        L.lib.mfc_set_flag(35, 0)
        print(line + f" | grid {lay.grid}x{lay.per_block} MT{lay.MT}", flush=True)
L.lib.mfc_set_flag(36, 0)
