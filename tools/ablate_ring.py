"""Ablation / knob sweep of conv3x3_ring_kernel: stream time per launch (mfc_program_profile).  python tools/ablate_ring.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mfcnet_amd import _lib as L
from sweep_conv2 import time_op
from bench_ring import build

MASKS = [(0, "full"), (1, "-mfma"), (2, "-store"), (4, "-dma"), (8, "-fixup"), (3, "-mfma-store"), (7, "-mfma-store-dma"), (15, "nothing")]
for (N, Cc, H, W) in [(24, 32, 120, 160), (24, 64, 60, 80)]:
    for v in ("plain", "stats", "xf+stats", "bn2", "acc+src+bn3"):
        for (mt, wgs) in ((4, 2), (2, 2), (2, 3)):
            if Cc == 64 and (mt != 4 or v in ("bn2", "acc+src+bn3")):
                continue
            L.lib.mfc_set_flag(31, mt); L.lib.mfc_set_flag(33, wgs)
            line = f"N{N} C{Cc} {H}x{W} {v:12s} MT{mt} wgs{wgs}"
            for m, name in MASKS:
                L.lib.mfc_set_flag(32, m)
                d, out, stats, lay, keep = build(N, Cc, H, W, v, 1)
                op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
                line += f" | {name} {time_op(op):5.1f}"
            L.lib.mfc_set_flag(32, 0)
            print(line + f" | grid {lay.grid}x{lay.per_block}", flush=True)
L.lib.mfc_set_flag(31, 4); L.lib.mfc_set_flag(33, 2)
