"""Launch geometry the library picks for the encoder convolutions of the benchmarked step (host query, no GPU): python tools/show_geometry.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
from mfcnet_amd import _lib as L
for (Cc, H, W) in [(32, 120, 160), (64, 60, 80), (128, 30, 40), (256, 15, 20), (48, 120, 160), (96, 60, 80), (192, 30, 40), (384, 15, 20)]:
    for coef, flags, tag in ((0, 0, "fwd conv1"), (16, 0, "fwd conv2 (xf)"), (0, L.CONV_WANT_FA, "dgrad (want FA)")):
        d = L.ConvDesc(16, 0, 16, 0, coef, 0, L.BF16, 24, H, W, Cc, Cc, H, W, Cc, Cc, H, W, 3, 3, -1, -1, 1, 1, 1, 0, 0, 1 if coef else 0, 8, 0, 0, 0)
        d.flags = flags
        lay = L.conv_layout(d)
        print(f"C{Cc:3d} {H}x{W} {tag:18s}: NW{lay.NW} MT{lay.MT} NT{lay.NT16 // 16} tile {lay.TH}x{lay.TW} KG{lay.KG} chunks{lay.nchunks} TAS{lay.TAS} Yb{lay.Yblocks} "
              f"slots{lay.nslots} lds{lay.lds_bytes // 1024}K grid{lay.grid} per_block{lay.per_block} fa{lay.fa}")
