"""A/B of the resident-workgroup count of the 32/48-channel 120x160 convolutions: time per launch at mfc_set_flag(4, grid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mfcnet_amd import _lib as L, ops
from sweep_conv2 import time_op
for (N, Cin, Cout, k, H, W) in [(24, 32, 32, 3, 120, 160), (24, 48, 48, 3, 120, 160)]:
    for xf in (0, 1):
        pad = k // 2
        x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        out = torch.zeros(N, H, W, ops.rup(Cout, 8), dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros(L.STAT_REPLICAS, 3, 2, out.shape[3], dtype=torch.float64, device="cuda")
        coef = torch.rand(3, 4, x.shape[3], device="cuda")
        line = f"{(N,Cin,Cout,k,H,W)} xf={xf}"
        for grid in (512, 768, 1024):
            L.lib.mfc_set_flag(4, grid)
            d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, coef.data_ptr() if xf else 0, stats.data_ptr(), L.BF16, N, H, W, x.shape[3], Cin, H, W,
                           out.shape[3], Cout, H, W, k, k, -pad, -pad, 1, 1, 1, 0, 0, 1 if xf else 0, N // 3, 0, 0, 0)
            wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
            lay = L.conv_layout(d)
            op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
            line += f" | grid{grid}: NT{lay.NT16//16} MT{lay.MT} {lay.grid}x{lay.per_block} {time_op(op):5.1f} us"
        L.lib.mfc_set_flag(4, 512)
        print(line, flush=True)
