"""Ablation of the conv kernel: stream time per launch (mfc_program_profile, 10 back-to-back launches)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
from sweep_conv2 import time_op
for (N, Cin, Cout, k, H, W) in [(24, 32, 32, 3, 120, 160), (24, 64, 64, 3, 60, 80), (24, 128, 128, 3, 30, 40), (24, 256, 256, 3, 15, 20),
                               (24, 48, 48, 3, 120, 160), (24, 96, 96, 3, 60, 80), (24, 192, 192, 3, 30, 40), (24, 384, 384, 3, 15, 20)]:
    pad = k // 2
    x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    out = torch.zeros(N, H, W, ops.rup(Cout, 8), dtype=torch.bfloat16, device="cuda")
    d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, 0, 0, L.BF16, N, H, W, x.shape[3], Cin, H, W, out.shape[3], Cout, H, W, k, k, -pad, -pad, 1, 1, 1, 0, 0, 0, N, 0, 0, 0)
    wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
    lay = L.conv_layout(d)
    op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
    line = f"{(N,Cin,Cout,k,H,W)} MT{lay.MT} NT{lay.NT16//16} KG{lay.KG} chunks{lay.nchunks} grid{lay.grid}x{lay.per_block}"
    for A, name in ((0, "full"), (1, "-dma"), (2, "-patch"), (4, "-store"), (8, "-mfma"), (3, "-dma-patch"), (7, "-dma-patch-store"), (15, "nothing"), (15 + 64, "nothing-epi"), (32, "prologue"), (16, "empty")):
        L.lib.mfc_set_flag(5, A)
        line += f" | {name}: {time_op(op):5.1f}"
    L.lib.mfc_set_flag(5, 0)
    print(line, flush=True)
