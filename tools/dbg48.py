import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "mfcnet-tracker_amd")
from mfcnet_amd import _lib as L, ops
from sweep_conv2 import time_op
N, Cc, H, W = 24, 48, 120, 160
x = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
w = torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.05
out = torch.zeros(N, H, W, Cc, device="cuda").bfloat16()
for ring48, mt in ((0, 2), (1, 2)):
    L.lib.mfc_set_flag(50, ring48); L.lib.mfc_set_flag(51, mt)
    for abl in ((0, 15, 31, 47, 79, 127) if ring48 else (0,)):
        L.lib.mfc_set_flag(32, abl)
        d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, 0, 0, L.BF16, N, H, W, Cc, Cc, H, W, Cc, Cc, H, W, 3, 3, -1, -1, 1, 1, 1, 0, 0, 0, 8, 0, 0, 0)
        wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
        op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
        t = time_op(op)
        print(f"ring48={ring48} MT{mt} ablate={abl}: {t:6.1f} us", flush=True)
L.lib.mfc_set_flag(32, 0); L.lib.mfc_set_flag(50, 1)
