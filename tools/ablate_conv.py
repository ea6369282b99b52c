import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
from bench_conv import timeit
for (N, Cin, Cout, k, s, H, W) in [(24, 32, 32, 3, 1, 120, 160), (24, 64, 64, 3, 1, 60, 80), (24, 96, 96, 3, 1, 60, 80), (24, 192, 192, 3, 1, 30, 40)]:
    dt = torch.bfloat16; pad = k // 2
    x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(dt)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    out = torch.zeros(N, H, W, ops.rup(Cout, 8), dtype=dt, device="cuda")
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    line = f"{(N,Cin,Cout,k,s,H,W)}"
    for A, name in ((0, "full"), (1, "-dma"), (2, "-patch"), (4, "-store"), (8, "-mfma"), (3, "-dma-patch"), (7, "-dma-patch-store"), (15, "nothing"), (16, "empty"), (32, "prologue"), (15+64, "noepi")):
        L.lib.mfc_set_flag(5, A)
        d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, 0, 0, ops.dt_of(x), N, H, W, x.shape[3], Cin, H, W, out.shape[3], Cout, H, W, k, k, -pad, -pad, s, 1, 1, 0, 0, 0, N, 0, 0, 0)
        wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
        t = timeit(lambda: L.call(L.lib.mfc_conv2d_fwd, d))
        line += f" | {name}: {t*1e6:5.1f}"
    L.lib.mfc_set_flag(5, 0)
    print(line, flush=True)
