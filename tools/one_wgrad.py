"""Run one wgrad shape repeatedly (for rocprofv3 --pmc).  python tools/one_wgrad.py N Cin Cout k s H W [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
N, Cin, Cout, k, s, H, W = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
dt = torch.bfloat16
pad = k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(dt)
dy = torch.randn(N, Ho, Wo, ops.rup(Cout, 8), device="cuda").to(dt)
d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, 0, ops.dt_of(x), N, H, W, x.shape[3], Cin, Ho, Wo, dy.shape[3], Cout,
                k, k, -pad, -pad, s, 0, N, 0, 0, 0)
dwp = torch.zeros(L.wgrad_parts(d) * k * k * ops.rup(Cout, 16) * ops.rup(Cin, 16), device="cuda")     # one slice per pixel split
d.dwp = dwp.data_ptr()
for _ in range(iters):
    L.call(L.lib.mfc_conv2d_wgrad, d)
torch.cuda.synchronize()
