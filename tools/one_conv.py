"""Run one conv shape repeatedly (for rocprofv3 --pmc).  python tools/one_conv.py N Cin Cout k s H W [iters] [fused]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
N, Cin, Cout, k, s, H, W = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
fused = len(sys.argv) > 9
dt = torch.bfloat16
pad = k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(dt)
w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
out = torch.zeros(N, Ho, Wo, ops.rup(Cout, 8), dtype=dt, device="cuda")
stats = torch.zeros(L.STAT_REPLICAS, 3, 2, out.shape[3], dtype=torch.float64, device="cuda")
coef = torch.rand(3, 4, x.shape[3], device="cuda")
d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, coef.data_ptr() if fused else 0, stats.data_ptr() if fused else 0,
               ops.dt_of(x), N, H, W, x.shape[3], Cin, Ho, Wo, out.shape[3], Cout, Ho, Wo, k, k, -pad, -pad, s, 1, 1, 0, 0, 1, N // 3, 0, 0, 0)
wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
for _ in range(iters):
    L.call(L.lib.mfc_conv2d_fwd, d)
torch.cuda.synchronize()
