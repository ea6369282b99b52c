import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
from bench_conv import timeit
for (N, Cin, Cout, k, s, H, W) in [(24, 32, 32, 3, 1, 120, 160), (24, 48, 48, 3, 1, 120, 160), (24, 64, 64, 3, 1, 60, 80), (24, 96, 96, 3, 1, 60, 80)]:
    dt = torch.bfloat16; pad = k // 2
    x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(dt)
    dy = torch.randn(N, H, W, ops.rup(Cout, 8), device="cuda").to(dt)
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    line = f"{(N,Cin,Cout,k,s,H,W)}"
    for S in (32, 64, 128, 256, 512, 1024):
        d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, 0, ops.dt_of(x), N, H, W, x.shape[3], Cin, H, W, dy.shape[3], Cout,
                        k, k, -pad, -pad, s, 0, N, 0, 0, S)
        dwp = torch.zeros(L.wgrad_parts(d) * k * k * ops.rup(Cout, 16) * ops.rup(Cin, 16), device="cuda")     # one slice per pixel split
        d.dwp = dwp.data_ptr()
        t = timeit(lambda: L.call(L.lib.mfc_conv2d_wgrad, d))
        line += f" | S{S}: {t*1e6:6.1f}us {flops/t/1e12:5.0f}TF"
    print(line, flush=True)
