"""Weight-gradient launches of the temporal head / stem (tiny channel counts, huge pixel counts): stream time vs the
workgroup target (mfc_set_flag(11)).  mfc_program_profile, 10 back-to-back launches."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
from sweep_conv2 import time_op

SHAPES = [(8, 15, 15, 3, 1, 480, 640, 1), (8, 15, 15, 11, 1, 480, 640, 0), (8, 15, 5, 1, 1, 480, 640, 1), (24, 3, 64, 3, 2, 480, 640, 0),
          (24, 64, 64, 3, 2, 240, 320, 1), (24, 32, 32, 3, 1, 120, 160, 1), (24, 64, 256, 1, 1, 120, 160, 1), (24, 64, 256, 1, 1, 120, 160, 0), (24, 256, 64, 1, 1, 120, 160, 0), (24, 128, 128, 1, 1, 60, 80, 0), (24, 256, 256, 1, 1, 120, 160, 0), (24, 480, 480, 1, 1, 120, 160, 0), (24, 720, 720, 1, 1, 120, 160, 0)]
for (N, Cin, Cout, k, s, H, W, xf) in SHAPES:
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, Ho, Wo, ops.rup(Cout, 8), device="cuda").to(torch.bfloat16)
    coef = torch.randn(N // 8 if N >= 8 else 1, 4, ops.rup(Cin, 8), device="cuda")
    Co16, Ci16 = ops.rup(Cout, 16), ops.rup(Cin, 16)
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    line = f"{(N,Cin,Cout,k,s,H,W,xf)}"
    for blocks in (0, 128, 64):            # flag 25/26: split-K GEMM kernel off / from 128 channels / from 64
        L.lib.mfc_set_flag(25, 1 if blocks else 0); L.lib.mfc_set_flag(26, blocks or 128)
        d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, coef.data_ptr() if xf else 0, L.BF16, N, H, W, x.shape[3], Cin, Ho, Wo, dy.shape[3], Cout,
                        k, k, -pad, -pad, s, 1 if xf else 0, 8 if N >= 8 else N, 0, 0, 0)
        parts = L.wgrad_parts(d)
        dwp = torch.zeros(parts * k * k * Co16 * Ci16, dtype=torch.float32, device="cuda")
        d.dwp = dwp.data_ptr()
        op = L.Op(); op.kind = L.OP_WGRAD; op.u.wgrad = d
        t = time_op(op)
        line += f" | {blocks}: parts {parts:4d} {t:7.1f}us {flops / t / 1e6:4.0f}TF"
    L.lib.mfc_set_flag(25, 1); L.lib.mfc_set_flag(26, 128)
    print(line, flush=True)
