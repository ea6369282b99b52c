"""Ablation of the bf16 wave-private wgrad kernel: stream time per launch (mfc_program_profile, 10 back-to-back launches)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops

def time_op(op, reps=10):
    prog = (L.Op * 1)(); prog[0] = op
    ms = (C.c_float * 1)()
    for _ in range(2):
        rc = L.lib.mfc_program_profile(prog, 1, reps, ms, L.stream_ptr()); assert rc == 0, rc
    return ms[0] * 1e3

for (N, Cin, Cout, k, s, H, W, xf) in [(24, 32, 32, 3, 1, 120, 160, 0), (24, 32, 32, 3, 1, 120, 160, 1), (24, 64, 64, 3, 1, 60, 80, 0), (24, 128, 128, 3, 1, 30, 40, 0), (24, 256, 256, 3, 1, 15, 20, 0), (24, 48, 48, 3, 1, 120, 160, 0)]:
    pad = k // 2
    x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, W, ops.rup(Cout, 8), device="cuda").to(torch.bfloat16)
    coef = torch.randn(3, 4, ops.rup(Cin, 8), device="cuda")
    Co16, Ci16 = ops.rup(Cout, 16), ops.rup(Cin, 16)
    d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, coef.data_ptr() if xf else 0, L.BF16, N, H, W, x.shape[3], Cin, H, W, dy.shape[3], Cout,
                    k, k, -pad, -pad, s, 1 if xf else 0, N // 3, 0, 0, 0)
    parts = L.wgrad_parts(d)
    dwp = torch.zeros(parts * k * k * Co16 * Ci16, dtype=torch.float32, device="cuda")      # one slice per pixel split
    d.dwp = dwp.data_ptr()
    op = L.Op(); op.kind = L.OP_WGRAD; op.u.wgrad = d
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    line = f"{(N,Cin,Cout,k,s,H,W,xf)} parts={parts}"
    for A, name in ((0, "full"), (1, "-gload"), (2, "-lds_st"), (3, "-gload-st"), (4, "-mfma"), (8, "-flush"), (12, "-mfma-flush"), (15, "nothing")):
        L.lib.mfc_set_flag(7, A)
        t = time_op(op)
        line += f" | {name}: {t:5.1f}"
    L.lib.mfc_set_flag(7, 0)
    print(line + f" | full {flops / time_op(op) / 1e6:.0f} TF/s", flush=True)
