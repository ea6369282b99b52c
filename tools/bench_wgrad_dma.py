"""Old (register-staged, mfc_set_flag(29, 0)) vs new (LDS-DMA ring) 3x3 weight-gradient kernel on the encoder shapes of the W32 / W48 step,
alone on the GPU, for several workgroup targets (mfc_set_flag(11)).  python tools/bench_wgrad_dma.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mfcnet_amd import _lib as L, ops
from bench_conv import timeit
SHAPES = [(24, 32, 120, 160), (24, 64, 60, 80), (24, 128, 30, 40), (24, 256, 15, 20), (24, 96, 60, 80), (24, 192, 30, 40), (24, 384, 15, 20)]
for (N, Cc, H, W) in SHAPES:
    for xf in (0, 1):
        x = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
        dy = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
        coef = torch.rand(3, 4, Cc, device="cuda") + 0.5
        flops = 2.0 * N * H * W * Cc * Cc * 9
        byt = 2.0 * N * H * W * Cc * 2
        line = f"C{Cc:3d} {H}x{W} xf{xf}"
        for blocks in (128, 256, 512):
            L.lib.mfc_set_flag(11, blocks)
            for flag in (0, 1, 2):            # 0 register-staged wave kernel, 1 LDS-DMA ring with 4 waves, 2 with 8 waves when xf
                L.lib.mfc_set_flag(29, 1 if flag else 0)
                L.lib.mfc_set_flag(38, 1 if flag == 2 else 0)
                if flag == 2 and not xf:
                    continue
                d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, coef.data_ptr() if xf else 0, L.BF16, N, H, W, Cc, Cc, H, W, Cc, Cc,
                                3, 3, -1, -1, 1, xf, 8, 0, 0, 0)
                dwp = torch.zeros(L.wgrad_parts(d) * 9 * Cc * Cc, device="cuda")
                d.dwp = dwp.data_ptr()
                t = timeit(lambda: L.call(L.lib.mfc_conv2d_wgrad, d))
                line += f" | b{blocks} {('old', 'dma', 'dma8')[flag]} {t*1e6:6.1f}us {flops/t/1e12:4.0f}TF {byt/t/1e9:5.0f}GB/s"
        print(line, flush=True)
L.lib.mfc_set_flag(11, 128); L.lib.mfc_set_flag(29, 1); L.lib.mfc_set_flag(38, 0)
