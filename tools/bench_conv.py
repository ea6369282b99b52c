"""Micro-benchmark of single conv / wgrad launches on the hot shapes (tuning aid, not part of the product path).
    python tools/bench_conv.py [bf16|fp32] [fwd|wgrad|both]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops

SHAPES = [  # N, Cin, Cout, k, s, H, W
    (24, 48, 48, 3, 1, 120, 160), (24, 96, 96, 3, 1, 60, 80), (24, 192, 192, 3, 1, 30, 40), (24, 384, 384, 3, 1, 15, 20),
    (24, 32, 32, 3, 1, 120, 160), (24, 64, 64, 3, 1, 60, 80), (24, 128, 128, 3, 1, 30, 40), (24, 256, 256, 3, 1, 15, 20),
    (24, 720, 720, 1, 1, 120, 160), (24, 64, 256, 1, 1, 120, 160), (24, 48, 96, 3, 2, 120, 160), (8, 15, 15, 11, 1, 480, 640),
]

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3

def main():
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
    what = sys.argv[2] if len(sys.argv) > 2 else "both"
    if len(sys.argv) > 3:
        L.lib.mfc_set_flag(2, int(sys.argv[3]))
    if len(sys.argv) > 4:
        L.lib.mfc_set_flag(3, int(sys.argv[4]))
    for (N, Cin, Cout, k, s, H, W) in SHAPES:
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(dt)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        out = torch.zeros(N, Ho, Wo, ops.rup(Cout, 8), dtype=dt, device="cuda")
        stats = torch.zeros(L.STAT_REPLICAS, 3, 2, out.shape[3], dtype=torch.float64, device="cuda")
        coef = torch.rand(3, 4, x.shape[3], device="cuda")
        flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        line = f"{str((N,Cin,Cout,k,s,H,W)):38s}"
        if what in ("fwd", "both"):
            for label, cf, st in (("plain", None, None), ("fused", coef, stats)):
                d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, cf.data_ptr() if cf is not None else 0,
                               st.data_ptr() if st is not None else 0, ops.dt_of(x), N, H, W, x.shape[3], Cin, Ho, Wo, out.shape[3], Cout,
                               Ho, Wo, k, k, -pad, -pad, s, 1, 1, 0, 0, 1, N // 3 if N % 3 == 0 else N, 0, 0, 0)
                if st is not None and N % 3:
                    continue
                wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
                if label == "plain":
                    lay = L.conv_layout(d)
                    line += f" [NT{lay.NT16//16} MT{lay.MT} {lay.TH}x{lay.TW} KG{lay.KG} TAS{lay.TAS} lds{lay.lds_bytes//1024}K g{lay.grid}x{lay.per_block}]"
                t = timeit(lambda: L.call(L.lib.mfc_conv2d_fwd, d))
                line += f" | fwd-{label} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF"
        if what in ("wgrad", "both"):
            dy = torch.randn(N, Ho, Wo, ops.rup(Cout, 8), device="cuda").to(dt)
            d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, 0, ops.dt_of(x), N, H, W, x.shape[3], Cin, Ho, Wo, dy.shape[3], Cout,
                            k, k, -pad, -pad, s, 0, N, 0, 0, 0)
            dwp = torch.zeros(L.wgrad_parts(d) * k * k * ops.rup(Cout, 16) * ops.rup(Cin, 16), device="cuda")     # one slice per pixel split
            d.dwp = dwp.data_ptr()
            t = timeit(lambda: L.call(L.lib.mfc_conv2d_wgrad, d))
            line += f" | wgrad {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF"
        print(line, flush=True)

if __name__ == "__main__":
    main()
