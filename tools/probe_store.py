"""Do write-through (sc1) or non-temporal output stores shorten a convolution launch + its kernel boundary?  (MI355X_MICROARCH.md prices a dependent
boundary at + dirty bytes / 6 TB/s: a plain store leaves its line dirty in the XCD's L2 until the end-of-kernel release writes it back.)
Ring kernel, plain variant, 32 @ 120x160 and 64 @ 60x80, N = 24: 20 back-to-back launches of a ping-pong chain (launch i reads what launch i-1 wrote).
    python tools/probe_store.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops

G = 3
for (N, Cc, H, W) in [(24, 32, 120, 160), (24, 64, 60, 80)]:
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(N, H, W, Cc, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.zeros_like(a)
    w = torch.randn(Cc, Cc, 3, 3, device="cuda", generator=g) * 0.02
    stats = torch.zeros(L.STAT_REPLICAS, G, 2, Cc, dtype=torch.float64, device="cuda")
    ds = []
    for src, dst in ((a, b), (b, a)):
        ds.append(L.ConvDesc(src.data_ptr(), 0, dst.data_ptr(), 0, 0, stats.data_ptr(), L.BF16, N, H, W, Cc, Cc, H, W, Cc, Cc, H, W, 3, 3, -1, -1, 1, 1, 1, 0, 0, 0, N // G, 0, 0, 0))
    wp = ops.pack_weight(w, ds[0], "fwd")
    for d in ds:
        d.wp = wp.data_ptr()
    ref = None
    for mode, name in ((0, "plain"), (32, "sc1 (write-through)"), (64, "nt")):
        L.lib.mfc_set_flag(32, mode)
        a.copy_(torch.randn(N, H, W, Cc, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.bfloat16))
        L.call(L.lib.mfc_conv2d_fwd, ds[0])
        torch.cuda.synchronize()
        out = b.clone()
        if ref is None:
            ref = out
        same = bool(torch.equal(ref, out))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            e0.record()
            for i in range(20):
                L.call(L.lib.mfc_conv2d_fwd, ds[i & 1])
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        print(f"C{Cc} {H}x{W} stores {name:20s}: {best:6.2f} us per launch (chain of 20), output identical to plain: {same}", flush=True)
    L.lib.mfc_set_flag(32, 0)
