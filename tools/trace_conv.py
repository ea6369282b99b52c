"""(diagnostic build only: `make -C mfcnet-tracker_amd/csrc TRACE=1`) cycle trace of wave 0 of one workgroup of the conv kernel."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops
NAMES = {13: "epi: tile base done", 14: "epi: row start", 10: "loop top", 11: "dma issued", 12: "patch loads issued", 0: "start", 1: "prologue done", 2: "stage top (prefetch issued)", 3: "k-loop done", 4: "epilogue done", 5: "barrier A", 6: "patch stored", 7: "dma waited", 8: "barrier B", 9: "end"}
SHAPES = [(24, 32, 32, 3, 120, 160, 0), (24, 64, 64, 3, 60, 80, 0), (24, 128, 128, 3, 30, 40, 0), (24, 48, 48, 3, 120, 160, 0), (24, 96, 96, 3, 60, 80, 0), (24, 192, 192, 3, 30, 40, 0), (24, 384, 384, 3, 15, 20, 0)]
for (N, Cin, Cout, k, H, W, xf) in SHAPES:
    pad = k // 2
    x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    out = torch.zeros(N, H, W, ops.rup(Cout, 8), dtype=torch.bfloat16, device="cuda")
    d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, 0, 0, L.BF16, N, H, W, x.shape[3], Cin, H, W, out.shape[3], Cout, H, W, k, k, -pad, -pad, 1, 1, 1, 0, 0, 0, N, 0, 0, 0)
    wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
    for _ in range(3):
        L.call(L.lib.mfc_conv2d_fwd, d)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 4096)()
    L.lib.mfc_conv_trace_read.argtypes = [C.c_void_p]
    assert L.lib.mfc_conv_trace_read(buf) == 0
    n = int(buf[4095])
    print(f"--- {(N,Cin,Cout,k,H,W)} : {n} trace points")
    t0 = buf[1]
    prev = t0
    for i in range(min(n, int(os.environ.get('TRACE_POINTS', '40')))):
        tag, t = buf[2 * i], buf[2 * i + 1]
        print(f"  {t - t0:8d} (+{t - prev:6d})  {NAMES.get(tag, tag)}")
        prev = t
