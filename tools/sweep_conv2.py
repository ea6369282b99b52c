"""Conv geometry sweep with stream-time measurement (mfc_program_profile, 10 back-to-back launches)."""
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

SHAPES = [(24, 64, 256, 1, 120, 160), (24, 64, 128, 1, 120, 160), (24, 480, 480, 1, 120, 160), (24, 720, 720, 1, 120, 160), (24, 256, 64, 1, 120, 160), (24, 256, 256, 1, 120, 160), (24, 128, 128, 1, 60, 80), (24, 192, 384, 1, 30, 40)]
SHAPES_OLD = [(24, 32, 32, 3, 120, 160), (24, 64, 64, 3, 60, 80), (24, 128, 128, 3, 30, 40), (24, 256, 256, 3, 15, 20), (24, 480, 480, 1, 120, 160), (24, 64, 256, 1, 120, 160), (24, 96, 96, 3, 60, 80), (24, 192, 192, 3, 30, 40), (24, 384, 384, 3, 15, 20), (8, 15, 15, 11, 480, 640), (8, 15, 15, 3, 480, 640)]
COMBOS = [0, 1]           # flag 23: big 1x1 convolutions through the plain-GEMM kernel
def main():
  for (N, Cin, Cout, k, H, W) in SHAPES:
      pad = k // 2
      x = torch.randn(N, H, W, ops.rup(Cin, 8), device="cuda").to(torch.bfloat16)
      w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
      out = torch.zeros(N, H, W, ops.rup(Cout, 8), dtype=torch.bfloat16, device="cuda")
      flops = 2.0 * N * H * W * Cout * Cin * k * k
      line = f"{(N,Cin,Cout,k,H,W)}"
      for nw8 in COMBOS:
          L.lib.mfc_set_flag(23, nw8); L.lib.mfc_set_flag(24, 64)
          d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, 0, 0, L.BF16, N, H, W, x.shape[3], Cin, H, W, out.shape[3], Cout, H, W, k, k, -pad, -pad, 1, 1, 1, 0, 0, 0, N, 0, 0, 0)
          try:
              wp = ops.pack_weight(w, d, "fwd"); d.wp = wp.data_ptr()
              lay = L.conv_layout(d)
              op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
              t = time_op(op)
              line += f" | gemm={nw8}: NW{lay.NW} MT{lay.MT} KG{lay.KG} chunks{lay.nchunks} TAS{lay.TAS} {lay.lds_bytes//1024}K g{lay.grid} {t:6.1f}us {flops/t/1e6:4.0f}TF"
          except Exception as e:
              line += f" | gemm={nw8}: n/a {e}"
      print(line, flush=True)

if __name__ == "__main__":
    main()
