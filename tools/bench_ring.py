"""conv3x3_ring_kernel vs conv_igemm_kernel on the BasicBlock shapes: stream time per launch (mfc_program_profile, 10 back-to-back
launches) and the largest difference between the two kernels' outputs (same inputs, same packed weights' source).
    python tools/bench_ring.py [--quick]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mfcnet_amd import _lib as L, ops
from sweep_conv2 import time_op

G = 3
SHAPES = [(24, 32, 120, 160), (24, 64, 60, 80), (12, 32, 120, 160), (24, 32, 45, 60), (24, 64, 23, 30)]
VARIANTS = ["plain", "stats", "xf+stats", "acc", "bn2", "acc+src+bn3"]


def build(N, Cc, H, W, variant, ring):
    L.lib.mfc_set_flag(30, 1 if ring else 0)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(N, H, W, Cc, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(Cc, Cc, 3, 3, device="cuda", generator=g) * 0.05
    out = (torch.randn(N, H, W, Cc, device="cuda", generator=g) if "acc" in variant else torch.zeros(N, H, W, Cc, device="cuda")).to(torch.bfloat16)
    keep = [x, w, out]
    d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, 0, 0, L.BF16, N, H, W, Cc, Cc, H, W, Cc, Cc, H, W, 3, 3, -1, -1, 1, 1, 1, 0, 0, 0, N // G, 0, 0, 0)
    stats = None
    if "stats" in variant or "bn" in variant:
        stats = torch.zeros(L.STAT_REPLICAS, G, 2, Cc, dtype=torch.float64, device="cuda"); keep.append(stats)
        d.out_stats = stats.data_ptr()
    if "xf" in variant:
        coef = torch.rand(G, 4, Cc, device="cuda", generator=g) + 0.25; coef[:, 1] -= 0.8; keep.append(coef)
        d.in_coef, d.in_relu = coef.data_ptr(), 1
    if "acc" in variant:
        d.accumulate = 1
    if "src" in variant:
        src = torch.randn(N, H, W, Cc, device="cuda", generator=g).to(torch.bfloat16); keep.append(src)
        d.acc_src = src.data_ptr()
    if "bn" in variant:
        y = torch.randn(N, H, W, Cc, device="cuda", generator=g).to(torch.bfloat16)
        cf = torch.rand(G, 4, Cc, device="cuda", generator=g) + 0.5; cf[:, 1] -= 1.0; cf[:, 2] -= 1.0
        keep += [y, cf]
        d.bn_y, d.bn_coef = y.data_ptr(), cf.data_ptr()
        d.bn_mask_mode = 2 if "bn2" in variant else 3
        if d.bn_mask_mode == 3:
            bits = torch.randint(0, 256, (N * H * W * Cc // 8,), device="cuda", generator=g).to(torch.uint8); keep.append(bits)
            d.bn_bits = bits.data_ptr()
    if variant not in ("plain", "stats", "xf+stats"):
        d.flags = L.CONV_WANT_FA
    lay = L.conv_layout(d)
    wp = ops.pack_weight(w, d, "fwd"); keep.append(wp)
    d.wp = wp.data_ptr()
    return d, out, stats, lay, keep


def main():
    quick = "--quick" in sys.argv
    only = [a for a in sys.argv[1:] if a.startswith("C")]
    for (N, Cc, H, W) in SHAPES[:2] if quick else SHAPES:
        if only and f"C{Cc}" not in only:
            continue
        flops = 2.0 * N * H * W * Cc * Cc * 9
        byts = 2.0 * N * H * W * Cc * 2
        for v in VARIANTS:
            res = {}
            for ring in (0, 1):
                d, out, stats, lay, keep = build(N, Cc, H, W, v, ring)
                o0 = out.clone()
                L.call(L.lib.mfc_conv2d_fwd, d)
                torch.cuda.synchronize()
                o1 = out.float().clone()
                s1 = stats.sum(0).clone() if stats is not None else None
                op = L.Op(); op.kind = L.OP_CONV; op.u.conv = d
                if "acc" in v and not d.acc_src:
                    t = float("nan")          # (an in-place accumulate cannot be repeated)
                    try:
                        out.copy_(o0); t = time_op(op)
                    except Exception:
                        pass
                else:
                    t = time_op(op)
                res[ring] = (o1, s1, t, lay)
            (oa, sa, ta, la), (ob, sb, tb, lb) = res[0], res[1]
            err = float((oa - ob).abs().max() / (oa.abs().max() + 1e-9))
            serr = float((sa - sb).abs().max() / (sa.abs().max() + 1e-9)) if sa is not None else 0.0
            print(f"N{N} C{Cc} {H}x{W} {v:12s} igemm {ta:6.1f} us ({flops/ta/1e6:4.0f} TF, {byts/ta/1e3:5.0f} GB/s) | ring MT{lb.MT} grid{lb.grid}x{lb.per_block} "
                  f"{tb:6.1f} us ({flops/tb/1e6:4.0f} TF, {byts/tb/1e3:5.0f} GB/s) | max diff {err:.2e} stats {serr:.2e}", flush=True)
    L.lib.mfc_set_flag(30, 1)


if __name__ == "__main__":
    main()
