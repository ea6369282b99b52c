"""Per-record stream time of one training step (forward + backward programs), grouped by op kind and shape.

    python tools/profile_step.py [width] [reps] [B]

Uses mfc_program_profile (a HIP event between records); with reps>1 every record is launched `reps` times back to back
so the launch gap is amortised.  Prints the groups sorted by total time, with TFLOP/s for conv/wgrad records and GB/s for
the elementwise ones.
"""
import ctypes as C
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch  # noqa: E402
import mfcnet_amd as mfc  # noqa: E402
from mfcnet_amd import _lib as L  # noqa: E402

KIND = {1: "conv", 2: "wgrad", 3: "bnfin", 4: "combine", 5: "bnbwd_reduce", 6: "bnbwd_fin", 7: "bnbwd_apply", 8: "mask_add",
        9: "head_fwd", 10: "head_bwd", 11: "bias_grad", 12: "memset", 13: "pack", 14: "unpack", 15: "nchw2nhwc", 16: "nhwc2nchw", 17: "wgrad_batch", 18: "bnfin_batch"}


def describe(o, esz):
    k = o.kind
    if k == 1:
        d = o.u.conv
        fl = 2.0 * d.N * d.Hl * d.Wl * d.Cout * d.TA * d.TB * d.Cin
        tag = f"conv N{d.N} {d.Cin}->{d.Cout} k{d.TA}x{d.TB} in{d.Hin}x{d.Win} loc{d.Hl}x{d.Wl} s{d.in_stride} os{d.out_sh}" \
              f"{' xf' if d.in_coef else ''}{' st' if d.out_stats else ''}{' acc' if d.accumulate else ''}{' b' if d.bias else ''}"
        by = d.N * (d.Hin * d.Win * d.Cin_p / (d.out_sh * d.out_sw) + d.Hl * d.Wl * d.Cout_p) * esz
        return tag, fl, by
    if k == 2:
        d = o.u.wgrad
        fl = 2.0 * d.N * d.Hout * d.Wout * d.Cout * d.TA * d.TB * d.Cin
        tag = f"wgrad N{d.N} {d.Cin}->{d.Cout} k{d.TA}x{d.TB} in{d.Hin}x{d.Win} out{d.Hout}x{d.Wout} s{d.in_stride}{' xf' if d.in_coef else ''}"
        by = d.N * (d.Hin * d.Win * d.Cin_p + d.Hout * d.Wout * d.Cout_p) * esz
        return tag, fl, by
    if k == 4:
        d = o.u.combine
        by = d.N * d.out.H * d.out.W * d.C * esz
        src = 0
        for i in range(d.nsrc):
            src += d.N * d.src[i].H * d.src[i].W * d.C * esz
        res = ",".join(f"{d.src[i].H}" for i in range(d.nsrc))
        return f"combine N{d.N} C{d.C} {d.out.H}x{d.out.W} src[{res}]", 0.0, by + src
    if k in (5, 7):
        d = o.u.bnbwd
        n = d.N * d.g.H * d.g.W * d.C * esz
        mult = (2 if k == 5 else 3) + (1 if d.mask_mode else 0)
        return f"{KIND[k]} N{d.N} C{d.C} {d.g.H}x{d.g.W} mm{d.mask_mode}", 0.0, n * mult
    if k == 8:
        d = o.u.maskadd
        n = d.N * d.g.H * d.g.W * d.C * esz
        nd = d.N * d.dst.H * d.dst.W * d.C * esz
        return f"mask_add N{d.N} C{d.C} g{d.g.H}x{d.g.W} dst{d.dst.H}x{d.dst.W} mm{d.mask_mode} acc{d.accumulate}", 0.0, n * (1 + d.mask_mode) + nd * (1 + d.accumulate)
    if k == 12:
        return "memset", 0.0, float(o.u.raw.n)
    if k == 11:
        r = o.u.raw
        return f"bias_grad npix{r.n} Cp{r.i[1]}", 0.0, float(r.n * r.i[1] * esz)
    return KIND.get(k, str(k)), 0.0, 0.0


def main():
    for env, flag in (("MFC_CONV_NW8", 19), ("MFC_WGRAD_DMA48_X2", 48), ("MFC_CONV_RING48", 50), ("MFC_RING48_MT", 51), ("MFC_WGRAD_DMA48", 47), ("MFC_WGRAD_DMA_S2", 46), ("MFC_BNRED_MINPX", 42), ("MFC_BNRED_THREADS", 41), ("MFC_EW_ABLATE", 40), ("MFC_APPLYFIN_BLOCKS", 39), ("MFC_WGRAD_XF8", 38), ("MFC_BNRED_BLOCKS", 27), ("MFC_WGRAD_BLOCKS", 11)):    # tuning switches (include/mfcnet_hip.h)
        if os.environ.get(env):
            L.lib.mfc_set_flag(flag, int(os.environ[env]))
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    T, H, W = 3, 480, 640
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False, width=width, compute_dtype="bf16").to(dev).train()
    frames = [torch.randn(B, 3, H, W, device=dev) for _ in range(T)]
    mask = torch.randint(0, 5, (B, H, W), device=dev)
    for _ in range(2):
        model.zero_grad()
        loss, _ = mfc.mfc_loss(model(frames), mask)
        loss.backward()
    torch.cuda.synchronize()
    plan = model._plan if hasattr(model, "_plan") else next(iter(model._plans.values()))
    groups = defaultdict(lambda: [0.0, 0, 0.0, 0.0])
    total = 0.0
    for name, prog in (("fwd", plan.fwd_prog), ("bwd", plan.bwd_prog)):
        n = len(prog)
        ms = (C.c_float * n)()
        rc = L.lib.mfc_program_profile(prog, n, reps, ms, L.stream_ptr())
        assert rc == 0, rc
        t = sum(ms)
        total += t
        print(f"{name}: {n} records, {t:.2f} ms (event-bracketed{', x%d reps' % reps if reps > 1 else ''})")
        for i in range(n):
            tag, fl, by = describe(prog[i], 2)
            g = groups[(name, tag)]
            g[0] += ms[i]; g[1] += 1; g[2] += fl; g[3] += by
    bykind = defaultdict(float)
    for (name, tag), g in groups.items():
        bykind[(name, tag.split()[0])] += g[0]
    print("---- by kind ----")
    for k, v in sorted(bykind.items(), key=lambda kv: -kv[1]):
        print(f"{k[0]:4s} {k[1]:14s} {v:8.3f} ms  {100 * v / total:5.1f}%")
    print("---- by shape (all) ----")
    for (name, tag), g in sorted(groups.items(), key=lambda kv: -kv[1][0])[:400]:
        rate = f"{g[2] / g[0] / 1e9:7.0f} TF/s" if g[2] else (f"{g[3] / g[0] / 1e6:7.0f} GB/s" if g[3] else "")
        print(f"{name:4s} {g[0]:8.3f} ms {100 * g[0] / total:5.1f}% n={g[1]:4d} avg {1e3 * g[0] / g[1]:7.1f} us {rate:>14s}  {tag}")


if __name__ == "__main__":
    main()
