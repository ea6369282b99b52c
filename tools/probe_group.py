"""What would ONE launch for the independent branches of a HighResolutionModule (hrnet.py:242-243) buy?  Emulation with the existing
kernels: the four BasicBlock convolutions of one layer index (32 @ 120x160, 64 @ 60x80, 128 @ 30x40, 256 @ 15x20, N = 24) as chains of
L dependent launches per branch,
  serial       one stream, full grids (what a one-stream program does)
  lanes        one stream per branch, full grids (what the lane streams do: every launch asks for all 512 workgroup slots)
  partitioned  one stream per branch, every launch sized to its SHARE of the workgroup slots -- the co-residency a grouped launch
               would have (minus its single launch overhead)
    python tools/probe_group.py [--nw8] [--share a,b,c,d] [--layers L]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
from mfcnet_amd import _lib as L, ops

G = 3
BR = [(24, 32, 120, 160), (24, 64, 60, 80), (24, 128, 30, 40), (24, 256, 15, 20)]


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def set_grid(total):
    """persistent workgroups of the next launches: conv_igemm (flag 4; the 8-wave form halves it itself) and the ring kernel (flag 52)"""
    L.lib.mfc_set_flag(4, total)
    L.lib.mfc_set_flag(52, total)


def build(N, Cc, H, W, grid, xf):
    set_grid(grid)
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(N, H, W, Cc, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.zeros(N, H, W, Cc, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cc, Cc, 3, 3, device="cuda", generator=g) * 0.02
    stats = torch.zeros(L.STAT_REPLICAS, G, 2, Cc, dtype=torch.float64, device="cuda")
    coef = torch.rand(G, 4, Cc, device="cuda", generator=g) * 0.2 + 0.9
    coef[:, 1] = 0.0
    ds = []
    for (src, dst) in ((a, b), (b, a)):
        d = L.ConvDesc(src.data_ptr(), 0, dst.data_ptr(), 0, coef.data_ptr() if xf else 0, stats.data_ptr(), L.BF16, N, H, W, Cc, Cc, H, W, Cc, Cc, H, W,
                       3, 3, -1, -1, 1, 1, 1, 0, 0, 1 if xf else 0, N // G, 0, 0, 0)
        ds.append(d)
    wp = ops.pack_weight(w, ds[0], "fwd")
    for d in ds:
        d.wp = wp.data_ptr()
    lay = L.conv_layout(ds[0])
    return ds, lay, (a, b, w, stats, coef, wp)


def launch(d, st):
    rc = L.lib.mfc_conv2d_fwd(C.byref(d), C.c_void_p(st.cuda_stream))
    assert rc == 0, rc


def run(mode, descs, grids, streams, layers, reps):
    main = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event() for _ in streams]

    def once():
        if mode == "serial":
            for l in range(layers):
                for b, ds in enumerate(descs):
                    set_grid(grids[b])
                    launch(ds[l & 1], main)
            return
        fork = torch.cuda.Event(); fork.record(main)
        for b, ds in enumerate(descs):
            st = streams[b]
            st.wait_event(fork)
        for l in range(layers):
            for b, ds in enumerate(descs):
                set_grid(grids[b])
                launch(ds[l & 1], streams[b])
        for b, st in enumerate(streams):
            evs[b].record(st); main.wait_event(evs[b])

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0.record(main)
    for _ in range(reps):
        once()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    layers, reps = int(arg("--layers", 8)), int(arg("--reps", 20))
    nw8 = "--nw8" in sys.argv
    xf = "--xf" in sys.argv
    nbr = int(arg("--branches", 4))
    L.lib.mfc_set_flag(19, 90 if nw8 else 0)
    br = BR[:nbr]
    streams = [torch.cuda.Stream() for _ in br]
    flops = sum(2.0 * N * H * W * Cc * Cc * 9 for (N, Cc, H, W) in br) * layers
    shares = [s.strip() for s in arg("--share", "full;128,128,128,128;192,128,96,96;160,128,112,112;224,128,80,80;128,96,144,144").split(";")]
    print(f"# {nbr} branches x {layers} dependent 3x3 launches each, N=24, {'8-wave igemm allowed' if nw8 else '4-wave geometries only'}, xf={int(xf)}; us per layer index (all branches)")
    for sh in shares:
        grids = [512] * nbr if sh == "full" else [int(v) for v in sh.split(",")][:nbr]
        built = [build(*b, grid=g, xf=xf) for b, g in zip(br, grids)]
        descs = [x[0] for x in built]
        geo = " | ".join(f"C{b[1]} NW{x[1].NW} grid{x[1].grid}x{x[1].per_block} lds{x[1].lds_bytes // 1024}K" for b, x in zip(br, built))
        if sh == "full":
            # each branch alone, serial
            for b, ds in enumerate(descs):
                t = run("serial", [ds], [grids[b]], [], layers, reps) / layers
                print(f"alone C{br[b][1]:3d}: {t:7.1f} us per launch")
            t = run("serial", descs, grids, [], layers, reps) / layers
            print(f"serial full grids          : {t:7.1f} us  ({flops / layers / t / 1e6:5.0f} TF/s)   {geo}")
            t = run("lanes", descs, grids, streams, layers, reps) / layers
            print(f"{nbr} streams full grids       : {t:7.1f} us  ({flops / layers / t / 1e6:5.0f} TF/s)")
        else:
            t = run("lanes", descs, grids, streams, layers, reps) / layers
            print(f"{nbr} streams share {sh:18s}: {t:7.1f} us  ({flops / layers / t / 1e6:5.0f} TF/s)   {geo}")
            t = run("serial", descs, grids, [], layers, reps) / layers
            print(f"   (same grids, one stream)   : {t:7.1f} us")
    L.lib.mfc_set_flag(4, 512); L.lib.mfc_set_flag(52, 0); L.lib.mfc_set_flag(19, 90)


if __name__ == "__main__":
    main()
