"""The element-wise launches the BENCHMARKED plan actually makes (BASELINE.json configs[2]: T=3, B=8, 480x640, bf16, HRNet-w32 and -w48): every
distinct mfc_combine_fwd / mfc_bnbwd_reduce / mfc_bnbwd_apply / mfc_mask_add record of that plan -- with the plan's own descriptor
(views, channel slices, mask modes, 1-bit masks, accumulate flags, fused finalize, separable-adjoint scratch, N = 24 images at
120x160 ... 15x20 and the 480-channel full-resolution tensors) -- is run once on seeded tensors and compared with PyTorch CPU fp32
operators applied to the same bf16-rounded inputs: F.interpolate(bilinear, align_corners=False) and its autograd adjoint, the
BatchNorm / ReLU backward formulas of include/mfcnet_hip.h.  Companion of tests/test_gpu_plan_kernels.py (the convolution launches).
Reference operators replaced: models/hrnet.py:62-72 (BN / ReLU / residual add), :245-260 (fuse sums with bilinear up-sampling), :464-469.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1.5e-2
DT = torch.bfloat16


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rnd(gen, N, Cp, H, W, scale=1.0):
    """fp32 NCHW tensor rounded to bf16, and its device NHWC image"""
    t = (torch.randn(N, Cp, H, W, generator=gen) * scale).to(DT).float()
    return t, t.permute(0, 2, 3, 1).contiguous().to(DT).cuda()


def to_nchw(dev_nhwc):
    return dev_nhwc.float().permute(0, 3, 1, 2).cpu()


def vkey(v):
    return (v.H, v.W, v.Cp, v.c_off, bool(v.coef), bool(v.ptr))


def sign_bits(dev_nhwc):
    b = (dev_nhwc.float() > 0).to(torch.int32).reshape(-1, 8)
    w = torch.tensor([1 << e for e in range(8)], dtype=torch.int32, device=b.device)
    return (b * w).sum(1).to(torch.uint8).contiguous()


def plan_records(width=32, B=8, H=480, W=640, T=3):
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd.plan import Plan
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False, width=width, compute_dtype="bf16").train()
    pl = Plan(m, B, H, W, False, False, True, True, True, torch.device("cpu"), dry=True)
    recs = {}
    for rec in list(pl.fwd) + list(pl.bwd):
        kind, d = rec[0], rec[1]
        if kind == L.OP_COMBINE:
            key = ("combine", vkey(d.out), tuple(vkey(d.src[i]) for i in range(d.nsrc)), d.nsrc, d.relu, d.N, d.C, d.images_per_group, bool(d.maskbits))
        elif kind in (L.OP_BNBWD_REDUCE, L.OP_BNBWD_APPLY):
            key = ("reduce" if kind == L.OP_BNBWD_REDUCE else "apply", vkey(d.g), vkey(d.y), vkey(d.mask), vkey(d.dy), d.mask_mode, d.N, d.C,
                   d.images_per_group, d.accumulate, bool(d.fin_dgamma), d.fin_C, d.fin_training, bool(d.bcoef), d.dy.ptr == d.g.ptr)
        elif kind == L.OP_MASK_ADD:
            key = ("mask_add", vkey(d.g), vkey(d.mask), vkey(d.dst), d.mask_mode, d.N, d.C, d.accumulate, bool(d.scratch))
        else:
            continue
        recs.setdefault(key, d)
    return recs


def coef_block(gen, G, Cp):
    """[G][4][Cp]: scale, shift, mean, rstd"""
    c = torch.zeros(G, 4, Cp)
    c[:, 0] = torch.rand(G, Cp, generator=gen) + 0.5
    c[:, 1] = torch.randn(G, Cp, generator=gen) * 0.3
    c[:, 2] = torch.randn(G, Cp, generator=gen) * 0.2
    c[:, 3] = torch.rand(G, Cp, generator=gen) + 0.5
    return c


def per_group(t, G):          # [N, C, H, W] -> [G, ipg, C, H, W]
    return t.view(G, t.shape[0] // G, *t.shape[1:])


def run_combine(d0, L, gen):
    d = type(d0).from_buffer_copy(d0)
    N, Cc, G = d.N, d.C, d.N // d.images_per_group
    H, W = d.out.H, d.out.W
    out_c, out_d = rnd(gen, N, d.out.Cp, H, W)
    keep = [out_d]
    acc = torch.zeros(N, Cc, H, W)
    for k in range(d.nsrc):
        s = d.src[k]
        sc, sd = rnd(gen, N, s.Cp, s.H, s.W)
        term = sc[:, s.c_off:s.c_off + Cc]
        if (s.H, s.W) != (H, W):
            term = F.interpolate(term, size=(H, W), mode="bilinear", align_corners=False)
        cf_ptr = 0
        if s.coef:
            cf = coef_block(gen, G, s.Cp)
            cfd = cf.cuda()
            keep.append(cfd)
            cf_ptr = cfd.data_ptr()
            tg = per_group(term, G)
            term = (tg * cf[:, 0, s.c_off:s.c_off + Cc].view(G, 1, Cc, 1, 1) + cf[:, 1, s.c_off:s.c_off + Cc].view(G, 1, Cc, 1, 1)).reshape(N, Cc, H, W)
        keep.append(sd)
        d.src[k] = L.View(sd.data_ptr(), cf_ptr, s.H, s.W, s.Cp, s.c_off)
        acc = acc + term
    ref = F.relu(acc) if d.relu == 1 else acc
    assert d.relu in (0, 1)
    d.out = L.View(out_d.data_ptr(), 0, H, W, d.out.Cp, d.out.c_off)
    bits = None
    if d.maskbits:
        bits = torch.full((N * H * W * d.out.Cp // 8,), 0xAA, dtype=torch.uint8, device="cuda")
        d.maskbits = bits.data_ptr()
    L.call(L.lib.mfc_combine_fwd, d)
    got = to_nchw(out_d)
    lo, hi = d.out.c_off, d.out.c_off + Cc
    assert relerr(got[:, lo:hi], ref) < TOL
    untouched = torch.cat([got[:, :lo], got[:, hi:]], 1)
    assert torch.equal(untouched, torch.cat([out_c[:, :lo], out_c[:, hi:]], 1))          # channels outside the slice keep their values
    if bits is not None:
        want = sign_bits(out_d).view(N * H * W, d.out.Cp // 8)
        have = bits.view(N * H * W, d.out.Cp // 8)
        assert torch.equal(have[:, lo // 8:hi // 8], want[:, lo // 8:hi // 8])


def bn_inputs(d, L, gen):
    """tensors of a BatchNorm-backward record and the reference g*m, yhat, scale"""
    N, Cc, G = d.N, d.C, d.N // d.images_per_group
    H, W = d.y.H, d.y.W
    assert (d.g.H, d.g.W) == (H, W)
    g_c, g_d = rnd(gen, N, d.g.Cp, H, W)
    y_c, y_d = rnd(gen, N, d.y.Cp, H, W)
    cf = coef_block(gen, G, d.y.Cp)
    cfd = cf.cuda()
    ys = slice(d.y.c_off, d.y.c_off + Cc)
    gv, yv = g_c[:, d.g.c_off:d.g.c_off + Cc], y_c[:, ys]
    e5 = lambda t: t[:, ys].view(G, 1, Cc, 1, 1)
    keep = [g_d, y_d, cfd]
    mview = L.View(0, 0, 0, 0, 0, 0)
    if d.mask_mode == 0:
        m = torch.ones(N, Cc, H, W)
    elif d.mask_mode == 2:
        m = ((per_group(yv, G) * e5(cf[:, 0]) + e5(cf[:, 1])) > 0).float().reshape(N, Cc, H, W)
    else:
        ms = d.mask
        assert (ms.H, ms.W) == (H, W)
        m_c, m_d = rnd(gen, N, ms.Cp, H, W)
        m = (m_c[:, ms.c_off:ms.c_off + Cc] > 0).float()
        if d.mask_mode == 3:
            bits = sign_bits(m_d)
            keep.append(bits)
            mview = L.View(bits.data_ptr(), 0, H, W, ms.Cp, ms.c_off)
        else:
            keep.append(m_d)
            mview = L.View(m_d.data_ptr(), 0, H, W, ms.Cp, ms.c_off)
    gm = gv * m
    yhat = ((per_group(yv, G) - e5(cf[:, 2])) * e5(cf[:, 3])).reshape(N, Cc, H, W)
    s1 = per_group(gm, G).sum((1, 3, 4))                          # [G, C]
    s2 = per_group(gm * yhat, G).sum((1, 3, 4))
    d.g = L.View(g_d.data_ptr(), 0, H, W, d.g.Cp, d.g.c_off)
    d.y = L.View(y_d.data_ptr(), cfd.data_ptr(), H, W, d.y.Cp, d.y.c_off)
    d.mask = mview
    return dict(N=N, C=Cc, G=G, H=H, W=W, g_c=g_c, g_d=g_d, gm=gm, yhat=yhat, s1=s1, s2=s2, cf=cf, ys=ys, keep=keep)


def run_reduce(d0, L, gen):
    d = type(d0).from_buffer_copy(d0)
    has_dy = bool(d0.dy.ptr)
    t = bn_inputs(d, L, gen)
    N, Cc, G, H, W = t["N"], t["C"], t["G"], t["H"], t["W"]
    bst = torch.zeros(L.STAT_REPLICAS, G, 2, d.y.Cp, dtype=torch.float64, device="cuda")
    d.bstats = bst.data_ptr()
    if has_dy:
        old_c, old_d = rnd(gen, N, d.dy.Cp, H, W)
        d.dy = L.View(old_d.data_ptr(), 0, H, W, d.dy.Cp, d.dy.c_off)
    else:
        d.dy = L.View(0, 0, 0, 0, 0, 0)
    L.call(L.lib.mfc_bnbwd_reduce, d)
    st = bst.sum(0).cpu().float()
    assert relerr(st[:, 0, t["ys"]], t["s1"]) < 5 * TOL and relerr(st[:, 1, t["ys"]], t["s2"]) < 5 * TOL
    if has_dy:
        lo = d.dy.c_off
        want = t["gm"] + (old_c[:, lo:lo + Cc] if d.accumulate else 0.0)
        assert relerr(to_nchw(old_d)[:, lo:lo + Cc], want) < 2 * TOL


def run_apply(d0, L, gen):
    d = type(d0).from_buffer_copy(d0)
    alias = d0.dy.ptr == d0.g.ptr
    t = bn_inputs(d, L, gen)
    N, Cc, G, H, W, ys = t["N"], t["C"], t["G"], t["H"], t["W"], t["ys"]
    count = float(d.images_per_group * H * W)
    train = d.fin_training if d.fin_dgamma else 1
    c1 = (t["s1"] / count) if train else torch.zeros_like(t["s1"])
    c2 = (t["s2"] / count) if train else torch.zeros_like(t["s2"])
    scale = t["cf"][:, 0, ys]
    ref = (scale.view(G, 1, Cc, 1, 1) * (per_group(t["gm"], G) - c1.view(G, 1, Cc, 1, 1) - per_group(t["yhat"], G) * c2.view(G, 1, Cc, 1, 1))).reshape(N, Cc, H, W)
    keep = []
    if d.fin_dgamma:
        bst = torch.zeros(L.STAT_REPLICAS, G, 2, d.y.Cp, dtype=torch.float64, device="cuda")
        bst[0, :, 0, ys], bst[L.STAT_REPLICAS - 1, :, 1, ys] = t["s1"].double().cuda(), t["s2"].double().cuda()
        dg, db = torch.full((d.fin_C,), 9.0, device="cuda"), torch.full((d.fin_C,), 9.0, device="cuda")
        d.bstats, d.bcoef, d.fin_dgamma, d.fin_dbeta, d.fin_count = bst.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), count
        keep += [bst, dg, db]
    else:
        bco = torch.zeros(G, 2, d.y.Cp)
        bco[:, 0, ys], bco[:, 1, ys] = c1, c2
        bcod = bco.cuda()
        d.bcoef = bcod.data_ptr()
        d.bstats = 0                # (unused with explicit coefficients; the dry plan's placeholder must not travel -- descriptor hardening refuses it)
        keep.append(bcod)
    if alias:
        out_d = t["g_d"]
        d.dy = L.View(out_d.data_ptr(), 0, H, W, d.g.Cp, d.dy.c_off)
    else:
        _, out_d = rnd(gen, N, d.dy.Cp, H, W)
        d.dy = L.View(out_d.data_ptr(), 0, H, W, d.dy.Cp, d.dy.c_off)
    L.call(L.lib.mfc_bnbwd_apply, d)
    lo = d.dy.c_off
    assert relerr(to_nchw(out_d)[:, lo:lo + Cc], ref) < 2 * TOL
    if d.fin_dgamma:
        nC = d.fin_C
        assert relerr(dg.cpu(), t["s2"].sum(0)[:nC]) < 1e-4 and relerr(db.cpu(), t["s1"].sum(0)[:nC]) < 1e-4


def run_mask_add(d0, L, gen):
    d = type(d0).from_buffer_copy(d0)
    N, Cc = d.N, d.C
    gH, gW, dH, dW = d.g.H, d.g.W, d.dst.H, d.dst.W
    g_c, g_d = rnd(gen, N, d.g.Cp, gH, gW)
    gv = g_c[:, d.g.c_off:d.g.c_off + Cc]
    keep = [g_d]
    mview = L.View(0, 0, 0, 0, 0, 0)
    if d.mask_mode == 0:
        m = torch.ones(N, Cc, gH, gW)
    else:
        ms = d.mask
        assert (ms.H, ms.W) == (gH, gW) and d.mask_mode in (1, 3)
        m_c, m_d = rnd(gen, N, ms.Cp, gH, gW)
        m = (m_c[:, ms.c_off:ms.c_off + Cc] > 0).float()
        src = sign_bits(m_d) if d.mask_mode == 3 else m_d
        keep.append(src)
        mview = L.View(src.data_ptr(), 0, gH, gW, ms.Cp, ms.c_off)
    gm = gv * m
    if (dH, dW) == (gH, gW):
        add = gm
    else:                                       # adjoint of the bilinear up-sampling dst -> g resolution
        x = torch.zeros(N, Cc, dH, dW, requires_grad=True)
        F.interpolate(x, size=(gH, gW), mode="bilinear", align_corners=False).backward(gm)
        add = x.grad
    old_c, old_d = rnd(gen, N, d.dst.Cp, dH, dW)
    lo = d.dst.c_off
    want = add + (old_c[:, lo:lo + Cc] if d.accumulate else 0.0)
    d.g = L.View(g_d.data_ptr(), 0, gH, gW, d.g.Cp, d.g.c_off)
    d.mask = mview
    d.dst = L.View(old_d.data_ptr(), 0, dH, dW, d.dst.Cp, lo)
    if d0.scratch:
        scr = torch.zeros(N * gH * dW * Cc, dtype=torch.float32, device="cuda")
        d.scratch = scr.data_ptr()
        keep.append(scr)
    L.call(L.lib.mfc_mask_add, d)
    got = to_nchw(old_d)
    assert relerr(got[:, lo:lo + Cc], want) < 2 * TOL
    assert torch.equal(torch.cat([got[:, :lo], got[:, lo + Cc:]], 1), torch.cat([old_c[:, :lo], old_c[:, lo + Cc:]], 1))


RUN = {"combine": run_combine, "reduce": run_reduce, "apply": run_apply, "mask_add": run_mask_add}


@pytest.mark.parametrize("width", [32, 48])
def test_every_elementwise_record_of_the_benchmarked_plan(width):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mfcnet_amd import _lib as L
    recs = plan_records(width)
    kinds = [k[0] for k in recs]
    assert kinds.count("combine") >= 10 and kinds.count("apply") >= 8 and kinds.count("reduce") >= 4 and kinds.count("mask_add") >= 8
    for i, (key, d) in enumerate(recs.items()):
        gen = torch.Generator().manual_seed(500 + i)
        try:
            RUN[key[0]](d, L, gen)
        except AssertionError as e:
            raise AssertionError(f"{key}: {e}") from e
        torch.cuda.synchronize()
    print(f"w{width}: {len(recs)} distinct element-wise records ({', '.join(f'{kinds.count(k)} {k}' for k in RUN)}) of the B=8 480x640 bf16 plan match CPU fp32")
