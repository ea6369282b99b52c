"""The launches the BENCHMARKED plans actually make (BASELINE.json configs[2]: T=3, B=8, 480x640, bf16, HRNet-w32 and -w48; and the per-GPU
share of configs[4]: T=5, B=8, 720x960, fp16, W48): every
distinct convolution node of that plan -- forward launch, data-gradient launch(es) and weight-gradient launch, with the plan's own
descriptors (so the geometry the library's search picks for N = 24 images at 120x160 ... 15x20, persistent ranges, XCD remap,
8-wave forms, split-K slices) -- is run once on seeded tensors and compared with PyTorch's CPU fp32 operators (F.conv2d + autograd)
applied to the SAME bf16-rounded inputs.  Tolerance 1.5e-2 of the output scale (bf16 output rounding; fp32 accumulation), as in
tests/test_gpu_ops.py, whose toy shapes need not reach these instantiations.
Reference operators replaced: models/hrnet.py:58-74 (BasicBlock convs), :95-115, :200-230, :334-351; multiframe_model.py:191-201.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1.5e-2                     # bf16; set per case by the test (fp16: 2e-3)
DT16 = torch.bfloat16            # 16-bit storage type of the case being run


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def bf(t):
    """round to the 16-bit storage type of the case (bf16 or fp16)"""
    return t.to(DT16).float()


GEOM = ("dtype", "N", "Hin", "Win", "Cin_p", "Cin", "Hout", "Wout", "Cout_p", "Cout", "Hl", "Wl", "TA", "TB", "dh0", "dw0", "in_stride",
        "out_sh", "out_sw", "out_oh", "out_ow", "in_relu", "images_per_group", "accumulate", "bn_mask_mode", "flags")


def conv_key(d):
    return tuple(getattr(d, n) for n in GEOM) + (bool(d.bias), bool(d.in_coef), bool(d.out_stats), bool(d.acc_src), bool(d.bn_y))


def _fields(s, skip):
    return tuple(getattr(s, n) for n, _ in s._fields_[skip:])


def node_key(ci):
    g = ci.wg
    return (conv_key(ci.fwd), tuple(conv_key(d) for d in ci.dgrad), _fields(g, 4) + (bool(g.in_coef),))


def unique_nodes(width, B=8, H=480, W=640, T=3, dtype="bf16", aux=False):
    import mfcnet_amd as mfc
    from mfcnet_amd.plan import Plan
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False, width=width, compute_dtype=dtype, optflow_inputs=aux,
                            depth_inputs=aux).train()
    pl = Plan(m, B, H, W, aux, aux, True, True, True, torch.device("cpu"), dry=True)
    seen = {}
    for op in pl.ops:
        if op[0] == "conv":
            seen.setdefault(node_key(op[3]), op[3])
    return list(seen.values())


def clone_desc(d):
    c = type(d)()
    C.memmove(C.byref(c), C.byref(d), C.sizeof(d))
    return c


def nhwc(t_nchw, Cp):
    """CPU fp32 NCHW (already rounded to the storage type) -> device 16-bit NHWC with zero channel padding"""
    N, Cc, H, W = t_nchw.shape
    out = torch.zeros(N, H, W, Cp, dtype=DT16)
    out[..., :Cc] = t_nchw.permute(0, 2, 3, 1).to(DT16)
    return out.cuda()


def nchw(t_nhwc, Cc):
    return t_nhwc[..., :Cc].permute(0, 3, 1, 2).float().cpu()


def run_node(ci, L, ops, seed):
    f = ci.fwd
    N, Hin, Win, Cin, Cout, k, s, pad = f.N, f.Hin, f.Win, f.Cin, f.Cout, ci.k, ci.stride, ci.pad
    G, ipg = f.N // f.images_per_group, f.images_per_group
    g = torch.Generator().manual_seed(seed)
    x = bf(torch.randn(N, Cin, Hin, Win, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k))
    bias = torch.randn(Cout, generator=g) if f.bias else None
    coef = None
    xa = x
    if f.in_coef:
        scale, shift = torch.rand(G, Cin, generator=g) + 0.5, torch.randn(G, Cin, generator=g) * 0.3
        coef = torch.zeros(G, 4, f.Cin_p)
        coef[:, 0, :Cin], coef[:, 1, :Cin] = scale, shift
        xa = x.view(G, ipg, Cin, Hin, Win) * scale.view(G, 1, Cin, 1, 1) + shift.view(G, 1, Cin, 1, 1)
        xa = bf((F.relu(xa) if f.in_relu else xa).reshape(N, Cin, Hin, Win))      # the kernels round the transformed operand to bf16
    xa = xa.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(xa, wr, bias, stride=s, padding=pad)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    what = f"{ci.wname} {Cin}->{Cout} k{k} s{s} {Hin}x{Win} N{N}"

    # ---- forward launch, the plan's descriptor with test tensors
    xd = nhwc(x, f.Cin_p)
    d = clone_desc(f)
    out = torch.zeros(N, f.Hout, f.Wout, f.Cout_p, dtype=DT16, device="cuda")
    stats = torch.zeros(L.STAT_REPLICAS, G, 2, f.Cout_p, dtype=torch.float64, device="cuda") if f.out_stats else None
    coef_d = coef.cuda() if coef is not None else None
    bias_d = bias.cuda() if bias is not None else None
    d.inp, d.out = xd.data_ptr(), out.data_ptr()
    d.bias = bias_d.data_ptr() if bias_d is not None else 0
    d.in_coef = coef_d.data_ptr() if coef_d is not None else 0
    d.out_stats = stats.data_ptr() if stats is not None else 0
    wp = ops.pack_weight(w.cuda(), d, "fwd")
    d.wp = wp.data_ptr()
    L.call(L.lib.mfc_conv2d_fwd, d)
    yo = nchw(out, Cout)
    assert relerr(yo, y.detach()) < TOL, ("fwd", what, relerr(yo, y.detach()))
    if f.Cout_p > Cout:
        assert float(out[..., Cout:].float().abs().max()) == 0.0, ("fwd padding", what)
    if stats is not None:
        st = stats.sum(0).cpu()
        yr = y.detach().view(G, ipg, Cout, f.Hout, f.Wout)
        assert relerr(st[:, 0, :Cout], yr.sum((1, 3, 4))) < 5 * TOL, ("stats sum", what)
        assert relerr(st[:, 1, :Cout], (yr * yr).sum((1, 3, 4))) < 5 * TOL, ("stats sumsq", what)

    # ---- data-gradient launch(es): 1 for stride 1, 4 output-parity classes for stride 2 -- with the epilogue fusions the plan gave them
    #      (mfc_conv_desc.acc_src: running sum read from another tensor; bn_y: BatchNorm-backward mask + statistics)
    dyd = nhwc(dy, f.Cout_p)
    if ci.dgrad:
        d0 = ci.dgrad[0]
        acc, from_src, fused, mode = d0.accumulate, bool(d0.acc_src), bool(d0.bn_y), d0.bn_mask_mode
        base = bf(torch.randn(N, Cin, Hin, Win, generator=g)) if acc else torch.zeros(N, Cin, Hin, Win)
        dx = nhwc(torch.zeros_like(base) if from_src else base, f.Cin_p)
        srcd = nhwc(base, f.Cin_p) if from_src else None
        ref = xa.grad + base
        if fused:
            ybn = bf(torch.randn(N, Cin, Hin, Win, generator=g))
            signsrc = bf(torch.randn(N, Cin, Hin, Win, generator=g))
            sc2, sh2 = torch.rand(G, Cin, generator=g) + 0.5, torch.randn(G, Cin, generator=g) * 0.3
            mean, rstd = torch.randn(G, Cin, generator=g) * 0.2, torch.rand(G, Cin, generator=g) + 0.5
            cf2 = torch.zeros(G, 4, f.Cin_p)
            cf2[:, 0, :Cin], cf2[:, 1, :Cin], cf2[:, 2, :Cin], cf2[:, 3, :Cin] = sc2, sh2, mean, rstd
            e5 = lambda t: t.view(G, 1, Cin, 1, 1)
            yv = ybn.view(G, ipg, Cin, Hin, Win)
            m = ((yv * e5(sc2) + e5(sh2)) > 0).float() if mode == 2 else (signsrc > 0).float().view(G, ipg, Cin, Hin, Win) if mode == 3 \
                else torch.ones(G, ipg, Cin, Hin, Win)
            gm_ref = ref.view(G, ipg, Cin, Hin, Win) * m
            s1_ref, s2_ref = gm_ref.sum((1, 3, 4)), (gm_ref * (yv - e5(mean)) * e5(rstd)).sum((1, 3, 4))
            ref = gm_ref.view(N, Cin, Hin, Win)
            yd2, cf2d = nhwc(ybn, f.Cin_p), cf2.cuda()
            sgn = nhwc(signsrc, f.Cin_p)
            bits = ((sgn.float() > 0).to(torch.int32).reshape(-1, 8) * torch.tensor([1 << e for e in range(8)], dtype=torch.int32, device="cuda")).sum(1).to(torch.uint8).contiguous()
            bst = torch.zeros(L.STAT_REPLICAS, G, 2, f.Cin_p, dtype=torch.float64, device="cuda")
        keep = []
        for dg in ci.dgrad:
            q = clone_desc(dg)
            q.inp, q.out, q.bias, q.in_coef, q.out_stats = dyd.data_ptr(), dx.data_ptr(), 0, 0, 0
            q.acc_src = srcd.data_ptr() if from_src else 0
            if fused:
                q.bn_y, q.bn_coef, q.out_stats, q.bn_bits = yd2.data_ptr(), cf2d.data_ptr(), bst.data_ptr(), bits.data_ptr() if mode == 3 else 0
            mode_s, cls = ("dgrad", (0, 0)) if s == 1 else (("dgrad_s2_all", (0, 0)) if (q.flags & L.CONV_S2_CLASSES) else ("dgrad_s2", (q.out_oh, q.out_ow)))
            wq = ops.pack_weight(w.cuda(), q, mode_s, cls)
            keep.append(wq)
            q.wp = wq.data_ptr()
            L.call(L.lib.mfc_conv2d_fwd, q)
        torch.cuda.synchronize()
        e = relerr(nchw(dx, Cin), ref)
        assert e < (2 if (acc or fused) else 1) * TOL, ("dgrad", what, e, fused, mode, from_src)
        if fused:
            st = bst.sum(0).cpu()
            assert relerr(st[:, 0, :Cin], s1_ref) < 5 * TOL, ("dgrad fused sum g*m", what)
            assert relerr(st[:, 1, :Cin], s2_ref) < 5 * TOL, ("dgrad fused sum g*m*yhat", what)

    # ---- weight-gradient launch (+ unpack of its partial-sum slices)
    wg = clone_desc(ci.wg)
    parts = L.wgrad_parts(wg)
    Co16, Ci16 = ops.rup(Cout, 16), ops.rup(Cin, 16)
    dwp = torch.zeros(parts * k * k * Co16 * Ci16, dtype=torch.float32, device="cuda")
    wg.x, wg.dy, wg.dwp = xd.data_ptr(), dyd.data_ptr(), dwp.data_ptr()
    wg.in_coef = coef_d.data_ptr() if coef_d is not None else 0
    L.call(L.lib.mfc_conv2d_wgrad, wg)
    dw = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device="cuda")
    ops._run_jobs([dict(src=dwp.data_ptr(), dst=dw.data_ptr(), Cout=Cout, Cin=Cin, KH=k, KW=k, Co16=Co16, Ci16=Ci16, nparts=parts)],
                  L.UnpackJob, L.lib.mfc_unpack_wgrad)
    e = relerr(dw.cpu(), wr.grad)
    assert e < TOL, ("wgrad", what, e, parts)
    return what


CASES = [(32, "bf16", 8, 480, 640, 3, False),          # BASELINE.json configs[2] (the metric's model)
         (48, "bf16", 8, 480, 640, 3, False),          # the same with the reference's widths
         (32, "bf16", 4, 480, 640, 3, True),           # the per-GPU share of configs[3]: B=4, depth + optical-flow inputs (N = 12: other geometries)
         (48, "fp16", 8, 720, 960, 5, False)]          # the per-GPU share of configs[4]: T=5, 720x960, fp16 (N = 40 images per launch)


@pytest.mark.parametrize("width,dtype,B,H,W,T,aux", CASES, ids=["w32-bf16-480x640", "w48-bf16-480x640", "w32-bf16-b4-flow-depth", "w48-fp16-t5-720x960"])
def test_every_conv_node_of_the_benchmarked_plan(width, dtype, B, H, W, T, aux):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mfcnet_amd import _lib as L, ops
    global DT16, TOL
    DT16, TOL = (torch.float16, 2e-3) if dtype == "fp16" else (torch.bfloat16, 1.5e-2)
    nodes = unique_nodes(width, B, H, W, T, dtype, aux)
    assert 30 <= len(nodes) <= 80
    done = [run_node(ci, L, ops, 100 + i) for i, ci in enumerate(nodes)]
    nfused = sum(1 for ci in nodes if ci.dgrad and ci.dgrad[0].bn_y)
    assert nfused >= (10 if width == 32 else 4)          # the fused data-gradient epilogues of the plan are among them
    print(f"w{width}: {len(done)} distinct convolution nodes (fwd + dgrad + wgrad; {nfused} with a fused BatchNorm-backward epilogue) "
          f"of the B={B} T={T} {H}x{W} {dtype} plan match CPU fp32")
