"""Per-kernel parity tests: every C-ABI entry point against PyTorch CPU fp32 operators
(F.conv2d / autograd / F.batch_norm / F.interpolate / torch.optim.Adam) on the same seeded inputs.

Tolerances: fp32 kernels 2e-4 max-abs relative to the output scale (they use exact-fp32 MFMA, the
difference is summation order); bf16 kernels are checked against the fp32 operator applied to the SAME
bf16-rounded inputs, 1.5e-2 relative to the output scale (bf16 output rounding, fp32 accumulation).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import mfcnet_amd
    from mfcnet_amd import _lib, ops
    return mfcnet_amd, _lib, ops


DT = [torch.float32, torch.bfloat16, torch.float16]
H16 = (torch.bfloat16, torch.float16)          # 16-bit storage: same kernels, layouts and fusions
TOL = {torch.float32: 2e-4, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}


def rnd(dtype, *shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(dtype).float() if dtype in H16 else x


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def pack_sign_bits(t_nhwc):
    """uint8 [N*H*W*Cp/8]: bit e of byte (pixel*Cp + c)/8 = t[pixel][c + e] > 0 (the layout mfc_combine_fwd writes)."""
    b = (t_nhwc.float() > 0).to(torch.int32).reshape(-1, 8)
    w = torch.tensor([1 << e for e in range(8)], dtype=torch.int32, device=b.device)
    return (b * w).sum(1).to(torch.uint8).contiguous()


CONV_CASES = [  # N, Cin, Cout, k, s, H, W
    (2, 48, 48, 3, 1, 24, 40), (2, 3, 64, 3, 2, 64, 96), (2, 64, 64, 3, 2, 32, 48), (3, 96, 192, 3, 2, 15, 20),
    (2, 384, 384, 3, 1, 15, 20), (2, 256, 64, 1, 1, 16, 24), (1, 720, 720, 1, 1, 12, 20), (2, 720, 5, 1, 1, 16, 24),
    (2, 15, 15, 11, 1, 32, 48), (1, 22, 15, 11, 1, 24, 40), (2, 15, 5, 1, 1, 16, 16), (1, 48, 96, 3, 2, 23, 30),
    (2, 32, 32, 3, 1, 30, 40), (1, 128, 256, 3, 2, 30, 40), (2, 192, 48, 1, 1, 8, 10),
    (3, 64, 256, 1, 1, 40, 56), (2, 32, 96, 3, 1, 24, 40),      # single-stage launches with several cout blocks (weights of all blocks resident)
    (2, 480, 480, 1, 1, 16, 24), (2, 256, 200, 1, 1, 16, 16), (1, 136, 520, 1, 1, 16, 32), (3, 720, 720, 1, 1, 16, 32), (2, 64, 256, 1, 1, 16, 16),   # big 1x1: the plain-GEMM kernel (bf16)
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(M, case, dtype):
    _, L, ops = M
    N, Cin, Cout, k, s, H, W = case
    x = rnd(dtype, N, Cin, H, W, seed=1)
    w = rnd(dtype, Cout, Cin, k, k, seed=2, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(torch.float32, Cout, seed=3)
    ref = F.conv2d(x, w, b, stride=s, padding=k // 2)
    y = ops.conv2d(ops.to_nhwc(x, dtype), w.cuda(), k, s, bias=b.cuda())
    out = ops.to_nchw(y, Cout).cpu()
    assert relerr(out, ref) < TOL[dtype]
    assert float(y[..., Cout:].float().abs().max() if y.shape[3] > Cout else 0.0) == 0.0   # channel padding stays zero


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(48, 96, 3, 20, 28), (64, 256, 1, 40, 56), (64, 256, 1, 16, 32), (136, 520, 1, 16, 16)])   # (last two: plain-GEMM kernel in bf16)
def test_conv_fused_bn_relu_input_and_stats(M, dtype, cfg):
    """consumer-side BN-apply+ReLU (zero padding AFTER the transform) and the epilogue statistics."""
    _, L, ops = M
    Cin, Cout, k, H, W = cfg
    N, G = 4, 2
    x = rnd(dtype, N, Cin, H, W, seed=4)
    w = rnd(dtype, Cout, Cin, k, k, seed=5, scale=0.05)
    scale = torch.rand(G, Cin) + 0.5
    shift = torch.randn(G, Cin) * 0.3
    coef = torch.zeros(G, 4, Cin)
    coef[:, 0], coef[:, 1] = scale, shift
    xa = torch.cat([F.relu(x[g * 2:(g + 1) * 2] * scale[g].view(1, -1, 1, 1) + shift[g].view(1, -1, 1, 1)) for g in range(G)])
    if dtype in H16:
        xa = xa.to(dtype).float()
    ref = F.conv2d(xa, w, None, padding=k // 2)
    stats = torch.zeros(L.STAT_REPLICAS, G, 2, Cout, dtype=torch.float64, device="cuda")
    y = ops.conv2d(ops.to_nhwc(x, dtype), w.cuda(), k, 1, in_coef=coef.cuda(), in_relu=True, ipg=2, stats=stats)
    out = ops.to_nchw(y, Cout).cpu()
    assert relerr(out, ref) < TOL[dtype]
    st = stats.sum(0).cpu()
    for g in range(G):
        r = ref[g * 2:(g + 1) * 2]
        assert relerr(st[g, 0], r.sum((0, 2, 3))) < 5 * TOL[dtype]
        assert relerr(st[g, 1], (r * r).sum((0, 2, 3))) < 5 * TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(480, 480, 16, 16), (136, 520, 16, 32), (256, 720, 8, 32)])
def test_conv1x1_bias_and_stats_without_input_transform(M, dtype, cfg):
    """last_layer[0] (hrnet.py:334-351): 1x1 conv + bias whose epilogue feeds the following BatchNorm's statistics; in bf16 this
    shape class runs in the plain-GEMM kernel (conv_gemm1x1.hip), in fp32 in the general one."""
    _, L, ops = M
    Cin, Cout, H, W = cfg
    N, G = 6, 3
    x = rnd(dtype, N, Cin, H, W, seed=14)
    w = rnd(dtype, Cout, Cin, 1, 1, seed=15, scale=1.0 / np.sqrt(Cin))
    b = rnd(torch.float32, Cout, seed=16)
    ref = F.conv2d(x, w, b)
    stats = torch.zeros(L.STAT_REPLICAS, G, 2, ops.rup(Cout, 8), dtype=torch.float64, device="cuda")
    y = ops.conv2d(ops.to_nhwc(x, dtype), w.cuda(), 1, 1, bias=b.cuda(), ipg=2, stats=stats)
    assert relerr(ops.to_nchw(y, Cout).cpu(), ref) < TOL[dtype]
    st = stats.sum(0).cpu()
    for g in range(G):
        r = ref[g * 2:(g + 1) * 2]
        assert relerr(st[g, 0, :Cout], r.sum((0, 2, 3))) < 5 * TOL[dtype]
        assert relerr(st[g, 1, :Cout], (r * r).sum((0, 2, 3))) < 5 * TOL[dtype]
    assert float(st[:, :, Cout:].abs().max() if st.shape[2] > Cout else 0.0) == 0.0


DG_CASES = [(2, 48, 48, 3, 1, 24, 40), (2, 64, 64, 3, 2, 32, 48), (1, 48, 96, 3, 2, 23, 30), (2, 96, 48, 1, 1, 15, 20),
            (1, 22, 15, 11, 1, 24, 40), (2, 192, 384, 3, 2, 30, 40), (2, 256, 96, 3, 2, 16, 24),
            (2, 256, 192, 1, 1, 16, 16), (1, 480, 480, 1, 1, 16, 32), (2, 256, 64, 1, 1, 16, 16)]      # (1x1 with >= 128 channels: plain-GEMM kernel, incl. accumulate)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", DG_CASES)
def test_conv_dgrad(M, case, dtype):
    _, L, ops = M
    N, Cin, Cout, k, s, H, W = case
    x = rnd(dtype, N, Cin, H, W, seed=6).requires_grad_(True)
    w = rnd(dtype, Cout, Cin, k, k, seed=7, scale=1.0 / np.sqrt(Cin * k * k))
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = rnd(dtype, *y.shape, seed=8)
    y.backward(dy)
    dx = ops.conv2d_dgrad(ops.to_nhwc(dy, dtype), w.cuda(), k, s, (H, W))
    assert relerr(ops.to_nchw(dx, Cin).cpu(), x.grad) < TOL[dtype]
    # accumulate flag
    dx2 = ops.conv2d_dgrad(ops.to_nhwc(dy, dtype), w.cuda(), k, s, (H, W), accumulate_into=dx.clone())
    assert relerr(ops.to_nchw(dx2, Cin).cpu(), 2 * x.grad) < 2 * TOL[dtype]


S2_MERGED_CASES = [(2, 64, 64, 24, 32), (1, 48, 96, 23, 30), (2, 192, 384, 30, 40), (3, 32, 128, 15, 21), (2, 256, 96, 16, 24), (1, 3, 64, 33, 47)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", S2_MERGED_CASES)
def test_stride2_dgrad_as_one_launch_over_the_parity_classes(M, case, dtype):
    """MFC_CONV_S2_CLASSES (round 4): the data gradient of a 3x3 / stride-2 convolution (fuse-layer down-sampling chains hrnet.py:211-230,
    transitions :361-386, the stem :426-431) as ONE launch whose units are (parity class, cout block, tile) with per-class weight images
    (2x2 taps, the missing ones zero) -- against autograd's data gradient, odd sizes included, and with the accumulate option."""
    _, L, ops = M
    N, Cin, Cout, H, W = case
    x = rnd(dtype, N, Cin, H, W, seed=6).requires_grad_(True)
    w = rnd(dtype, Cout, Cin, 3, 3, seed=7, scale=1.0 / np.sqrt(Cin * 9))
    y = F.conv2d(x, w, None, stride=2, padding=1)
    dy = rnd(dtype, *y.shape, seed=8)
    y.backward(dy)
    dx = ops.conv2d_dgrad(ops.to_nhwc(dy, dtype), w.cuda(), 3, 2, (H, W), merged_s2=True)
    assert relerr(ops.to_nchw(dx, Cin).cpu(), x.grad) < TOL[dtype]
    if dx.shape[3] > Cin:
        assert float(dx[..., Cin:].float().abs().max()) == 0.0
    dx4 = ops.conv2d_dgrad(ops.to_nhwc(dy, dtype), w.cuda(), 3, 2, (H, W))                    # the four-launch form
    assert relerr(dx.float(), dx4.float()) < TOL[dtype]
    dx2 = ops.conv2d_dgrad(ops.to_nhwc(dy, dtype), w.cuda(), 3, 2, (H, W), accumulate_into=dx.clone(), merged_s2=True)
    assert relerr(ops.to_nchw(dx2, Cin).cpu(), 2 * x.grad) < 2 * TOL[dtype]


FUSED_DG_CASES = [  # N, G, Cin, Cout, k, s, H, W, mask mode, accumulate from another tensor
    (6, 3, 32, 32, 3, 1, 24, 40, 2, False), (6, 3, 32, 32, 3, 1, 24, 40, 3, True), (4, 2, 64, 64, 3, 1, 17, 21, 3, True),
    (4, 2, 64, 64, 3, 1, 17, 21, 0, False), (6, 3, 64, 128, 3, 2, 30, 40, 2, False), (3, 3, 128, 128, 3, 1, 15, 20, 3, True),
    (6, 3, 64, 32, 1, 1, 20, 28, 2, False), (3, 1, 32, 64, 3, 2, 23, 30, 2, True),
    # HRNet-W48's channel counts: 3- / 6-tile cout blocks, fused through the non-transposed epilogue (round 4)
    (6, 3, 48, 48, 3, 1, 24, 40, 3, True), (4, 2, 96, 96, 3, 1, 17, 21, 2, False), (3, 3, 192, 192, 3, 1, 15, 20, 3, True),
    (3, 1, 384, 384, 3, 1, 8, 12, 0, False), (6, 3, 48, 96, 3, 2, 30, 40, 2, False), (2, 2, 96, 192, 3, 2, 15, 21, 3, True),
    # wide 1x1 data gradients (layer1's 64 -> 256 bottleneck convolutions, hrnet.py:77-111): the plain-GEMM kernel's fused epilogue
    (6, 3, 256, 64, 1, 1, 16, 24, 3, True), (6, 3, 256, 64, 1, 1, 16, 24, 0, False), (3, 3, 256, 128, 1, 1, 16, 16, 2, False),
    (2, 1, 512, 64, 1, 1, 16, 16, 3, True)]


@pytest.mark.parametrize("merged", [False, True], ids=["", "s2-one-launch"])
@pytest.mark.parametrize("dt16", H16, ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", FUSED_DG_CASES)
def test_dgrad_fused_bn_backward_reduce(dt16, M, case, merged):
    """mfc_conv_desc.bn_y / acc_src: a data-gradient launch that masks its result, accumulates the BatchNorm-backward statistics
    (sum g*m, sum g*m*yhat per group and channel) in its epilogue and reads its running sum from another tensor -- against autograd's
    data gradient with the mask / the sums applied on the CPU (the work mfc_bnbwd_reduce otherwise does in a second sweep)."""
    _, L, ops = M
    N, G, Cin, Cout, k, s, H, W, mode, acc = case
    if merged and s != 2:
        pytest.skip("merged form exists for stride-2 data gradients only")
    ipg = N // G
    x = rnd(dt16, N, Cin, H, W, seed=51).requires_grad_(True)
    w = rnd(dt16, Cout, Cin, k, k, seed=52, scale=1.0 / np.sqrt(Cin * k * k))
    yo = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = rnd(dt16, *yo.shape, seed=53)
    yo.backward(dy)
    base = rnd(dt16, N, Cin, H, W, seed=54) if acc else torch.zeros(N, Cin, H, W)
    g_ref = x.grad + base                                      # the completed gradient
    ybn = rnd(dt16, N, Cin, H, W, seed=55)           # pre-BN tensor of the BatchNorm whose backward is fused
    g = torch.Generator().manual_seed(56)
    scale, shift = torch.rand(G, Cin, generator=g) + 0.5, torch.randn(G, Cin, generator=g) * 0.3
    mean, rstd = torch.randn(G, Cin, generator=g) * 0.2, torch.rand(G, Cin, generator=g) + 0.5
    coef = torch.stack([scale, shift, mean, rstd], 1).contiguous()          # [G][4][Cin]
    e5 = lambda t: t.view(G, 1, Cin, 1, 1)
    yv = ybn.view(G, ipg, Cin, H, W)
    signsrc = rnd(dt16, N, Cin, H, W, seed=57)
    if mode == 2:
        m = ((yv * e5(scale) + e5(shift)) > 0).float()
    elif mode == 3:
        m = (signsrc > 0).float().view(G, ipg, Cin, H, W)
    else:
        m = torch.ones(G, ipg, Cin, H, W)
    gm_ref = g_ref.view(G, ipg, Cin, H, W) * m
    s1_ref = gm_ref.sum((1, 3, 4))
    s2_ref = (gm_ref * (yv - e5(mean)) * e5(rstd)).sum((1, 3, 4))
    # ---- the launch(es)
    dyd, yd, srcd = ops.to_nhwc(dy, dt16), ops.to_nhwc(ybn, dt16), ops.to_nhwc(base, dt16)
    bits = pack_sign_bits(ops.to_nhwc(signsrc, dt16))
    dx = torch.zeros(N, H, W, Cin, dtype=dt16, device="cuda")
    bstats = torch.zeros(L.STAT_REPLICAS, G, 2, Cin, dtype=torch.float64, device="cuda")
    coef_d = coef.cuda()
    Ho, Wo, pad = yo.shape[2], yo.shape[3], k // 2
    Cop = -(-Cout // 8) * 8                   # (channel pitch of the gradient the launch reads)
    keep, launched = [], 0
    classes = [(0, 0)] if (s == 1 or merged) else [(a, b) for a in range(2) for b in range(2)]
    for (ph, pw) in classes:
        if merged:       # MFC_CONV_S2_CLASSES: the four parity classes in one launch
            d = L.ConvDesc(dyd.data_ptr(), 0, dx.data_ptr(), 0, 0, bstats.data_ptr(), ops.dt_of(dx), N, Ho, Wo, Cop, Cout, H, W, Cin, Cin, (H + 1) // 2, (W + 1) // 2,
                           2, 2, 0, 0, 1, 2, 2, 0, 0, 0, ipg, 1 if acc else 0, 0, 0)
            d.flags = L.CONV_S2_CLASSES
            mode_s, cls = "dgrad_s2_all", (0, 0)
        elif s == 1:
            d = L.ConvDesc(dyd.data_ptr(), 0, dx.data_ptr(), 0, 0, bstats.data_ptr(), ops.dt_of(dx), N, Ho, Wo, Cop, Cout, H, W, Cin, Cin, H, W,
                           k, k, -(k - 1 - pad), -(k - 1 - pad), 1, 1, 1, 0, 0, 0, ipg, 1 if acc else 0, 0, 0)
            mode_s, cls = "dgrad", (0, 0)
        else:
            ta, _, dh0 = ops.s2_class(k, pad, ph)
            tb, _, dw0 = ops.s2_class(k, pad, pw)
            Hl, Wl = (H - ph + 1) // 2, (W - pw + 1) // 2
            d = L.ConvDesc(dyd.data_ptr(), 0, dx.data_ptr(), 0, 0, bstats.data_ptr(), ops.dt_of(dx), N, Ho, Wo, Cop, Cout, H, W, Cin, Cin, Hl, Wl,
                           ta, tb, dh0, dw0, 1, 2, 2, ph, pw, 0, ipg, 1 if acc else 0, 0, 0)
            mode_s, cls = "dgrad_s2", (ph, pw)
        d.flags |= L.CONV_WANT_FA
        if L.conv_layout(d).fa != 1:
            pytest.skip("no fusable geometry for this shape")
        d.acc_src = srcd.data_ptr() if acc else 0
        d.bn_y, d.bn_coef, d.bn_mask_mode, d.bn_bits = yd.data_ptr(), coef_d.data_ptr(), mode, bits.data_ptr() if mode == 3 else 0
        wp = ops.pack_weight(w.cuda(), d, mode_s, cls)
        keep.append(wp)
        d.wp = wp.data_ptr()
        L.call(L.lib.mfc_conv2d_fwd, d)
        launched += 1
    torch.cuda.synchronize()
    gm = ops.to_nchw(dx, Cin).cpu().view(G, ipg, Cin, H, W)
    assert relerr(gm, gm_ref) < 2 * TOL[dt16]
    st = bstats.sum(0).cpu()
    assert relerr(st[:, 0], s1_ref) < 5 * TOL[dt16]
    assert relerr(st[:, 1], s2_ref) < 5 * TOL[dt16]
    # a launch that cannot take the fusion must refuse it rather than drop it
    d2 = L.ConvDesc(dyd.data_ptr(), 0, dx.data_ptr(), 0, 0, bstats.data_ptr(), L.F32, N, Ho, Wo, Cop, Cout, H, W, Cin, Cin, H, W,
                    k, k, -(k - 1 - pad), -(k - 1 - pad), 1, 1, 1, 0, 0, 0, ipg, 0, 0, 0)
    d2.bn_y, d2.bn_coef, d2.wp = yd.data_ptr(), coef_d.data_ptr(), keep[0].data_ptr()
    assert L.lib.mfc_conv2d_fwd(C.byref(d2), L.stream_ptr()) < 0


WG_CASES = [(2, 48, 48, 3, 1, 24, 40), (2, 3, 64, 3, 2, 64, 96), (2, 96, 96, 3, 1, 30, 40), (3, 96, 192, 3, 2, 15, 20),
            (2, 384, 384, 3, 1, 15, 20), (2, 256, 64, 1, 1, 16, 24), (1, 720, 720, 1, 1, 12, 20), (2, 720, 5, 1, 1, 16, 24),
            (2, 15, 15, 11, 1, 32, 48), (1, 22, 15, 11, 1, 24, 40), (1, 48, 96, 3, 2, 23, 30), (2, 32, 64, 3, 1, 17, 21),
            (4, 480, 480, 1, 1, 16, 16), (3, 136, 520, 1, 1, 16, 32), (2, 720, 720, 1, 1, 24, 32), (4, 256, 128, 1, 1, 16, 32)]      # (1x1, >= 128 channels: split-K GEMM kernel in bf16)


@pytest.mark.parametrize("tr", [1, 0])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(M, case, dtype, tr):
    _, L, ops = M
    if dtype == torch.float32 and tr == 0:
        pytest.skip("tr flag only affects bf16")
    N, Cin, Cout, k, s, H, W = case
    x = rnd(dtype, N, Cin, H, W, seed=9)
    w = rnd(dtype, Cout, Cin, k, k, seed=10, scale=0.05).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = rnd(dtype, *y.shape, seed=11)
    y.backward(dy)
    L.lib.mfc_set_flag(1, tr)
    try:
        dw = ops.conv2d_wgrad(ops.to_nhwc(x, dtype), ops.to_nhwc(dy, dtype), Cout, Cin, k, s)
    finally:
        L.lib.mfc_set_flag(1, 1)
    assert relerr(dw.cpu(), w.grad) < TOL[dtype]


DMA_CASES = [(3, 32, 32, 15, 20), (2, 64, 32, 23, 30), (2, 32, 64, 5, 6), (4, 96, 96, 12, 40), (2, 32, 32, 4, 8), (6, 64, 64, 17, 9)]


@pytest.mark.parametrize("dt16", H16, ids=["bf16", "fp16"])
@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("case", DMA_CASES)
def test_wgrad_dma_kernel(dt16, M, case, xf):
    """conv_wgrad_dma.hip (3x3 / stride 1, channel counts that are multiples of 32, bf16): ragged images (partial 4x8 sub-tiles on both
    axes, images narrower than a tile, every tile an edge tile), several channel blocks, with and without the fused BatchNorm + ReLU
    of the producer (two statistic groups); against autograd on the same bf16-rounded operands, and against the register-staged
    kernel it replaces (mfc_set_flag(29, 0))."""
    _, L, ops = M
    N, Cin, Cout, H, W = case
    G = 2 if N % 2 == 0 else 1
    x = rnd(dt16, N, Cin, H, W, seed=41)
    xa, coef = x, None
    if xf:
        scale, shift = torch.rand(G, Cin) + 0.5, torch.randn(G, Cin) * 0.3
        coef = torch.zeros(G, 4, Cin)
        coef[:, 0], coef[:, 1] = scale, shift
        xa = F.relu(x.view(G, N // G, Cin, H, W) * scale.view(G, 1, Cin, 1, 1) + shift.view(G, 1, Cin, 1, 1)).reshape(N, Cin, H, W)
        xa = xa.to(dt16).float()
    w = rnd(dt16, Cout, Cin, 3, 3, seed=42, scale=0.05).requires_grad_(True)
    y = F.conv2d(xa, w, None, padding=1)
    dy = rnd(dt16, *y.shape, seed=43)
    y.backward(dy)
    res = []
    for flag in (1, 0):
        L.lib.mfc_set_flag(29, flag)
        try:
            res.append(ops.conv2d_wgrad(ops.to_nhwc(x, dt16), ops.to_nhwc(dy, dt16), Cout, Cin, 3, 1,
                                        in_coef=coef.cuda() if xf else None, in_relu=xf, ipg=N // G).cpu())
        finally:
            L.lib.mfc_set_flag(29, 1)
    assert relerr(res[0], w.grad) < TOL[dt16]
    assert relerr(res[0], res[1]) < 1e-4            # same products, fp32 accumulation in another order


S2_CASES = [  # N, Cin, Cout, H, W  (3x3 / stride 2 / pad 1)
    (2, 32, 64, 24, 40), (2, 32, 32, 23, 31), (1, 64, 128, 9, 17), (2, 48, 96, 21, 30), (4, 48, 48, 16, 16), (2, 96, 192, 7, 5),
    (2, 32, 256, 30, 40), (1, 64, 64, 2, 3), (2, 128, 64, 18, 34),
]


@pytest.mark.parametrize("dt16", H16, ids=["bf16", "fp16"])
@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("case", S2_CASES)
def test_wgrad_dma_stride2_kernel(dt16, M, case, xf):
    """conv_wgrad_dma_s2.hip (3x3 / stride 2: the down-sampling convolutions of the fuse layers and transitions, hrnet.py:200-230): odd and even
    image sizes (the last input row / column is or is not read), images smaller than a sub-tile, channel counts that are multiples of 16
    but not of 32 (HRNet-W48: blocks that stick out of the tensor), several channel blocks, with and without the producer's fused
    BatchNorm + ReLU; against autograd on the same rounded operands and against the register-staged kernel it replaces (mfc_set_flag(46, 0))."""
    _, L, ops = M
    N, Cin, Cout, H, W = case
    G = 2 if N % 2 == 0 else 1
    x = rnd(dt16, N, Cin, H, W, seed=51)
    xa, coef = x, None
    if xf:
        scale, shift = torch.rand(G, Cin) + 0.5, torch.randn(G, Cin) * 0.3
        coef = torch.zeros(G, 4, Cin)
        coef[:, 0], coef[:, 1] = scale, shift
        xa = F.relu(x.view(G, N // G, Cin, H, W) * scale.view(G, 1, Cin, 1, 1) + shift.view(G, 1, Cin, 1, 1)).reshape(N, Cin, H, W)
        xa = xa.to(dt16).float()
    w = rnd(dt16, Cout, Cin, 3, 3, seed=52, scale=0.05).requires_grad_(True)
    y = F.conv2d(xa, w, None, stride=2, padding=1)
    dy = rnd(dt16, *y.shape, seed=53)
    y.backward(dy)
    res = []
    for flag in (1, 0):
        L.lib.mfc_set_flag(46, flag)
        try:
            res.append(ops.conv2d_wgrad(ops.to_nhwc(x, dt16), ops.to_nhwc(dy, dt16), Cout, Cin, 3, 2,
                                        in_coef=coef.cuda() if xf else None, in_relu=xf, ipg=N // G).cpu())
        finally:
            L.lib.mfc_set_flag(46, 1)
    assert relerr(res[0], w.grad) < TOL[dt16]
    assert relerr(res[0], res[1]) < 1e-4            # same products, fp32 accumulation in another order


C48_CASES = [(2, 48, 48, 21, 37), (1, 48, 48, 3, 5), (4, 48, 48, 16, 16), (2, 48, 144, 9, 12), (2, 144, 48, 10, 13), (6, 48, 48, 33, 24)]      # N, Cin, Cout, H, W


@pytest.mark.parametrize("dt16", H16, ids=["bf16", "fp16"])
@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("case", C48_CASES)
def test_wgrad_dma48_kernel(dt16, M, case, xf):
    """conv_wgrad_dma48.hip (3x3 / stride 1 with channel counts that are multiples of 48 but not of 32 -- the 48 -> 48 BasicBlocks of HRNet-W48,
    hrnet.py:297-333): ragged and tiny images, one to three 48 x 48 channel blocks, one to three statistic groups, with and without the
    producer's fused BatchNorm + ReLU; against autograd on the same rounded operands and against the register-staged kernel it replaces
    (mfc_set_flag(47, 0))."""
    _, L, ops = M
    N, Cin, Cout, H, W = case
    G = 3 if N % 3 == 0 else (2 if N % 2 == 0 else 1)
    x = rnd(dt16, N, Cin, H, W, seed=61)
    xa, coef = x, None
    if xf:
        scale, shift = torch.rand(G, Cin) + 0.5, torch.randn(G, Cin) * 0.3
        coef = torch.zeros(G, 4, Cin)
        coef[:, 0], coef[:, 1] = scale, shift
        xa = F.relu(x.view(G, N // G, Cin, H, W) * scale.view(G, 1, Cin, 1, 1) + shift.view(G, 1, Cin, 1, 1)).reshape(N, Cin, H, W)
        xa = xa.to(dt16).float()
    w = rnd(dt16, Cout, Cin, 3, 3, seed=62, scale=0.05).requires_grad_(True)
    y = F.conv2d(xa, w, None, padding=1)
    dy = rnd(dt16, *y.shape, seed=63)
    y.backward(dy)
    res = []
    for flag in (1, 0):
        L.lib.mfc_set_flag(47, flag)
        try:
            res.append(ops.conv2d_wgrad(ops.to_nhwc(x, dt16), ops.to_nhwc(dy, dt16), Cout, Cin, 3, 1,
                                        in_coef=coef.cuda() if xf else None, in_relu=xf, ipg=N // G).cpu())
        finally:
            L.lib.mfc_set_flag(47, 1)
    assert relerr(res[0], w.grad) < TOL[dt16]
    assert relerr(res[0], res[1]) < 1e-4            # same products, fp32 accumulation in another order


def test_wgrad_fused_input_transform(M):
    _, L, ops = M
    N, Cin, Cout, H, W = 2, 48, 48, 16, 24
    x = rnd(torch.float32, N, Cin, H, W, seed=12)
    scale, shift = torch.rand(1, Cin) + 0.5, torch.randn(1, Cin) * 0.3
    coef = torch.zeros(1, 4, Cin)
    coef[:, 0], coef[:, 1] = scale, shift
    xa = F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    w = rnd(torch.float32, Cout, Cin, 3, 3, seed=13, scale=0.05).requires_grad_(True)
    y = F.conv2d(xa, w, None, padding=1)
    dy = rnd(torch.float32, *y.shape, seed=14)
    y.backward(dy)
    dw = ops.conv2d_wgrad(ops.to_nhwc(x), ops.to_nhwc(dy), Cout, Cin, 3, 1, in_coef=coef.cuda(), in_relu=True)
    assert relerr(dw.cpu(), w.grad) < 2e-4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(64, 256, 16, 32), (136, 520, 16, 16), (256, 64, 8, 32)])
def test_wgrad_1x1_fused_input_transform_groups(M, dtype, cfg):
    """1x1 weight gradient with the producer's BatchNorm + ReLU applied to x on the fly, one coefficient set per statistics group
    (layer1's 64 -> 256 convolutions, hrnet.py:95-115); in bf16 this is the split-K GEMM kernel (wgrad_gemm1x1.hip)."""
    _, L, ops = M
    Cin, Cout, H, W = cfg
    N, G = 6, 3
    x = rnd(dtype, N, Cin, H, W, seed=21)
    scale, shift = torch.rand(G, Cin) + 0.5, torch.randn(G, Cin) * 0.3
    coef = torch.zeros(G, 4, ops.rup(Cin, 8))
    coef[:, 0, :Cin], coef[:, 1, :Cin] = scale, shift
    xa = torch.cat([F.relu(x[g * 2:(g + 1) * 2] * scale[g].view(1, -1, 1, 1) + shift[g].view(1, -1, 1, 1)) for g in range(G)])
    if dtype in H16:
        xa = xa.to(dtype).float()
    w = rnd(dtype, Cout, Cin, 1, 1, seed=22, scale=0.05).requires_grad_(True)
    y = F.conv2d(xa, w, None)
    dy = rnd(dtype, *y.shape, seed=23)
    y.backward(dy)
    dw = ops.conv2d_wgrad(ops.to_nhwc(x, dtype), ops.to_nhwc(dy, dtype), Cout, Cin, 1, 1, in_coef=coef.cuda(), in_relu=True, ipg=2)
    assert relerr(dw.cpu(), w.grad) < TOL[dtype]


def test_bn_finalize_matches_batch_norm(M):
    _, L, ops = M
    G, ipg, Cc, H, W = 3, 2, 48, 10, 12
    y = torch.randn(G * ipg, Cc, H, W, generator=torch.Generator().manual_seed(15)) * 2 + 0.5
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc)
    rm, rv = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    outs = [F.batch_norm(y[g * ipg:(g + 1) * ipg], rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5) for g in range(G)]
    stats = torch.zeros(L.STAT_REPLICAS, G, 2, Cc, dtype=torch.float64)
    for g in range(G):
        yy = y[g * ipg:(g + 1) * ipg]
        stats[g % L.STAT_REPLICAS, g, 0] = yy.sum((0, 2, 3))
        stats[(g + 5) % L.STAT_REPLICAS, g, 1] = (yy * yy).sum((0, 2, 3))
    stats, coef = stats.cuda(), torch.zeros(G, 4, Cc, device="cuda")
    dg, db, drm, drv = gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    d = L.BnFinDesc(stats.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), drm.data_ptr(), drv.data_ptr(),
                    nbt.data_ptr(), Cc, Cc, G, 1, float(ipg * H * W), 1e-5, 0.1)
    L.call(L.lib.mfc_bn_finalize, d)
    coef = coef.cpu()
    for g in range(G):
        yy = y[g * ipg:(g + 1) * ipg]
        mine = yy * coef[g, 0].view(1, -1, 1, 1) + coef[g, 1].view(1, -1, 1, 1)
        assert float((mine - outs[g]).abs().max()) < 2e-5
    assert float((drm.cpu() - rm_ref).abs().max()) < 1e-6 and float((drv.cpu() - rv_ref).abs().max()) < 1e-5
    assert int(nbt) == G
    # eval mode: coefficients from running stats
    d.training = 0
    L.call(L.lib.mfc_bn_finalize, d)
    ref = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, False, 0.1, 1e-5)
    c = coef.cpu() if False else torch.empty(0)
    coef2 = torch.zeros(G, 4, Cc, device="cuda")
    d.coef = coef2.data_ptr()
    L.call(L.lib.mfc_bn_finalize, d)
    c = coef2.cpu()
    assert float((y * c[0, 0].view(1, -1, 1, 1) + c[0, 1].view(1, -1, 1, 1) - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("dtype", DT)
def test_combine_residual_bilinear(M, dtype):
    """fuse-layer sum: identity + BN(up-sampled conv output) + BN(same-res conv output), ReLU (hrnet.py:245-260)."""
    _, L, ops = M
    N, Cc, H, W = 2, 48, 24, 40
    a = rnd(dtype, N, Cc, H, W, seed=16)
    b = rnd(dtype, N, Cc, H // 2, W // 2, seed=17)
    c = rnd(dtype, N, Cc, H, W, seed=18)
    e = rnd(dtype, N, Cc, 5, 7, seed=19)               # non-integer scale
    cb = torch.zeros(1, 4, Cc); cb[0, 0] = torch.rand(Cc) + 0.5; cb[0, 1] = torch.randn(Cc) * 0.2
    cc = torch.zeros(1, 4, Cc); cc[0, 0] = torch.rand(Cc) + 0.5; cc[0, 1] = torch.randn(Cc) * 0.2
    aff = lambda t, cf: t * cf[0, 0].view(1, -1, 1, 1) + cf[0, 1].view(1, -1, 1, 1)
    ref = F.relu(a + F.interpolate(aff(b, cb), size=(H, W), mode="bilinear", align_corners=False) + aff(c, cc)
                 + F.interpolate(e, size=(H, W), mode="bilinear", align_corners=False))
    ta, tb, tc, te = (ops.to_nhwc(t, dtype) for t in (a, b, c, e))
    out = torch.zeros_like(ta)
    cbd, ccd = cb.cuda(), cc.cuda()
    d = L.CombineDesc()
    d.out = ops.view(out)
    d.src[0], d.src[1], d.src[2], d.src[3] = ops.view(ta), ops.view(tb, cbd), ops.view(tc, ccd), ops.view(te)
    d.nsrc, d.relu, d.dtype, d.N, d.C, d.images_per_group = 4, 1, ops.dt_of(ta), N, Cc, N
    bits = torch.zeros(out.numel() // 8, dtype=torch.uint8, device="cuda")
    if dtype in H16:
        d.maskbits = bits.data_ptr()
    L.call(L.lib.mfc_combine_fwd, d)
    assert relerr(ops.to_nchw(out, Cc).cpu(), ref) < TOL[dtype]
    if dtype in H16:      # the 1-bit image of the output's sign, for the backward pass
        assert torch.equal(bits, pack_sign_bits(out))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", DT)
def test_bn_backward(M, dtype, mode):
    """BN(+ReLU / +residual ReLU) backward: reduce -> finalize -> apply vs autograd (train-mode BN, 2 groups)."""
    _, L, ops = M
    G, ipg, Cc, H, W = 2, 2, 48, 12, 20
    N = G * ipg
    y = rnd(dtype, N, Cc, H, W, seed=20).requires_grad_(True)
    res = rnd(dtype, N, Cc, H, W, seed=21)
    gamma = (torch.rand(Cc) + 0.5).requires_grad_(True)
    beta = (torch.randn(Cc) * 0.2).requires_grad_(True)
    zs = [F.batch_norm(y[g * ipg:(g + 1) * ipg], None, None, gamma, beta, True, 0.1, 1e-5) for g in range(G)]
    z = torch.cat(zs)
    a = z if mode == 0 else (F.relu(z + res) if mode in (1, 3) else F.relu(z))
    ga = rnd(dtype, N, Cc, H, W, seed=22)
    a.backward(ga)
    coef = torch.zeros(G, 4, Cc)
    for g in range(G):
        yy = y.detach()[g * ipg:(g + 1) * ipg]
        mean, var = yy.mean((0, 2, 3)), yy.var((0, 2, 3), unbiased=False)
        rstd = 1 / torch.sqrt(var + 1e-5)
        coef[g, 0], coef[g, 1], coef[g, 2], coef[g, 3] = gamma.detach() * rstd, beta.detach() - mean * gamma.detach() * rstd, mean, rstd
    ty, tg = ops.to_nhwc(y.detach(), dtype), ops.to_nhwc(ga, dtype)
    ta = ops.to_nhwc(a.detach(), dtype)
    coefd = coef.cuda()
    bst = torch.zeros(L.STAT_REPLICAS, G, 2, Cc, dtype=torch.float64, device="cuda")
    bco = torch.zeros(G, 2, Cc, device="cuda")
    dgam, dbet = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    dy = torch.zeros_like(ty)
    d = L.BnBwdDesc()
    d.g, d.y, d.dy = ops.view(tg), ops.view(ty, coefd), ops.view(dy)
    if mode == 1:
        d.mask = ops.view(ta)
    if mode == 3:            # the 1-bit image of the same mask (bf16 only)
        if dtype not in H16:
            pytest.skip("1-bit masks are a bf16 feature")
        bits = pack_sign_bits(ta)
        d.mask = L.View(bits.data_ptr(), 0, ta.shape[1], ta.shape[2], ta.shape[3], 0)
    d.bstats, d.bcoef = bst.data_ptr(), bco.data_ptr()
    d.mask_mode, d.dtype, d.N, d.C, d.images_per_group, d.accumulate = mode, ops.dt_of(ty), N, Cc, ipg, 0
    tol = TOL[dtype] * 2
    # the reduce pass optionally writes the masked gradient g*m (= the residual branch's gradient): plain and accumulating
    gm = torch.full_like(ty, 7.0)
    gm1 = torch.ones_like(ty)
    dr = L.BnBwdDesc.from_buffer_copy(d)
    dr.dy, dr.accumulate = ops.view(gm), 0
    L.call(L.lib.mfc_bnbwd_reduce, dr)
    gm_ref = ga * (a.detach() > 0) if mode != 0 else ga
    assert relerr(ops.to_nchw(gm, Cc).cpu(), gm_ref) < tol
    bst.zero_()
    dr.dy, dr.accumulate = ops.view(gm1), 1
    L.call(L.lib.mfc_bnbwd_reduce, dr)
    assert relerr(ops.to_nchw(gm1, Cc).cpu() - 1.0, gm_ref) < tol
    f = L.BnBwdFinDesc(bst.data_ptr(), bco.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), Cc, Cc, G, 1, float(ipg * H * W))
    L.call(L.lib.mfc_bnbwd_finalize, f)
    L.call(L.lib.mfc_bnbwd_apply, d)
    assert relerr(ops.to_nchw(dy, Cc).cpu(), y.grad) < tol
    assert relerr(dgam.cpu(), gamma.grad) < tol and relerr(dbet.cpu(), beta.grad) < tol
    # the finalize fused into the apply launch (every workgroup sums the replicas itself) gives the same dy and parameter gradients
    dy3 = torch.zeros_like(ty)
    dgam3, dbet3 = torch.full_like(dgam, 9.0), torch.full_like(dbet, 9.0)
    dfu = L.BnBwdDesc.from_buffer_copy(d)
    dfu.dy, dfu.bcoef = ops.view(dy3), 0
    dfu.fin_dgamma, dfu.fin_dbeta, dfu.fin_C, dfu.fin_training, dfu.fin_count = dgam3.data_ptr(), dbet3.data_ptr(), Cc, 1, float(ipg * H * W)
    L.call(L.lib.mfc_bnbwd_apply, dfu)
    assert relerr(ops.to_nchw(dy3, Cc).cpu(), y.grad) < tol
    assert relerr(dgam3.cpu(), dgam.cpu()) < 1e-5 and relerr(dbet3.cpu(), dbet.cpu()) < 1e-5
    # ... and the apply pass may read g*m back instead of g and the mask
    dy2 = torch.zeros_like(ty)
    da = L.BnBwdDesc.from_buffer_copy(d)
    da.g, da.dy, da.mask_mode = ops.view(gm), ops.view(dy2), 0
    L.call(L.lib.mfc_bnbwd_apply, da)
    assert relerr(ops.to_nchw(dy2, Cc).cpu(), y.grad) < tol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("hw", [((12, 20), (24, 40)), ((5, 7), (24, 40)), ((3, 5), (24, 40)), ((23, 30), (92, 120)), ((24, 40), (24, 40))])
def test_mask_add_adjoint_bilinear(M, dtype, hw):
    _, L, ops = M
    (hs, ws), (H, W) = hw
    N, Cc = 2, 48
    src = rnd(dtype, N, Cc, hs, ws, seed=23).requires_grad_(True)
    other = rnd(dtype, N, Cc, H, W, seed=24)
    up = F.interpolate(src, size=(H, W), mode="bilinear", align_corners=False) if (hs, ws) != (H, W) else src
    out = F.relu(up + other)
    g = rnd(dtype, N, Cc, H, W, seed=25)
    out.backward(g)
    tg, tout = ops.to_nhwc(g, dtype), ops.to_nhwc(out.detach(), dtype)
    dst = torch.ones(N, hs, ws, Cc, dtype=dtype, device="cuda")
    d = L.MaskAddDesc()
    d.g, d.mask, d.dst = ops.view(tg), ops.view(tout), ops.view(dst)
    d.mask_mode, d.dtype, d.N, d.C, d.accumulate = 1, ops.dt_of(tg), N, Cc, 1
    L.call(L.lib.mfc_mask_add, d)
    assert relerr(ops.to_nchw(dst, Cc).cpu() - 1.0, src.grad) < TOL[dtype] * 2
    if (hs, ws) != (H, W):          # the separable two-pass form (fp32 workspace [N, H, ws, C]) gives the same adjoint
        dst2 = torch.ones(N, hs, ws, Cc, dtype=dtype, device="cuda")
        scratch = torch.empty(N * H * ws * Cc, dtype=torch.float32, device="cuda")
        d.dst, d.scratch = ops.view(dst2), scratch.data_ptr()
        L.call(L.lib.mfc_mask_add, d)
        assert relerr(ops.to_nchw(dst2, Cc).cpu() - 1.0, src.grad) < TOL[dtype] * 2
    if dtype in H16:     # the mask as a 1-bit image (mask_mode 3), every kernel form
        bits = pack_sign_bits(tout)
        for use_scratch in ((False, True) if (hs, ws) != (H, W) else (False,)):
            dst3 = torch.ones(N, hs, ws, Cc, dtype=dtype, device="cuda")
            d3 = L.MaskAddDesc()
            d3.g, d3.dst = ops.view(tg), ops.view(dst3)
            d3.mask = L.View(bits.data_ptr(), 0, H, W, tout.shape[3], 0)
            d3.mask_mode, d3.dtype, d3.N, d3.C, d3.accumulate = 3, ops.dt_of(tg), N, Cc, 1
            if use_scratch:
                d3.scratch = scratch.data_ptr()
            L.call(L.lib.mfc_mask_add, d3)
            assert relerr(ops.to_nchw(dst3, Cc).cpu() - 1.0, src.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("flow,depth,shape", [(False, False, (2, 3, 12, 20)), (True, True, (2, 3, 12, 20)), (False, True, (2, 3, 12, 20)),
                                              (False, False, (8, 3, 120, 160)), (True, True, (4, 3, 120, 160)), (False, False, (1, 5, 180, 240))],
                         ids=["rgb", "flow+depth", "depth", "bench-b8-480x640", "cfg3-b4-480x640-flow+depth", "cfg4-t5-720x960"])
def test_head_gather_fwd_bwd(M, dtype, flow, depth, shape):
    """x4 up-sample + temporal concat (hrnet.py:473-474, multiframe_model.py:459-469) and its adjoint -- toy shapes and the launches of the
    benchmarked configurations (BASELINE.json configs[2], [3], [4])."""
    _, L, ops = M
    B, T, Hs, Ws = shape
    nc = 5
    H, W = 4 * Hs, 4 * Ws
    lg = rnd(dtype, T * B, nc, Hs, Ws, seed=26).requires_grad_(True)
    fl = [rnd(torch.float32, B, 2, H, W, seed=30 + i) for i in range(T - 1)] if flow else []
    dp = [rnd(torch.float32, B, 1, H, W, seed=40 + i) for i in range(T)] if depth else []
    ups = [F.interpolate(lg[t * B:(t + 1) * B], size=(H, W), mode="bilinear", align_corners=False) for t in range(T)]
    ref = torch.cat(ups + fl + dp, 1)
    cin = ref.shape[1]
    Cp = (cin + 7) // 8 * 8
    tl = ops.to_nhwc(lg.detach(), dtype)
    xh = torch.full((B, H, W, Cp), 7.0, dtype=dtype, device="cuda")
    fld, dpd = [f.cuda() for f in fl], [p.cuda() for p in dp]
    d = L.HeadDesc()
    d.logits, d.xh = tl.data_ptr(), xh.data_ptr()
    for i, f in enumerate(fld):
        d.flow[i] = f.data_ptr()
    for i, p in enumerate(dpd):
        d.depth[i] = p.data_ptr()
    d.dtype, d.B, d.T, d.nc, d.Hs, d.Ws, d.Lp, d.H, d.W, d.Cp, d.warp = ops.dt_of(tl), B, T, nc, Hs, Ws, 8, H, W, Cp, 0
    L.call(L.lib.mfc_head_gather_fwd, d)
    got = ops.to_nchw(xh, cin).cpu()
    want = ref.detach().to(dtype).float() if dtype in H16 else ref.detach()
    assert relerr(got, want) < TOL[dtype]
    assert float(xh[..., cin:].float().abs().max() if Cp > cin else 0) == 0
    g = rnd(dtype, B, cin, H, W, seed=27)
    ref.backward(g)
    tg = ops.to_nhwc(g, dtype)
    dl = torch.zeros_like(tl)
    d.xh = tg.data_ptr()
    L.check(L.lib.mfc_head_gather_bwd(C.byref(d), dl.data_ptr(), L.stream_ptr()))
    assert relerr(ops.to_nchw(dl, nc).cpu(), lg.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("depth", [False, True])
def test_head_gather_warp_fwd_bwd(M, depth):
    """MultiFrameNetBasic flow warp (multiframe_model.py:89-170, incl. the 576x720 grid quirk) fused into the head gather:
    forward against F.grid_sample on the up-sampled logits, backward against autograd."""
    _, L, ops = M
    B, T, nc, Hs, Ws = 2, 3, 5, 12, 20
    H, W = 4 * Hs, 4 * Ws
    dtype = torch.float32
    lg = rnd(dtype, T * B, nc, Hs, Ws, seed=50).requires_grad_(True)
    fl = [rnd(dtype, B, 2, H, W, seed=51 + i, scale=3.0) for i in range(T - 1)]
    dp = [rnd(dtype, B, 1, H, W, seed=55 + i).abs() for i in range(T)] if depth else []
    gy, gx = torch.meshgrid(torch.arange(0, 576), torch.arange(0, 720), indexing="ij")
    grid = torch.stack((2.0 * gx / 719 - 1.0, 2.0 * gy / 575 - 1.0), 0).float().unsqueeze(0)[:, :, :H, :W]

    def warp(m, flow):
        g = (grid + torch.stack((flow[:, 0] / ((W - 1) / 2.0), flow[:, 1] / ((H - 1) / 2.0)), 1)).permute(0, 2, 3, 1)
        return F.grid_sample(m, g, mode="bilinear", padding_mode="zeros", align_corners=True)

    ups = [F.interpolate(lg[t * B:(t + 1) * B], size=(H, W), mode="bilinear", align_corners=False) for t in range(T)]
    segs = [ups[0]] + [warp(ups[t], fl[t - 1]) for t in range(1, T)]
    deps = ([dp[0]] + [warp(dp[t], fl[t - 1]) for t in range(1, T)]) if depth else []
    ref = torch.cat(segs + deps, 1)
    cin = ref.shape[1]
    Cp = (cin + 7) // 8 * 8
    tl = ops.to_nhwc(lg.detach(), dtype)
    xh = torch.zeros(B, H, W, Cp, dtype=dtype, device="cuda")
    fld, dpd = [f.cuda() for f in fl], [p.cuda() for p in dp]
    d = L.HeadDesc()
    d.logits, d.xh = tl.data_ptr(), xh.data_ptr()
    for i, f in enumerate(fld):
        d.flow[i] = f.data_ptr()
    for i, p in enumerate(dpd):
        d.depth[i] = p.data_ptr()
    d.dtype, d.B, d.T, d.nc, d.Hs, d.Ws, d.Lp, d.H, d.W, d.Cp, d.warp = ops.dt_of(tl), B, T, nc, Hs, Ws, 8, H, W, Cp, 1
    L.call(L.lib.mfc_head_gather_fwd, d)
    assert relerr(ops.to_nchw(xh, cin).cpu(), ref.detach()) < 2e-4
    g = rnd(dtype, B, cin, H, W, seed=60)
    ref.backward(g)
    tg = ops.to_nhwc(g, dtype)
    dl = torch.zeros_like(tl)
    scratch = torch.empty(B * H * W * (T - 1) * nc, dtype=torch.float32, device="cuda")
    d.xh = tg.data_ptr()
    d.depth[7] = scratch.data_ptr()
    L.check(L.lib.mfc_head_gather_bwd(C.byref(d), dl.data_ptr(), L.stream_ptr()))
    assert relerr(ops.to_nchw(dl, nc).cpu(), lg.grad) < 5e-4


@pytest.mark.parametrize("shape", [(2, 40, 56), (8, 480, 640), (1, 720, 960)], ids=["toy", "bench-b8-480x640", "720x960"])
def test_loss_fwd_bwd(M, shape):
    """fused log_softmax + weighted NLL + soft-Jaccard and its gradient (src/loss.py:31-63) against the oracle, incl. the benchmarked batch (2.46 M
    pixels: the fp64 sum cells keep the 26 sums exact to fp32 rounding) -- and bit-identical when repeated."""
    mfc, L, ops = M
    from oracle import mfcnet_oracle as O
    B, H, W = shape
    nc = 5
    g = torch.Generator().manual_seed(28)
    logits = (torch.randn(B, nc, H, W, generator=g) * 2).requires_grad_(True)
    target = torch.randint(0, nc, (B, H, W), generator=g)
    tot, parts = O.total_loss(logits, target, nc)
    tot.backward()
    lg = logits.detach().cuda().requires_grad_(True)
    loss, acc = mfc.mfc_loss(lg, target.cuda())
    loss.backward()
    acc = acc.cpu()
    assert abs(float(acc[26]) - float(parts["loss_nll"])) < 1e-5
    assert abs(float(acc[27]) - float(parts["loss_soft_jaccard"])) < 1e-5
    assert abs(float(loss) - float(tot)) < 1e-5
    assert relerr(lg.grad.cpu(), logits.grad) < 1e-4
    lg2 = logits.detach().cuda().requires_grad_(True)
    loss2, acc2 = mfc.mfc_loss(lg2, target.cuda())
    loss2.backward()
    assert torch.equal(acc2.cpu(), acc) and torch.equal(lg2.grad, lg.grad)


def test_loss_out_of_range_labels_are_ignored_and_counted(M):
    """A target outside [0, nc) (a 255 'ignore' value, a corrupt mask) must not index class_w / logits out of bounds (nn.NLLLoss
    raises on it, src/loss.py:31-43): the kernels drop those pixels from the NLL term, count them in acc[29], and
    mfc_loss(check_labels=True) raises like the reference.  In-range pixels keep their exact contribution: the loss equals the
    reference's NLL with ignore_index on the bad pixels (+ the Jaccard term with those pixels in no class)."""
    mfc, L, ops = M
    import torch.nn.functional as F
    B, nc, H, W = 2, 5, 24, 40
    g = torch.Generator().manual_seed(29)
    logits = (torch.randn(B, nc, H, W, generator=g) * 2)
    target = torch.randint(0, nc, (B, H, W), generator=g)
    bad = target.clone()
    bad[0, 3, 5], bad[1, 7, 9], bad[1, 0, 0] = 255, -1, nc
    lg = logits.cuda().requires_grad_(True)
    loss, acc = mfc.mfc_loss(lg, bad.cuda())
    loss.backward()
    assert int(acc[29]) == 3 and bool(torch.isfinite(lg.grad).all()) and bool(torch.isfinite(loss))
    logp = F.log_softmax(logits, 1)
    tg = bad.clone()
    tg[(bad < 0) | (bad >= nc)] = -100
    nll = F.nll_loss(logp, tg, weight=torch.tensor([1.0, 1000.0, 1000.0, 1000.0, 1000.0]), ignore_index=-100)
    assert abs(float(acc[26]) - float(nll)) < 1e-5
    with pytest.raises(IndexError):
        mfc.mfc_loss(lg.detach(), bad.cuda(), check_labels=True)
    _, acc_ok = mfc.mfc_loss(lg.detach(), target.cuda(), check_labels=True)
    assert int(acc_ok[29]) == 0


def test_loss_global_batch_split(M):
    """Data-parallel form of the loss: two 'ranks' (two halves of the batch on one GPU) run mfc_loss_partial, their 26 sums
    are added (what dist.allreduce_loss_sums does over RCCL), mfc_loss_finalize / mfc_loss_bwd then use the global sums.
    Loss and logit gradients must equal the reference loss of the whole batch (src/engine.py:64-66 after the gather)."""
    mfc, L, ops = M
    from oracle import mfcnet_oracle as O
    B, nc, H, W = 4, 5, 24, 40
    g = torch.Generator().manual_seed(31)
    logits = (torch.randn(B, nc, H, W, generator=g) * 2).requires_grad_(True)
    target = torch.randint(0, nc, (B, H, W), generator=g)
    tot, parts = O.total_loss(logits, target, nc)
    tot.backward()
    cw = torch.tensor(O.DEFAULT_CLASS_WEIGHTS, dtype=torch.float32, device="cuda")
    halves, accs = [], []
    for r in range(2):
        lg = logits.detach()[2 * r:2 * r + 2].contiguous().cuda()
        tg = target[2 * r:2 * r + 2].contiguous().cuda()
        acc = torch.empty(L.LOSS_ACC_FLOATS, device="cuda")
        d = L.LossDesc(lg.data_ptr(), tg.data_ptr(), cw.data_ptr(), acc.data_ptr(), 0, 2, nc, H, W, 0.7, 0.3, 1.0)
        L.call(L.lib.mfc_loss_partial, d)
        halves.append((lg, tg)); accs.append(acc[:32])
    glob = accs[0] + accs[1]
    ref_sums = O.loss_partial_sums(logits.detach(), target)
    assert relerr(glob[:26].cpu().double(), ref_sums[:26]) < 1e-5
    grads = []
    for (lg, tg) in halves:
        acc = glob.clone()
        dl = torch.empty_like(lg)
        d = L.LossDesc(lg.data_ptr(), tg.data_ptr(), cw.data_ptr(), acc.data_ptr(), dl.data_ptr(), 2, nc, H, W, 0.7, 0.3, 1.0)
        L.call(L.lib.mfc_loss_finalize, d)
        assert abs(float(acc[28]) - float(tot)) < 1e-5 and abs(float(acc[27]) - float(parts["loss_soft_jaccard"])) < 1e-5
        L.call(L.lib.mfc_loss_bwd, d)
        grads.append(dl.cpu())
    assert relerr(torch.cat(grads), logits.grad) < 1e-4


def test_wgrad_batch_matches_single_launches(M):
    """mfc_conv2d_wgrad_batch: n weight gradients of identical geometry in one launch (one with the fused input transform)
    equal the n single launches -- both write deterministic partial-sum slices, so the totals agree to fp32 summation order."""
    _, L, ops = M
    dtype = torch.bfloat16
    N, Cin, Cout, k, H, W = 6, 32, 32, 3, 24, 40
    n = 3
    xs = [rnd(dtype, N, Cin, H, W, seed=60 + i) for i in range(n)]
    dys = [rnd(dtype, N, Cout, H, W, seed=70 + i) for i in range(n)]
    coef = torch.randn(3, 4, Cin).cuda()
    tx = [ops.to_nhwc(x, dtype) for x in xs]
    tdy = [ops.to_nhwc(d, dtype) for d in dys]
    Co16, Ci16 = ops.rup(Cout, 16), ops.rup(Cin, 16)
    singles = [ops.conv2d_wgrad(tx[i], tdy[i], Cout, Cin, k, 1, in_coef=coef if i == 1 else None, in_relu=(i == 1), ipg=2) for i in range(n)]
    descs = (L.WgradDesc * n)()
    bufs, outs = [], []
    for i in range(n):
        descs[i] = L.WgradDesc(tx[i].data_ptr(), tdy[i].data_ptr(), 0, coef.data_ptr() if i == 1 else 0, L.BF16, N, H, W, tx[i].shape[3], Cin,
                               H, W, tdy[i].shape[3], Cout, k, k, -1, -1, 1, 1 if i == 1 else 0, 2, 0, 0, 0, n, 0)
    parts = L.wgrad_parts(descs[0])
    for i in range(n):
        buf = torch.zeros(parts * k * k * Co16 * Ci16, dtype=torch.float32, device="cuda")
        bufs.append(buf)
        descs[i].dwp, descs[i].splits = buf.data_ptr(), parts
    L.check(L.lib.mfc_conv2d_wgrad_batch(descs, n, L.stream_ptr()), "mfc_conv2d_wgrad_batch")
    for i in range(n):
        dw = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device="cuda")
        ops._run_jobs([dict(src=bufs[i].data_ptr(), dst=dw.data_ptr(), Cout=Cout, Cin=Cin, KH=k, KW=k, Co16=Co16, Ci16=Ci16, nparts=parts)],
                      L.UnpackJob, L.lib.mfc_unpack_wgrad)
        assert relerr(dw.cpu(), singles[i].cpu()) < 1e-5, i
    # mismatching geometry / batch counts are rejected
    descs[1].Hin += 1
    assert L.lib.mfc_conv2d_wgrad_batch(descs, n, L.stream_ptr()) == -1


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(3, 5, 37, 41), (2, 480, 24, 40), (1, 72, 7, 9)])
def test_bias_grad_atomics_and_slices(M, dtype, case):
    """mfc_bias_grad (fp32 atomics into db) and mfc_bias_grad_slices (per-workgroup partial sums, summed by the caller in a fixed order -- the
    form the plan uses, so that the gradient arena repeats bit for bit): both equal the per-channel sum over pixels; the slices repeat exactly."""
    _, L, ops = M
    N, Cc, H, W = case
    dy = rnd(dtype, N, Cc, H, W, seed=81)
    ref = dy.sum((0, 2, 3))
    dyd = ops.to_nhwc(dy, dtype)
    Cp = dyd.shape[3]
    db = torch.zeros(Cc, device="cuda")
    L.check(L.lib.mfc_bias_grad(dyd.data_ptr(), db.data_ptr(), ops.dt_of(dyd), N * H * W, Cp, Cc, L.stream_ptr()), "bias_grad")
    assert relerr(db.cpu(), ref) < 1e-5
    outs = []
    for _ in range(2):
        sl = torch.full((16, Cp), float("nan"), device="cuda")
        L.check(L.lib.mfc_bias_grad_slices(dyd.data_ptr(), sl.data_ptr(), ops.dt_of(dyd), N * H * W, Cp, Cc, 16, L.stream_ptr()), "bias_grad_slices")
        outs.append(sl.clone())
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())
    assert relerr(outs[0].sum(0)[:Cc].cpu(), ref) < 1e-5 and float(outs[0][:, Cc:].abs().max() if Cp > Cc else 0.0) == 0.0


def test_guarded_adam_skips_a_step_with_non_finite_gradients(M):
    """mfc_grad_check + mfc_adam_step_guarded (the overflow guard of a loss-scaled fp16 step): finite gradients -> the same update as
    mfc_adam_step, bit for bit; one Inf or NaN anywhere in the arena -> parameters and both moments untouched, counter incremented."""
    _, L, ops = M
    n = 100003
    g0 = torch.Generator().manual_seed(9)
    p0, g = torch.randn(n, generator=g0).cuda(), (torch.randn(n, generator=g0) * 512).cuda()
    flag = torch.zeros(2, dtype=torch.int32, device="cuda")

    def run(guarded, grad):
        p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        args = (p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0 / 512)
        if guarded:
            L.check(L.lib.mfc_grad_check(grad.data_ptr(), n, flag.data_ptr(), L.stream_ptr()), "check")
            L.check(L.lib.mfc_adam_step_guarded(*args, flag.data_ptr(), L.stream_ptr()), "adam")
        else:
            L.check(L.lib.mfc_adam_step(*args, L.stream_ptr()), "adam")
        torch.cuda.synchronize()
        return p, m, v
    a, b = run(False, g), run(True, g)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and flag.tolist() == [0, 0] and not torch.equal(a[0], p0)
    for bad, where in ((float("inf"), 0), (float("nan"), n - 1), (-float("inf"), n // 2)):
        gb = g.clone(); gb[where] = bad
        p, m, v = run(True, gb)
        assert torch.equal(p, p0) and float(m.abs().max()) == 0 and float(v.abs().max()) == 0 and int(flag[0]) == 1
    assert int(flag[1]) == 3
    run(True, g)
    assert flag.tolist() == [0, 3]


def test_adam_matches_torch(M):
    _, L, ops = M
    n = 10007
    g = torch.Generator().manual_seed(29)
    p0 = torch.randn(n, generator=g)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3)
    P, Mm, V = torch.zeros(10008, device="cuda"), torch.zeros(10008, device="cuda"), torch.zeros(10008, device="cuda")
    P[:n] = p0.cuda()
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        p.grad = grad.clone()
        opt.step()
        G = torch.zeros(10008, device="cuda"); G[:n] = grad.cuda()
        L.check(L.lib.mfc_adam_step(P.data_ptr(), G.data_ptr(), Mm.data_ptr(), V.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, step, 1.0, L.stream_ptr()))
    assert float((P[:n].cpu() - p.detach()).abs().max()) < 1e-6


def test_invalid_descriptors_are_rejected(M):
    _, L, ops = M
    d = L.ConvDesc()
    assert L.lib.mfc_conv2d_fwd(C.byref(d), None) == -1
    assert L.lib.mfc_program_run(None, 1, None) == -1
    x = torch.zeros(1, 8, 8, 12, device="cuda")      # channel pitch not a multiple of 8
    w = torch.zeros(8, 12, 3, 3, device="cuda")
    with pytest.raises(L.MfcError):
        ops.conv2d(x, w, 3, 1)


def test_placeholder_pointers_are_refused_not_faulted():
    """Round 3's aborts (gpurun_out/r03_t5.log, r03_t6.log): descriptors cloned from a DRY plan carried its placeholder address (1 << 30 + offset) in
    mfc_conv_desc.in_fin / mfc_combine_desc.fin (fields of round 3's folded BatchNorm finalize, removed in round 4); bn_fold_prologue dereferenced it and the GPU faulted at the next synchronisation.  With the
    descriptor hardening on (the GPU test session's default) such a launch is refused."""
    from mfcnet_amd import _lib as L, ops
    import ctypes as C
    assert L.lib.mfc_set_flag(53, 1) == 0
    x = torch.randn(2, 8, 8, 32, device="cuda").to(torch.bfloat16)
    out = torch.zeros_like(x)
    coef = torch.ones(1, 4, 32, device="cuda")
    w = torch.randn(32, 32, 3, 3, device="cuda") * 0.1
    d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), 0, coef.data_ptr(), 0, L.BF16, 2, 8, 8, 32, 32, 8, 8, 32, 32, 8, 8, 3, 3, -1, -1, 1, 1, 1, 0, 0, 1, 2, 0, 0, 0)
    wp = ops.pack_weight(w, d, "fwd")
    d.wp = wp.data_ptr()
    assert L.lib.mfc_conv2d_fwd(C.byref(d), L.stream_ptr()) == 0
    d.in_coef = (1 << 30) + 4096                      # a dry plan's placeholder address
    assert L.lib.mfc_conv2d_fwd(C.byref(d), L.stream_ptr()) == -1
    d.in_coef = coef.data_ptr()
    host = torch.zeros(64)
    d.bias = host.data_ptr()                          # a host tensor's address
    assert L.lib.mfc_conv2d_fwd(C.byref(d), L.stream_ptr()) == -1
    cd = L.CombineDesc()
    cd.out = L.View(out.data_ptr(), 0, 8, 8, 32, 0)
    cd.src[0] = L.View(x.data_ptr(), 0, 8, 8, 32, 0)
    cd.nsrc, cd.relu, cd.dtype, cd.N, cd.C, cd.images_per_group = 1, 1, L.BF16, 2, 32, 2
    assert L.lib.mfc_combine_fwd(C.byref(cd), L.stream_ptr()) == 0
    cd.src[0] = L.View(x.data_ptr(), (1 << 30) + 128, 8, 8, 32, 0)          # a placeholder coefficient block
    assert L.lib.mfc_combine_fwd(C.byref(cd), L.stream_ptr()) == -1
    torch.cuda.synchronize()
