"""conv3x3_ring_kernel (csrc/conv3x3_ring.hip: the 32 -> 32 / 64 -> 64 3x3 stride-1 convolutions of the BasicBlocks, models/hrnet.py:58-74,
forward and data gradient) on shapes the benchmarked plan does not have: ragged images (partial 16-row / 16-column tiles on both axes, images
smaller than one tile), one to eight statistic groups, both 16-bit storage types, both pixel-tile heights, and every epilogue the plan can ask
for (statistics, fused input transform, accumulate, the BatchNorm-backward fusions with the recomputed mask and with the 1-bit mask image).

Two checks per case: (1) against CPU fp32 operators on the same rounded inputs (tolerance = the storage type's rounding of the output);
(2) against conv_igemm_kernel on the same descriptor (mfc_set_flag(30, 0)): the two kernels accumulate each output in the same order, so
the outputs must be the SAME BITS and the fp64 statistic cells equal to 1e-6 relative (their fp32 partial sums group pixels differently)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H16 = (torch.bfloat16, torch.float16)
TOL = {torch.bfloat16: 1.5e-2, torch.float16: 2e-3}
VARIANTS = ["plain", "stats", "xf+stats", "acc", "bn2", "acc+src+bn3"]
# N, C, H, W, groups
SHAPES = [(6, 32, 21, 37, 3), (3, 32, 7, 9, 1), (8, 32, 16, 16, 8), (2, 32, 33, 17, 2), (6, 64, 19, 23, 3), (2, 64, 5, 40, 1), (4, 64, 32, 32, 4)]


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import mfcnet_amd
    from mfcnet_amd import _lib, ops
    return mfcnet_amd, _lib, ops


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def sign_bits(t_nhwc):
    b = (t_nhwc.float() > 0).to(torch.int32).reshape(-1, 8)
    w = torch.tensor([1 << e for e in range(8)], dtype=torch.int32, device=b.device)
    return (b * w).sum(1).to(torch.uint8).contiguous()


def run(L, ops, dtype, shape, variant, ring, mt=4):
    N, Cc, H, W, G = shape
    ipg = N // G
    L.lib.mfc_set_flag(30, 1 if ring else 0)
    L.lib.mfc_set_flag(31, mt)
    g = torch.Generator().manual_seed(11)
    rd = lambda *s: torch.randn(*s, generator=g).to(dtype).float()
    x = rd(N, Cc, H, W)
    w = rd(Cc, Cc, 3, 3) * 0.1
    xd = ops.to_nhwc(x, dtype)
    out0 = rd(N, Cc, H, W) if "acc" in variant else torch.zeros(N, Cc, H, W)
    out = ops.to_nhwc(out0, dtype)
    keep = [xd, out]
    d = L.ConvDesc(xd.data_ptr(), 0, out.data_ptr(), 0, 0, 0, ops.dt_of(xd), N, H, W, Cc, Cc, H, W, Cc, Cc, H, W, 3, 3, -1, -1, 1, 1, 1, 0, 0,
                   0, ipg, 0, 0, 0)
    ref_in = x
    stats = None
    if "stats" in variant or "bn" in variant:
        stats = torch.zeros(L.STAT_REPLICAS, G, 2, Cc, dtype=torch.float64, device="cuda")
        d.out_stats = stats.data_ptr()
    if "xf" in variant:
        coef = torch.zeros(G, 4, Cc)
        coef[:, 0] = torch.rand(G, Cc, generator=g) + 0.5
        coef[:, 1] = torch.randn(G, Cc, generator=g) * 0.4
        cd = coef.cuda(); keep.append(cd)
        d.in_coef, d.in_relu = cd.data_ptr(), 1
        t = x.view(G, ipg, Cc, H, W) * coef[:, 0].view(G, 1, Cc, 1, 1) + coef[:, 1].view(G, 1, Cc, 1, 1)
        ref_in = F.relu(t).reshape(N, Cc, H, W).to(dtype).float()          # the kernels round the transformed operand to the storage type
    ref = F.conv2d(ref_in, w, padding=1)
    src0 = None
    if "acc" in variant:
        d.accumulate = 1
        base = out0
        if "src" in variant:
            src0 = rd(N, Cc, H, W)
            sd = ops.to_nhwc(src0, dtype); keep.append(sd)
            d.acc_src = sd.data_ptr()
            base = src0
        ref = ref + base
    ref_stats = None
    if "bn" in variant:
        y = rd(N, Cc, H, W)
        yd = ops.to_nhwc(y, dtype); keep.append(yd)
        cf = torch.zeros(G, 4, Cc)
        cf[:, 0] = torch.rand(G, Cc, generator=g) + 0.5; cf[:, 1] = torch.randn(G, Cc, generator=g) * 0.3
        cf[:, 2] = torch.randn(G, Cc, generator=g) * 0.2; cf[:, 3] = torch.rand(G, Cc, generator=g) + 0.5
        cfd = cf.cuda(); keep.append(cfd)
        d.bn_y, d.bn_coef = yd.data_ptr(), cfd.data_ptr()
        yg = y.view(G, ipg, Cc, H, W)
        if "bn2" in variant:
            d.bn_mask_mode = 2
            m = ((yg * cf[:, 0].view(G, 1, Cc, 1, 1) + cf[:, 1].view(G, 1, Cc, 1, 1)) > 0).float().reshape(N, Cc, H, W)
        else:
            d.bn_mask_mode = 3
            mt_ = rd(N, Cc, H, W)
            md = ops.to_nhwc(mt_, dtype)
            bits = sign_bits(md); keep.append(bits)
            d.bn_bits = bits.data_ptr()
            m = (mt_ > 0).float()
        ref = ref * m
        yh = ((yg - cf[:, 2].view(G, 1, Cc, 1, 1)) * cf[:, 3].view(G, 1, Cc, 1, 1)).reshape(N, Cc, H, W)
        rg = ref.view(G, ipg, Cc, H, W)
        ref_stats = torch.stack([rg.sum((1, 3, 4)), (rg * yh.view(G, ipg, Cc, H, W)).sum((1, 3, 4))], 1)      # [G][2][C]
    elif stats is not None:
        rg = ref.view(G, ipg, Cc, H, W)
        ref_stats = torch.stack([rg.sum((1, 3, 4)), (rg * rg).sum((1, 3, 4))], 1)
    if variant not in ("plain", "stats", "xf+stats"):
        d.flags = L.CONV_WANT_FA
    lay = L.conv_layout(d)
    wp = ops.pack_weight(w.cuda(), d, "fwd"); keep.append(wp)
    d.wp = wp.data_ptr()
    L.call(L.lib.mfc_conv2d_fwd, d)
    torch.cuda.synchronize()
    return out.clone(), (stats.sum(0).cpu() if stats is not None else None), ref, ref_stats, lay


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "N%d_C%d_%dx%d_G%d" % s)
@pytest.mark.parametrize("dtype", H16, ids=["bf16", "fp16"])
def test_ring_kernel_vs_cpu_and_vs_igemm(M, dtype, shape, variant):
    _, L, ops = M
    N, Cc, H, W, G = shape
    fused = variant not in ("plain", "stats", "xf+stats")
    try:
        o_r, s_r, ref, ref_s, lay_r = run(L, ops, dtype, shape, variant, ring=True)
        o_i, s_i, _, _, lay_i = run(L, ops, dtype, shape, variant, ring=False)
    finally:
        L.lib.mfc_set_flag(30, 1); L.lib.mfc_set_flag(31, 4)
    # which kernel ran: the ring layout packs [tap][granule][cout] with one chunk and one cout block
    is_ring = (lay_r.nchunks, lay_r.Yblocks, lay_r.NT16, lay_r.nslots) == (1, 1, Cc, 9 * Cc // 8)
    # (64 channels with the data-gradient fusions stay on conv_igemm, whose single-stage 64-channel layout happens to be the same image)
    assert is_ring or (Cc == 64 and fused), (lay_r.nchunks, lay_r.Yblocks, lay_r.NT16, lay_r.nslots)
    got = ops.to_nchw(o_r, Cc).cpu()
    assert relerr(got, ref) < TOL[dtype], (variant, relerr(got, ref))
    assert torch.equal(o_r, o_i)
    if ref_s is not None:
        assert relerr(s_r[..., :Cc].float(), ref_s) < 5 * TOL[dtype]
        assert relerr(s_r.double(), s_i.double()) < 1e-6


@pytest.mark.parametrize("variant", ["stats", "bn2"])
def test_ring_kernel_two_row_tiles(M, variant):
    """mfc_set_flag(31, 2): the 8-row pixel tile (32 channels only) is a different instantiation with its own DMA tables"""
    _, L, ops = M
    shape = (6, 32, 21, 37, 3)
    try:
        o2, s2, ref, ref_s, lay2 = run(L, ops, torch.bfloat16, shape, variant, ring=True, mt=2)
        o4, s4, _, _, lay4 = run(L, ops, torch.bfloat16, shape, variant, ring=True, mt=4)
    finally:
        L.lib.mfc_set_flag(30, 1); L.lib.mfc_set_flag(31, 4)
    assert (lay2.MT, lay4.MT) == (2, 4)
    assert torch.equal(o2, o4) and relerr(ops.to_nchw(o2, 32).cpu(), ref) < TOL[torch.bfloat16]
    assert relerr(s2.double(), s4.double()) < 1e-6
