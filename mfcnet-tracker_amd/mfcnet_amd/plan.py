"""Execution plan of the MFCNet hot path: a static, flat program of HIP kernel launches.

The reference evaluates `HRNetMulti{Large,Basic}.forward` (models/multiframe_model.py:424-471) as a
Python loop over T frames of a nested nn.Module tree (models/hrnet.py:425-476) and lets autograd
replay ~5000 operator nodes backwards.  Here the same arithmetic is *planned once* per
(batch, resolution, inputs, mode): every activation, gradient, statistic and packed-weight image gets
a fixed address in one of five HBM arenas (288 GB per GPU: nothing is recomputed or re-allocated), the T
frames are batched as N = T*B images with T BatchNorm statistic groups, and the forward and backward
passes become two arrays of `mfc_op` records that libmfcnet_hip executes from ONE host call each
(mfc_program_run) -- no per-op Python, no autograd graph, capturable in a hipGraph.

Fusions the plan encodes:
  * conv -> BN -> ReLU -> conv chains never materialise the normalised activation: the producer's
    conv epilogue accumulates per-(group,channel) sum / sum-of-squares, `BNFIN` turns them into
    scale/shift, and the consumer conv (and its wgrad) applies scale/shift/ReLU while staging its input;
  * residual adds, fuse-layer sums, bilinear up-sampling and the final ReLU are one `COMBINE` pass;
  * the x4 logit up-sampling, the temporal concat and flow/depth inputs are one `HEAD_FWD` pass.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .arch import branch_widths, head_in_channels

BN_EPS, BN_MOMENTUM = 1e-5, 0.1          # nn.BatchNorm defaults; hrnet.py:34
BIAS_PARTS = 64                           # partial-sum slices of a bias gradient (one per workgroup of mfc_bias_grad_slices)


def rup(x, m):
    return (x + m - 1) // m * m


class Arena:
    def __init__(self, name):
        self.name, self.size, self.base, self.buf = name, 0, 0, None

    def alloc(self, nbytes, align=256):
        off = rup(self.size, align)
        self.size = off + nbytes
        return off

    def reset(self):
        self.size = 0


@dataclass
class Ten:
    N: int
    H: int
    W: int
    C: int
    Cp: int
    ipg: int
    ptr: int
    nbytes: int
    name: str = ""
    needs_grad: bool = True
    grad: Optional["Ten"] = None
    grad_init: bool = False
    bits: int = 0                 # 1-bit image of (tensor > 0), written by the ReLU'd combine that produced it (bf16 training)
    last_dgrad: Optional[list] = None     # backward emission: descriptors of the data-gradient launch(es) that wrote this tensor's gradient LAST
    grad_src: Optional["Ten"] = None      # backward emission: the gradient of this tensor starts from that tensor's gradient (not yet copied)


@dataclass
class BN:
    name: str
    C: int
    Cp: int
    G: int
    training: bool
    count: float
    stats: int
    coef: int
    bstats: int
    bcoef: int


@dataclass
class Act:
    t: Ten
    bn: Optional[BN] = None      # virtual: consumers apply bn (+relu) on the fly
    relu: bool = False


@dataclass
class Term:
    t: Ten
    bn: Optional[BN] = None
    c_off: int = 0               # channel offset inside t (slices)


@dataclass
class ConvInfo:
    wname: str
    cout: int
    cin: int
    k: int
    stride: int
    pad: int
    bias: Optional[str]
    dgrad: List = field(default_factory=list)     # data-gradient launch descriptors (1 for stride 1, 4 parity classes for stride 2)
    dwp: int = 0
    wg: Optional[object] = None                   # weight-gradient launch descriptor (x / dy pointers filled at backward emission)
    fwd: Optional[object] = None                  # the forward launch descriptor
    slice_bytes: int = 0                          # one partial-sum slice of the packed gradient image
    unpack: Optional[dict] = None                 # unpack job fields (source / slice count are set with the launch)


class _Recs(list):
    """Program records (kind, descriptor); append() tags the plan's current lane."""

    def __init__(self, plan):
        super().__init__()
        self.plan = plan

    def append(self, item):
        super().append((item[0], item[1], self.plan.cur_lane))


class _Ops(list):
    """Forward graph nodes; remembers the lane each one was emitted on (its backward records run on the same lane)."""

    def __init__(self, plan):
        super().__init__()
        self.plan, self.lanes = plan, []

    def append(self, item):
        super().append(item)
        self.lanes.append(self.plan.cur_lane)


class Plan:
    def __init__(self, model, B, H, W, has_flow, has_depth, base_training, head_training, need_backward, device,
                 base_frozen=False, dry=False):
        """dry=True: plan only (descriptors, geometry, sizes) with placeholder addresses -- no device memory, nothing runnable;
        works without a GPU (tests enumerate the launches of a configuration this way)."""
        self.m = model
        # base_frozen: no parameter of the per-frame network requires a gradient (the reference's default mode,
        # scripts/train_multiframe_detection.py:159-165): the backward program stops at the temporal head's input -- no
        # head-gather adjoint, no data gradient of the 11x11 convolution, nothing of the per-frame network
        self.base_frozen = bool(base_frozen) and not getattr(model, "single", False)
        self.B, self.H, self.W = B, H, W
        self.T, self.nc, self.width = model.num_frames, model.num_classes, model.width
        self.basic = model.basic
        self.single = getattr(model, "single", False)              # single-frame HighResolutionNet: no temporal head
        self.ext_base = getattr(model, "base_kind", "hrnet") != "hrnet"      # per-frame logits come from outside (ResUnet_VB): head only
        self.base_prefix = "" if self.single else "base_model."
        self.has_flow, self.has_depth = has_flow, has_depth
        self.dtype = model.compute_dtype
        self.esz = 2 if L.is16(self.dtype) else 4
        self.E = 16 // self.esz
        self.base_training, self.head_training = base_training, head_training
        self.need_backward = need_backward
        self.device = device
        self.fuse_bn = model.fuse_bn
        # weight gradients of identical geometry can share launches (mfc_conv2d_wgrad_batch).  Measured on the W32 step: 3 % SLOWER
        # (533 vs 545 frames/s) -- deferring a branch's eight gradients to one launch starves the detached stream -- so off by default
        self.batch_wgrad = getattr(model, "batch_wgrad", False)
        self.fuse_fin = getattr(model, "fuse_bn_finalize", True)     # BN-backward finalize inside the apply launch (C <= 128)
        self.mask_bits = getattr(model, "relu_mask_bits", True)     # 1-bit ReLU masks for the BatchNorm backward (bf16)
        self.lanes = getattr(model, "parallel_branches", True)      # branch-parallel lanes of the program (include/mfcnet_hip.h, mfc_op.lane)
        # BatchNorm-backward reduce passes folded into the epilogue of the data-gradient launch that completes the gradient (bf16):
        # mfc_conv_desc.bn_y / acc_src (include/mfcnet_hip.h)
        self.fuse_bnred = bool(getattr(model, "fuse_bnbwd_reduce", True)) and L.is16(self.dtype) and need_backward
        # data gradient of a 3x3 / stride-2 convolution as ONE launch over its four output parity classes instead of four launches (round 4)
        self.merge_s2 = bool(getattr(model, "merge_s2_dgrad", True))
        self.arenas = {k: Arena(k) for k in ("act", "stats", "bstats", "dwp", "misc")}
        self.dry = dry
        self._build()                       # sizing pass
        if dry:
            base = 1 << 30
            for a in self.arenas.values():
                a.base = base
                base += rup(a.size, 1 << 20) + (1 << 20)
            self._build()
            self.bytes_total = sum(a.size for a in self.arenas.values())
            return
        for a in self.arenas.values():
            a.buf = torch.empty(max(a.size, 256) + 256, dtype=torch.uint8, device=device)
            a.base = rup(a.buf.data_ptr(), 256)
        self.arenas["misc"].buf.zero_()
        self.arenas["act"].buf.zero_()
        self.arenas["dwp"].buf.zero_()        # once: cells outside the channel blocks are never written by the wgrad launches
        self._build()                       # real pass: pointers
        self._finalize()

    # ------------------------------------------------------------------ allocation helpers
    def _alloc(self, arena, nbytes):
        a = self.arenas[arena]
        return a.base + a.alloc(nbytes)

    def tensor(self, N, H, W, C, ipg, name="", needs_grad=True) -> Ten:
        Cp = rup(C, 8)
        nb = N * H * W * Cp * self.esz
        return Ten(N, H, W, C, Cp, ipg, self._alloc("act", nb), nb, name, needs_grad)

    def grad_of(self, t: Ten) -> Ten:
        if t.grad is None:
            t.grad = Ten(t.N, t.H, t.W, t.C, t.Cp, t.ipg, self._alloc("act", t.nbytes), t.nbytes, "d" + t.name)
        return t.grad

    def pptr(self, name):         # fp32 parameter pointer
        return self.m._P.data_ptr() + 4 * self.m._poff[name]

    def gptr(self, name):         # fp32 parameter-gradient pointer (into the flat gradient arena)
        return self.grad_base + 4 * self.m._poff[name]

    def _grad_ready(self, name):  # the record just appended is the last writer of this parameter's gradient
        self._ready[name] = len(self.bwd) - 1

    def bptr(self, name):         # running stats
        return self.m._RS.data_ptr() + 4 * self.m._boff[name]

    def new_bn(self, name, like: Ten, training) -> BN:
        G = like.N // like.ipg
        C_, Cp = like.C, like.Cp
        R = L.STAT_REPLICAS
        return BN(name, C_, Cp, G, training, float(like.ipg * like.H * like.W),
                  self._alloc("stats", R * G * 2 * Cp * L.STAT_BYTES), self._alloc("misc", G * 4 * Cp * 4),
                  self._alloc("bstats", R * G * 2 * Cp * L.STAT_BYTES) if self.need_backward else 0,
                  self._alloc("misc", G * 2 * Cp * 4) if self.need_backward else 0)

    def view(self, t: Ten, bn: Optional[BN] = None, c_off=0) -> L.View:
        return L.View(t.ptr, bn.coef if bn else 0, t.H, t.W, t.Cp, c_off)

    # ------------------------------------------------------------------ conv bookkeeping
    def _pack(self, desc, src, cout, cin, k, **tap):
        """allocate the packed image the launch `desc` reads, queue its pack job, point the descriptor at it"""
        lay = L.conv_layout(desc)
        assert (lay.TA, lay.TB) == (tap["TA"], tap["TB"])
        dst = self._alloc("act", lay.bytes)
        job = dict(src=src, dst=dst, Cout=cout, Cin=cin, KH=k, KW=k, **tap)
        job.update(L.pack_job_fields(lay))
        self.pack_jobs.append(job)
        desc.wp = dst

    def conv_info(self, fwd_desc, xt, y, wname, cout, cin, k, stride, bias, in_coef=0, in_relu=0) -> ConvInfo:
        ci = ConvInfo(wname, cout, cin, k, stride, k // 2, (wname[:-6] + "bias") if bias else None)
        pad = ci.pad
        src = self.pptr(wname)
        self._pack(fwd_desc, src, cout, cin, k, TA=k, TB=k, kh0=0, kh_step=1, kw0=0, kw_step=1, mode=0)
        if self.need_backward:
            ci.dgrad = []
            if xt.needs_grad:
                if stride == 1:
                    d = L.ConvDesc(16, 0, 16, 0, 0, 0, self.dtype, y.N, y.H, y.W, y.Cp, cout, xt.H, xt.W, xt.Cp, xt.C, xt.H, xt.W,
                                   k, k, -(k - 1 - pad), -(k - 1 - pad), 1, 1, 1, 0, 0, 0, y.ipg, 0, 0, 0)
                    d.flags = L.CONV_WANT_FA if self.fuse_bnred else 0      # (before packing: the weight image follows the geometry)
                    if in_coef and k == 3:     # x is the never-materialised conv -> BN -> ReLU tensor inside a residual block (hrnet.py:62-69, :99-106):
                        d.flags |= L.CONV_NEVER_ACC          # one consumer, so this launch is the only writer of its gradient (checked when the backward is emitted)
                    self._pack(d, src, cout, cin, k, TA=k, TB=k, kh0=k - 1, kh_step=-1, kw0=k - 1, kw_step=-1, mode=1)
                    ci.dgrad.append(d)
                elif stride == 2 and k == 3 and pad == 1 and self.merge_s2:
                    # ONE launch over the four output parity classes (MFC_CONV_S2_CLASSES, include/mfcnet_hip.h): every class is the same
                    # 2x2-tap problem on the dy grid with its own weight image (taps it does not have are packed as zeros)
                    d = L.ConvDesc(16, 0, 16, 0, 0, 0, self.dtype, y.N, y.H, y.W, y.Cp, cout, xt.H, xt.W, xt.Cp, xt.C, (xt.H + 1) // 2, (xt.W + 1) // 2,
                                   2, 2, 0, 0, 1, 2, 2, 0, 0, 0, y.ipg, 0, 0, 0)
                    d.flags = L.CONV_S2_CLASSES | (L.CONV_WANT_FA if self.fuse_bnred else 0)
                    lay = L.conv_layout(d)
                    dst = self._alloc("act", lay.bytes)
                    for ph in range(2):
                        for pw in range(2):
                            job = dict(src=src, dst=dst + (2 * ph + pw) * (lay.bytes // 4), Cout=cout, Cin=cin, KH=k, KW=k, TA=2, TB=2,
                                       kh0=1 + ph, kh_step=-2, kw0=1 + pw, kw_step=-2, mode=1)
                            job.update(L.pack_job_fields(lay))
                            self.pack_jobs.append(job)
                    d.wp = dst
                    ci.dgrad.append(d)
                else:
                    assert stride == 2
                    for ph in range(2):
                        for pw in range(2):
                            ta, kh0, dh0 = self._s2_class(k, pad, ph)
                            tb, kw0, dw0 = self._s2_class(k, pad, pw)
                            Hl, Wl = (xt.H - ph + 1) // 2, (xt.W - pw + 1) // 2
                            if Hl <= 0 or Wl <= 0:
                                continue
                            d = L.ConvDesc(16, 0, 16, 0, 0, 0, self.dtype, y.N, y.H, y.W, y.Cp, cout, xt.H, xt.W, xt.Cp, xt.C, Hl, Wl,
                                           ta, tb, dh0, dw0, 1, 2, 2, ph, pw, 0, y.ipg, 0, 0, 0)
                            d.flags = L.CONV_WANT_FA if self.fuse_bnred else 0
                            self._pack(d, src, cout, cin, k, TA=ta, TB=tb, kh0=kh0, kh_step=-2, kw0=kw0, kw_step=-2, mode=1)
                            ci.dgrad.append(d)
            Co16, Ci16 = rup(cout, 16), rup(cin, 16)
            # the pixel axis is split over `parts` workgroup sets, each writing its own partial-sum slice (no atomics);
            # the unpack job adds the slices up
            ci.wg = L.WgradDesc(16, 16, 16, in_coef, self.dtype, xt.N, xt.H, xt.W, xt.Cp, xt.C, y.H, y.W, y.Cp, cout, k, k, -pad, -pad,
                                stride, in_relu, xt.ipg, 0, 0, 0)
            ci.slice_bytes = k * k * Co16 * Ci16 * 4
            ci.unpack = dict(dst=self.gptr(wname), Cout=cout, Cin=cin, KH=k, KW=k, Co16=Co16, Ci16=Ci16)
            # (the partial-sum buffer is sized when the launch is emitted: gradients of identical geometry share launches,
            #  and the number of slices depends on how many share one -- see _flush_wgrads)
        return ci

    # ------------------------------------------------------------------ weight-gradient launches
    def _queue_wgrad(self, ci: ConvInfo):
        """Weight gradients are detached work (nothing reads them before their unpack, which runs behind them on the same stream), so those of identical geometry are
        collected and launched together, up to 8 per launch (mfc_conv2d_wgrad_batch): the 8 3x3 convolutions of a module
        branch become ONE launch whose workgroups each walk 8x more pixels of one problem."""
        w = ci.wg
        key = (self.cur_lane, w.dtype, w.N, w.Hin, w.Win, w.Cin_p, w.Cin, w.Hout, w.Wout, w.Cout_p, w.Cout, w.TA, w.TB, w.dh0, w.dw0,
               w.in_stride, w.images_per_group)
        lst = self._wg_pending.setdefault(key, [])
        lst.append(ci)
        if len(lst) == L.WGRAD_MAXBATCH or not self.batch_wgrad:
            self._flush_wgrads(key)

    def _flush_wgrads(self, key=None):
        keys = [key] if key is not None else list(self._wg_pending)
        for kk in keys:
            lst = self._wg_pending.pop(kk, [])
            if not lst:
                continue
            n = len(lst)
            parts = -1
            if n > 1:
                lst[0].wg.batch = n
                parts = L.lib.mfc_conv2d_wgrad_parts(C.byref(lst[0].wg))       # < 0: this geometry is not batchable
            groups = [lst] if parts > 0 else [[ci] for ci in lst]
            for grp in groups:
                m = len(grp)
                for ci in grp:
                    ci.wg.batch = m if m > 1 else 0
                pr = L.wgrad_parts(grp[0].wg)
                for ci in grp:
                    ci.wg.splits = pr if m > 1 else 0
                    ci.dwp = self._alloc("dwp", pr * ci.slice_bytes)
                    ci.wg.dwp = ci.dwp
                    self.unpack_jobs.append(dict(src=ci.dwp, nparts=pr, **ci.unpack))
                if m == 1:
                    self.bwd.append((L.OP_WGRAD, grp[0].wg))
                else:
                    arr = (L.WgradDesc * m)(*[ci.wg for ci in grp])
                    self._wg_arrays.append(arr)                                 # the record holds a HOST pointer to this array
                    r = L.RawOp(C.addressof(arr), 0, 0, 0)
                    r.i[0] = m
                    self.bwd.append((L.OP_WGRAD_BATCH, r))
                if self.lanes:      # detached: nothing reads the partial sums before their unpack (same stream, in order)
                    k_, d_, ln_ = self.bwd[-1]
                    self.bwd[-1] = (k_, d_, ln_ | L.LANE_ASYNC)
                for ci in grp:
                    self._grad_ready(ci.wname)
                    self.unpack_jobs[-len(grp) + grp.index(ci)]["_name"] = ci.wname

    @staticmethod
    def _s2_class(k, pad, ph):
        """stride-2 data-gradient, output parity class ph: (#taps, first kh, dh0); taps kh = kh0 - 2a."""
        par = (ph + pad) % 2
        khmax = k - 1 if (k - 1) % 2 == par else k - 2
        return khmax // 2 + 1, khmax, (ph + pad - khmax) // 2

    # ------------------------------------------------------------------ forward graph primitives
    def conv(self, x: Act, wname, cout, k, stride=1, bias=False, bn_name=None, training=True) -> Act:
        xt = x.t
        pad = k // 2
        Ho, Wo = (xt.H + 2 * pad - k) // stride + 1, (xt.W + 2 * pad - k) // stride + 1
        y = self.tensor(xt.N, Ho, Wo, cout, xt.ipg, wname)
        bn = self.new_bn(bn_name, y, training) if bn_name else None
        d = L.ConvDesc(xt.ptr, 0, y.ptr, self.pptr(wname[:-6] + "bias") if bias else 0, x.bn.coef if x.bn else 0,
                       bn.stats if (bn and bn.training) else 0, self.dtype, xt.N, xt.H, xt.W, xt.Cp, xt.C,
                       Ho, Wo, y.Cp, cout, Ho, Wo, k, k, -pad, -pad, stride, 1, 1, 0, 0,
                       1 if x.relu else 0, xt.ipg, 0, 0, 0)
        ci = self.conv_info(d, xt, y, wname, cout, xt.C, k, stride, bias, x.bn.coef if x.bn else 0, 1 if x.relu else 0)
        ci.fwd = d
        self.fwd.append((L.OP_CONV, d))
        self.ops.append(("conv", x, y, ci, bn))
        if bn:
            fd = L.BnFinDesc(bn.stats, bn.coef, self.pptr(bn_name + ".weight"), self.pptr(bn_name + ".bias"),
                             self.bptr(bn_name + ".running_mean"), self.bptr(bn_name + ".running_var"),
                             self.m._NBT.data_ptr() + 8 * self.m._noff[bn_name + ".num_batches_tracked"],
                             bn.C, bn.Cp, bn.G, 1 if bn.training else 0, bn.count, BN_EPS, BN_MOMENTUM)
            self.fwd.append((L.OP_BNFIN, fd))
            self.ops.append(("bnfin", y, bn))
        return Act(y, bn, False)

    def cbr(self, x: Act, conv_name, bn_name, cout, k, stride, relu, training) -> Act:
        a = self.conv(x, conv_name + ".weight", cout, k, stride, False, bn_name, training)
        a.relu = relu
        if not self.fuse_bn:
            return self.combine([Term(a.t, a.bn)], relu)
        return a

    def combine(self, terms: List[Term], relu, out: Optional[Ten] = None, out_c_off=0, C=None, size=None) -> Act:
        t0 = terms[0].t
        H, W = size if size else (t0.H, t0.W)
        C = C if C is not None else t0.C
        if out is None:
            out = self.tensor(t0.N, H, W, C, t0.ipg, "cmb")
        Cs = rup(C, self.E)
        d = L.CombineDesc()
        d.out = L.View(out.ptr, 0, out.H, out.W, out.Cp, out_c_off)
        for i, tm in enumerate(terms):
            d.src[i] = self.view(tm.t, tm.bn, tm.c_off)
        d.nsrc, d.relu, d.dtype, d.N, d.C, d.images_per_group = len(terms), 1 if relu else 0, self.dtype, out.N, Cs, out.ipg
        if relu and self.need_backward and L.is16(self.dtype) and self.mask_bits:
            # the backward of the summed terms (BatchNorm reduce, masked adds / up-sampling adjoints) needs only the sign of this
            # output: keep it as one bit per element (1/16 of the tensor) so that those passes read one tensor less
            if not out.bits:
                out.bits = self._alloc("act", out.N * out.H * out.W * out.Cp // 8)
            d.maskbits = out.bits
        self.fwd.append((L.OP_COMBINE, d))
        self.ops.append(("combine", terms, out, out_c_off, Cs, relu))
        return Act(out, None, False)

    # ------------------------------------------------------------------ the network
    def _build(self):
        for a in self.arenas.values():
            a.reset()
        self.cur_lane = 0
        self.fwd, self.bwd, self.ops = _Recs(self), _Recs(self), _Ops(self)
        self.pack_jobs, self.unpack_jobs = [], []
        m, B, T, H, W, nc = self.m, self.B, self.T, self.H, self.W, self.nc
        self.grad_base = m._G.data_ptr()
        f32 = lambda n: self._alloc("misc", 4 * n)
        self.in_frames = [f32(B * 3 * H * W) for _ in range(T)]
        self.in_flow = [f32(B * 2 * H * W) for _ in range(T - 1)] if self.has_flow else []
        self.in_depth = [f32(B * H * W) for _ in range(T)] if self.has_depth else []
        self.out_buf = f32(B * nc * H * W)
        self.gout_buf = f32(B * nc * H * W) if self.need_backward else 0

        if self.ext_base:
            # ---- the per-frame network runs outside this plan (ResUnet_VB, its own programs): its FULL-resolution logits [T*B, nc, H, W]
            #      come in as one NCHW fp32 tensor; the gradient w.r.t. them goes back out the same way
            self.in_logits = f32(T * B * nc * H * W)
            self.gin_buf = f32(T * B * nc * H * W) if (self.need_backward and not self.base_frozen) else 0
            logits = self.tensor(T * B, H, W, nc, B, "ext_logits", needs_grad=not self.base_frozen)
            r = L.RawOp(self.in_logits, logits.ptr, 0, 0)
            r.i[0:7] = [self.dtype, T * B, nc, H, W, logits.Cp, 0]
            self.fwd.append((L.OP_NCHW2NHWC, r))
            self.ext_logits = logits
        else:
            # ---- input bridge: T NCHW fp32 frames -> one NHWC [T*B, H, W, 8] tensor (multiframe_model.py:459)
            x0 = self.tensor(T * B, H, W, 3, B, "frames", needs_grad=False)
            for t in range(T):
                r = L.RawOp(self.in_frames[t], x0.ptr + t * B * H * W * x0.Cp * self.esz, 0, 0)
                r.i[0:7] = [self.dtype, B, 3, H, W, x0.Cp, 0]
                self.fwd.append((L.OP_NCHW2NHWC, r))
            logits = self._hrnet(Act(x0))
        self.n_base_ops = len(self.ops)               # forward graph nodes [0, n_base_ops) belong to the per-frame network
        # ---- head input: x4 up-sample + temporal concat (+flow, +depth, +warp)
        basic_warp = self.basic and self.has_flow
        cin = head_in_channels(self.basic, nc, T, self.has_flow, self.has_depth)
        xh = self.tensor(B, H, W, cin, B, "head_in", needs_grad=not self.base_frozen)
        hd = L.HeadDesc()
        hd.logits, hd.xh = logits.ptr, xh.ptr
        for i, p in enumerate(self.in_flow):
            hd.flow[i] = p
        for i, p in enumerate(self.in_depth):
            hd.depth[i] = p
        hd.dtype, hd.B, hd.T, hd.nc, hd.Hs, hd.Ws, hd.Lp = self.dtype, B, T, nc, logits.H, logits.W, logits.Cp
        hd.H, hd.W, hd.Cp, hd.warp = H, W, xh.Cp, 1 if basic_warp else 0
        self.fwd.append((L.OP_HEAD_FWD, hd))
        self.ops.append(("head", logits, xh, hd))
        if self.single:
            # single-frame HighResolutionNet (hrnet.py:473-474): the x4 up-sampled logits ARE the output
            o = xh
        else:
            # ---- temporal aggregation head (multiframe_model.py:191-202)
            q = "multiframe_net.multiframe_net."
            ht = self.head_training
            a = self.cbr(Act(xh), q + "0", q + "1", T * nc, 11, 1, True, ht)
            a = self.cbr(a, q + "3", q + "4", T * nc, 3, 1, True, ht)
            a = self.cbr(a, q + "6", q + "7", T * nc, 3, 1, True, ht)
            o = self.conv(a, q + "9.weight", nc, 1).t
        r = L.RawOp(o.ptr, self.out_buf, 0, 0)
        r.i[0:6] = [self.dtype, B, nc, H, W, o.Cp]
        self.fwd.append((L.OP_NHWC2NCHW, r))
        self.ops.append(("out", o))
        if self.need_backward:
            self._emit_backward()

    def _hrnet(self, x: Act) -> Ten:
        """models/hrnet.py:425-476 as plan primitives; returns the low-resolution logits [T*B, H/4, W/4, nc]."""
        p, tr = self.base_prefix, self.base_training
        Wd = branch_widths(self.width)
        x = self.cbr(x, p + "conv1", p + "bn1", 64, 3, 2, True, tr)
        x = self.cbr(x, p + "conv2", p + "bn2", 64, 3, 2, True, tr)
        for b in range(4):                                   # Bottleneck, hrnet.py:95-115
            q = f"{p}layer1.{b}."
            o = self.cbr(x, q + "conv1", q + "bn1", 64, 1, 1, True, tr)
            o = self.cbr(o, q + "conv2", q + "bn2", 64, 3, 1, True, tr)
            o = self.conv(o, q + "conv3.weight", 256, 1, 1, False, q + "bn3", tr)
            if b == 0:
                r = self.conv(x, q + "downsample.0.weight", 256, 1, 1, False, q + "downsample.1", tr)
                res = Term(r.t, r.bn)
            else:
                res = Term(x.t)
            x = self.combine([Term(o.t, o.bn), res], True)

        def mat(a: Act) -> Act:                              # materialise relu(bn(y)) (needed as a residual)
            return self.combine([Term(a.t, a.bn)], a.relu) if a.bn is not None else a

        def transition(name, ys, pre, cur):                  # hrnet.py:353-389, 434-461
            out = []
            for i, c in enumerate(cur):
                if i < len(pre):
                    out.append(mat(self.cbr(ys[i], f"{p}{name}.{i}.0", f"{p}{name}.{i}.1", c, 3, 1, True, tr))
                               if c != pre[i] else ys[i])
                else:
                    v = ys[-1]
                    for j in range(i + 1 - len(pre)):
                        co = c if j == i - len(pre) else pre[-1]
                        v = self.cbr(v, f"{p}{name}.{i}.{j}.0", f"{p}{name}.{i}.{j}.1", co, 3, 2, True, tr)
                    out.append(mat(v))
            return out

        def module(q, xs, ch, final_cat=None):               # HighResolutionModule.forward, hrnet.py:238-262
            nb = len(xs)
            xs = list(xs)
            for i in range(nb):
                v = xs[i]
                self.cur_lane = i + 1 if (nb > 1 and self.lanes) else 0      # the branches of a module are independent (hrnet.py:242-243)
                for b in range(4):                           # BasicBlock, hrnet.py:58-74
                    r = f"{q}branches.{i}.{b}."
                    o = self.cbr(v, r + "conv1", r + "bn1", ch[i], 3, 1, True, tr)
                    o = self.conv(o, r + "conv2.weight", ch[i], 3, 1, False, r + "bn2", tr)
                    v = self.combine([Term(o.t, o.bn), Term(v.t)], True)
                xs[i] = v
            self.cur_lane = 0
            one_join = self.lanes and getattr(self.m, "fuse_one_join", True)
            outs, sums = [], []
            for i in range(nb):
                terms = []
                for j in range(nb):
                    r = f"{q}fuse_layers.{i}.{j}."
                    # path (i, j) reads branch j and is the only writer of grad(branch j) among the paths: one lane per SOURCE
                    # branch, forward and backward -- so the paths follow their branch on its lane without a join in between
                    self.cur_lane = j + 1 if (self.lanes and j != i) else 0
                    if j == i:
                        terms.append(Term(xs[j].t))
                    elif j > i:                              # 1x1 + BN, then bilinear up-sampling
                        o = self.conv(xs[j], r + "0.weight", ch[i], 1, 1, False, r + "1", tr)
                        terms.append(Term(o.t, o.bn))
                    else:                                    # chain of strided 3x3
                        v = xs[j]
                        for k in range(i - j):
                            last = k == i - j - 1
                            co = ch[i] if last else ch[j]
                            if last:
                                v = self.conv(v, f"{r}{k}.0.weight", co, 3, 2, False, f"{r}{k}.1", tr)
                            else:
                                v = self.cbr(v, f"{r}{k}.0", f"{r}{k}.1", co, 3, 2, True, tr)
                        terms.append(Term(v.t, v.bn))
                self.cur_lane = 0
                if one_join:
                    sums.append(terms)
                else:
                    # reference sums in j order starting from j = 0
                    outs.append(self.combine(terms, True, size=(xs[i].t.H, xs[i].t.W), C=ch[i]))
            if one_join:
                # Every path of the module first (nb*(nb-1) of them, spread over the lanes), ONE join, then sum i on lane i+1:
                # the next module's branch i follows it on the same lane, so a module costs one join (forward and backward)
                # where "paths of output i -> join -> sum i" costs nb+1.
                self.join()
                for i in range(nb):
                    self.cur_lane = i + 1
                    outs.append(self.combine(sums[i], True, size=(xs[i].t.H, xs[i].t.W), C=ch[i]))
                self.cur_lane = 0        # (no record is emitted here: the section stays open for the next module's branches)
            return outs

        ys = transition("transition1", [x], [256], Wd[:2])
        ys = module(f"{p}stage2.0.", ys, Wd[:2])
        ys = transition("transition2", ys, Wd[:2], Wd[:3])
        for mi in range(4):
            ys = module(f"{p}stage3.{mi}.", ys, Wd[:3])
        ys = transition("transition3", ys, Wd[:3], Wd[:4])
        for mi in range(3):
            ys = module(f"{p}stage4.{mi}.", ys, Wd[:4])
        # 4-way up-sample + concat (hrnet.py:464-469): four slice-writing combines
        t0 = ys[0].t
        cat = self.tensor(t0.N, t0.H, t0.W, sum(Wd), t0.ipg, "cat")
        off = 0
        for j in range(4):
            self.combine([Term(ys[j].t)], False, out=cat, out_c_off=off, C=Wd[j], size=(t0.H, t0.W))
            off += Wd[j]
        o = self.conv(Act(cat), p + "last_layer.0.weight", sum(Wd), 1, 1, True, p + "last_layer.1", tr)
        o.relu = True
        if not self.fuse_bn:
            o = self.combine([Term(o.t, o.bn)], True)
        return self.conv(o, p + "last_layer.3.weight", self.nc, 1, 1, True).t

    def join(self):
        """A lane-0 record that launches nothing: the interpreter joins the side lanes there.  Needed between two runs of lane
        records when the second reads what SEVERAL lanes of the first wrote."""
        self.cur_lane = 0
        self.fwd.append((L.OP_JOIN, L.RawOp(0, 0, 0, 0)))
        self.ops.append(("join",))

    # ------------------------------------------------------------------ backward program
    def _bn_backward(self, bn: BN, y: Ten, g_view: L.View, mask_mode, mask_view, dy_view: L.View, C, gm_view=None, gm_acc=0,
                     reduce_fused=False):
        """reduce -> finalize -> apply.  With gm_view the reduce pass also writes the masked gradient g*m there (the
        identity branch of the same sum, += if gm_acc); when it was a plain write, the apply pass reads g*m back from it
        instead of g and the mask (one tensor less).  reduce_fused: the data-gradient launch that completed g has already masked it
        and accumulated the statistics (mfc_conv_desc.bn_y): no reduce record, g is read with mask_mode 0."""
        def desc(g, mode, mask, dy, acc):
            d = L.BnBwdDesc()
            d.g, d.y = g, self.view(y, bn)
            if dy is not None:
                d.dy = dy
            if mask is not None:
                d.mask = mask
            d.bstats, d.bcoef = bn.bstats, bn.bcoef
            d.mask_mode, d.dtype, d.N, d.C, d.images_per_group, d.accumulate = mode, self.dtype, y.N, C, y.ipg, acc
            return d
        if not reduce_fused:
            self.bwd.append((L.OP_BNBWD_REDUCE, desc(g_view, mask_mode, mask_view, gm_view, gm_acc)))
        fused_fin = self.fuse_fin and C <= 128 and bn.C <= C and bn.G <= 8
        if not fused_fin:
            self.bwd.append((L.OP_BNBWD_FIN, L.BnBwdFinDesc(bn.bstats, bn.bcoef, self.gptr(bn.name + ".weight"),
                                                            self.gptr(bn.name + ".bias"), bn.C, bn.Cp, bn.G,
                                                            1 if bn.training else 0, bn.count)))
        if gm_view is not None and not gm_acc:
            ad = desc(gm_view, 0, None, dy_view, 0)
        else:
            ad = desc(g_view, mask_mode, mask_view, dy_view, 0)
        if fused_fin:       # the finalize rides in the apply launch (one dependent launch less per BatchNorm)
            ad.fin_dgamma, ad.fin_dbeta = self.gptr(bn.name + ".weight"), self.gptr(bn.name + ".bias")
            ad.fin_C, ad.fin_training, ad.fin_count = bn.C, 1 if bn.training else 0, bn.count
        self.bwd.append((L.OP_BNBWD_APPLY, ad))
        self._grad_ready(bn.name + ".weight")
        self._grad_ready(bn.name + ".bias")

    def _fusable(self, descs) -> bool:
        """the data-gradient launches `descs` (those that wrote a gradient tensor last) can take the epilogue fusions"""
        return bool(descs) and self.fuse_bnred and all(L.conv_layout(d).fa == 1 for d in descs)

    def _materialise_grad_src(self, t: Ten):
        """a pending "the gradient of t starts from the gradient of another tensor" that no data-gradient launch can read in place:
        make the copy (what the BatchNorm-backward reduce pass used to write)"""
        src = t.grad_src
        if src is None:
            return
        t.grad_src = None
        sg = self.grad_of(t)
        md = L.MaskAddDesc()
        md.g = L.View(src.ptr, 0, src.H, src.W, src.Cp, 0)
        md.dst = L.View(sg.ptr, 0, sg.H, sg.W, sg.Cp, 0)
        md.mask_mode, md.dtype, md.N, md.C, md.accumulate = 0, self.dtype, sg.N, rup(t.C, self.E), 1 if t.grad_init else 0
        self.bwd.append((L.OP_MASK_ADD, md))
        t.grad_init, t.last_dgrad = True, None

    def _emit_backward(self):
        E, Cs = self.E, None
        self._wg_pending, self._wg_arrays = {}, []
        self._ready = {}
        n_ops = len(self.ops)
        for ridx, (op, lane) in enumerate(zip(reversed(self.ops), reversed(self.ops.lanes))):
            kind = op[0]
            if self.base_frozen and (n_ops - 1 - ridx < self.n_base_ops or kind == "head"):
                continue                # frozen per-frame network: its gradients are neither needed nor computed
            if lane != self.cur_lane:
                # leaving a lane: what is still queued there is launched first (from the lane that produced its inputs; after a
                # parallel section the first serial record joins the side lanes before anything runs)
                self._flush_wgrads()
            self.cur_lane = lane
            if kind == "join":
                self.bwd.append((L.OP_JOIN, L.RawOp(0, 0, 0, 0)))
            elif kind == "out":
                o = op[1]
                g = self.grad_of(o)
                r = L.RawOp(self.gout_buf, g.ptr, 0, 0)
                r.i[0:7] = [self.dtype, self.B, self.nc, self.H, self.W, g.Cp, 0]
                self.bwd.append((L.OP_NCHW2NHWC, r))
                o.grad_init = True
            elif kind == "head":
                _, logits, xh, hd = op
                hb = L.HeadBwd()
                C.memmove(C.byref(hb.d), C.byref(hd), C.sizeof(L.HeadDesc))
                hb.d.xh = self.grad_of(xh).ptr
                hb.dlogits = self.grad_of(logits).ptr
                if hb.d.warp:          # fp32 scratch for the scatter form of the grid_sample adjoint
                    hb.d.depth[7] = self._alloc("misc", self.B * self.H * self.W * (self.T - 1) * self.nc * 4)
                self.bwd.append((L.OP_HEAD_BWD, hb))
                logits.grad_init = True
            elif kind == "bnfin":
                _, y, bn = op
                if getattr(y, "virtual_consumed", False):
                    g = self.grad_of(y)
                    assert y.grad_init, y.name
                    gv = L.View(g.ptr, 0, g.H, g.W, g.Cp, 0)
                    mmode = 2 if getattr(y, "virtual_relu", False) else 0
                    if self._fusable(y.last_dgrad):
                        # the launch(es) that completed d(relu(bn(y))) mask it and accumulate the statistics in their epilogue
                        for d in y.last_dgrad:
                            d.bn_y, d.bn_coef, d.out_stats, d.bn_mask_mode, d.bn_bits = y.ptr, bn.coef, bn.bstats, mmode, 0
                        self._bn_backward(bn, y, gv, 0, None, gv, rup(y.C, E), reduce_fused=True)
                    else:
                        self._bn_backward(bn, y, gv, mmode, None, gv, rup(y.C, E))
                    y.last_dgrad = None
            elif kind == "combine":
                _, terms, out, out_c_off, Cs, relu = op
                g = self.grad_of(out)
                assert out.grad_init, out.name
                gv = L.View(g.ptr, 0, g.H, g.W, g.Cp, out_c_off)
                mv = L.View(out.ptr, 0, out.H, out.W, out.Cp, out_c_off) if relu else None
                mode = 1 if relu else 0
                # (same-resolution BatchNorm terms read the 1-bit image of the mask when the forward wrote one)
                bmode, bmv = (3, L.View(out.bits, 0, out.H, out.W, out.Cp, out_c_off)) if (relu and out.bits) else (mode, mv)
                same = lambda tm: tm.t.H == out.H and tm.t.W == out.W
                for tm in terms:                    # (a term whose gradient is still "borrowed" gets its own copy before anything adds to it)
                    if tm.t.needs_grad and tm.t.grad_src is not None:
                        self._materialise_grad_src(tm.t)
                # The identity term of a ReLU'd sum receives exactly g*m.  If its gradient buffer is still unwritten, let the
                # BN-backward reduce pass of a same-resolution BN term write it (no separate mask pass), and let every other
                # term read g*m back from there instead of g and the mask.
                idt = next((tm for tm in terms if relu and tm.bn is None and same(tm) and tm.t.needs_grad and not tm.t.grad_init), None)
                fuse_bn = next((tm for tm in terms if tm.bn is not None and same(tm) and tm.t.needs_grad), None) if idt else None
                order = list(terms)
                if idt is not None:          # materialise g*m first
                    first = fuse_bn if fuse_bn is not None else idt
                    order.remove(first)
                    order.insert(0, first)
                    if fuse_bn is not None:
                        order.remove(idt)
                gmv = None
                for tm in order:
                    s = tm.t
                    if not s.needs_grad:
                        continue
                    sg = self.grad_of(s)
                    g_in, m_in, md_in = (gmv, None, 0) if gmv is not None else (gv, bmv, bmode)
                    if tm.bn is None or not same(tm):
                        md = L.MaskAddDesc()
                        md.g, md.dst = g_in, L.View(sg.ptr, 0, sg.H, sg.W, sg.Cp, tm.c_off)
                        if m_in is not None:
                            md.mask = m_in
                        acc = 1 if (s.grad_init and tm.bn is None) else 0
                        md.mask_mode, md.dtype, md.N, md.C, md.accumulate = md_in, self.dtype, out.N, Cs, acc
                        if not same(tm):        # adjoint of an up-sampling: separable two-pass form, fp32 workspace [N, H, w, C]
                            md.scratch = self._alloc("act", out.N * out.H * s.W * Cs * 4)
                        self.bwd.append((L.OP_MASK_ADD, md))
                        s.grad_init, s.last_dgrad = True, None
                        if tm is idt:
                            gmv = L.View(sg.ptr, 0, sg.H, sg.W, sg.Cp, tm.c_off)
                        if tm.bn is not None:       # up-sampled BN term: BN backward at the low resolution, in place
                            sv = L.View(sg.ptr, 0, sg.H, sg.W, sg.Cp, tm.c_off)
                            self._bn_backward(tm.bn, s, sv, 0, None, sv, Cs)
                    else:
                        assert not s.grad_init, s.name
                        dyv = L.View(sg.ptr, 0, sg.H, sg.W, sg.Cp, tm.c_off)
                        if tm is fuse_bn and bmode == 3 and len(terms) == 2 and out_c_off == 0 and tm.c_off == 0 and idt.c_off == 0 \
                                and Cs == rup(out.C, E) == rup(idt.t.C, E) and self._fusable(out.last_dgrad):
                            # residual block tail (hrnet.py:71-72): the data-gradient launch that completed d(out) has masked it in place
                            # and holds the statistics; the identity branch's gradient STARTS from that tensor (read in place by the
                            # first launch that adds to it: mfc_conv_desc.acc_src) -- no reduce pass, no copy
                            for d in out.last_dgrad:
                                d.bn_y, d.bn_coef, d.out_stats, d.bn_mask_mode, d.bn_bits = s.ptr, tm.bn.coef, tm.bn.bstats, 3, out.bits
                            self._bn_backward(tm.bn, s, gv, 0, None, dyv, Cs, reduce_fused=True)
                            idt.t.grad_src = g
                            gmv = gv
                        elif tm is fuse_bn:
                            ig = self.grad_of(idt.t)
                            gmv = L.View(ig.ptr, 0, ig.H, ig.W, ig.Cp, idt.c_off)
                            self._bn_backward(tm.bn, s, gv, bmode, bmv, dyv, Cs, gm_view=gmv, gm_acc=0)
                            idt.t.grad_init = True
                            idt.t.last_dgrad = None
                        else:
                            self._bn_backward(tm.bn, s, g_in, md_in, m_in, dyv, Cs)
                        s.grad_init, s.last_dgrad = True, None
            elif kind == "conv":
                _, x, y, ci, bn = op
                dy = self.grad_of(y)
                assert y.grad_init, y.name
                xt = x.t
                k, s, pad = ci.k, ci.stride, ci.pad
                ci.wg.x, ci.wg.dy = xt.ptr, dy.ptr
                self._queue_wgrad(ci)
                if ci.bias:
                    # partial sums per workgroup (no atomics), added up in a fixed order by the unpack launch of the bucket, like the weight
                    # gradients' slices: the gradient arena is the same, bit for bit, whatever order the workgroups finish in
                    NP = BIAS_PARTS
                    sl = self._alloc("dwp", NP * y.Cp * 4)
                    r = L.RawOp(dy.ptr, sl, 0, y.N * y.H * y.W)
                    r.i[0:4] = [self.dtype, y.Cp, ci.cout, NP]
                    self.bwd.append((L.OP_BIAS_GRAD, r))
                    self.unpack_jobs.append(dict(src=sl, nparts=NP, dst=self.gptr(ci.bias), Cout=ci.cout, Cin=1, KH=1, KW=1, Co16=y.Cp, Ci16=1,
                                                 _name=ci.bias))
                    self._grad_ready(ci.bias)
                if xt.needs_grad:
                    dx = self.grad_of(xt)
                    src = None
                    if xt.grad_src is not None:
                        # the gradient of xt starts from another tensor's (the masked output gradient of a residual block): the first
                        # data-gradient launch reads it in place; any other kind of launch needs a copy first
                        if not xt.grad_init and self._fusable(ci.dgrad):
                            src, xt.grad_src = xt.grad_src, None
                        else:
                            self._materialise_grad_src(xt)
                    acc = 1 if (xt.grad_init or src is not None) else 0
                    for d in ci.dgrad:
                        if (d.flags & L.CONV_NEVER_ACC) and (acc or src is not None):
                            raise L.MfcError(f"{ci.wname}: a data gradient planned as the only writer of its result would have to accumulate")
                        d.inp, d.out, d.accumulate = dy.ptr, dx.ptr, acc
                        d.acc_src = src.ptr if src is not None else 0
                        self.bwd.append((L.OP_CONV, d))
                    xt.grad_init, xt.last_dgrad = True, list(ci.dgrad)
                    if x.bn is not None:
                        xt.virtual_consumed, xt.virtual_relu = True, x.relu

        self._flush_wgrads()
        self.cur_lane = 0
        if self.ext_base and not self.base_frozen:
            g = self.grad_of(self.ext_logits)
            r = L.RawOp(g.ptr, self.gin_buf, 0, 0)
            r.i[0:6] = [self.dtype, self.T * self.B, self.nc, self.H, self.W, g.Cp]
            self.bwd.append((L.OP_NHWC2NCHW, r))

    # ------------------------------------------------------------------ program arrays

    def _jobs(self, jobs, cls, per_block):
        arr = (cls * len(jobs))()
        b0 = 0
        for i, j in enumerate(jobs):
            for k_, v in j.items():
                if not k_.startswith("_"):
                    setattr(arr[i], k_, v)
            if cls is L.PackJob:
                total = (j["TA"] // j["TAS"]) * j["nchunks"] * j["Yblocks"] * j["nslots"] * j["NT16"]
            else:
                total = j["KH"] * j["KW"] * j["Co16"] * j["Ci16"]
            nb = -(-total // per_block)
            arr[i].block0, arr[i].nblocks = b0, nb
            b0 += nb
        dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        return dev, len(jobs), b0

    def _program(self, recs):
        arr = (L.Op * len(recs))()
        for i, rec in enumerate(recs):
            kind, d = rec[0], rec[1]
            arr[i].kind = kind
            arr[i].lane = rec[2] if len(rec) > 2 else 0
            C.memmove(C.byref(arr[i].u), C.byref(d), C.sizeof(d))
        return arr

    def _finalize(self):
        st, bs, dw = self.arenas["stats"], self.arenas["bstats"], self.arenas["dwp"]
        self._pack_dev, npk, nbk = self._jobs(self.pack_jobs, L.PackJob, 256)
        pro = []
        r = L.RawOp(st.base, 0, 0, st.size)
        pro.append((L.OP_MEMSET, r))
        r = L.RawOp(self._pack_dev.data_ptr(), 0, 0, 0)
        r.i[0:3] = [npk, nbk, self.dtype]
        pro.append((L.OP_PACK, r))
        # eval-mode BatchNorm finalizes read only the running statistics: one table launch in front of the first convolution
        # instead of one one-block launch between every conv and its consumer
        ev = [rec for rec in self.fwd if rec[0] == L.OP_BNFIN and not rec[1].training]
        if any(rec[1].G > 8 or rec[1].G < 1 for rec in ev):
            raise L.MfcError("BatchNorm finalize table: a row has more than 8 statistic groups (num_frames > 8 is not supported)")
        if len(ev) > 1 and getattr(self.m, "batch_eval_bnfin", True):
            tab = (L.BnFinDesc * len(ev))(*[rec[1] for rec in ev])
            self._bnfin_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
            r = L.RawOp(self._bnfin_dev.data_ptr(), 0, 0, 0)
            r.i[0:2] = [len(ev), max(rec[1].Cp for rec in ev)]
            pro.append((L.OP_BNFIN_BATCH, r))
            keep = set(id(rec) for rec in ev)
            self.fwd[:] = [rec for rec in self.fwd if id(rec) not in keep]
        self.n_fwd_ops = len(self.fwd) + len(pro)
        self.fwd_prog = self._program(pro + self.fwd)
        if self.need_backward:
            pro = [(L.OP_MEMSET, L.RawOp(bs.base, 0, 0, bs.size)),
                   (L.OP_MEMSET, L.RawOp(self.grad_base, 0, 0, self.m._G.numel() * 4))]
            self._build_grad_buckets(pro)
        self.bytes_total = sum(a.size for a in self.arenas.values())

    def _build_grad_buckets(self, pro, nbuckets=6):
        """Cut the backward program into segments so that the flat gradient arena becomes final bucket by bucket, from its end
        (the temporal head, first in backward) to its start (the stem): bucket b = a contiguous range of the arena whose last
        writer record closes segment b, followed by the unpack of that range's weight gradients.  A data-parallel run starts
        the RCCL all-reduce of a bucket as soon as its segment has been enqueued (dist.GradBucketReducer), overlapping it with
        the rest of the backward; a single-GPU run just executes the segments back to back."""
        names = [n for n in self.m._poff]                         # arena order
        total = self.m._np
        last = len(self.bwd) - 1
        ready = {n: self._ready.get(n, -1) for n in names}        # -1: never written (frozen part of the model)
        # bucket boundaries at parameter starts, ~equal sizes
        bounds, acc_sz, target = [0], 0, max(1, total // nbuckets)
        for n in names:
            off = self.m._poff[n]
            if off - bounds[-1] >= target and off < total:
                bounds.append(off)
        bounds.append(total)
        buckets = []
        for b in range(len(bounds) - 1):
            lo, hi = bounds[b], bounds[b + 1]
            members = [n for n in names if lo <= self.m._poff[n] < hi]
            rdy = max([ready[n] for n in members] + [-1])
            # close the segment at a serial record (lane 0) so that no parallel section is cut in two
            while rdy >= 0 and rdy < last and (self.bwd[rdy + 1][2] & 0xff) != 0:
                rdy += 1
            buckets.append(dict(lo=lo, hi=hi, ready=rdy, members=set(members)))
        order = sorted(range(len(buckets)), key=lambda i: (buckets[i]["ready"], -buckets[i]["lo"]))
        self.bwd_segments, self._unpack_devs = [], []
        start = 0
        for rank, bi in enumerate(order):
            bk = buckets[bi]
            end = last + 1 if rank == len(order) - 1 else max(start, bk["ready"] + 1)
            recs = list(self.bwd[start:end])
            jobs = [j for j in self.unpack_jobs if j["_name"] in bk["members"]]
            if jobs:
                dev, nup, nbu = self._jobs(jobs, L.UnpackJob, 256)
                self._unpack_devs.append(dev)
                r = L.RawOp(dev.data_ptr(), 0, 0, 0)
                r.i[0:2] = [nup, nbu]
                recs.append((L.OP_UNPACK, r, L.LANE_ASYNC if self.lanes else 0))      # behind its weight gradients on the detached stream
            if rank == 0:
                recs = pro + recs
            self.bwd_segments.append((self._program(recs), bk["lo"], bk["hi"]))
            start = end
        assert start == last + 1
        # the same backward as ONE program: what a run without a bucket hook executes.
        # Its unpack work is cut into chunks that follow their weight-gradient launches ON THE DETACHED STREAM (same stream, in order, so
        # no event is needed): the slices of the first layers of the backward are summed while the chain is still running, and
        # only the last chunk is left when the chain ends (one final unpack used to add 0.5 ms after the join).
        recs, cursor, first, nwg = list(pro), 0, 0, 0
        self._unpack_chunks = []

        def flush(lane):
            nonlocal first
            jobs = self.unpack_jobs[first:cursor]
            if not jobs:
                return
            dev, nup, nbu = self._jobs(jobs, L.UnpackJob, 256)
            self._unpack_chunks.append(dev)
            r = L.RawOp(dev.data_ptr(), 0, 0, 0)
            r.i[0:2] = [nup, nbu]
            recs.append((L.OP_UNPACK, r, (lane | L.LANE_ASYNC) if self.lanes else 0))
            first = cursor

        chunk = getattr(self.m, "unpack_chunk", 48)              # weight gradients per detached unpack launch
        for rec in self.bwd:
            recs.append(rec)
            if rec[0] == L.OP_WGRAD:
                cursor += 1; nwg += 1
            elif rec[0] == L.OP_WGRAD_BATCH:
                cursor += rec[1].i[0]; nwg += 1
            else:
                if rec[0] == L.OP_BIAS_GRAD and rec[1].i[3] > 0:
                    cursor += 1                                  # (its partial-sum slices ride in the next unpack)
                continue
            if chunk > 0 and nwg % chunk == 0:
                flush(rec[2] & 0xff)
        assert cursor == len(self.unpack_jobs)
        flush(0)
        self.bwd_prog = self._program(recs)

    # ------------------------------------------------------------------ execution
    def _io(self, off, shape):
        a = self.arenas["misc"]
        n = 1
        for s in shape:
            n *= s
        o = off - a.buf.data_ptr()
        return a.buf[o:o + 4 * n].view(torch.float32).view(shape)

    def _run(self, prog, flags):
        """one host call per program, on the model's own interpreter context (mfc_ctx: its side streams / events are not shared with any other
        model or thread, include/mfcnet_hip.h)"""
        ctx = getattr(self.m, "_ctx", None)
        if ctx is None or ctx.device_index != self.device.index:
            ctx = self.m._ctx = L.Ctx(self.device.index if self.device.index is not None else torch.cuda.current_device())
        return L.lib.mfc_program_run_ctx(ctx.handle, prog, len(prog), L.stream_ptr(), flags)

    def run_forward(self, frames, flow, depth, ext_logits=None):
        B, T, H, W = self.B, self.T, self.H, self.W
        if self.ext_base:
            self._io(self.in_logits, (T * B, self.nc, H, W)).copy_(ext_logits.detach())
        else:
            for t in range(T):
                self._io(self.in_frames[t], (B, 3, H, W)).copy_(frames[t])
        for i, p in enumerate(self.in_flow):
            self._io(p, (B, 2, H, W)).copy_(flow[i])
        for i, p in enumerate(self.in_depth):
            self._io(p, (B, 1, H, W)).copy_(depth[i])
        rc = self._run(self.fwd_prog, 0)
        if rc != 0:
            raise L.MfcError(f"forward program failed: record {(-rc) // 1000 - 1 if rc <= -1000 else '?'} status {rc}")
        return self._io(self.out_buf, (B, self.nc, H, W)).clone()

    def run_backward(self, grad_out):
        self._io(self.gout_buf, (self.B, self.nc, self.H, self.W)).copy_(grad_out)
        hook = getattr(self.m, "grad_bucket_hook", None)
        if hook is None:
            rc = self._run(self.bwd_prog, 0)
            if rc != 0:
                raise L.MfcError(f"backward program failed: record {(-rc) // 1000 - 1 if rc <= -1000 else '?'} status {rc}")
            return self._ext_grad()
        nseg = len(self.bwd_segments)
        for i, (prog, lo, hi) in enumerate(self.bwd_segments):
            # every segment but the last leaves the detached stream un-joined (the chain does not wait for the weight gradients of the
            # segment); the bucket's consumer orders itself after them with mfc_wait_detached
            defer = self.lanes and i + 1 < nseg
            rc = self._run(prog, L.RUN_DEFER_JOIN if defer else 0)
            if rc != 0:
                raise L.MfcError(f"backward program failed: record {(-rc) // 1000 - 1 if rc <= -1000 else '?'} status {rc}")
            if hook is not None:
                # gradients [lo, hi) of the flat arena are final for a stream that (1) waits for the current stream and (2) has called
                # mfc_wait_detached -- dist.GradBucketReducer does both for the stream the all-reduce is ordered after
                hook(lo, hi)
        return self._ext_grad()

    def _ext_grad(self):
        """gradient w.r.t. the externally computed per-frame logits (None when the plan owns the per-frame network or it is frozen)"""
        if not (self.ext_base and self.need_backward and not self.base_frozen):
            return None
        return self._io(self.gin_buf, (self.T * self.B, self.nc, self.H, self.W)).clone()
