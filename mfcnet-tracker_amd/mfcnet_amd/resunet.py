"""ResUnet_VB on MI355X (inference): the weight-standardised-conv / GroupNorm / SiLU residual U-Net of models/resunet.py:97-180.

Same constructor arguments, attribute paths and `state_dict()` keys as the reference class (`init_conv`, `downs.{i}.0.block{1,2}.{proj,norm}`,
`downs.{i}.1.1`, `ups...`, `mid_block`, `final_res_block`, `output_layer`), so a checkpoint of the reference loads strictly; `forward(x)`
takes one `[B, channels, H, W]` fp32 tensor (H, W multiples of 2^(len(dim_mults)-1)) and returns `[B, out_dim, H, W]` -- with
`out_dim = num_classes` the shape contract of a per-frame `base_model` (multiframe_model.py:459-461).  All arithmetic runs in libmfcnet_hip.so through static programs of `mfc_op` records
(the convolutions are the same implicit-GEMM kernels as HRNet's; new here: mfc_ws_normalize, mfc_gn_finalize, the SiLU mode of
mfc_combine_fwd, mfc_upsample_nearest2x).  Under `torch.enable_grad()` the output carries a grad_fn whose backward is a second program
(data / weight gradients through the same convolution kernels; GroupNorm + SiLU backward = mfc_bnbwd_reduce with the SiLU-derivative
mask, mfc_gnbwd_finalize, mfc_bnbwd_apply in GroupNorm mode; mfc_ws_backward; mfc_upsample_nearest2x_bwd) and fills `.grad` of every
parameter, so `torch.optim` optimizers train it (round 3; `HRNetMultiLarge(..., base_model="resunet_vb")` uses it as the per-frame network).

GroupNorm on the conv epilogue's statistics: a convolution launched with images_per_group = 1 leaves per-(image, channel) sums; mfc_gn_finalize
folds the C/groups channels of each group into mean / rstd and writes the same [image][scale, shift, mean, rstd][channel] coefficient block the
BatchNorm path uses, so the normalisation itself costs no extra pass: it rides in the SiLU pass.  The pixel-unshuffle + 1x1 convolution of
`Downsample` (resunet.py:40-44) is ONE 2x2 / stride-2 convolution: the 1x1 weight [co, (c p1 p2)] IS a [co, c, 2, 2] kernel in memory.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .model import _DT, _get_node
from .plan import Arena, Ten, rup

GN_EPS, WS_EPS = 1e-5, 1e-5          # nn.GroupNorm default; WeightStandardizedConv2d for fp32 inputs (resunet.py:52)


def resunet_entries(channels, dim, out_dim, dim_mults, init_dim=None):
    """(name, shape) of every parameter in the reference's registration order (resunet.py:107-155; `ups` is registered before `mid_block`)."""
    init_dim = init_dim or dim
    dims = [init_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    t = []

    def conv(name, cout, cin, k):
        t.extend([(name + ".weight", (cout, cin, k, k)), (name + ".bias", (cout,))])

    def block(name, cin, cout):
        conv(name + ".proj", cout, cin, 3)
        t.extend([(name + ".norm.weight", (cout,)), (name + ".norm.bias", (cout,))])

    def resblock(name, cin, cout):
        block(name + ".block1", cin, cout)
        block(name + ".block2", cout, cout)
        if cin != cout:
            conv(name + ".res_conv", cout, cin, 1)

    conv("init_conv", init_dim, channels, 7)
    n = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        resblock(f"downs.{i}.0", ci, ci)
        conv(f"downs.{i}.1.1", co, ci * 4, 1) if i < n - 1 else conv(f"downs.{i}.1", co, ci, 3)
    for i, (ci, co) in enumerate(reversed(in_out)):
        resblock(f"ups.{i}.0", co + ci, co)
        conv(f"ups.{i}.1.1", ci, co, 3) if i < n - 1 else conv(f"ups.{i}.1", ci, co, 3)
    resblock("mid_block", dims[-1], dims[-1])
    resblock("final_res_block", dim * 2, dim)
    conv("output_layer", out_dim, dim, 1)
    return t


class _Conv:
    """bookkeeping of one convolution for the backward program"""
    def __init__(self, x, y, wname, wsrc, raw_w, cout, cin, k, stride, pad, bname, standardised):
        self.x, self.y, self.wname, self.wsrc, self.raw_w = x, y, wname, wsrc, raw_w
        self.cout, self.cin, self.k, self.stride, self.pad, self.bname, self.standardised = cout, cin, k, stride, pad, bname, standardised


class _Program:
    """Forward (and, with need_backward, backward) pass as flat arrays of mfc_op records over fixed arenas (the machinery of plan.Plan).
    Backward = the adjoint of every forward record in reverse order (resunet.py:46-180 under loss.backward()):
      conv            -> weight gradient (mfc_conv2d_wgrad + unpack; through mfc_ws_backward for the weight-standardised ones), bias gradient,
                         data gradient (the same convolution kernels with flipped / parity-class weight images)
      GroupNorm+SiLU  -> mfc_bnbwd_reduce (mask_mode 4 = SiLU derivative, per-image sums) -> mfc_gnbwd_finalize -> mfc_bnbwd_apply (gn_mode)
      h + res, cat    -> mfc_mask_add (copies / accumulates channel slices)
      nearest x2      -> mfc_upsample_nearest2x_bwd
    Gradients of tensors with several consumers (the block input feeds block1 and the residual; the skip tensors feed the down path and
    a later cat) accumulate: the first contribution writes, the others add."""

    def __init__(self, model, B, H, W, device, need_backward=False):
        self.m, self.B, self.H, self.W, self.device = model, B, H, W, device
        self.need_backward = need_backward
        self.dtype = model.compute_dtype
        self.esz = 2 if L.is16(self.dtype) else 4
        self.E = 16 // self.esz
        self.arenas = {k: Arena(k) for k in ("act", "stats", "misc", "bstats", "dwp")}
        self._build()
        for a in self.arenas.values():
            a.buf = torch.zeros(max(a.size, 256) + 256, dtype=torch.uint8, device=device)
            a.base = rup(a.buf.data_ptr(), 256)
        self._build()
        self._finalize()

    # ---- allocation
    def _alloc(self, arena, nbytes):
        a = self.arenas[arena]
        return a.base + a.alloc(nbytes)

    def tensor(self, N, H, W, C_, name="", needs_grad=True) -> Ten:
        Cp = rup(C_, 8)
        nb = N * H * W * Cp * self.esz
        return Ten(N, H, W, C_, Cp, 1, self._alloc("act", nb), nb, name, needs_grad and self.need_backward)

    def grad_of(self, t: Ten) -> Ten:
        if t.grad is None:
            t.grad = Ten(t.N, t.H, t.W, t.C, t.Cp, 1, self._alloc("act", t.nbytes), t.nbytes, "d" + t.name)
        return t.grad

    def p(self, name):
        return self.m._params[name].data_ptr()

    def g(self, name):
        return self.m._G.data_ptr() + 4 * self.m._goff[name]

    # ---- primitives
    def _pack(self, d, src, cout, cin, k, **tap):
        lay = L.conv_layout(d)
        assert (lay.TA, lay.TB) == (tap["TA"], tap["TB"])
        dst = self._alloc("act", lay.bytes)
        job = dict(src=src, dst=dst, Cout=cout, Cin=cin, KH=k, KW=k, **tap)
        job.update(L.pack_job_fields(lay))
        self.pack_jobs.append(job)
        d.wp = dst

    def conv(self, x: Ten, wname, cout, cin, k, stride, pad, stats=0, wsrc=None) -> Ten:
        """wsrc: the standardised copy of the weights (None: the parameter itself)"""
        Ho, Wo = (x.H + 2 * pad - k) // stride + 1, (x.W + 2 * pad - k) // stride + 1
        y = self.tensor(x.N, Ho, Wo, cout, wname)
        bname = wname[:-6] + "bias"
        src = wsrc if wsrc is not None else self.p(wname)
        d = L.ConvDesc(x.ptr, 0, y.ptr, self.p(bname), 0, stats, self.dtype, x.N, x.H, x.W, x.Cp, cin, Ho, Wo, y.Cp, cout, Ho, Wo,
                       k, k, -pad, -pad, stride, 1, 1, 0, 0, 0, 1, 0, 0, 0)         # images_per_group = 1: per-image statistics
        self._pack(d, src, cout, cin, k, TA=k, TB=k, kh0=0, kh_step=1, kw0=0, kw_step=1, mode=0)
        self.recs.append((L.OP_CONV, d))
        self.tape.append(("conv", _Conv(x, y, wname, src, self.p(wname), cout, cin, k, stride, pad, bname, wsrc is not None)))
        return y

    def combine(self, terms, act, out: Optional[Ten] = None, out_c_off=0, C_=None) -> Ten:
        t0 = terms[0][0]
        C_ = C_ if C_ is not None else t0.C
        if out is None:
            out = self.tensor(t0.N, t0.H, t0.W, C_, "cmb")
        d = L.CombineDesc()
        d.out = L.View(out.ptr, 0, out.H, out.W, out.Cp, out_c_off)
        for i, (t, coef) in enumerate(terms):
            d.src[i] = L.View(t.ptr, coef, t.H, t.W, t.Cp, 0)
        d.nsrc, d.relu, d.dtype, d.N, d.C, d.images_per_group = len(terms), act, self.dtype, out.N, rup(C_, self.E), 1
        self.recs.append((L.OP_COMBINE, d))
        return out

    def block(self, x: Ten, name, cin, cout) -> Ten:
        """Block.forward (resunet.py:69-76): weight-standardised 3x3 conv -> GroupNorm -> SiLU."""
        wn = name + ".proj.weight"
        ws = self._alloc("misc", cout * cin * 9 * 4)               # the standardised weights of this step
        r = L.RawOp(self.p(wn), ws, 0, 0)
        r.i[0], r.i[1], r.i[2] = cout, cin * 9, struct.unpack("i", struct.pack("f", self.m.ws_eps))[0]
        self.pro.append((L.OP_WSNORM, r))
        Cp = rup(cout, 8)
        stats = self._alloc("stats", L.STAT_REPLICAS * x.N * 2 * Cp * L.STAT_BYTES)
        y = self.conv(x, wn, cout, cin, 3, 1, 1, stats, wsrc=ws)
        coef = self._alloc("misc", x.N * 4 * Cp * 4)
        self.recs.append((L.OP_GNFIN, L.GnFinDesc(stats, coef, self.p(name + ".norm.weight"), self.p(name + ".norm.bias"), cout, Cp, x.N,
                                                  self.m.groups, float(y.H * y.W), GN_EPS)))
        out = self.combine([(y, coef)], 2)
        self.tape.append(("gnsilu", y, out, coef, name + ".norm"))
        return out

    def resblock(self, x: Ten, name, cin, cout) -> Ten:
        """ResnetBlock.forward (resunet.py:87-95)."""
        h = self.block(self.block(x, name + ".block1", cin, cout), name + ".block2", cout, cout)
        res = x if cin == cout else self.conv(x, name + ".res_conv.weight", cout, cin, 1, 1, 0)
        out = self.combine([(h, 0), (res, 0)], 0)
        self.tape.append(("add", [h, res], out))
        return out

    def cat(self, a: Ten, b: Ten) -> Ten:
        out = self.tensor(a.N, a.H, a.W, a.C + b.C, "cat")
        self.combine([(a, 0)], 0, out=out, out_c_off=0, C_=a.C)
        self.combine([(b, 0)], 0, out=out, out_c_off=a.C, C_=b.C)
        self.tape.append(("cat", [(a, 0), (b, a.C)], out))
        return out

    # ---- the network (resunet.py:157-180)
    def _build(self):
        for a in self.arenas.values():
            a.reset()
        self.recs, self.pro, self.pack_jobs, self.tape, self.bwd, self.unpack_jobs = [], [], [], [], [], []
        m, B, H, W = self.m, self.B, self.H, self.W
        self.in_buf = self._alloc("misc", B * m.channels * H * W * 4)
        self.out_buf = self._alloc("misc", B * m.out_dim * H * W * 4)
        self.gout_buf = self._alloc("misc", B * m.out_dim * H * W * 4) if self.need_backward else 0
        x = self.tensor(B, H, W, m.channels, "input", needs_grad=False)
        r = L.RawOp(self.in_buf, x.ptr, 0, 0)
        r.i[0:7] = [self.dtype, B, m.channels, H, W, x.Cp, 0]
        self.recs.append((L.OP_NCHW2NHWC, r))
        dims = [m.init_dim] + [m.dim * k for k in m.dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        n = len(in_out)
        x = self.conv(x, "init_conv.weight", m.init_dim, m.channels, 7, 1, 3)
        r0, hs = x, []
        for i, (ci, co) in enumerate(in_out):
            x = self.resblock(x, f"downs.{i}.0", ci, ci)
            hs.append(x)
            if i < n - 1:      # pixel-unshuffle + 1x1 == one 2x2 / stride-2 convolution over the same weight memory
                x = self.conv(x, f"downs.{i}.1.1.weight", co, ci, 2, 2, 0)
            else:
                x = self.conv(x, f"downs.{i}.1.weight", co, ci, 3, 1, 1)
        x = self.resblock(x, "mid_block", dims[-1], dims[-1])
        for i, (ci, co) in enumerate(reversed(in_out)):
            x = self.resblock(self.cat(x, hs.pop()), f"ups.{i}.0", co + ci, co)
            if i < n - 1:
                up = self.tensor(x.N, 2 * x.H, 2 * x.W, x.C, "up")
                rr = L.RawOp(x.ptr, up.ptr, 0, 0)
                rr.i[0:5] = [self.dtype, x.N, x.H, x.W, x.Cp]
                self.recs.append((L.OP_UPNEAR, rr))
                self.tape.append(("upnear", x, up))
                x = self.conv(up, f"ups.{i}.1.1.weight", ci, co, 3, 1, 1)
            else:
                x = self.conv(x, f"ups.{i}.1.weight", ci, co, 3, 1, 1)
        x = self.resblock(self.cat(x, r0), "final_res_block", m.dim * 2, m.dim)
        x = self.conv(x, "output_layer.weight", m.out_dim, m.dim, 1, 1, 0)
        r = L.RawOp(x.ptr, self.out_buf, 0, 0)
        r.i[0:6] = [self.dtype, B, m.out_dim, H, W, x.Cp]
        self.recs.append((L.OP_NHWC2NCHW, r))
        self.tape.append(("out", x))
        if self.need_backward:
            self._emit_backward()

    # ---- backward program
    @staticmethod
    def _s2_class(k, pad, ph):
        par = (ph + pad) % 2
        khmax = k - 1 if (k - 1) % 2 == par else k - 2
        return khmax // 2 + 1, khmax, (ph + pad - khmax) // 2

    def _view(self, t: Ten, c_off=0):
        return L.View(t.ptr, 0, t.H, t.W, t.Cp, c_off)

    def _mask_add(self, g: Ten, g_off, dst: Ten, C_):
        md = L.MaskAddDesc()
        gd = self.grad_of(dst)
        md.g, md.dst = self._view(g, g_off), self._view(gd)
        md.mask_mode, md.dtype, md.N, md.C, md.accumulate = 0, self.dtype, dst.N, rup(C_, self.E), 1 if dst.grad_init else 0
        self.bwd.append((L.OP_MASK_ADD, md))
        dst.grad_init = True

    def _emit_backward(self):
        E = self.E
        for op in reversed(self.tape):
            kind = op[0]
            if kind == "out":
                o = op[1]
                g = self.grad_of(o)
                r = L.RawOp(self.gout_buf, g.ptr, 0, 0)
                r.i[0:7] = [self.dtype, self.B, self.m.out_dim, self.H, self.W, g.Cp, 0]
                self.bwd.append((L.OP_NCHW2NHWC, r))
                o.grad_init = True
            elif kind == "add":
                _, terms, out = op
                for t in terms:
                    self._mask_add(self.grad_of(out), 0, t, t.C)
            elif kind == "cat":
                _, parts, out = op
                for t, off in parts:
                    if t.needs_grad:
                        self._mask_add(self.grad_of(out), off, t, t.C)
            elif kind == "upnear":
                _, x, up = op
                gx, gu = self.grad_of(x), self.grad_of(up)
                r = L.RawOp(gu.ptr, gx.ptr, 0, 0)
                r.i[0:6] = [self.dtype, x.N, x.H, x.W, x.Cp, 1 if x.grad_init else 0]
                self.bwd.append((L.OP_UPNEAR_BWD, r))
                x.grad_init = True
            elif kind == "gnsilu":
                _, y, out, coef, nname = op
                Cs = rup(y.C, E)
                g, gy = self.grad_of(out), self.grad_of(y)
                R = L.STAT_REPLICAS
                bstats = self._alloc("bstats", R * y.N * 2 * y.Cp * L.STAT_BYTES)
                bcoef = self._alloc("misc", y.N * 2 * y.Cp * 4)

                def desc(dy):
                    d = L.BnBwdDesc()
                    d.g, d.y = self._view(g), L.View(y.ptr, coef, y.H, y.W, y.Cp, 0)
                    if dy is not None:
                        d.dy = dy
                    d.bstats, d.bcoef = bstats, bcoef
                    d.mask_mode, d.dtype, d.N, d.C, d.images_per_group, d.accumulate = 4, self.dtype, y.N, Cs, 1, 0
                    return d
                self.bwd.append((L.OP_BNBWD_REDUCE, desc(None)))
                self.bwd.append((L.OP_GNBWD_FIN, L.GnBwdFinDesc(bstats, bcoef, self.p(nname + ".weight"), self.g(nname + ".weight"), self.g(nname + ".bias"),
                                                                y.C, y.Cp, y.N, self.m.groups, float(y.H * y.W), 0)))
                ad = desc(self._view(gy))
                ad.gn_mode = 1
                self.bwd.append((L.OP_BNBWD_APPLY, ad))
                y.grad_init = True
            elif kind == "conv":
                ci: _Conv = op[1]
                x, y, k, s, pad = ci.x, ci.y, ci.k, ci.stride, ci.pad
                dy = self.grad_of(y)
                assert y.grad_init, ci.wname
                # ---- weight gradient (partial-sum slices -> unpack -> reference layout; through the standardisation for WS convs)
                Co16, Ci16 = rup(ci.cout, 16), rup(ci.cin, 16)
                wg = L.WgradDesc(x.ptr, dy.ptr, 16, 0, self.dtype, x.N, x.H, x.W, x.Cp, x.C, y.H, y.W, y.Cp, ci.cout, k, k, -pad, -pad,
                                 s, 0, 1, 0, 0, 0)
                parts = L.wgrad_parts(wg)
                wg.dwp = self._alloc("dwp", parts * k * k * Co16 * Ci16 * 4)
                self.bwd.append((L.OP_WGRAD, wg))
                dst = self._alloc("misc", ci.cout * ci.cin * k * k * 4) if ci.standardised else self.g(ci.wname)
                self.unpack_jobs.append(dict(src=wg.dwp, nparts=parts, dst=dst, Cout=ci.cout, Cin=ci.cin, KH=k, KW=k, Co16=Co16, Ci16=Ci16))
                self.bwd.append(("unpack", len(self.unpack_jobs) - 1))
                if ci.standardised:
                    r = L.RawOp(ci.raw_w, dst, self.g(ci.wname), 0)
                    r.i[0], r.i[1], r.i[2] = ci.cout, ci.cin * k * k, struct.unpack("i", struct.pack("f", self.m.ws_eps))[0]
                    self.bwd.append((L.OP_WSBWD, r))
                r = L.RawOp(dy.ptr, self.g(ci.bname), 0, y.N * y.H * y.W)
                r.i[0:4] = [self.dtype, y.Cp, ci.cout, 0]
                self.bwd.append((L.OP_BIAS_GRAD, r))
                # ---- data gradient
                if x.needs_grad:
                    dx = self.grad_of(x)
                    acc = 1 if x.grad_init else 0
                    if s == 1:
                        d = L.ConvDesc(dy.ptr, 0, dx.ptr, 0, 0, 0, self.dtype, y.N, y.H, y.W, y.Cp, ci.cout, x.H, x.W, x.Cp, x.C, x.H, x.W,
                                       k, k, -(k - 1 - pad), -(k - 1 - pad), 1, 1, 1, 0, 0, 0, 1, acc, 0, 0)
                        self._pack(d, ci.wsrc, ci.cout, ci.cin, k, TA=k, TB=k, kh0=k - 1, kh_step=-1, kw0=k - 1, kw_step=-1, mode=1)
                        self.bwd.append((L.OP_CONV, d))
                    else:
                        for ph in range(2):
                            for pw in range(2):
                                ta, kh0, dh0 = self._s2_class(k, pad, ph)
                                tb, kw0, dw0 = self._s2_class(k, pad, pw)
                                Hl, Wl = (x.H - ph + 1) // 2, (x.W - pw + 1) // 2
                                if Hl <= 0 or Wl <= 0:
                                    continue
                                d = L.ConvDesc(dy.ptr, 0, dx.ptr, 0, 0, 0, self.dtype, y.N, y.H, y.W, y.Cp, ci.cout, x.H, x.W, x.Cp, x.C, Hl, Wl,
                                               ta, tb, dh0, dw0, 1, 2, 2, ph, pw, 0, 1, acc, 0, 0)
                                self._pack(d, ci.wsrc, ci.cout, ci.cin, k, TA=ta, TB=tb, kh0=kh0, kh_step=-2, kw0=kw0, kw_step=-2, mode=1)
                                self.bwd.append((L.OP_CONV, d))
                    x.grad_init = True

    def _jobs(self, jobs, cls, single=False):
        arr = (cls * len(jobs))()
        b0 = 0
        for i, j in enumerate(jobs):
            for k_, v in j.items():
                setattr(arr[i], k_, v)
            if cls is L.PackJob:
                total = (j["TA"] // j["TAS"]) * j["nchunks"] * j["Yblocks"] * j["nslots"] * j["NT16"]
            else:
                total = j["KH"] * j["KW"] * j["Co16"] * j["Ci16"]
            nb = -(-total // 256)
            arr[i].block0, arr[i].nblocks = (0 if single else b0), nb
            b0 += nb
        return arr, b0

    def _prog(self, recs):
        prog = (L.Op * len(recs))()
        for i, (kind, d) in enumerate(recs):
            prog[i].kind, prog[i].lane = kind, 0
            C.memmove(C.byref(prog[i].u), C.byref(d), C.sizeof(d))
        return prog

    def _finalize(self):
        jobs, b0 = self._jobs(self.pack_jobs, L.PackJob)
        self._pack_dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
        st = self.arenas["stats"]
        pro = [(L.OP_MEMSET, L.RawOp(st.base, 0, 0, st.size))] + self.pro
        r = L.RawOp(self._pack_dev.data_ptr(), 0, 0, 0)
        r.i[0:3] = [len(self.pack_jobs), b0, self.dtype]
        pro.append((L.OP_PACK, r))
        self.prog = self._prog(pro + self.recs)
        if self.need_backward:
            uj, _ = self._jobs(self.unpack_jobs, L.UnpackJob, single=True)       # every job is launched on its own, right behind its weight gradient
            self._unpack_dev = torch.frombuffer(bytearray(bytes(uj)), dtype=torch.uint8).to(self.device)
            bs = self.arenas["bstats"]
            recs = [(L.OP_MEMSET, L.RawOp(bs.base, 0, 0, max(bs.size, 16))),
                    (L.OP_MEMSET, L.RawOp(self.m._G.data_ptr(), 0, 0, self.m._G.numel() * 4))]
            for kind, d in self.bwd:
                if kind == "unpack":
                    r = L.RawOp(self._unpack_dev.data_ptr() + d * C.sizeof(L.UnpackJob), 0, 0, 0)
                    r.i[0:2] = [1, uj[d].nblocks]
                    recs.append((L.OP_UNPACK, r))
                else:
                    recs.append((kind, d))
            self.bwd_prog = self._prog(recs)

    def _io(self, off, shape):
        a = self.arenas["misc"]
        n = 1
        for s in shape:
            n *= s
        o = off - a.buf.data_ptr()
        return a.buf[o:o + 4 * n].view(torch.float32).view(shape)

    def run(self, x):
        m = self.m
        self._io(self.in_buf, (self.B, m.channels, self.H, self.W)).copy_(x)
        rc = L.lib.mfc_program_run(self.prog, len(self.prog), L.stream_ptr())
        if rc != 0:
            raise L.MfcError(f"ResUnet_VB program failed: record {(-rc) // 1000 - 1 if rc <= -1000 else '?'} status {rc}")
        return self._io(self.out_buf, (self.B, m.out_dim, self.H, self.W)).clone()

    def run_backward(self, gout):
        self._io(self.gout_buf, (self.B, self.m.out_dim, self.H, self.W)).copy_(gout)
        rc = L.lib.mfc_program_run(self.bwd_prog, len(self.bwd_prog), L.stream_ptr())
        if rc != 0:
            raise L.MfcError(f"ResUnet_VB backward program failed: record {(-rc) // 1000 - 1 if rc <= -1000 else '?'} status {rc}")


class _ResFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, prog, x):
        ctx.model, ctx.prog = model, prog
        return prog.run(x)

    @staticmethod
    def backward(ctx, gout):
        ctx.model._run_backward(ctx.prog, gout)
        return None, None, None, None


class ResUnet_VB(nn.Module):
    """models/resunet.py:97-180 (constructor signature of the reference; `compute_dtype` selects fp32 parity / bf16 throughput kernels)."""

    def __init__(self, channels, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), resnet_block_groups=8, compute_dtype="fp32"):
        super().__init__()
        self.channels, self.dim, self.init_dim = channels, dim, init_dim if init_dim is not None else dim
        self.out_dim = out_dim if out_dim is not None else channels          # resunet.py:152-153
        self.dim_mults, self.groups = tuple(dim_mults), resnet_block_groups
        self.compute_dtype = _DT[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype
        if any((dim * k) % 8 for k in self.dim_mults) or self.init_dim % 8:
            raise ValueError("ResUnet_VB on MI355X: channel counts must be multiples of 8 (16-byte NHWC granules)")
        for c in [self.init_dim, dim] + [dim * k for k in self.dim_mults]:
            if c % resnet_block_groups:          # what nn.GroupNorm(groups, c) raises in the reference's constructor (resunet.py:64)
                raise ValueError("num_channels must be divisible by num_groups")
        # WeightStandardizedConv2d.forward picks eps from the INPUT dtype (resunet.py:52): 1e-5 for float32, 1e-3 otherwise.  This module
        # always takes float32 tensors at its boundary, so 1e-5 whatever the storage type of the kernels (compute_dtype is an
        # implementation detail below the boundary; a reference run under autocast would see half inputs and use 1e-3 -- set ws_eps then).
        self.ws_eps = WS_EPS
        self._params: Dict[str, nn.Parameter] = {}
        for name, shape in resunet_entries(channels, dim, self.out_dim, self.dim_mults, self.init_dim):
            path = name.split(".")
            node, leaf = _get_node(self, path[:-1]), path[-1]
            prm = nn.Parameter(torch.empty(shape))
            node.register_parameter(leaf, prm)
            self._params[name] = prm
        self._default_init()
        self._progs = {}
        self._goff, off = {}, 0
        for name, prm in self._params.items():
            self._goff[name] = off
            off += (prm.numel() + 3) // 4 * 4
        self._gtotal = off
        self._G = torch.zeros(1)
        self._anchor = torch.zeros(1, requires_grad=True)

    def _default_init(self):
        """PyTorch default initialisers of nn.Conv2d / nn.GroupNorm (the reference calls no init function)."""
        import math
        with torch.no_grad():
            for name, prm in self._params.items():
                if name.endswith("norm.weight"):
                    prm.fill_(1.0)
                elif name.endswith("norm.bias"):
                    prm.zero_()
                elif name.endswith(".weight"):
                    b = 1.0 / math.sqrt(prm.shape[1] * prm.shape[2] * prm.shape[3])
                    prm.uniform_(-b, b)
                    self._params[name[:-6] + "bias"].uniform_(-b, b)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._params = {n: p for n, p in self.named_parameters()}
        self._progs = {}
        dev = next(iter(self._params.values())).device
        self._G = torch.zeros(self._gtotal, dtype=torch.float32, device=dev)          # flat gradient buffer the backward kernels write
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        return self

    def _run_backward(self, prog, gout):
        """backward program, then autograd's rule per parameter: a `.grad` that is still set accumulates, None gets a fresh tensor"""
        prog.run_backward(gout)
        for name, prm in self._params.items():
            if not prm.requires_grad:
                continue
            off = self._goff[name]
            g = self._G[off:off + prm.numel()].view(prm.shape)
            if prm.grad is None:
                prm.grad = g.clone()
            else:
                prm.grad.add_(g)

    def forward(self, captimgs, *args, **kwargs):
        x = captimgs
        if not x.is_cuda or not next(iter(self._params.values())).is_cuda:
            raise L.MfcError("the MI355X ResUnet_VB runs on a GPU only (no CPU fallback): move model and input to cuda")
        B, c, H, W = x.shape
        q = 2 ** (len(self.dim_mults) - 1)
        if c != self.channels or H % q or W % q:
            raise ValueError(f"input must be [B,{self.channels},H,W] with H, W multiples of {q}, got {tuple(x.shape)}")
        need_bwd = torch.is_grad_enabled() and any(p.requires_grad for p in self._params.values())
        key = (B, H, W, x.device, need_bwd, tuple(p.data_ptr() for p in list(self._params.values())[:4]))
        prog = self._progs.get(key)
        if prog is None:
            if len(self._progs) >= 2:
                self._progs.clear()
            prog = self._progs[key] = _Program(self, B, H, W, x.device, need_backward=need_bwd)
        if need_bwd:
            return _ResFn.apply(self._anchor, self, prog, x.detach().float())
        return prog.run(x.detach().float())
