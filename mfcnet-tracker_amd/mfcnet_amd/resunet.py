"""ResUnet_VB on MI355X (inference): the weight-standardised-conv / GroupNorm / SiLU residual U-Net of models/resunet.py:97-180.

Same constructor arguments, attribute paths and `state_dict()` keys as the reference class (`init_conv`, `downs.{i}.0.block{1,2}.{proj,norm}`,
`downs.{i}.1.1`, `ups...`, `mid_block`, `final_res_block`, `output_layer`), so a checkpoint of the reference loads strictly; `forward(x)`
takes one `[B, channels, H, W]` fp32 tensor (H, W multiples of 2^(len(dim_mults)-1)) and returns `[B, out_dim, H, W]` -- with
`out_dim = num_classes` the shape contract of a per-frame `base_model` (multiframe_model.py:459-461).  Nothing in the reference constructs
this class, so only the forward pass is built: all arithmetic in libmfcnet_hip.so through a static program of `mfc_op` records
(the convolutions are the same implicit-GEMM kernels as HRNet's; new here: mfc_ws_normalize, mfc_gn_finalize, the SiLU mode of
mfc_combine_fwd, mfc_upsample_nearest2x).  There is no backward: parameters are exposed for checkpoints, outputs carry no grad_fn.

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


class _Program:
    """The forward pass as a flat array of mfc_op records over fixed arenas (the machinery of plan.Plan, without a backward)."""

    def __init__(self, model, B, H, W, device):
        self.m, self.B, self.H, self.W, self.device = model, B, H, W, device
        self.dtype = model.compute_dtype
        self.esz = 2 if L.is16(self.dtype) else 4
        self.E = 16 // self.esz
        self.arenas = {k: Arena(k) for k in ("act", "stats", "misc")}
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

    def tensor(self, N, H, W, C_, name="") -> Ten:
        Cp = rup(C_, 8)
        nb = N * H * W * Cp * self.esz
        return Ten(N, H, W, C_, Cp, 1, self._alloc("act", nb), nb, name, False)

    def p(self, name):
        return self.m._params[name].data_ptr()

    # ---- primitives
    def conv(self, x: Ten, wsrc, cout, cin, k, stride, pad, bias, stats=0) -> Ten:
        Ho, Wo = (x.H + 2 * pad - k) // stride + 1, (x.W + 2 * pad - k) // stride + 1
        y = self.tensor(x.N, Ho, Wo, cout)
        d = L.ConvDesc(x.ptr, 0, y.ptr, bias, 0, stats, self.dtype, x.N, x.H, x.W, x.Cp, cin, Ho, Wo, y.Cp, cout, Ho, Wo,
                       k, k, -pad, -pad, stride, 1, 1, 0, 0, 0, 1, 0, 0, 0)         # images_per_group = 1: per-image statistics
        lay = L.conv_layout(d)
        dst = self._alloc("act", lay.bytes)
        job = dict(src=wsrc, dst=dst, Cout=cout, Cin=cin, KH=k, KW=k, TA=k, TB=k, kh0=0, kh_step=1, kw0=0, kw_step=1, mode=0)
        job.update(L.pack_job_fields(lay))
        self.pack_jobs.append(job)
        d.wp = dst
        self.recs.append((L.OP_CONV, d))
        return y

    def combine(self, terms, act, out: Optional[Ten] = None, out_c_off=0, C_=None) -> Ten:
        t0 = terms[0][0]
        C_ = C_ if C_ is not None else t0.C
        if out is None:
            out = self.tensor(t0.N, t0.H, t0.W, C_)
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
        r.i[0], r.i[1], r.i[2] = cout, cin * 9, struct.unpack("i", struct.pack("f", WS_EPS))[0]
        self.pro.append((L.OP_WSNORM, r))
        Cp = rup(cout, 8)
        stats = self._alloc("stats", L.STAT_REPLICAS * x.N * 2 * Cp * L.STAT_BYTES)
        y = self.conv(x, ws, cout, cin, 3, 1, 1, self.p(name + ".proj.bias"), stats)
        coef = self._alloc("misc", x.N * 4 * Cp * 4)
        self.recs.append((L.OP_GNFIN, L.GnFinDesc(stats, coef, self.p(name + ".norm.weight"), self.p(name + ".norm.bias"), cout, Cp, x.N,
                                                  self.m.groups, float(y.H * y.W), GN_EPS)))
        return self.combine([(y, coef)], 2)

    def resblock(self, x: Ten, name, cin, cout) -> Ten:
        """ResnetBlock.forward (resunet.py:87-95)."""
        h = self.block(self.block(x, name + ".block1", cin, cout), name + ".block2", cout, cout)
        res = x if cin == cout else self.conv(x, self.p(name + ".res_conv.weight"), cout, cin, 1, 1, 0, self.p(name + ".res_conv.bias"))
        return self.combine([(h, 0), (res, 0)], 0)

    def cat(self, a: Ten, b: Ten) -> Ten:
        out = self.tensor(a.N, a.H, a.W, a.C + b.C)
        self.combine([(a, 0)], 0, out=out, out_c_off=0, C_=a.C)
        self.combine([(b, 0)], 0, out=out, out_c_off=a.C, C_=b.C)
        return out

    # ---- the network (resunet.py:157-180)
    def _build(self):
        for a in self.arenas.values():
            a.reset()
        self.recs, self.pro, self.pack_jobs = [], [], []
        m, B, H, W = self.m, self.B, self.H, self.W
        self.in_buf = self._alloc("misc", B * m.channels * H * W * 4)
        self.out_buf = self._alloc("misc", B * m.out_dim * H * W * 4)
        x = self.tensor(B, H, W, m.channels, "input")
        r = L.RawOp(self.in_buf, x.ptr, 0, 0)
        r.i[0:7] = [self.dtype, B, m.channels, H, W, x.Cp, 0]
        self.recs.append((L.OP_NCHW2NHWC, r))
        dims = [m.init_dim] + [m.dim * k for k in m.dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        n = len(in_out)
        x = self.conv(x, self.p("init_conv.weight"), m.init_dim, m.channels, 7, 1, 3, self.p("init_conv.bias"))
        r0, hs = x, []
        for i, (ci, co) in enumerate(in_out):
            x = self.resblock(x, f"downs.{i}.0", ci, ci)
            hs.append(x)
            if i < n - 1:      # pixel-unshuffle + 1x1 == one 2x2 / stride-2 convolution over the same weight memory
                x = self.conv(x, self.p(f"downs.{i}.1.1.weight"), co, ci, 2, 2, 0, self.p(f"downs.{i}.1.1.bias"))
            else:
                x = self.conv(x, self.p(f"downs.{i}.1.weight"), co, ci, 3, 1, 1, self.p(f"downs.{i}.1.bias"))
        x = self.resblock(x, "mid_block", dims[-1], dims[-1])
        for i, (ci, co) in enumerate(reversed(in_out)):
            x = self.resblock(self.cat(x, hs.pop()), f"ups.{i}.0", co + ci, co)
            if i < n - 1:
                up = self.tensor(x.N, 2 * x.H, 2 * x.W, x.C)
                rr = L.RawOp(x.ptr, up.ptr, 0, 0)
                rr.i[0:5] = [self.dtype, x.N, x.H, x.W, x.Cp]
                self.recs.append((L.OP_UPNEAR, rr))
                x = self.conv(up, self.p(f"ups.{i}.1.1.weight"), ci, co, 3, 1, 1, self.p(f"ups.{i}.1.1.bias"))
            else:
                x = self.conv(x, self.p(f"ups.{i}.1.weight"), ci, co, 3, 1, 1, self.p(f"ups.{i}.1.bias"))
        x = self.resblock(self.cat(x, r0), "final_res_block", m.dim * 2, m.dim)
        x = self.conv(x, self.p("output_layer.weight"), m.out_dim, m.dim, 1, 1, 0, self.p("output_layer.bias"))
        r = L.RawOp(x.ptr, self.out_buf, 0, 0)
        r.i[0:6] = [self.dtype, B, m.out_dim, H, W, x.Cp]
        self.recs.append((L.OP_NHWC2NCHW, r))

    def _finalize(self):
        jobs = (L.PackJob * len(self.pack_jobs))()
        b0 = 0
        for i, j in enumerate(self.pack_jobs):
            for k_, v in j.items():
                setattr(jobs[i], k_, v)
            total = (j["TA"] // j["TAS"]) * j["nchunks"] * j["Yblocks"] * j["nslots"] * j["NT16"]
            jobs[i].block0, jobs[i].nblocks = b0, -(-total // 256)
            b0 += jobs[i].nblocks
        self._pack_dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
        st = self.arenas["stats"]
        pro = [(L.OP_MEMSET, L.RawOp(st.base, 0, 0, st.size))] + self.pro
        r = L.RawOp(self._pack_dev.data_ptr(), 0, 0, 0)
        r.i[0:3] = [len(self.pack_jobs), b0, self.dtype]
        pro.append((L.OP_PACK, r))
        recs = pro + self.recs
        self.prog = (L.Op * len(recs))()
        for i, (kind, d) in enumerate(recs):
            self.prog[i].kind, self.prog[i].lane = kind, 0
            C.memmove(C.byref(self.prog[i].u), C.byref(d), C.sizeof(d))

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


class ResUnet_VB(nn.Module):
    """models/resunet.py:97-180 (constructor signature of the reference; `compute_dtype` selects fp32 parity / bf16 throughput kernels)."""

    def __init__(self, channels, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), resnet_block_groups=8, compute_dtype="fp32"):
        super().__init__()
        self.channels, self.dim, self.init_dim = channels, dim, init_dim if init_dim is not None else dim
        self.out_dim = out_dim if out_dim is not None else channels          # resunet.py:152-153
        self.dim_mults, self.groups = tuple(dim_mults), resnet_block_groups
        self.compute_dtype = _DT[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype
        if any((dim * k) % 8 for k in self.dim_mults) or self.init_dim % 8 or dim % resnet_block_groups:
            raise ValueError("ResUnet_VB on MI355X: channel counts must be multiples of 8 (16-byte NHWC granules) and of the GroupNorm group count")
        self._params: Dict[str, nn.Parameter] = {}
        for name, shape in resunet_entries(channels, dim, self.out_dim, self.dim_mults, self.init_dim):
            path = name.split(".")
            node, leaf = _get_node(self, path[:-1]), path[-1]
            prm = nn.Parameter(torch.empty(shape))
            node.register_parameter(leaf, prm)
            self._params[name] = prm
        self._default_init()
        self._progs = {}

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
        return self

    def forward(self, captimgs, *args, **kwargs):
        x = captimgs
        if not x.is_cuda or not next(iter(self._params.values())).is_cuda:
            raise L.MfcError("the MI355X ResUnet_VB runs on a GPU only (no CPU fallback): move model and input to cuda")
        B, c, H, W = x.shape
        q = 2 ** (len(self.dim_mults) - 1)
        if c != self.channels or H % q or W % q:
            raise ValueError(f"input must be [B,{self.channels},H,W] with H, W multiples of {q}, got {tuple(x.shape)}")
        key = (B, H, W, x.device, tuple(p.data_ptr() for p in list(self._params.values())[:4]))
        prog = self._progs.get(key)
        if prog is None:
            self._progs.clear()
            prog = self._progs[key] = _Program(self, B, H, W, x.device)
        return prog.run(x.detach().float())
