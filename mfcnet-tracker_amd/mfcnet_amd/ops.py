"""Single-op Python wrappers over the C ABI (used by the parity tests and for bring-up).

Tensors are torch CUDA tensors used purely as device memory: NHWC `[N,H,W,Cp]` activations in
float32 or bfloat16, fp32 parameters in the reference layout.  Nothing here computes with torch.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def rup(x, m):
    return (x + m - 1) // m * m


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    if t.dtype == torch.float16:
        return L.F16
    raise L.MfcError(f"unsupported dtype {t.dtype}")


def to_nhwc(x_nchw: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """[N,C,H,W] fp32 (any device) -> device NHWC [N,H,W,Cp] through mfc_nchw_to_nhwc."""
    x = x_nchw.detach().to("cuda", torch.float32).contiguous()
    N, Cc, H, W = x.shape
    Cp = rup(Cc, 8)
    out = torch.zeros(N, H, W, Cp, dtype=dtype, device="cuda")
    L.check(L.lib.mfc_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), dt_of(out), N, Cc, H, W, Cp, 0, 1, L.stream_ptr()))
    return out


def to_nchw(x_nhwc: torch.Tensor, Cc: int) -> torch.Tensor:
    N, H, W, Cp = x_nhwc.shape
    out = torch.empty(N, Cc, H, W, dtype=torch.float32, device="cuda")
    L.check(L.lib.mfc_nhwc_to_nchw(x_nhwc.data_ptr(), out.data_ptr(), dt_of(x_nhwc), N, Cc, H, W, Cp, L.stream_ptr()))
    return out


def _run_jobs(jobs, cls, fn, *extra):
    arr = (cls * len(jobs))()
    b0 = 0
    for i, j in enumerate(jobs):
        for k, v in j.items():
            setattr(arr[i], k, v)
        total = ((j["TA"] // j["TAS"]) * j["nchunks"] * j["Yblocks"] * j["nslots"] * j["NT16"] if cls is L.PackJob
                 else j["KH"] * j["KW"] * j["Co16"] * j["Ci16"])
        arr[i].block0, arr[i].nblocks = b0, -(-total // 256)
        b0 += arr[i].nblocks
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    L.check(fn(dev.data_ptr(), len(jobs), b0, *extra, L.stream_ptr()))
    torch.cuda.current_stream().synchronize()       # jobs table must outlive the launch


def s2_class(k, pad, ph):
    par = (ph + pad) % 2
    khmax = k - 1 if (k - 1) % 2 == par else k - 2
    return khmax // 2 + 1, khmax, (ph + pad - khmax) // 2


def pack_weight(w: torch.Tensor, desc: L.ConvDesc, mode="fwd", cls=(0, 0)) -> torch.Tensor:
    """Pack fp32 [Cout,Cin,k,k] weights (cuda) into the image the launch `desc` reads.
    mode: 'fwd' | 'dgrad' (stride 1) | 'dgrad_s2' (output parity class cls) | 'dgrad_s2_all' (MFC_CONV_S2_CLASSES: the four class images of a
    3x3 / stride-2 data gradient, one after the other; taps outside the filter are zero)."""
    Cout, Cin, k, _ = w.shape
    pad = k // 2
    lay = L.conv_layout(desc)
    if mode == "dgrad_s2_all":
        assert k == 3 and (lay.TA, lay.TB) == (2, 2) and lay.bytes % 4 == 0
        out = torch.empty(lay.bytes, dtype=torch.uint8, device="cuda")
        jobs = []
        for ph in range(2):
            for pw in range(2):
                j = dict(TA=2, TB=2, kh0=1 + ph, kh_step=-2, kw0=1 + pw, kw_step=-2, mode=1)
                j.update(L.pack_job_fields(lay))
                j.update(src=w.data_ptr(), dst=out.data_ptr() + (2 * ph + pw) * (lay.bytes // 4), Cout=Cout, Cin=Cin, KH=k, KW=k)
                jobs.append(j)
        _run_jobs(jobs, L.PackJob, L.lib.mfc_pack_weights, desc.dtype)
        return out
    if mode == "fwd":
        j = dict(TA=k, TB=k, kh0=0, kh_step=1, kw0=0, kw_step=1, mode=0)
    elif mode == "dgrad":
        j = dict(TA=k, TB=k, kh0=k - 1, kh_step=-1, kw0=k - 1, kw_step=-1, mode=1)
    else:
        ta, kh0, _ = s2_class(k, pad, cls[0])
        tb, kw0, _ = s2_class(k, pad, cls[1])
        j = dict(TA=ta, TB=tb, kh0=kh0, kh_step=-2, kw0=kw0, kw_step=-2, mode=1)
    assert (j["TA"], j["TB"]) == (lay.TA, lay.TB)
    j.update(L.pack_job_fields(lay))
    out = torch.empty(lay.bytes, dtype=torch.uint8, device="cuda")
    j.update(src=w.data_ptr(), dst=out.data_ptr(), Cout=Cout, Cin=Cin, KH=k, KW=k)
    _run_jobs([j], L.PackJob, L.lib.mfc_pack_weights, desc.dtype)
    return out


def conv2d(x, w, k, stride=1, bias=None, in_coef=None, in_relu=False, ipg=None, stats=None, tile=(0, 0)):
    """Forward conv on NHWC x [N,H,W,Cp] with fp32 reference-layout weights w [Cout,Cin,k,k]."""
    N, H, W, Cp = x.shape
    Cout, Cin = w.shape[0], w.shape[1]
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.zeros(N, Ho, Wo, rup(Cout, 8), dtype=x.dtype, device="cuda")
    d = L.ConvDesc(x.data_ptr(), 0, out.data_ptr(), bias.data_ptr() if bias is not None else 0,
                   in_coef.data_ptr() if in_coef is not None else 0, stats.data_ptr() if stats is not None else 0,
                   dt_of(x), N, H, W, Cp, Cin, Ho, Wo, out.shape[3], Cout, Ho, Wo, k, k, -pad, -pad, stride, 1, 1, 0, 0,
                   1 if in_relu else 0, ipg or N, 0, tile[0], tile[1])
    wp = pack_weight(w, d, "fwd")
    d.wp = wp.data_ptr()
    L.call(L.lib.mfc_conv2d_fwd, d)
    return out


def conv2d_dgrad(dy, w, k, stride, in_hw, accumulate_into=None, merged_s2=False):
    """Data gradient: dy NHWC [N,Ho,Wo,Cop] -> dx NHWC [N,H,W,Cip].  merged_s2: a 3x3 / stride-2 gradient as ONE launch over its four output
    parity classes (MFC_CONV_S2_CLASSES) instead of four launches."""
    N, Ho, Wo, Cop = dy.shape
    Cout, Cin = w.shape[0], w.shape[1]
    H, W = in_hw
    pad = k // 2
    dx = accumulate_into if accumulate_into is not None else torch.zeros(N, H, W, rup(Cin, 8), dtype=dy.dtype, device="cuda")
    acc = 1 if accumulate_into is not None else 0
    if stride == 1:
        d = L.ConvDesc(dy.data_ptr(), 0, dx.data_ptr(), 0, 0, 0, dt_of(dy), N, Ho, Wo, Cop, Cout, H, W, dx.shape[3],
                       Cin, H, W, k, k, -(k - 1 - pad), -(k - 1 - pad), 1, 1, 1, 0, 0, 0, N, acc, 0, 0)
        wp = pack_weight(w, d, "dgrad")
        d.wp = wp.data_ptr()
        L.call(L.lib.mfc_conv2d_fwd, d)
    elif merged_s2:
        d = L.ConvDesc(dy.data_ptr(), 0, dx.data_ptr(), 0, 0, 0, dt_of(dy), N, Ho, Wo, Cop, Cout, H, W, dx.shape[3], Cin, (H + 1) // 2, (W + 1) // 2,
                       2, 2, 0, 0, 1, 2, 2, 0, 0, 0, N, acc, 0, 0)
        d.flags = L.CONV_S2_CLASSES
        wp = pack_weight(w, d, "dgrad_s2_all")
        d.wp = wp.data_ptr()
        L.call(L.lib.mfc_conv2d_fwd, d)
        torch.cuda.synchronize()
    else:
        keep = []
        for ph in range(2):
            for pw in range(2):
                ta, _, dh0 = s2_class(k, pad, ph)
                tb, _, dw0 = s2_class(k, pad, pw)
                Hl, Wl = (H - ph + 1) // 2, (W - pw + 1) // 2
                d = L.ConvDesc(dy.data_ptr(), 0, dx.data_ptr(), 0, 0, 0, dt_of(dy), N, Ho, Wo, Cop, Cout, H, W,
                               dx.shape[3], Cin, Hl, Wl, ta, tb, dh0, dw0, 1, 2, 2, ph, pw, 0, N, acc, 0, 0)
                wp = pack_weight(w, d, "dgrad_s2", (ph, pw))
                keep.append(wp)
                d.wp = wp.data_ptr()
                L.call(L.lib.mfc_conv2d_fwd, d)
        torch.cuda.synchronize()
    return dx


def conv2d_wgrad(x, dy, Cout, Cin, k, stride=1, in_coef=None, in_relu=False, ipg=None, splits=0):
    """Weight gradient in the reference layout [Cout,Cin,k,k] (fp32)."""
    N, H, W, Cp = x.shape
    _, Ho, Wo, Cop = dy.shape
    pad = k // 2
    Co16, Ci16 = rup(Cout, 16), rup(Cin, 16)
    d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), 0, in_coef.data_ptr() if in_coef is not None else 0, dt_of(x),
                    N, H, W, Cp, Cin, Ho, Wo, Cop, Cout, k, k, -pad, -pad, stride, 1 if in_relu else 0, ipg or N, 0, 0, splits)
    parts = L.wgrad_parts(d)
    dwp = torch.zeros(parts * k * k * Co16 * Ci16, dtype=torch.float32, device="cuda")      # partial-sum slices
    d.dwp = dwp.data_ptr()
    L.call(L.lib.mfc_conv2d_wgrad, d)
    dw = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device="cuda")
    _run_jobs([dict(src=dwp.data_ptr(), dst=dw.data_ptr(), Cout=Cout, Cin=Cin, KH=k, KW=k, Co16=Co16, Ci16=Ci16, nparts=parts)],
              L.UnpackJob, L.lib.mfc_unpack_wgrad)
    return dw


def view(t: torch.Tensor, coef=None, c_off=0) -> L.View:
    return L.View(t.data_ptr(), coef.data_ptr() if coef is not None else 0, t.shape[1], t.shape[2], t.shape[3], c_off)
