"""ctypes binding of libmfcnet_hip.so (C ABI declared in include/mfcnet_hip.h).

The library is the product: there is NO fallback.  If it is missing, importing
this module raises, and every op fails loudly rather than silently running
PyTorch operators instead.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  MUST precede the CDLL below: PyTorch bundles its own libamdhip64.so.7 (same soname as
#                     /opt/rocm's); whichever loads first serves the whole process, and torch's allocations must
#                     live in the same HIP runtime instance as our launches.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFC_LIB") or os.path.join(_HERE, "libmfcnet_hip.so")      # (MFC_LIB: another build of the same library, for A/B measurements)

F32, BF16, F16 = 0, 1, 2


def is16(dtype: int) -> bool:
    """16-bit storage (bf16 or fp16): 8 channels per 16-byte granule, MFMA 16x16x32 kernels."""
    return dtype in (BF16, F16)
STAT_REPLICAS = int(os.environ.get("MFC_STAT_REPLICAS", "8"))      # must equal the library's MFC_STAT_REPLICAS (A/B builds: make REPLICAS=n OUT=...)
STAT_BYTES = 8            # sizeof(mfc_stat_t): the statistic cells are fp64
LOSS_ACC_FLOATS = 96      # MFC_LOSS_ACC_FLOATS: 32 result floats + the scratch of mfc_loss_partial

# op kinds (mfc_op_kind)
OP_CONV, OP_WGRAD, OP_BNFIN, OP_COMBINE, OP_BNBWD_REDUCE, OP_BNBWD_FIN, OP_BNBWD_APPLY, OP_MASK_ADD = range(1, 9)
LANE_ASYNC = 0x100
OP_HEAD_FWD, OP_HEAD_BWD, OP_BIAS_GRAD, OP_MEMSET, OP_PACK, OP_UNPACK, OP_NCHW2NHWC, OP_NHWC2NCHW = range(9, 17)
OP_WGRAD_BATCH = 17
OP_BNFIN_BATCH = 18
OP_WSNORM, OP_GNFIN, OP_UPNEAR = 19, 20, 21
OP_GNBWD_FIN, OP_WSBWD, OP_UPNEAR_BWD, OP_JOIN = 22, 23, 24, 25
WGRAD_MAXBATCH = 8
CONV_WANT_FA = 1
CONV_NEVER_ACC = 4
CONV_S2_CLASSES = 2      # mfc_conv_desc.flags: data gradient of a 3x3 / stride-2 convolution, all four output parity classes in one launch
RUN_DEFER_JOIN = 1

i32, u64, f32, vp = C.c_int32, C.c_uint64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [("inp", vp), ("wp", vp), ("out", vp), ("bias", vp), ("in_coef", vp), ("out_stats", vp),
                ("dtype", i32), ("N", i32), ("Hin", i32), ("Win", i32), ("Cin_p", i32), ("Cin", i32),
                ("Hout", i32), ("Wout", i32), ("Cout_p", i32), ("Cout", i32), ("Hl", i32), ("Wl", i32),
                ("TA", i32), ("TB", i32), ("dh0", i32), ("dw0", i32), ("in_stride", i32),
                ("out_sh", i32), ("out_sw", i32), ("out_oh", i32), ("out_ow", i32),
                ("in_relu", i32), ("images_per_group", i32), ("accumulate", i32), ("TH", i32), ("TW", i32),
                ("acc_src", vp), ("bn_y", vp), ("bn_coef", vp), ("bn_bits", vp), ("bn_mask_mode", i32), ("flags", i32)]


class PackJob(C.Structure):
    _fields_ = [("src", u64), ("dst", u64), ("Cout", i32), ("Cin", i32), ("KH", i32), ("KW", i32),
                ("TA", i32), ("TB", i32), ("kh0", i32), ("kh_step", i32), ("kw0", i32), ("kw_step", i32),
                ("mode", i32), ("KG", i32), ("nchunks", i32), ("NT16", i32), ("Yblocks", i32), ("nslots", i32),
                ("TAS", i32), ("block0", i32), ("nblocks", i32), ("pad_", i32)]


class ConvLayout(C.Structure):
    _fields_ = [("KG", i32), ("nchunks", i32), ("NT16", i32), ("Yblocks", i32), ("nslots", i32), ("TA", i32), ("TB", i32),
                ("lds_bytes", i32), ("bytes", C.c_int64), ("MT", i32), ("TH", i32), ("TW", i32), ("grid", i32),
                ("per_block", i32), ("TAS", i32), ("NW", i32), ("fa", i32)]


class WgradDesc(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("dwp", vp), ("in_coef", vp),
                ("dtype", i32), ("N", i32), ("Hin", i32), ("Win", i32), ("Cin_p", i32), ("Cin", i32),
                ("Hout", i32), ("Wout", i32), ("Cout_p", i32), ("Cout", i32),
                ("TA", i32), ("TB", i32), ("dh0", i32), ("dw0", i32), ("in_stride", i32),
                ("in_relu", i32), ("images_per_group", i32), ("TH", i32), ("TW", i32), ("splits", i32), ("batch", i32), ("pad_", i32)]


class UnpackJob(C.Structure):
    _fields_ = [("src", u64), ("dst", u64), ("Cout", i32), ("Cin", i32), ("KH", i32), ("KW", i32),
                ("Co16", i32), ("Ci16", i32), ("block0", i32), ("nblocks", i32), ("nparts", i32), ("pad_", i32)]


class BnFinDesc(C.Structure):
    _fields_ = [("stats", vp), ("coef", vp), ("gamma", vp), ("beta", vp), ("running_mean", vp),
                ("running_var", vp), ("num_batches_tracked", vp),
                ("C", i32), ("Cp", i32), ("G", i32), ("training", i32), ("count", f32), ("eps", f32), ("momentum", f32)]


class GnFinDesc(C.Structure):
    _fields_ = [("stats", vp), ("coef", vp), ("gamma", vp), ("beta", vp), ("C", i32), ("Cp", i32), ("N", i32), ("groups", i32),
                ("count", f32), ("eps", f32)]


class View(C.Structure):
    _fields_ = [("ptr", u64), ("coef", u64), ("H", i32), ("W", i32), ("Cp", i32), ("c_off", i32)]


class CombineDesc(C.Structure):
    _fields_ = [("out", View), ("src", View * 4), ("nsrc", i32), ("relu", i32), ("dtype", i32),
                ("N", i32), ("C", i32), ("images_per_group", i32), ("maskbits", u64)]


class BnBwdDesc(C.Structure):
    _fields_ = [("g", View), ("y", View), ("mask", View), ("dy", View), ("bstats", vp), ("bcoef", vp),
                ("mask_mode", i32), ("dtype", i32), ("N", i32), ("C", i32), ("images_per_group", i32), ("accumulate", i32),
                ("fin_dgamma", vp), ("fin_dbeta", vp), ("fin_C", i32), ("fin_training", i32), ("fin_count", f32), ("gn_mode", i32)]


class GnBwdFinDesc(C.Structure):
    _fields_ = [("bstats", vp), ("bcoef", vp), ("gamma", vp), ("dgamma", vp), ("dbeta", vp),
                ("C", i32), ("Cp", i32), ("N", i32), ("groups", i32), ("count", f32), ("pad_", i32)]


class BnBwdFinDesc(C.Structure):
    _fields_ = [("bstats", vp), ("bcoef", vp), ("dgamma", vp), ("dbeta", vp),
                ("C", i32), ("Cp", i32), ("G", i32), ("training", i32), ("count", f32)]


class MaskAddDesc(C.Structure):
    _fields_ = [("g", View), ("mask", View), ("dst", View),
                ("mask_mode", i32), ("dtype", i32), ("N", i32), ("C", i32), ("accumulate", i32), ("pad_", i32), ("scratch", vp)]


class HeadDesc(C.Structure):
    _fields_ = [("logits", vp), ("flow", vp * 8), ("depth", vp * 8), ("xh", vp),
                ("dtype", i32), ("B", i32), ("T", i32), ("nc", i32), ("Hs", i32), ("Ws", i32), ("Lp", i32),
                ("H", i32), ("W", i32), ("Cp", i32), ("warp", i32)]


class HeadBwd(C.Structure):
    _fields_ = [("d", HeadDesc), ("dlogits", u64)]


class LossDesc(C.Structure):
    _fields_ = [("logits", vp), ("target", vp), ("class_w", vp), ("acc", vp), ("dlogits", vp),
                ("B", i32), ("nc", i32), ("H", i32), ("W", i32), ("w_nll", f32), ("w_jac", f32), ("grad_scale", f32), ("pad_", i32),
                ("grad_scale_dev", vp)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 120), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("launches", C.c_int64)]


class RawOp(C.Structure):
    _fields_ = [("a", u64), ("b", u64), ("c", u64), ("n", C.c_int64), ("i", i32 * 12)]


class OpUnion(C.Union):
    _fields_ = [("conv", ConvDesc), ("wgrad", WgradDesc), ("bnfin", BnFinDesc), ("gnfin", GnFinDesc), ("combine", CombineDesc),
                ("bnbwd", BnBwdDesc), ("bnbwdfin", BnBwdFinDesc), ("maskadd", MaskAddDesc), ("head", HeadDesc),
                ("headbwd", HeadBwd), ("raw", RawOp), ("bytes", C.c_uint8 * 248)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("lane", i32), ("u", OpUnion)]


# every symbol include/mfcnet_hip.h declares
EXPORTS = ["mfc_conv2d_fwd", "mfc_conv2d_layout", "mfc_conv2d_lds_bytes", "mfc_pack_weights", "mfc_conv2d_wgrad", "mfc_conv2d_wgrad_parts", "mfc_conv2d_wgrad_batch", "mfc_unpack_wgrad",
           "mfc_bn_finalize", "mfc_bn_finalize_batch", "mfc_ws_normalize", "mfc_gn_finalize", "mfc_upsample_nearest2x", "mfc_gnbwd_finalize", "mfc_ws_backward", "mfc_upsample_nearest2x_bwd", "mfc_combine_fwd", "mfc_bnbwd_reduce", "mfc_bnbwd_apply", "mfc_bnbwd_finalize",
           "mfc_mask_add", "mfc_bias_grad", "mfc_bias_grad_slices", "mfc_nchw_to_nhwc", "mfc_nhwc_to_nchw", "mfc_head_gather_fwd",
           "mfc_head_gather_bwd", "mfc_loss_fwd", "mfc_loss_partial", "mfc_loss_finalize", "mfc_loss_bwd", "mfc_confusion_counts", "mfc_adam_step", "mfc_adam_step_guarded", "mfc_grad_check", "mfc_program_run", "mfc_program_run_ex", "mfc_wait_detached", "mfc_program_profile", "mfc_graph_capture", "mfc_graph_launch", "mfc_graph_destroy",
           "mfc_ctx_create", "mfc_ctx_destroy", "mfc_program_run_ctx", "mfc_wait_detached_ctx",
           "mfc_op_size", "mfc_version", "mfc_prof_enable", "mfc_prof_collect", "mfc_prof_dump"]
# include/mfcnet_hip_tuning.h (tools / A-B measurements only: not part of the product interface)
TUNING_EXPORTS = ["mfc_set_flag"]


class MfcError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise MfcError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                       f"`make -C mfcnet-tracker_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS + TUNING_EXPORTS:
        if not hasattr(lib, name):
            raise MfcError(f"{LIB_PATH} does not export {name}")
    lib.mfc_version.restype = C.c_char_p
    for name in EXPORTS + TUNING_EXPORTS:
        if name not in ("mfc_version",):
            getattr(lib, name).restype = C.c_int
    lib.mfc_program_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.mfc_program_run_ex.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]
    lib.mfc_program_profile.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mfc_graph_capture.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mfc_wait_detached.argtypes = [C.c_void_p]
    lib.mfc_ctx_create.argtypes = [C.c_int32, C.c_void_p]
    lib.mfc_ctx_destroy.argtypes = [C.c_void_p]
    lib.mfc_program_run_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]
    lib.mfc_wait_detached_ctx.argtypes = [C.c_void_p, C.c_void_p]
    lib.mfc_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    lib.mfc_graph_destroy.argtypes = [C.c_void_p]
    lib.mfc_bias_grad.argtypes = [vp, vp, i32, C.c_int64, i32, i32, vp]
    lib.mfc_bias_grad_slices.argtypes = [vp, vp, i32, C.c_int64, i32, i32, i32, vp]
    lib.mfc_nchw_to_nhwc.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.mfc_nhwc_to_nchw.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.mfc_pack_weights.argtypes = [vp, i32, i32, i32, vp]
    lib.mfc_unpack_wgrad.argtypes = [vp, i32, i32, vp]
    lib.mfc_adam_step.argtypes = [vp, vp, vp, vp, C.c_int64, f32, f32, f32, f32, i32, f32, vp]
    lib.mfc_adam_step_guarded.argtypes = [vp, vp, vp, vp, C.c_int64, f32, f32, f32, f32, i32, f32, vp, vp]
    lib.mfc_grad_check.argtypes = [vp, C.c_int64, vp, vp]
    lib.mfc_head_gather_bwd.argtypes = [vp, vp, vp]
    lib.mfc_confusion_counts.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    for name in ("mfc_conv2d_fwd", "mfc_conv2d_wgrad", "mfc_bn_finalize", "mfc_combine_fwd", "mfc_bnbwd_reduce",
                 "mfc_bnbwd_apply", "mfc_bnbwd_finalize", "mfc_mask_add", "mfc_head_gather_fwd", "mfc_loss_fwd",
                 "mfc_loss_partial", "mfc_loss_finalize", "mfc_loss_bwd"):
        getattr(lib, name).argtypes = [vp, vp]
    lib.mfc_bn_finalize_batch.argtypes = [vp, i32, i32, vp]
    lib.mfc_ws_normalize.argtypes = [vp, vp, i32, i32, f32, vp]
    lib.mfc_gn_finalize.argtypes = [vp, vp]
    lib.mfc_upsample_nearest2x.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.mfc_conv2d_lds_bytes.argtypes = [vp]
    lib.mfc_conv2d_wgrad_parts.argtypes = [vp]
    lib.mfc_conv2d_wgrad_batch.argtypes = [vp, i32, vp]
    lib.mfc_conv2d_layout.argtypes = [vp, vp]
    lib.mfc_set_flag.argtypes = [i32, i32]
    lib.mfc_prof_enable.argtypes = [i32]
    lib.mfc_prof_collect.argtypes = [vp, i32]
    lib.mfc_prof_dump.argtypes = [C.c_char_p]
    if lib.mfc_op_size() != C.sizeof(Op):
        raise MfcError(f"mfc_op size mismatch: library {lib.mfc_op_size()} vs python mirror {C.sizeof(Op)}")
    return lib


lib = _load()


class Ctx:
    """An interpreter context (mfc_ctx_create): the side streams / events a model's programs run on.  One per model; destroyed with it."""

    def __init__(self, device_index: int):
        self.handle = C.c_void_p()
        check_rc = lib.mfc_ctx_create(int(device_index), C.byref(self.handle))
        if check_rc != 0:
            raise MfcError(f"mfc_ctx_create(device {device_index}) failed with status {check_rc}")
        self.device_index = int(device_index)

    def __deepcopy__(self, memo):          # a copied model gets a context of its own (handles are never shared)
        return Ctx(self.device_index)

    def __reduce__(self):
        return (Ctx, (self.device_index,))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                lib.mfc_ctx_destroy(h)
            except Exception:
                pass


def check(rc: int, what: str = "mfc call"):
    if rc != 0:
        raise MfcError(f"{what} failed with status {rc}")


def wgrad_parts(desc) -> int:
    """Number of partial-sum slices mfc_conv2d_wgrad writes for `desc` (sizes its dwp buffer)."""
    n = lib.mfc_conv2d_wgrad_parts(C.byref(desc))
    if n <= 0:
        raise MfcError(f"mfc_conv2d_wgrad_parts failed with status {n}")
    return n


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(fn, desc, what=None):
    check(fn(C.byref(desc), stream_ptr()), what or fn.__name__)


def conv_layout(desc: ConvDesc) -> ConvLayout:
    """Blocking of the packed weight image the launch described by `desc` will read."""
    lay = ConvLayout()
    check(lib.mfc_conv2d_layout(C.byref(desc), C.byref(lay)), "mfc_conv2d_layout")
    return lay


def pack_job_fields(lay: ConvLayout) -> dict:
    return dict(KG=lay.KG, nchunks=lay.nchunks, NT16=lay.NT16, Yblocks=lay.Yblocks, nslots=lay.nslots, TAS=lay.TAS)


def prof_collect(cap: int = 256):
    """Rows of the event profiler as dicts {name, ms, flops, bytes, launches} (see mfc_prof_collect), largest total time first."""
    arr = (ProfEntry * cap)()
    n = lib.mfc_prof_collect(arr, cap)
    if n < 0:
        raise MfcError(f"mfc_prof_collect failed with status {n}")
    rows = [dict(name=arr[i].name.decode(), ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes, launches=arr[i].launches) for i in range(n)]
    return sorted(rows, key=lambda r: -r["ms"])
