"""Host-side mirror of the reference model interface for the MFCNet hot path.

Same names, constructor arguments, `forward(x, optflow=None, depth=None)` signature, sub-module
attributes (`.base_model`, `.multiframe_net`), `state_dict` keys and error behaviour as
  models/multiframe_model.py:408-471   HRNetMultiBasic / HRNetMultiLarge
  models/__init__.py:54-87             get_multiframe_segmentation_model(args)
so scripts/train_multiframe_detection.py and src/engine.py call it unchanged.  All arithmetic runs in
libmfcnet_hip.so (hand-written gfx950 kernels) through a static plan (plan.py); PyTorch provides device
memory, streams and the autograd hook only.  There is no PyTorch/CPU fallback: without a GPU or without
the built library, forward() raises.

MI355X-first layout: all fp32 master parameters live in ONE flat arena `_P` (base_model segment first,
then multiframe_net), gradients in a matching arena `_G`; every nn.Parameter / .grad is a view.  One
RCCL all-reduce over `_G` and one fused Adam launch per lr group replace 931 per-tensor operations.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .arch import Entry, head_entries, hrnet_entries

_DT = {"fp32": L.F32, "float32": L.F32, "f32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16,
       "fp16": L.F16, "float16": L.F16, "f16": L.F16, "half": L.F16}


class _Node(nn.Module):
    """Plain container; the tree of these reproduces the reference's dotted parameter names."""

    def forward(self, *a, **k):
        raise NotImplementedError("sub-modules of the HIP model are parameter containers; call the top-level model")

    def __getitem__(self, idx):          # nn.Sequential / nn.ModuleList style access: model.base_model.last_layer[3]
        return self._modules[str(idx)]


def _get_node(root: nn.Module, path: List[str]) -> nn.Module:
    m = root
    for p in path:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    return m


class _MFCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, plan, frames, flow, depth):
        ctx.model, ctx.plan = model, plan
        return plan.run_forward(frames, flow, depth)

    @staticmethod
    def backward(ctx, gout):
        ctx.model._run_backward(ctx.plan, gout)
        return None, None, None, None, None, None


class _MFCHeadFn(torch.autograd.Function):
    """temporal head on externally computed per-frame logits (base_model = "resunet_vb"): the gradient w.r.t. the logits flows on into the
    per-frame network's own autograd node"""
    @staticmethod
    def forward(ctx, logits, anchor, model, plan, flow, depth):
        ctx.model, ctx.plan = model, plan
        return plan.run_forward(None, flow, depth, ext_logits=logits)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.model._run_backward(ctx.plan, gout)
        return g, None, None, None, None, None


class HRNetMultiHIP(nn.Module):
    """HRNetMultiLarge / HRNetMultiBasic on MI355X (constructor mirrors multiframe_model.py:409,442).
    base_model = "hrnet" (the reference hard-codes HighResolutionNet, multiframe_model.py:414,447) or "resunet_vb": models/resunet.py's
    ResUnet_VB(channels=3, dim=resunet_dim, out_dim=num_classes) as the per-frame network -- the class the reference ships as the
    alternative GroupNorm / SiLU base; its parameters are ordinary nn.Parameters under `base_model.` (train them with torch.optim; FlatAdam
    covers the temporal head's arena), its full-resolution logits feed the same head gather / 4-conv head."""

    basic = False
    single = False          # True: single-frame HighResolutionNet (no temporal head), see HighResolutionNetHIP

    def __init__(self, num_classes=2, num_frames=1, pretrained=True, loadpath=None, optflow_inputs=False,
                 depth_inputs=False, width=48, compute_dtype="fp32", fuse_bn=True, base_model="hrnet", resunet_dim=16):
        super().__init__()
        if base_model not in ("hrnet", "resunet_vb") or (base_model != "hrnet" and self.single):
            raise ValueError(f"base_model {base_model!r} not recognized")
        self.base_kind = base_model
        self.num_classes, self.num_frames = num_classes, num_frames
        self.pretrained = pretrained                      # ignored by the reference too (multiframe_model.py:413)
        self.optflow_inputs, self.depth_inputs = optflow_inputs, depth_inputs
        self.width = width
        self.compute_dtype = _DT[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype
        self.fuse_bn = fuse_bn
        self.parallel_branches = True      # run the independent branches of each HRNet module on parallel HIP streams
        if self.single:
            self._entries: List[Entry] = hrnet_entries(width, num_classes, p="")
        elif base_model == "resunet_vb":
            self._entries = head_entries(self.basic, num_classes, num_frames, optflow_inputs, depth_inputs)
        else:
            self._entries = hrnet_entries(width, num_classes) + head_entries(
                self.basic, num_classes, num_frames, optflow_inputs, depth_inputs)
        self._poff: Dict[str, int] = {}
        self._boff: Dict[str, int] = {}
        self._noff: Dict[str, int] = {}
        np_, nb, nn_ = 0, 0, 0
        for e in self._entries:
            if e.is_param:
                self._poff[e.name] = np_
                np_ += (e.numel + 3) // 4 * 4            # 16-byte aligned segments
            elif e.kind in ("bn_rm", "bn_rv"):
                self._boff[e.name] = nb
                nb += (e.numel + 3) // 4 * 4
            elif e.kind == "bn_nbt":
                self._noff[e.name] = nn_
                nn_ += 1
        self._n_base = np_ if self.single else min(off for n, off in self._poff.items() if n.startswith("multiframe_net."))
        self._np, self._nb, self._nn = np_, nb, nn_
        if base_model == "resunet_vb":
            from .resunet import ResUnet_VB
            self.add_module("base_model", ResUnet_VB(channels=3, dim=resunet_dim, out_dim=num_classes, compute_dtype=self.compute_dtype))
        else:
            self.add_module("base_model", _Node())
        self.add_module("multiframe_net", _Node())
        self._alloc_flat(torch.device("cpu"))
        self._default_init()
        self._plans = {}

    # ------------------------------------------------------------------ flat arenas
    def _alloc_flat(self, device, src: Optional[Dict[str, torch.Tensor]] = None):
        P = torch.zeros(self._np, dtype=torch.float32, device=device)
        G = torch.zeros(self._np, dtype=torch.float32, device=device)
        RS = torch.zeros(max(self._nb, 1), dtype=torch.float32, device=device)
        NBT = torch.zeros(max(self._nn, 1), dtype=torch.int64, device=device)
        object.__setattr__(self, "_P", P)
        object.__setattr__(self, "_G", G)
        object.__setattr__(self, "_RS", RS)
        object.__setattr__(self, "_NBT", NBT)
        object.__setattr__(self, "_anchor", torch.zeros((), dtype=torch.float32, device=device, requires_grad=True))
        for e in self._entries:
            path = e.name.split(".")
            node, leaf = _get_node(self, path[:-1]), path[-1]
            old = src.get(e.name) if src is not None else None
            if e.is_param:
                v = P[self._poff[e.name]:self._poff[e.name] + e.numel].view(e.shape)
                if old is not None:
                    v.copy_(old.to(torch.float32))
                if leaf in node._parameters and node._parameters[leaf] is not None:
                    prm = node._parameters[leaf]
                    prm.data = v
                    prm.grad = None
                else:
                    node.register_parameter(leaf, nn.Parameter(v))
            else:
                if e.kind == "bn_nbt":
                    v = NBT[self._noff[e.name]:self._noff[e.name] + 1].view(())
                elif e.kind == "grid":
                    v = _warp_grid().to(device)
                else:
                    v = RS[self._boff[e.name]:self._boff[e.name] + e.numel].view(e.shape)
                if old is not None and e.kind != "grid":
                    v.copy_(old.to(v.dtype))
                if leaf in node._buffers:
                    node._buffers[leaf] = v
                else:
                    node.register_buffer(leaf, v)
        self._plans = {}

    def _default_init(self):
        """PyTorch default initialisers (the reference keeps them: hrnet.py never calls init_weights)."""
        with torch.no_grad():
            sd = dict(self.named_parameters())
            for e in self._entries:
                if e.kind == "conv_w":
                    fan_in = e.shape[1] * e.shape[2] * e.shape[3]
                    b = 1.0 / math.sqrt(fan_in)
                    sd[e.name].uniform_(-b, b)
                    bname = e.name[:-6] + "bias"
                    if bname in sd:
                        sd[bname].uniform_(-b, b)
                elif e.kind == "bn_w":
                    sd[e.name].fill_(1.0)
            bufs = dict(self.named_buffers())
            for e in self._entries:
                if e.kind == "bn_rv":
                    bufs[e.name].fill_(1.0)

    def _replicate_for_data_parallel(self):
        """nn.DataParallel on a multi-GPU node (scripts/train_multiframe_detection.py:107-110 wraps unconditionally) replicates the module
        onto every visible device from one process; this model owns per-device arenas, plans and streams, so that cannot work -- say so
        instead of failing somewhere inside replicate().  With ONE visible device nn.DataParallel never replicates and works unchanged."""
        raise L.MfcError("the MI355X MFCNet cannot be replicated by nn.DataParallel: run one process per GPU (torchrun / bench.py --gpus N) "
                         "and wrap the model in mfcnet_amd.DataParallel(model), which exposes .module like nn.DataParallel and "
                         "all-reduces the gradients over RCCL (INTEGRATION.md)")

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        cur = {n: p.data for n, p in self.named_parameters()}
        cur.update({n: b for n, b in self.named_buffers()})
        dev = next(iter(cur.values())).device
        self._alloc_flat(dev, cur)
        return self

    # ------------------------------------------------------------------ execution
    def _get_plan(self, B, H, W, has_flow, has_depth, need_bwd, device):
        from .plan import Plan
        base_tr = self.training if self.single else self.base_model.training
        # the reference's default mode freezes the per-frame network (`param.requires_grad = False` for base_model,
        # scripts/train_multiframe_detection.py:159-165): then its backward is not part of the program at all
        frozen = need_bwd and not self.single and not any(p.requires_grad for p in self.base_model.parameters())
        key = (B, H, W, has_flow, has_depth, base_tr, self.multiframe_net.training, need_bwd,
               self.compute_dtype, self.fuse_bn, frozen)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 2:
                self._plans.clear()
                torch.cuda.empty_cache()
            plan = Plan(self, B, H, W, has_flow, has_depth, base_tr, self.multiframe_net.training,
                        need_bwd, device, base_frozen=frozen)
            self._plans[key] = plan
        return plan

    def forward(self, x, optflow=None, depth=None):
        frames = list(x)
        if len(frames) != self.num_frames:
            raise ValueError(f"expected {self.num_frames} frames, got {len(frames)}")
        if not frames[0].is_cuda or not self._P.is_cuda:
            raise L.MfcError("the MI355X model runs on a GPU only (no CPU fallback): move model and inputs to cuda")
        B, c, H, W = frames[0].shape
        if c != 3 or H % 4 or W % 4 or H < 32 or W < 32:
            raise ValueError(f"frames must be [B,3,H,W] with H,W multiples of 4 and >= 32, got {tuple(frames[0].shape)}")
        ext = self.base_kind == "resunet_vb"
        if (optflow is not None) != bool(self.optflow_inputs) or (depth is not None) != bool(self.depth_inputs):
            raise ValueError("optflow/depth inputs do not match the model's optflow_inputs/depth_inputs")
        if optflow is not None and len(optflow) != self.num_frames - 1:
            raise ValueError("optflow must hold num_frames-1 tensors")
        if depth is not None and len(depth) != self.num_frames:
            raise ValueError("depth must hold num_frames tensors")
        need_bwd = torch.is_grad_enabled()
        plan = self._get_plan(B, H, W, optflow is not None, depth is not None, need_bwd, frames[0].device)
        if ext:
            # GroupNorm statistics are per image, so the T frames go through the per-frame network as one batch of T*B images
            # (multiframe_model.py:459-461 calls it frame by frame: the same arithmetic)
            logits = self.base_model(torch.cat(frames, 0))
            if need_bwd:
                return _MFCHeadFn.apply(logits, self._anchor, self, plan, optflow, depth)
            return plan.run_forward(None, optflow, depth, ext_logits=logits)
        if need_bwd:
            return _MFCFn.apply(self._anchor, self, plan, frames, optflow, depth)
        return plan.run_forward(frames, optflow, depth)

    def _run_backward(self, plan, gout):
        """Runs the backward program and exposes the flat gradient arena as the `.grad` of every parameter that requires one.
        autograd semantics: a parameter whose `.grad` is still set (no zero_grad() since the last backward) ACCUMULATES; one
        whose `.grad` is None gets a fresh gradient.  Parameters with requires_grad=False never get a `.grad` (and when the whole
        per-frame network is frozen its backward is not even computed, see _get_plan)."""
        named = [(n, p) for n, p in self.named_parameters() if n in self._poff]      # (an external per-frame network keeps its own gradients)
        live = [(n, p) for n, p in named if p.requires_grad and p.grad is not None]
        if live and getattr(self, "grad_bucket_hook", None) is not None:
            raise L.MfcError("gradient accumulation across backward passes cannot be combined with a gradient-bucket hook "
                             "(the buckets are reduced while the pass runs): call zero_grad() before every backward")
        old = None
        if live:
            old = self._G.clone()
            if len(live) != sum(1 for _, p in named if p.requires_grad):
                # mixed state: only the parameters that still hold a gradient accumulate
                keep = torch.zeros_like(old)
                for n, p in live:
                    off = self._poff[n]
                    keep[off:off + p.numel()] = old[off:off + p.numel()]
                old = keep
        gin = plan.run_backward(gout)
        if old is not None:
            self._G.add_(old)
        for n, p in named:
            if p.requires_grad and p.grad is None and n in self._poff:
                off = self._poff[n]
                p.grad = self._G[off:off + p.numel()].view(p.shape)
        return gin

    # segments of the flat arenas = the reference's two optimizer groups
    def flat_segments(self):
        if self.single:
            return {"base_model": (0, self._np)}
        return {"base_model": (0, self._n_base), "multiframe_net": (self._n_base, self._np)}


class HRNetMultiLarge(HRNetMultiHIP):
    basic = False


class HighResolutionNetHIP(HRNetMultiHIP):
    """Single-frame `HighResolutionNet` (models/hrnet.py:271-476) with the `last_layer` of the reference's 'HRNet' model type
    (models/__init__.py:38-46: 1x1 conv + BN + ReLU + 1x1 conv to `num_classes`): `forward(x)` takes ONE `[B,3,H,W]` tensor and
    returns `[B,num_classes,H,W]` logits (the x4 up-sampled map, hrnet.py:473-474).  Its `state_dict()` has the HRNet keys
    without prefix, i.e. exactly what `model.base_model.load_state_dict(ckpt['model'])` of the multi-frame models consumes
    (scripts/train_multiframe_detection.py:115-118).  Same kernels, same plan, T = 1, no temporal head."""
    single = True

    def __init__(self, num_classes=5, width=48, compute_dtype="fp32", fuse_bn=True):
        super().__init__(num_classes=num_classes, num_frames=1, pretrained=False, loadpath=None, optflow_inputs=False,
                         depth_inputs=False, width=width, compute_dtype=compute_dtype, fuse_bn=fuse_bn)

    def forward(self, x):
        if isinstance(x, (list, tuple)):
            raise ValueError("the single-frame HRNet takes one [B,3,H,W] tensor")
        return super().forward([x])


class HRNetMultiBasic(HRNetMultiHIP):
    basic = True


def _warp_grid():
    """multiframe_model.py:172-185."""
    H, W = 576, 720
    y, x = torch.meshgrid(torch.arange(0, H), torch.arange(0, W), indexing="ij")
    return torch.stack((2.0 * x / (W - 1) - 1.0, 2.0 * y / (H - 1) - 1.0), dim=0).float().unsqueeze(0)


def get_tooltip_segmentation_model(args, **kw):
    """models/__init__.py:24-52 for the 'HRNet' model type (the single-frame base the multi-frame models are seeded from).
    The reference also loads `models/hrnet_cs_8090_torch11.pth` before swapping `last_layer`; no pretrained file exists here,
    so weights start from the default initialisers (load a checkpoint with `load_model_weights`)."""
    for k in ("width", "compute_dtype", "fuse_bn"):
        if k not in kw and hasattr(args, k):
            kw[k] = getattr(args, k)
    if args.model_type == "HRNet":
        return HighResolutionNetHIP(num_classes=args.num_classes, **kw)
    raise ValueError(f"Model type {args.model_type} not recognized")


def get_multiframe_segmentation_model(args, **kw):
    """models/__init__.py:54-87 for the HRNetMulti-* model types (the only ones on the hot path).
    Extra keyword arguments (width, compute_dtype, fuse_bn) may also be given as attributes of `args`."""
    for k in ("width", "compute_dtype", "fuse_bn"):
        if k not in kw and hasattr(args, k):
            kw[k] = getattr(args, k)
    common = dict(num_classes=args.num_classes, num_frames=args.num_input_frames, pretrained=args.pretrained,
                  loadpath=args.load_wts_base_model, optflow_inputs=args.add_optflow_inputs,
                  depth_inputs=args.add_depth_inputs)
    if args.model_type == "HRNetMulti-Basic":
        return HRNetMultiBasic(**common, **kw)
    if args.model_type == "HRNetMulti-Large":
        return HRNetMultiLarge(**common, **kw)
    raise ValueError(f"Model type {args.model_type} not recognized")
