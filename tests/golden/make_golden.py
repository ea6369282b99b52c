"""Generate tests/golden/*.npz by importing the REFERENCE model in the build container.

Runs only where /root/reference exists (never on the GPU box).  The reference's
`models/hrnet.py`, `models/multiframe_model.py` and `src/loss.py` are imported
in place (SURVEY.md Appendix A recipe: empty stub modules stand in for the
absent torchvision / segmentation_models_pytorch imports, none of which the
HRNetMulti* classes use).  Weights come from the key-hash generator in
oracle/mfcnet_oracle.py, inputs from the same hash generator, so a fixture
stores only the case description and the reference's OUTPUTS.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mfcnet_oracle as O  # noqa: E402

REF = "/root/reference"


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tvm = stub("torchvision.models")
    tv.models = tvm
    tvm.segmentation = stub("torchvision.models.segmentation")
    stub("torchvision.models.segmentation.deeplabv3", DeepLabHead=object)
    stub("torchvision.models.segmentation.fcn", FCNHead=object)
    stub("segmentation_models_pytorch", Segformer=object)

    def load(name, path):
        s = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(s)
        sys.modules[name] = m
        s.loader.exec_module(m)
        return m

    load("hrnet", f"{REF}/models/hrnet.py")
    pkg = stub("models")
    pkg.__path__ = []
    stub("models.ternausnet", TernausNet11=object, TernausNet16=object)
    mf = load("models.multiframe_model", f"{REF}/models/multiframe_model.py")
    loss = load("ref_loss", f"{REF}/src/loss.py")
    return mf, loss


SENTINELS = [
    "base_model.conv1.weight", "base_model.bn1.weight", "base_model.layer1.0.conv2.weight",
    "base_model.layer1.0.downsample.0.weight", "base_model.transition1.1.0.0.weight",
    "base_model.stage2.0.branches.0.0.conv1.weight", "base_model.stage2.0.fuse_layers.0.1.0.weight",
    "base_model.stage2.0.fuse_layers.1.0.0.0.weight", "base_model.stage3.1.branches.2.3.conv2.weight",
    "base_model.stage3.3.fuse_layers.2.0.0.0.weight", "base_model.stage3.3.fuse_layers.2.0.1.1.bias",
    "base_model.stage4.2.branches.3.1.conv1.weight", "base_model.stage4.2.branches.3.1.bn1.weight",
    "base_model.stage4.0.fuse_layers.0.3.0.weight", "base_model.last_layer.0.weight",
    "base_model.last_layer.0.bias", "base_model.last_layer.1.bias", "base_model.last_layer.3.weight",
    "base_model.last_layer.3.bias",
    "multiframe_net.multiframe_net.0.weight", "multiframe_net.multiframe_net.1.weight",
    "multiframe_net.multiframe_net.3.weight", "multiframe_net.multiframe_net.6.weight",
    "multiframe_net.multiframe_net.7.bias", "multiframe_net.multiframe_net.9.weight",
]
BN_SENTINELS = ["base_model.bn1", "base_model.stage3.0.branches.1.2.bn2", "base_model.last_layer.1",
                "multiframe_net.multiframe_net.1", "multiframe_net.multiframe_net.7"]

# name, model_type, T, flow, depth, B, H, W, mode
CASES = [
    ("large_rgb_train", "HRNetMulti-Large", 3, False, False, 2, 64, 96, "train"),
    ("large_rgb_eval", "HRNetMulti-Large", 3, False, False, 2, 64, 96, "eval"),
    ("large_all_train", "HRNetMulti-Large", 3, True, True, 2, 64, 96, "train"),
    ("large_depth_eval", "HRNetMulti-Large", 3, False, True, 1, 64, 96, "eval"),
    ("large_flow_headonly", "HRNetMulti-Large", 3, True, False, 2, 64, 96, "headonly"),
    ("basic_rgb_train", "HRNetMulti-Basic", 3, False, False, 2, 64, 96, "train"),
    ("basic_all_train", "HRNetMulti-Basic", 3, True, True, 2, 64, 96, "train"),
    ("basic_flow_eval", "HRNetMulti-Basic", 3, True, False, 1, 96, 128, "eval"),
    ("large_t5_train", "HRNetMulti-Large", 5, True, True, 1, 96, 128, "train"),
    ("large_odd_train", "HRNetMulti-Large", 3, False, False, 1, 92, 120, "train"),   # 23x30 -> odd pyramid sizes
    ("large_480_train", "HRNetMulti-Large", 3, False, False, 1, 480, 640, "train"),
]


def run_case(mf, ref_loss, case):
    name, mtype, T, flow, depth, B, H, W, mode = case
    cls = mf.HRNetMultiLarge if "Large" in mtype else mf.HRNetMultiBasic
    net = cls(num_classes=5, num_frames=T, pretrained=False, loadpath=None, optflow_inputs=flow, depth_inputs=depth)
    table = O.mfcnet_table(mtype, 48, 5, T, flow, depth)
    sd = O.hashed_state(table)
    ref_keys = list(net.state_dict().keys())
    assert ref_keys == [t[0] for t in table], "state_dict key order/name mismatch"
    net.load_state_dict(sd, strict=True)
    frames, flows, depths, mask = O.synthetic_clip(name, B, T, H, W, flow, depth)
    out = {"meta": np.array([mtype, str(T), str(int(flow)), str(int(depth)), str(B), str(H), str(W), mode])}
    if mode == "eval":
        net.eval()
        with torch.no_grad():
            y = net(frames, optflow=flows, depth=depths)
    else:
        if mode == "train":
            net.train()
        else:                              # engine.py:25-26
            net.base_model.eval()
            net.multiframe_net.train()
        lr = 1e-4
        opt = torch.optim.Adam([{"params": net.base_model.parameters(), "lr": lr / T},
                                {"params": net.multiframe_net.parameters(), "lr": lr}])
        opt.zero_grad()
        y = net(frames, optflow=flows, depth=depths)
        logp = torch.nn.functional.log_softmax(y, dim=1)
        args = SimpleNamespace(class_weights=np.array(O.DEFAULT_CLASS_WEIGHTS), num_classes=5)
        loss, ld = ref_loss.get_loss(logp, mask, ["nll", "soft_jaccard"], [0.7, 0.3], args)
        loss.backward()
        named = dict(net.named_parameters())
        for s in SENTINELS:
            g = named[s].grad
            out["gradnorm/" + s] = np.float64(g.double().norm().item())
            flat = g.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 16).long()
            out["gradsample/" + s] = flat[idx].numpy().copy()
        opt.step()
        for s in SENTINELS:
            flat = named[s].detach().flatten()
            idx = torch.linspace(0, flat.numel() - 1, 16).long()
            out["paramsample/" + s] = flat[idx].numpy().copy()
        st = net.state_dict()
        for b in BN_SENTINELS:
            out["bn_mean/" + b] = st[b + ".running_mean"].numpy().copy()
            out["bn_var/" + b] = st[b + ".running_var"].numpy().copy()
            out["bn_count/" + b] = np.int64(st[b + ".num_batches_tracked"].item())
        for k, v in ld.items():
            out[k] = np.float64(v)
    y = y.detach()
    if H * W > 20000:      # big case: strided samples + per-channel moments
        out["logits_s8"] = y[:, :, ::8, ::8].numpy().copy()
        out["logits_chsum"] = y.double().sum(dim=(0, 2, 3)).numpy()
        out["logits_chabsmax"] = y.abs().amax(dim=(0, 2, 3)).numpy()
    else:
        out["logits"] = y.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", tuple(y.shape), {k: float(v) for k, v in out.items() if k.startswith("loss_")})


SINGLE_CASES = [            # name, B, H, W, mode   (single-frame 'HRNet' model type, models/__init__.py:38-46)
    ("hrnet_single_train", 2, 64, 96, "train"),
    ("hrnet_single_eval", 2, 64, 96, "eval"),
]
SINGLE_SENTINELS = ["conv1.weight", "layer1.0.conv2.weight", "stage3.1.branches.2.3.conv2.weight", "stage4.0.fuse_layers.0.3.0.weight",
                    "last_layer.0.weight", "last_layer.1.bias", "last_layer.3.weight", "last_layer.3.bias"]


def run_single_case(ref_loss, case):
    """The reference's single-frame model: HighResolutionNet() with last_layer swapped as get_tooltip_segmentation_model does
    for model_type 'HRNet' (models/__init__.py:38-46; the pretrained Cityscapes file it loads first does not exist here, the
    hashed weights stand in), stepped as scripts/train_toolpose_segmentation.py:162-163: log_softmax(model(x)) + get_loss."""
    import torch.nn as nn
    name, B, H, W, mode = case
    hr = sys.modules["hrnet"]
    net = hr.HighResolutionNet()
    cin = net.last_layer[0].in_channels
    net.last_layer = nn.Sequential(nn.Conv2d(cin, cin, kernel_size=1, stride=1, padding=0), hr.BatchNorm2d(cin, momentum=0.1),
                                   nn.ReLU(inplace=True), nn.Conv2d(cin, 5, kernel_size=1, stride=1, padding=0))
    table = O.hrnet_table(48, 5, "")
    assert list(net.state_dict().keys()) == [t[0] for t in table], "state_dict key order/name mismatch"
    net.load_state_dict(O.hashed_state(table), strict=True)
    frames, _, _, mask = O.synthetic_clip(name, B, 1, H, W, False, False)
    x = frames[0]
    out = {"meta": np.array(["HRNet", "1", "0", "0", str(B), str(H), str(W), mode])}
    if mode == "eval":
        net.eval()
        with torch.no_grad():
            y = net(x)
    else:
        net.train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        opt.zero_grad()
        y = net(x)
        logp = torch.nn.functional.log_softmax(y, dim=1)
        args = SimpleNamespace(class_weights=np.array(O.DEFAULT_CLASS_WEIGHTS), num_classes=5)
        loss, ld = ref_loss.get_loss(logp, mask, ["nll", "soft_jaccard"], [0.7, 0.3], args)
        loss.backward()
        named = dict(net.named_parameters())
        for s in SINGLE_SENTINELS:
            gr = named[s].grad
            out["gradnorm/" + s] = np.float64(gr.double().norm().item())
            flat = gr.flatten()
            out["gradsample/" + s] = flat[torch.linspace(0, flat.numel() - 1, 16).long()].numpy().copy()
        opt.step()
        for s in SINGLE_SENTINELS:
            flat = named[s].detach().flatten()
            out["paramsample/" + s] = flat[torch.linspace(0, flat.numel() - 1, 16).long()].numpy().copy()
        st = net.state_dict()
        for b in ("bn1", "last_layer.1"):
            out["bn_mean/" + b] = st[b + ".running_mean"].numpy().copy()
            out["bn_var/" + b] = st[b + ".running_var"].numpy().copy()
            out["bn_count/" + b] = np.int64(st[b + ".num_batches_tracked"].item())
        for k, v in ld.items():
            out[k] = np.float64(v)
    out["logits"] = y.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", tuple(y.shape))


METRIC_CASES = [            # (name, B, num_classes, H, W, seed, logit scale)
    ("metrics_b3", 3, 5, 40, 56, 101, 2.0),
    ("metrics_b1", 1, 5, 33, 47, 102, 1.0),
    ("metrics_ties", 2, 5, 24, 32, 103, 0.0),      # scale 0 -> quantised logits with many ties: argmax takes the first maximum
    ("metrics_empty", 2, 5, 16, 16, 104, 2.0),     # classes 3 and 4 absent from the masks: the epsilon terms decide
]


def metric_inputs(B, nc, H, W, seed, scale):
    """Deterministic (logits, mask) of a metrics case; shared with the tests through oracle.metric_case_inputs."""
    return O.metric_case_inputs(B, nc, H, W, seed, scale)


def run_metric_case(case):
    """Reference src/metrics.py::get_metrics (imported in place) on log-softmax outputs, as src/engine.py:137-139 calls it."""
    name, B, nc, H, W, seed, scale = case
    spec = importlib.util.spec_from_file_location("ref_metrics", f"{REF}/src/metrics.py")
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    logits, mask = metric_inputs(B, nc, H, W, seed, scale)
    out = torch.nn.functional.log_softmax(logits, dim=1)
    vals, md = rm.get_metrics(out, mask, ["iou", "dice"], SimpleNamespace(num_classes=nc))
    conf = rm.calculate_confusion_matrix_from_arrays(out.numpy().argmax(axis=1), mask.numpy(), nc)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), case=np.array([B, nc, H, W, seed], dtype=np.int64), scale=np.float64(scale),
                        iou=np.array(vals[0], dtype=np.float64), dice=np.array(vals[1], dtype=np.float64),
                        metric_iou=np.float64(md["metric_iou"]), metric_dice=np.float64(md["metric_dice"]),
                        confusion=conf.astype(np.int64))
    print(name, "ok", md)


if __name__ == "__main__":
    torch.manual_seed(0)
    only = sys.argv[1:]
    for c in METRIC_CASES:
        if only and c[0] not in only:
            continue
        run_metric_case(c)
    if only and all(o.startswith("metrics_") for o in only):
        sys.exit(0)
    mf, ref_loss = import_reference()
    for c in SINGLE_CASES:
        if only and c[0] not in only:
            continue
        run_single_case(ref_loss, c)
    for c in CASES:
        if only and c[0] not in only:
            continue
        run_case(mf, ref_loss, c)
