"""Shared helpers: load a golden fixture, rebuild its case, compare a candidate against it."""
import os

import numpy as np

from oracle import mfcnet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_CASES = ["large_rgb_train", "large_rgb_eval", "large_all_train", "large_depth_eval", "large_flow_headonly",
               "basic_rgb_train", "basic_all_train", "basic_flow_eval", "large_t5_train", "large_odd_train"]
BIG_CASES = ["large_480_train"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    mtype, T, flow, depth, B, H, W, mode = [str(v) for v in z["meta"]]
    cfg = dict(name=name, model_type=mtype, T=int(T), optflow=bool(int(flow)), depth=bool(int(depth)), B=int(B),
               H=int(H), W=int(W), mode=mode)
    return cfg, z


def case_inputs(cfg):
    return O.synthetic_clip(cfg["name"], cfg["B"], cfg["T"], cfg["H"], cfg["W"], cfg["optflow"], cfg["depth"])


def case_state(cfg, width=48):
    return O.hashed_state(O.mfcnet_table(cfg["model_type"], width, 5, cfg["T"], cfg["optflow"], cfg["depth"]))


def compare_logits(z, y, atol):
    """y: numpy [B,5,H,W]; returns max abs error vs the stored reference output."""
    if "logits" in z.files:
        err = float(np.abs(y - z["logits"]).max())
    else:
        err = float(np.abs(y[:, :, ::8, ::8] - z["logits_s8"]).max())
        cs = y.astype(np.float64).sum(axis=(0, 2, 3))
        scale = np.abs(z["logits_chsum"]).max() + 1.0
        assert np.abs(cs - z["logits_chsum"]).max() / scale < atol, "channel checksum mismatch"
    assert err <= atol, f"max abs logits error {err} > {atol}"
    return err


def sample16(t):
    import torch
    flat = t.detach().flatten().float().cpu()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    return flat[idx].numpy()
