"""Synthetic weights and inputs of the measurement contract (SURVEY.md 8(c) "weights-by-key-hash", 8(d)).

Real checkpoints and datasets cannot travel to the GPU box, and 65.9 M parameters are too big to commit: every state_dict entry is instead a
pure function of its KEY -- element i of entry `name` is splitmix64((crc32(name) << 32) + i) mapped to a range that depends on what the entry
is (convolution weights U(+-1/sqrt(fan_in)), biases U(+-0.1), BatchNorm gamma U(0.5, 1.5), beta / running mean U(+-0.2), running variance
U(0.5, 1.5): a fresh-initialised eval-mode network is degenerate, its running statistics being 0 / 1).  The CPU oracle's fixtures
(tests/golden/*.npz) were produced from the same function, so a model filled by `fill_hashed` IS the model of those fixtures; bench.py uses
it for the weights of every measured configuration.  tests/test_abi_and_host.py checks this generator against the oracle's, entry by entry."""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(key: str, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    seed = np.uint64(zlib.crc32(key.encode("utf-8"))) << np.uint64(32)
    with np.errstate(over="ignore"):
        bits = _splitmix64(seed + np.arange(n, dtype=np.uint64))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(np.float32)


def _entry(name: str, t: torch.Tensor, salt: str):
    """value range of a state_dict entry, decided by its name and rank (the reference's naming: hrnet.py / multiframe_model.py modules)"""
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    if name.endswith("grid"):                       # MultiFrameNetBasic's registered mesh grid (multiframe_model.py:172-185): not a parameter
        return None
    n, k = t.numel(), salt + name
    if name.endswith("running_mean"):
        v = hash_uniform(k, n, -0.2, 0.2)
    elif name.endswith("running_var"):
        v = hash_uniform(k, n, 0.5, 1.5)
    elif t.dim() == 4:
        v = hash_uniform(k, n, -1.0 / math.sqrt(t.shape[1] * t.shape[2] * t.shape[3]), 1.0 / math.sqrt(t.shape[1] * t.shape[2] * t.shape[3]))
    elif name.endswith(".weight"):                  # 1-D weight: BatchNorm gamma
        v = hash_uniform(k, n, 0.5, 1.5)
    elif name.endswith(".bias"):
        # a bias next to a 4-D weight of the same module is a convolution bias, otherwise BatchNorm beta
        return ("bias", k, n)
    else:
        raise KeyError(name)
    return torch.from_numpy(v.reshape(tuple(t.shape)).copy())


def hashed_state_for(model, salt: str = ""):
    """{key: tensor} for every entry of model.state_dict(), values by key hash (identical to the oracle's `hashed_state` of the same table)"""
    sd = model.state_dict()
    out = {}
    for name, t in sd.items():
        e = _entry(name, t, salt)
        if e is None:
            out[name] = t.detach().clone()
        elif isinstance(e, tuple):
            _, k, n = e
            conv_bias = name[:-4] + "weight" in sd and sd[name[:-4] + "weight"].dim() == 4
            v = hash_uniform(k, n, -0.1, 0.1) if conv_bias else hash_uniform(k, n, -0.2, 0.2)
            out[name] = torch.from_numpy(v.reshape(tuple(t.shape)).copy())
        else:
            out[name] = e
    return out


def fill_hashed(model, salt: str = ""):
    """load the key-hash state into `model` (strict) and return it"""
    model.load_state_dict(hashed_state_for(model, salt), strict=True)
    return model
