"""Generate tests/golden/resunet_vb_eval*.npz by importing the REFERENCE's models/resunet.py in the build container (never on the GPU box).
Weights come from the key-hash generator (oracle.hashed_state over oracle.resunet_table, salt "resunet"), the input from a seeded generator,
so the fixture stores only the case description and the reference's OUTPUT.

    python tests/golden/make_golden_resunet.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mfcnet_oracle as O  # noqa: E402

CASES = {"resunet_vb_eval": dict(channels=3, dim=16, out_dim=5, B=2, H=64, W=96, seed=5),
         "resunet_vb_eval_odd": dict(channels=3, dim=16, out_dim=5, B=3, H=40, W=72, seed=6)}


def case_input(cfg):
    g = torch.Generator().manual_seed(cfg["seed"])
    return torch.randn(cfg["B"], cfg["channels"], cfg["H"], cfg["W"], generator=g)


def main():
    spec = importlib.util.spec_from_file_location("ref_resunet", "/root/reference/models/resunet.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for name, cfg in CASES.items():
        net = m.ResUnet_VB(channels=cfg["channels"], dim=cfg["dim"], out_dim=cfg["out_dim"], dim_mults=(1, 2, 4, 8), resnet_block_groups=8).eval()
        tab = O.resunet_table(cfg["channels"], cfg["dim"], cfg["out_dim"])
        assert list(net.state_dict().keys()) == [t[0] for t in tab]
        net.load_state_dict(O.hashed_state(tab, salt="resunet"), strict=True)
        with torch.no_grad():
            y = net(case_input(cfg))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), logits=y.numpy().astype(np.float32),
                            meta=np.array([cfg[k] for k in ("channels", "dim", "out_dim", "B", "H", "W", "seed")], dtype=np.int64))
        print(name, tuple(y.shape), float(y.abs().max()))


if __name__ == "__main__":
    main()
