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
# training step (round 3): est -> log_softmax -> the reference's get_loss (0.7 NLL + 0.3 soft-Jaccard, src/loss.py) -> backward
TRAIN_CASES = {"resunet_vb_train": dict(channels=3, dim=16, out_dim=5, B=2, H=64, W=96, seed=7),
               "resunet_vb_train_odd": dict(channels=3, dim=16, out_dim=5, B=3, H=40, W=72, seed=8)}
SENTINELS = ["init_conv.weight", "init_conv.bias", "downs.0.0.block1.proj.weight", "downs.0.0.block1.proj.bias", "downs.0.0.block1.norm.weight",
             "downs.0.0.block1.norm.bias", "downs.1.1.1.weight", "downs.3.1.weight", "mid_block.block2.proj.weight", "mid_block.block2.norm.weight",
             "ups.0.0.res_conv.weight", "ups.0.0.block1.proj.weight", "ups.1.1.1.weight", "ups.3.1.weight", "final_res_block.res_conv.weight",
             "final_res_block.block2.norm.bias", "output_layer.weight", "output_layer.bias"]


def case_mask(cfg):
    g = torch.Generator().manual_seed(cfg["seed"] + 1000)
    return torch.randint(0, cfg["out_dim"], (cfg["B"], cfg["H"], cfg["W"]), generator=g)


def sample16(t):
    flat = t.detach().flatten().float()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    return flat[idx].numpy()


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
    from types import SimpleNamespace
    lspec = importlib.util.spec_from_file_location("ref_loss", "/root/reference/src/loss.py")
    ref_loss = importlib.util.module_from_spec(lspec)
    lspec.loader.exec_module(ref_loss)
    for name, cfg in TRAIN_CASES.items():
        net = m.ResUnet_VB(channels=cfg["channels"], dim=cfg["dim"], out_dim=cfg["out_dim"], dim_mults=(1, 2, 4, 8), resnet_block_groups=8).train()
        tab = O.resunet_table(cfg["channels"], cfg["dim"], cfg["out_dim"])
        net.load_state_dict(O.hashed_state(tab, salt="resunet"), strict=True)
        y = net(case_input(cfg))
        args = SimpleNamespace(class_weights=np.array([1.0, 1000.0, 1000.0, 1000.0, 1000.0]), num_classes=cfg["out_dim"])
        loss, ld = ref_loss.get_loss(torch.nn.functional.log_softmax(y, dim=1), case_mask(cfg), ["nll", "soft_jaccard"], [0.7, 0.3], args)
        loss.backward()
        out = dict(logits=y.detach().numpy().astype(np.float32), loss_total=np.float32(float(loss)),
                   meta=np.array([cfg[k] for k in ("channels", "dim", "out_dim", "B", "H", "W", "seed")], dtype=np.int64))
        named = dict(net.named_parameters())
        for pn, prm in named.items():
            out["gradnorm/" + pn] = np.float64(float(prm.grad.double().norm()))
        for pn in SENTINELS:
            out["gradsample/" + pn] = sample16(named[pn].grad)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, tuple(y.shape), "loss", float(loss), "grad norms", float(named["init_conv.weight"].grad.norm()), float(named["output_layer.weight"].grad.norm()))


if __name__ == "__main__":
    main()
