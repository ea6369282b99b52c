"""Driver entry points: build() compiles every HIP source for gfx950; smoke() runs one tiny
forward+backward of the MFCNet hot path on cuda:0 and checks it against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "mfcnet-tracker_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def build() -> None:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU) -> mfcnet_amd/libmfcnet_hip.so, then import."""
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4"], check=True)
    import mfcnet_amd  # noqa: F401  (raises if the library is missing or an export is absent)
    from mfcnet_amd import _lib
    print("built", _lib.LIB_PATH, _lib.lib.mfc_version().decode())
    # the oracle is pure Python (torch CPU ops): nothing to compile; import it as the build check
    from oracle import mfcnet_oracle  # noqa: F401


def smoke() -> None:
    import torch
    import mfcnet_amd as mfc
    from oracle import mfcnet_oracle as O
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    T, B, H, W = 3, 1, 64, 96
    sd = O.hashed_state(O.mfcnet_table("HRNetMulti-Large", 48, 5, T, False, False))
    frames, _, _, mask = O.synthetic_clip("smoke", B, T, H, W, False, False)
    net = O.Net(sd, "HRNetMulti-Large", 48, 5, T).train()
    yo = net(frames)
    lo, _ = O.total_loss(yo, mask, 5)
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0").train()
    y = m([f.to("cuda:0") for f in frames])
    loss, _ = mfc.mfc_loss(y, mask.to("cuda:0"))
    loss.backward()
    err = float((y.detach().cpu() - yo.detach()).abs().max())
    g = m.multiframe_net.multiframe_net[9].weight.grad
    assert err < 1e-3, f"logits differ from the oracle by {err}"
    assert abs(float(loss) - float(lo)) < 1e-4 and bool(torch.isfinite(g).all())
    print(f"smoke ok: max|logits - oracle| = {err:.2e}, loss {float(loss):.5f} (oracle {float(lo):.5f})")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
