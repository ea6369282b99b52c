"""Run-to-run reproducibility of the bf16 train-mode forward (statistic atomics are order-dependent in the last bit; anything coarser is a race).
python tools/train_noise.py [lanes 0|1] [dtype] [runs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd")); sys.path.insert(0, ROOT)
import torch
import mfcnet_amd as mfc
from mfcnet_amd import _lib as L
from golden_util import case_inputs, case_state
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L.lib.mfc_set_flag(9, lanes)
cfg = dict(name="bitscase", model_type="HRNetMulti-Large", T=3, optflow=False, depth=False, B=2, H=96, W=128, mode="train")
frames, flows, depths, mask = case_inputs(cfg)
m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, loadpath=None, width=48, compute_dtype=dtype)
m.load_state_dict(case_state(cfg, 48), strict=True)
m = m.cuda().train()
x = [f.cuda() for f in frames]
outs, grads = [], []
for _ in range(runs):
    m.zero_grad()
    y = m(x)
    loss, _ = mfc.mfc_loss(y, mask.cuda())
    loss.backward()
    outs.append(y.detach().float().cpu()); grads.append(m._G.detach().clone())
ref = outs[0]
distinct = []
for o in outs:
    if not any(float((o - d).abs().max()) < 1e-2 * float(ref.abs().max()) for d in distinct):
        distinct.append(o)
print(f"lanes={lanes} dtype={dtype} R={L.STAT_REPLICAS}: scale {float(ref.abs().max()):.3f}; max diffs to run 0:",
      " ".join(f"{float((o - ref).abs().max()):.4f}" for o in outs[1:]), f"-> {len(distinct)} distinct outcome(s) at 1 % of scale")
gn = float(grads[0].norm())
print("gradient arena, relative L2 distance to run 0:", " ".join(f"{float((g - grads[0]).norm()) / gn:.2e}" for g in grads[1:]),
      "| bit-identical:", all(torch.equal(g, grads[0]) for g in grads[1:]))
