"""Serial (event-bracketed, one record at a time) time of the step's programs by REGION: each maximal run of lane-0 records and each
parallel section, in program order -- shows what the serial head and tail of the backward (head, last layer, layer1, stem) are made of.
    python tools/region_times.py [width] [B] [reps]"""
import ctypes as C
import os
import sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import mfcnet_amd as mfc  # noqa: E402
from mfcnet_amd import _lib as L  # noqa: E402
from profile_step import describe, KIND  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T, H, W = 3, 480, 640
dev = torch.device("cuda")
torch.manual_seed(0)
model = mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False, width=width, compute_dtype="bf16").to(dev).train()
frames = [torch.randn(B, 3, H, W, device=dev) for _ in range(T)]
mask = torch.randint(0, 5, (B, H, W), device=dev)
for _ in range(2):
    model.zero_grad()
    loss, _ = mfc.mfc_loss(model(frames), mask)
    loss.backward()
torch.cuda.synchronize()
plan = model._plan if hasattr(model, "_plan") else next(iter(model._plans.values()))
for name, prog in (("fwd", plan.fwd_prog), ("bwd", plan.bwd_prog)):
    n = len(prog)
    ms = (C.c_float * n)()
    assert L.lib.mfc_program_profile(prog, n, reps, ms, L.stream_ptr()) == 0
    regions, cur = [], None
    for i in range(n):
        lane = prog[i].lane & 0xff
        det = bool(prog[i].lane & L.LANE_ASYNC)
        kind = "detached" if det else ("section" if lane else "serial")
        if det:                       # detached records overlap whatever region they sit in: counted apart
            regions.append(["detached", i, i, ms[i], {}])
            continue
        if cur is None or cur[0] != kind:
            cur = [kind, i, i, 0.0, defaultdict(float)]
            regions.append(cur)
        cur[2] = i
        cur[3] += ms[i]
        tag = describe(prog[i], 2)[0]
        cur[4][tag] += ms[i]
    det = sum(r[3] for r in regions if r[0] == "detached")
    print(f"==== {name}: {n} records, serial sum {sum(ms):.3f} ms, of which detached (weight gradients + unpack) {det:.3f} ms")
    # merge: consecutive non-detached regions of the same kind that were split only by detached records
    merged = []
    for r in regions:
        if r[0] == "detached":
            continue
        if merged and merged[-1][0] == r[0]:
            merged[-1][2] = r[2]; merged[-1][3] += r[3]
            for k, v in r[4].items():
                merged[-1][4][k] += v
        else:
            merged.append(r)
    for kind, a, b, t, tags in merged:
        print(f"{kind:8s} rec {a:5d}..{b:5d}  {t:7.3f} ms")
        if kind == "serial" and t > 0.25:
            for tag, v in sorted(tags.items(), key=lambda kv: -kv[1])[:14]:
                print(f"              {v:7.3f} ms  {tag}")
