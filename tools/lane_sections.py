"""Per parallel section (one HighResolutionModule's branches, or the per-source fuse paths) and lane: time from the fork to the lane's last
record, from ~100 events per step (mfc_prof_enable(2)) -- the step is not perturbed.  Shows which lane is the critical one per module.
    python tools/lane_sections.py [width] [batch]"""
import os, sys, csv
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch, mfcnet_amd as mfc
from mfcnet_amd import _lib as L
width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=width, compute_dtype="bf16").cuda().train()
opt = mfc.FlatAdam(m, lr=1e-4)
g = torch.Generator().manual_seed(1)
frames = [torch.randn(B, 3, 480, 640, generator=g).cuda() for _ in range(3)]
mask = torch.randint(0, 5, (B, 480, 640), generator=g).cuda()
def step():
    opt.zero_grad(); loss, _ = mfc.mfc_loss(m(frames), mask); loss.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"un-profiled step: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
L.lib.mfc_prof_enable(2)
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); t1 = time.perf_counter()
L.lib.mfc_prof_enable(0)
print(f"step with section markers: {(t1 - t0) * 1e3:.3f} ms")
path = os.path.join(ROOT, "gpurun_out", f"sections_w{width}_b{B}.csv")
os.makedirs(os.path.dirname(path), exist_ok=True)
assert L.lib.mfc_prof_dump(path.encode()) == 0
rows = [(r["name"], int(r["stream"]), float(r["start_us"]), float(r["end_us"])) for r in csv.DictReader(open(path))]
secs = defaultdict(dict)
for n, s, a, e in rows:
    lane, first = n.split("@")
    secs[(a, int(first))][lane] = e - a
tot_max = tot_main = 0.0
print("section start(ms)  first-record  lane spans (us)")
for (a, first), d in sorted(secs.items()):
    mx = max(d.values())
    tot_max += mx; tot_main += d.get("lane1", 0.0)
    print(f"{a / 1e3:9.3f}  rec {first:5d}  " + "  ".join(f"{k}:{v:7.1f}" for k, v in sorted(d.items())) + f"   critical {max(d, key=d.get)}")
print(f"sum over sections of the longest lane: {tot_max / 1e3:.3f} ms; of lane 1 alone: {tot_main / 1e3:.3f} ms; sections: {len(secs)}")
