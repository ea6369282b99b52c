"""Timeline of one multi-stream training step from the library's own event profiler (rocprofv3 serialises the streams, so it cannot show
this): per stream the busy time, the first / last launch, and the idle gaps of the main chain.
    python tools/lane_timeline.py [width] [batch]"""
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
L.lib.mfc_prof_enable(1); step(); torch.cuda.synchronize(); L.lib.mfc_prof_enable(0)
path = os.path.join(ROOT, "gpurun_out", f"timeline_w{width}_b{B}.csv")
os.makedirs(os.path.dirname(path), exist_ok=True)
assert L.lib.mfc_prof_dump(path.encode()) == 0
rows = [(r["name"], int(r["stream"]), float(r["start_us"]), float(r["end_us"])) for r in csv.DictReader(open(path))]
tend = max(r[3] for r in rows)
print(f"step (with events): {tend / 1e3:.3f} ms, {len(rows)} launches")
by = defaultdict(list)
for r in rows: by[r[1]].append(r)
for s, ks in sorted(by.items()):
    busy = sum(e - a for _, _, a, e in ks)
    fam = defaultdict(float)
    for n, _, a, e in ks: fam[n.split("<")[0]] += (e - a) / 1e3
    top = ", ".join(f"{k} {v:.2f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:6])
    print(f"stream {s}: {len(ks):5d} launches, busy {busy / 1e3:7.3f} ms, active {ks[0][2] / 1e3:7.3f} .. {max(k[3] for k in ks) / 1e3:7.3f} ms | {top}")
# phases on the main stream (0): forward end = head_gather_fwd, backward start etc.
for n, s, a, e in rows:
    if n.split("<")[0] in ("head_gather_fwd_kernel", "head_gather_bwd_kernel", "unpack_wgrad_kernel", "adam_kernel"):
        print(f"  {a / 1e3:8.3f} ms  stream {s}  {n}  {e - a:.1f} us")
# idle gaps of stream 0 (the chain): histogram
ks = sorted(by[0], key=lambda k: k[2])
gaps = [ks[i + 1][2] - ks[i][3] for i in range(len(ks) - 1)]
print("main stream: sum of gaps %.3f ms; gaps > 20 us: %d (%.3f ms); median gap %.2f us" % (sum(gaps) / 1e3, sum(1 for x in gaps if x > 20), sum(x for x in gaps if x > 20) / 1e3, sorted(gaps)[len(gaps) // 2]))
# per kernel family: mean duration in this (concurrent) run
fam = defaultdict(list)
for n, s, a, e in rows: fam[n].append(e - a)
for n, v in sorted(fam.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {sum(v) / 1e3:7.3f} ms  n={len(v):4d}  avg {sum(v) / len(v):7.1f} us  {n}")
