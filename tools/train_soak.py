"""Does the step train?  N optimisation steps on one fixed synthetic batch (the benchmarked shape): the loss must fall (the model memorises the
batch), stay finite, and -- the step being bit-reproducible -- a second run must print the same numbers.
python tools/train_soak.py [dtype] [steps] [width]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
import mfcnet_amd as mfc
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
width = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B, T, H, W, nc = 8, 3, 480, 640, 5
torch.manual_seed(1234)
m = mfc.HRNetMultiLarge(num_classes=nc, num_frames=T, pretrained=False, width=width, compute_dtype=dtype).cuda().train()
opt = mfc.FlatAdam(m, lr=1e-4)
g = torch.Generator().manual_seed(42)
frames = [torch.randn(B, 3, H, W, generator=g).cuda() for _ in range(T)]
mask = torch.randint(0, nc, (B, H, W), generator=g).cuda()
accs = []
t0 = time.time()
for i in range(steps):
    _, acc = mfc.train_step(m, opt, frames, mask)
    if i % 10 == 0 or i == steps - 1:
        accs.append((i, acc.clone()))
torch.cuda.synchronize()
dt = time.time() - t0
print(f"{dtype} w{width} B={B} T={T} {H}x{W}: {steps} steps in {dt:.1f} s ({1e3 * dt / steps:.1f} ms/step incl. host), skipped (fp16 guard): {opt.skipped_steps()}")
for i, a in accs:
    print(f"  step {i:4d}: total {float(a[28]):.5f}  nll {float(a[26]):.5f}  soft-jaccard {float(a[27]):.5f}")
first, last = float(accs[0][1][28]), float(accs[-1][1][28])
assert last == last and last < first, "the loss did not fall"
print(f"  loss {first:.4f} -> {last:.4f}; parameters finite: {bool(torch.isfinite(m._P).all())}")
