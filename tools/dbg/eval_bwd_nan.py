import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch, mfcnet_amd as mfc
for width in (16, 48):
    for mode in ("eval", "train", "headonly"):
        torch.manual_seed(7)
        model = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=width, compute_dtype="fp32").cuda()
        with torch.no_grad():
            model._RS.uniform_(0.5, 1.5)
        if mode == "eval": model.eval()
        elif mode == "train": model.train()
        else:
            model.train(); model.base_model.eval()
        g = torch.Generator().manual_seed(100)
        frames = [torch.randn(2, 3, 64, 96, generator=g).cuda() for _ in range(3)]
        mask = torch.randint(0, 5, (2, 64, 96), generator=g).cuda()
        loss, _ = mfc.mfc_loss(model(frames), mask)
        loss.backward()
        torch.cuda.synchronize()
        bad = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p.grad).all())]
        print(width, mode, "loss", float(loss), "nonfinite params:", len(bad), bad[:6], "gmax", float(model._G.abs().max()))
