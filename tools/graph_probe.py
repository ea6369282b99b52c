"""Experiment: does a captured hipGraph of the forward + backward programs run faster than mfc_program_run?"""
import ctypes as C, os, sys, time
if os.environ.get("HWQ"): os.environ["GPU_MAX_HW_QUEUES"] = os.environ["HWQ"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
import torch
import mfcnet_amd as mfc
from mfcnet_amd import _lib as L
lib = L.lib
width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B, T, H, W = (int(sys.argv[2]) if len(sys.argv) > 2 else 8), 3, 480, 640
dev = torch.device("cuda")
model = mfc.HRNetMultiLarge(num_classes=5, num_frames=T, pretrained=False, width=width, compute_dtype="bf16").to(dev).train()
frames = [torch.randn(B, 3, H, W, device=dev) for _ in range(T)]
mask = torch.randint(0, 5, (B, H, W), device=dev)
for _ in range(2):
    model.zero_grad(); loss, _ = mfc.mfc_loss(model(frames), mask); loss.backward()
torch.cuda.synchronize()
plan = next(iter(model._plans.values()))
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
def run_direct():
    assert lib.mfc_program_run(plan.fwd_prog, len(plan.fwd_prog), sp) == 0
    assert lib.mfc_program_run(plan.bwd_prog, len(plan.bwd_prog), sp) == 0
def timeit(fn, iters=10):
    fn(); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    s.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
print(f"direct (lanes): {timeit(run_direct):.2f} ms per fwd+bwd", flush=True)
sp0 = C.c_void_p(0)
def run_null():
    assert lib.mfc_program_run(plan.fwd_prog, len(plan.fwd_prog), sp0) == 0
    assert lib.mfc_program_run(plan.bwd_prog, len(plan.bwd_prog), sp0) == 0
def timeit0(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
print(f"direct (lanes, null stream): {timeit0(run_null):.2f} ms per fwd+bwd", flush=True)
gf, gb = C.c_void_p(), C.c_void_p()
rc = lib.mfc_graph_capture(plan.fwd_prog, len(plan.fwd_prog), sp, C.byref(gf)); print("capture fwd rc", rc, flush=True)
rc2 = lib.mfc_graph_capture(plan.bwd_prog, len(plan.bwd_prog), sp, C.byref(gb)); print("capture bwd rc", rc2, flush=True)
if rc == 0 and rc2 == 0:
    def run_graph():
        assert lib.mfc_graph_launch(gf, sp) == 0
        assert lib.mfc_graph_launch(gb, sp) == 0
    print(f"graph (lanes):  {timeit(run_graph):.2f} ms per fwd+bwd", flush=True)
lib.mfc_set_flag(9, 0)
print(f"direct (serial): {timeit(run_direct):.2f} ms", flush=True)
gf2, gb2 = C.c_void_p(), C.c_void_p()
if lib.mfc_graph_capture(plan.fwd_prog, len(plan.fwd_prog), sp, C.byref(gf2)) == 0 and lib.mfc_graph_capture(plan.bwd_prog, len(plan.bwd_prog), sp, C.byref(gb2)) == 0:
    def run_graph2():
        assert lib.mfc_graph_launch(gf2, sp) == 0
        assert lib.mfc_graph_launch(gb2, sp) == 0
    print(f"graph (serial):  {timeit(run_graph2):.2f} ms", flush=True)
