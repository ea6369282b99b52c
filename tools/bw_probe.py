"""How fast can simple streaming kernels go on this GPU for tensor sizes of the hot path (MALL-resident vs HBM)?"""
import torch
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3
for mb in (7, 15, 30, 59, 118, 236, 472, 944):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x); z = torch.empty_like(x)
    t1 = timeit(lambda: y.copy_(x))
    t2 = timeit(lambda: torch.add(x, y, out=z))
    t3 = timeit(lambda: x.float().sum()) if mb <= 236 else 0
    print(f"{mb:4d} MB tensors: copy {2*mb/1024/t1/1e3:6.2f} TB/s ({t1*1e6:6.1f} us) | add3 {3*mb/1024/t2/1e3:6.2f} TB/s ({t2*1e6:6.1f} us)", flush=True)
