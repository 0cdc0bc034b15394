"""sinkhorn_aggregate_kernel: time vs number of Sinkhorn iterations (per-iteration cost vs fixed part)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
B = 64
g = torch.Generator(device=dev).manual_seed(0)
S = torch.randn(B, 256, 64, device=dev, generator=g)
F = torch.randn(B, 256, 128, device=dev, generator=g)
t = torch.randn(B, 256, device=dev, generator=g)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for it in (1, 3, 6, 12):
    print(f"iters={it:2d}: {timeit(lambda: ops.salad_sinkhorn_aggregate(S, F, t, 1.0, it)):.1f} us", flush=True)
