"""Whole-call timing of vpr_knn_topk (bf16) and vpr_knn_topk_fp8 over gallery sizes (run under
rocprofv3 --kernel-trace to split the call into kernels: scripts/trace_medians.py <dir> 6)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
D, B, k = 8448, 64, 10
g = torch.Generator(device=dev).manual_seed(0)
def rows(n):
    out = torch.empty((n, D), dtype=torch.bfloat16, device=dev)
    for lo in range(0, n, 25000):
        m = min(25000, n - lo)
        out[lo:lo + m] = torch.nn.functional.normalize(torch.randn(m, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
    return out
def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
q = rows(B)
q8, qs = ops.quantize_fp8_rows(q.float())
for n in (100_000, 250_000, 500_000, 1_000_000):
    gal = rows(n)
    if n <= 500_000:
        t = timeit(lambda: ops.knn_topk(q, gal, k))
        print(f"bf16 N={n:8d}: {t:7.3f} ms  {n * D * 2 / t / 1e6:6.0f} GB/s", flush=True)
    g8, gs = ops.quantize_fp8_rows(gal.float()) if n <= 250_000 else (None, None)
    if g8 is None:                       # quantise in slabs: the f32 copy of 1M rows would be 34 GB
        g8 = torch.empty((n, D), dtype=torch.uint8, device=dev); gs = torch.empty(n, dtype=torch.float32, device=dev)
        for lo in range(0, n, 100_000):
            a, b = ops.quantize_fp8_rows(gal[lo:lo + 100_000].float())
            g8[lo:lo + 100_000] = a; gs[lo:lo + 100_000] = b
    del gal
    t = timeit(lambda: ops.knn_topk_fp8(q8, qs, g8, gs, k))
    print(f"fp8  N={n:8d}: {t:7.3f} ms  {n * D / t / 1e6:6.0f} GB/s", flush=True)
    del g8, gs
