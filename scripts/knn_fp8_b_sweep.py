"""fp8 score stage for > 64 queries: streaming kernel (one gallery pass per 64 queries) vs block-scaled fp8 GEMM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
D = 8448
g = torch.Generator(device=dev).manual_seed(0)
def rows8(n):
    q8 = torch.empty((n, D), dtype=torch.uint8, device=dev); sc = torch.empty(n, dtype=torch.float32, device=dev)
    for lo in range(0, n, 25000):
        m = min(25000, n - lo)
        a, b = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(m, D, device=dev, generator=g), dim=1))
        q8[lo:lo + m] = a; sc[lo:lo + m] = b
    return q8, sc
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, N) in [(128, 500000), (256, 250000), (512, 125000)]:
    q, qs = rows8(B); gal, gs = rows8(N)
    ws = ops.knn_workspace(B, N, D, 10, dev)
    line = f"B={B:4d} N={N:7d}:"
    for thr in (100000, 65):
        _lib.tuning_set("VPR_KNN_GEMM_MIN_B", int(thr))
        t = timeit(lambda: ops.knn_topk_fp8(q, qs, gal, gs, 10, 0, ws))
        line += f"  {'stream' if thr > 65 else 'gemm  '} {t:8.1f} us"
    print(line + f"   ({2*B*N*D/1e12:.2f} TFLOP)", flush=True)
