"""Score-stage time vs query-batch size on a fixed shard: streaming kernel (one gallery pass per 64 queries)
vs the MFMA GEMM path (VPR_KNN_GEMM_MIN_B picks the crossover)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
D = 8448
g = torch.Generator(device=dev).manual_seed(0)
def rows(n):
    out = torch.empty((n, D), dtype=torch.bfloat16, device=dev)
    for lo in range(0, n, 25000):
        m = min(25000, n - lo)
        out[lo:lo + m] = torch.nn.functional.normalize(torch.randn(m, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
    return out
def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, N) in [(128, 50000), (192, 33333), (256, 25000), (512, 12500), (256, 250000), (512, 125000)]:
    q, gal = rows(B), rows(N)
    ws = ops.knn_workspace(B, N, D, 10, dev)
    line = f"B={B:4d} N={N:6d}:"
    for thr, stages, g256 in ((100000, "2", "1"), (1, "2", "0"), (1, "3", "0"), (1, "2", "1")):
        _lib.tuning_set("VPR_KNN_GEMM_MIN_B", int(thr))
        _lib.tuning_set("VPR_GEMM_NT_STAGES", int(stages))
        _lib.tuning_set("VPR_KNN_FP8_GEMM256", int(g256))
        t = timeit(lambda: ops.knn_scores(q, gal, ws))
        line += f"  {'stream' if thr > 1 else ('gemm/' + stages + '-stage' if g256 == '0' else 'auto(256-tile when it fills)')} {t:7.1f} us"
    print(line, flush=True)
