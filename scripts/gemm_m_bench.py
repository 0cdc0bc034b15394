"""hipBLASLt (TunableOp-tuned) on the four backbone GEMM shapes: M = 64*257 vs M = 64*256 (+ other M)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd.backbone import gemm_autotune
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gemm_autotune(True, tuning=True, max_ms_per_gemm=int(os.environ.get('TUNE_MS', '150')))
for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
    line = f"N{N} K{K}:"
    for M in (16448, 16384, 64):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: F.linear(a, w, b))
        line += f"  M{M} {t:6.1f} us ({2*M*N*K/t/1e6:5.0f} TF)"
    print(line, flush=True)
