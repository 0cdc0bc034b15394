"""gemm256 vs tuned hipBLASLt on the backbone's exact-tile shapes (M = 64 * 256)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
from vpr_amd.backbone import gemm_autotune
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gemm_autotune(True, tuning=True)
M = 16384
for (N, K, gelu) in [(3072, 1024, False), (4096, 1024, True), (1024, 1024, False), (1024, 4096, False)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b16 = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    b32 = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if gelu:
        t_lt = timeit(lambda: torch._addmm_activation(b16, a, w.t(), use_gelu=True, out=out))
        t_lt_plain = timeit(lambda: torch.addmm(b16, a, w.t(), out=out))
    else:
        t_lt = timeit(lambda: torch.addmm(b16, a, w.t(), out=out)); t_lt_plain = t_lt
    t_256 = timeit(lambda: ops.gemm_nt_bf16(a, w, b32, False, torch.bfloat16, tile256=True))
    fl = 2 * M * N * K
    print(f"N{N} K{K}: hipBLASLt {t_lt:6.1f} us (plain {t_lt_plain:6.1f})   gemm256 (bias only) {t_256:6.1f} us ({fl/t_256/1e6:5.0f} TF)", flush=True)
