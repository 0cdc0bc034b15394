import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(16384, 1024, 1024), (4096, 4096, 4096), (8192, 8192, 8192), (16448, 3072, 1024), (16448, 1024, 4096), (16448, 4096, 1024)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    fl = 2 * M * N * K
    t128 = timeit(lambda: ops.gemm_nt_bf16(a, w, b, True, torch.bfloat16))
    t256 = timeit(lambda: ops.gemm_nt_bf16(a, w, b, True, torch.bfloat16, tile256=True))
    bb = b.to(torch.bfloat16)
    tlt = timeit(lambda: F.linear(a, w, bb))
    print(f"M{M} N{N} K{K}: 128-tile {t128:7.1f} us ({fl/t128/1e6:5.0f} TF)  256-tile {t256:7.1f} us ({fl/t256/1e6:5.0f} TF)  hipBLASLt {tlt:7.1f} us ({fl/tlt/1e6:5.0f} TF)")
