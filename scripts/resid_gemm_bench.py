"""Does folding the residual add into the proj / fc2 GEMM (beta = 1, in place) pay?  TunableOp on."""
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
M = 16448
gemm_autotune(True, tuning=True)
for (N, K) in [(1024, 1024), (1024, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, N, device=dev).to(torch.bfloat16)
    wt = w.t()
    t_lin = timeit(lambda: F.linear(a, w, b))
    t_add = timeit(lambda: x.addmm_(a, wt))
    y = torch.empty_like(x)
    t_add_out = timeit(lambda: torch.addmm(x, a, wt, out=y))
    print(f"N{N} K{K}: linear+bias {t_lin:7.1f} us   x.addmm_(a, Wt) {t_add:7.1f} us   addmm(out=y) {t_add_out:7.1f} us")
