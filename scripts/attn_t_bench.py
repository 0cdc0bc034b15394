"""attention_kernel time vs sequence length around the 16-query tile boundary (257 = 16 tiles + 1 row)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
B, H = 64, 16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for T in (240, 256, 257, 272, 288):
    qkv = torch.randn(B, T, 3 * H * 64, device=dev, dtype=torch.bfloat16)
    print(f"T={T}: {timeit(lambda: ops.attention_qkv_bf16(qkv, H)):.1f} us", flush=True)
