"""skinny_linear_kernel on the four cls-row shapes with weights that are cold in L2 (24 layers' worth, cycled)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
L = 24
shapes = {"qkv": (3072, 1024, 0), "proj": (1024, 1024, 2), "fc1": (4096, 1024, 1), "fc2": (1024, 4096, 2)}
ws = {k: [(torch.randn(n, kk, device=dev) * 0.02).to(torch.bfloat16) for _ in range(L)] for k, (n, kk, _) in shapes.items()}
bs = {k: torch.zeros(n, device=dev, dtype=torch.bfloat16) for k, (n, kk, _) in shapes.items()}
big = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device=dev)
def run(name):
    n, kk, mode = shapes[name]
    a = torch.randn(64, kk, device=dev).to(torch.bfloat16)
    out = torch.zeros(64, n, device=dev, dtype=torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        big.zero_()                       # flush L2 / Infinity Cache
        torch.cuda.synchronize()
        e0.record()
        for l in range(L):
            ops.skinny_linear_bf16(a, ws[name][l], None if mode == 2 else bs[name], out, mode)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / L * 1e3
for cfg in [("", ""), ("1", ""), ("4", ""), ("1", "16"), ("4", "16"), ("1", "4"), ("4", "4")]:
    for k, v in (("VPR_SKINNY_MBW", cfg[0]), ("VPR_SKINNY_NW", cfg[1])):
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    print(f"MBW={cfg[0] or 'auto':4s} NW={cfg[1] or 'auto':4s}: " + "  ".join(f"{n} {run(n):5.1f} us" for n in shapes), flush=True)
