"""A/B of the pose head's split-K factor (VPR_POSE_KS) and timing of the Swin pooler head, one process, interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, D, H = 64, 8448, 1024
x = torch.randn(B, D, device=dev, generator=g)
W1 = torch.randn(H, D, device=dev, generator=g) * 0.01
b1 = torch.zeros(H, device=dev)
W2 = torch.randn(4, H, device=dev, generator=g) * 0.05
b2 = torch.zeros(4, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ks_list = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,11,16,22,24,33").split(",")]
res = {k: [] for k in ks_list}
for r in range(6):
    for k in ks_list:
        if k: _lib.tuning_set("VPR_POSE_KS", int(k))
        else: _lib.tuning_set("VPR_POSE_KS", None)
        t = timeit(lambda: ops.pose_head(x, W1, b1, W2, b2, 2))
        res[k].append(t)
for k in ks_list:
    t = sorted(res[k])
    print(f"VPR_POSE_KS={k:3d}: median {t[len(t)//2]:6.1f} us  min {t[0]:6.1f} us")
for T in (49, 144):
    xs = torch.randn(256, T, 1024, device=dev, generator=g).to(torch.bfloat16)
    gm, bt = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    Wh, bh = torch.randn(4, 1024, device=dev, generator=g) * 0.03, torch.zeros(4, device=dev)
    t = timeit(lambda: ops.ln_meanpool_head(xs, gm, bt, 1e-5, Wh, bh, 2, want_pooled=False))
    print(f"ln_meanpool_head 256x{T}x1024 bf16: {t:6.1f} us = {xs.numel() * 2 / t / 1e6:6.0f} GB/s")
