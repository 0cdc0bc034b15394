"""Stress of the two arrival-counter protocols (pose head VPR_POSE_VARIANT=1, Sinkhorn VPR_SALAD_VARIANT=3): thousands of
back-to-back calls of mixed shapes on shared workspaces, every result compared bitwise with the first result of its case
(a lost or early arrival shows up as a wrong or stale row), counters checked zero at the end."""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vpr_amd import _lib, ops
import test_salad_gpu as T
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = random.Random(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000

heads = []
for B, D, hidden, n_out in ((64, 8448, 1024, 4), (7, 768, 384, 2), (130, 1024, 512, 2), (1, 64, 48, 1), (256, 1024, 512, 8), (33, 2048, 256, 3)):
    x = torch.randn(B, D, generator=g).to(dev)
    W1 = (torch.randn(hidden, D, generator=g) * 0.02).to(dev); b1 = torch.randn(hidden, generator=g).to(dev) * 0.1
    W2 = (torch.randn(n_out, hidden, generator=g) * 0.05).to(dev); b2 = torch.zeros(n_out).to(dev)
    heads.append([x, W1, b1, W2, b2, None])
salads = []
for B in (1, 5, 16, 64, 33):
    tokens = torch.randn(B, 257, 1024, generator=g).to(torch.bfloat16).to(dev)
    w = T._to_dev(T._weights(1024, seed=B), dev, 1.0)
    salads.append([tokens, w, None])
_lib.tuning_set("VPR_POSE_VARIANT", 1)
_lib.tuning_set("VPR_SALAD_VARIANT", 3)
bad = 0
for it in range(N):
    if rnd.random() < 0.6:
        c = rnd.choice(heads)
        out = ops.pose_head(*c[:5], -1, fused=True)
        if c[5] is None: c[5] = out.clone()
        elif not torch.equal(out, c[5]): bad += 1
    else:
        c = rnd.choice(salads)
        out, _ = ops.salad_aggregate(c[0], c[1], 3)
        if c[2] is None: c[2] = out.clone()
        elif not torch.equal(out, c[2]): bad += 1
    if it % 500 == 499:
        torch.cuda.synchronize(); print(f"{it + 1} calls, {bad} mismatches", flush=True)
torch.cuda.synchronize()
z1 = int(ops.workspace("pose_fused", 256, dev)[:4096].view(torch.int32).abs().sum())
z2 = int(ops.workspace("salad", 256, dev)[:4096].view(torch.int32).abs().sum())
print(f"done: {N} calls, {bad} mismatches, counter words left: pose {z1}, salad {z2}")
sys.exit(1 if bad or z1 or z2 else 0)
