"""One-off fuzz of vpr_salad_aggregate against oracle/salad.py: random batch sizes, widths, weight scales, dustbins and
Sinkhorn iteration counts (tolerance 1e-4 absolute on the descriptor, the north-star bar).  Test infrastructure."""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import salad as osalad
from vpr_amd import ops
import test_salad_gpu as T
dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    B = rnd.choice([1, 2, 3, 5, 8, 17, 33, 64, rnd.randint(1, 70)])
    C = 64 * rnd.randint(1, 24)
    std = rnd.choice([0.02, 0.02, 0.05, 0.08])
    dust = rnd.choice([1.0, 0.0, -2.0, rnd.uniform(-3, 3)])
    iters = rnd.choice([3, 3, 1, 2, 5])
    scale = rnd.choice([1.0, 1.0, 2.0, 0.5])
    g = torch.Generator().manual_seed(case)
    tokens = (torch.randn(B, 257, C, generator=g) * scale).to(torch.bfloat16)
    w = T._weights(C, seed=1000 + case, std=std)
    ref = osalad.salad_aggregate(tokens, w, dustbin=dust, iters=iters)
    out, out16 = ops.salad_aggregate(tokens.to(dev), T._to_dev(w, dev, dust), iters)
    err = (out.cpu().double() - ref).abs().max().item()
    worst = max(worst, err)
    ok = err < 1e-4 and bool(torch.isfinite(out).all()) and torch.equal(out16.cpu(), out.cpu().to(torch.bfloat16))
    print(f"case {case:2d} B={B:2d} C={C:4d} std={std} dust={dust:+.2f} iters={iters} scale={scale}: err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
print("worst", worst)
