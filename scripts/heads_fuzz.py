"""One-off fuzz of the head kernels against oracle/heads.py: random batch sizes, widths, hidden sizes, output counts
and pair offsets for vpr_pose_head (both first-layer forms, the single-Linear form) and vpr_ln_meanpool_head.
Shapes the C ABI declares unsupported must come back as a Python exception, never as a fault.  Test infrastructure."""
import os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import heads as oheads
from vpr_amd import ops
import test_heads_gpu as T
dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = 0.0
refused = 0
for case in range(n):
    g = torch.Generator().manual_seed(case)
    B = rnd.choice([1, 2, 7, 33, 64, 65, 130, 256, rnd.randint(1, 300)])
    D = 16 * rnd.randint(1, 600)
    hidden = rnd.choice([0, 32, 64, 96, 512, 1024, 32 * rnd.randint(1, 40)])
    n_out = rnd.randint(1, 8)
    off = rnd.choice([-1] + list(range(0, max(1, n_out - 1))))
    if off >= 0 and off + 2 > n_out:
        off = -1
    split = rnd.random() < 0.6
    x = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)      # descriptor-like: the contract's domain
    W2, b2 = T._linear_init(n_out, hidden if hidden else D, g)
    if hidden:
        W1, b1 = T._linear_init(hidden, D, g)
        ref = oheads.mlp_head(x, W1, b1, W2, b2, off)
        call = lambda: ops.pose_head(x.to(dev), W1.to(dev), b1.to(dev), W2.to(dev), b2.to(dev), off, split=split)
    else:
        ref = oheads.mlp_head(x, None, None, W2, b2, off)
        call = lambda: ops.pose_head(x.to(dev), None, None, W2.to(dev), b2.to(dev), off)
    try:
        out = call().cpu().double()
    except RuntimeError as e:
        refused += 1
        print(f"case {case}: B={B} D={D} hidden={hidden} n_out={n_out} off={off} split={split}: refused ({str(e)[:60]})", flush=True)
        continue
    err = (out - ref).abs().max().item()
    worst = max(worst, err)
    print(f"case {case}: B={B} D={D} hidden={hidden} n_out={n_out} off={off} split={split}: abs err {err:.2e}", flush=True)
    if not err < 1e-4:
        sys.exit(1)
for case in range(n // 2):
    g = torch.Generator().manual_seed(1000 + case)
    B, Tn = rnd.choice([1, 3, 8, 64, 256, rnd.randint(1, 300)]), rnd.choice([1, 5, 49, 144, rnd.randint(1, 200)])
    H = 64 * rnd.randint(1, 32)
    dtype = rnd.choice([torch.float32, torch.bfloat16])
    n_out, off = rnd.randint(1, 8), -1
    if n_out >= 2 and rnd.random() < 0.5:
        off = rnd.randint(0, n_out - 2)
    x = (torch.randn(B, Tn, H, generator=g) * 1.5 + 0.3).to(dtype)
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    Wh, bh = T._linear_init(n_out, H, g)
    pooled_ref, out_ref = oheads.ln_meanpool_head(x, gamma, beta, 1e-5, Wh, bh, off)
    try:
        pooled, out = ops.ln_meanpool_head(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, Wh.to(dev), bh.to(dev), off)
    except RuntimeError as e:
        refused += 1
        print(f"ln case {case}: B={B} T={Tn} H={H} {dtype} n_out={n_out} off={off}: refused ({str(e)[:60]})", flush=True)
        continue
    e1 = (pooled.cpu().double() - pooled_ref).abs().max().item()
    e2 = (out.cpu().double() - out_ref).abs().max().item()
    worst = max(worst, e2)
    print(f"ln case {case}: B={B} T={Tn} H={H} {dtype} n_out={n_out} off={off}: pooled {e1:.2e} out {e2:.2e}", flush=True)
    if not (e1 < 2e-5 and e2 < 1e-4):
        sys.exit(1)
print("done; worst", worst, "refused", refused)
