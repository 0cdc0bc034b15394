"""A/B of knn_scores_kernel variants in ONE process, interleaved rounds (cdna guide rule 24)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=100000)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--variants", default="0,1,11,12,13")
ap.add_argument("--rounds", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
D = 8448
g = torch.Generator(device=dev).manual_seed(0)
gal = torch.empty((a.N, D), dtype=torch.bfloat16, device=dev)
for lo in range(0, a.N, 25000):
    n = min(25000, a.N - lo)
    gal[lo:lo + n] = torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
q = torch.nn.functional.normalize(torch.randn(a.B, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
ws = ops.knn_workspace(a.B, a.N, D, 10, dev)
variants = [int(v) for v in a.variants.split(",")]
times = {v: [] for v in variants}
for r in range(a.rounds + 1):
    for v in variants:
        _lib.tuning_set("VPR_KNN_VARIANT", int(v))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.knn_scores(q, gal, ws)
        e0.record(); ops.knn_scores(q, gal, ws); ops.knn_scores(q, gal, ws); e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[v].append(e0.elapsed_time(e1) / 2)
byt = a.N * D * 2 + a.B * D * 2 + a.B * 80
for v in variants:
    t = sorted(times[v])
    print(f"variant {v:3d}: median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  -> {byt / (t[len(t)//2]*1e-3) / 1e9:7.0f} GB/s (median)")
