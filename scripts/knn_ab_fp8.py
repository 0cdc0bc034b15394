"""A/B of knn_scores_kernel variants on an e4m3 gallery (whole vpr_knn_topk_fp8 call), one process, interleaved."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=1000000)
ap.add_argument("--variants", default="0,3")
ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
D, B = 8448, 64
g = torch.Generator(device=dev).manual_seed(0)
g8 = torch.empty((a.N, D), dtype=torch.uint8, device=dev)
gs = torch.empty((a.N,), dtype=torch.float32, device=dev)
for lo in range(0, a.N, 65536):
    n = min(65536, a.N - lo)
    g8[lo:lo + n], gs[lo:lo + n] = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=1))
q8, qs = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1))
ws = ops.knn_workspace(B, a.N, D, 10, dev)
variants = [int(v) for v in a.variants.split(",")]
times = {v: [] for v in variants}
ref = None
for r in range(a.rounds + 1):
    for v in variants:
        _lib.tuning_set("VPR_KNN_VARIANT", int(v))
        out = ops.knn_topk_fp8(q8, qs, g8, gs, 10, 0, ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.knn_topk_fp8(q8, qs, g8, gs, 10, 0, ws); ops.knn_topk_fp8(q8, qs, g8, gs, 10, 0, ws); e1.record()
        torch.cuda.synchronize()
        if ref is None: ref = out
        assert torch.equal(out[1], ref[1]) and torch.equal(out[0], ref[0])
        if r > 0: times[v].append(e0.elapsed_time(e1) / 2)
byt = a.N * D + a.N * 4 + B * D + B * 80
for v in variants:
    t = sorted(times[v])
    print(f"variant {v}: median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us -> {byt / (t[len(t)//2]*1e-3) / 1e9:6.0f} GB/s")
