"""A/B of the K-walk rotation of gemm256's tiles (VPR_GEMM256_STAGGER): plain 256-tile GEMMs at the SALAD layer-1 and ViT-L
shapes, the fused SALAD MLP stage, the gathered-batch kNN calls of an 8-GPU job's shard.  us per call, two repetitions."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
from vpr_amd.modules import DinoV2Salad
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


cases = {}
M = 16384
for (N, K) in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    cases[f"gemm256 {M}x{N}x{K}"] = (lambda a=a, w=w, b=b: ops.gemm_nt_bf16(a, w, b, False, torch.bfloat16, tile256=True))
torch.manual_seed(0)
ext = DinoV2Salad("vit_large").to(dev).to(torch.bfloat16).eval()
patch = torch.randn(64, 256, 1024, device=dev).to(torch.bfloat16)
cls = torch.randn(64, 1024, device=dev).to(torch.bfloat16)
wts = ext.aggregator.pack()
cases["salad_aggregate_split B=64"] = lambda: ops.salad_aggregate_split(patch, cls, wts, 3, True)
q = torch.nn.functional.normalize(torch.randn(512, 8448, device=dev), dim=1)
g16 = torch.nn.functional.normalize(torch.randn(12500, 8448, device=dev), dim=1).to(torch.bfloat16)
q16 = q.to(torch.bfloat16)
ws16 = ops.knn_workspace(512, 12500, 8448, 10, dev)
cases["knn_topk bf16 512x12500"] = lambda: ops.knn_topk(q16, g16, 10, 0, ws16)
g8, gs = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(125000, 8448, device=dev), dim=1))
q8, qs = ops.quantize_fp8_rows(q)
cases["knn_topk_fp8 512x125000"] = lambda: ops.knn_topk_fp8(q8, qs, g8, gs, 10, 0)
res = {k: {} for k in cases}
for rep in range(2):
    for mode in (0, 1, 2, 3, 4, 5):
        _lib.tuning_set("VPR_GEMM256_STAGGER", mode)
        for k, fn in cases.items():
            res[k].setdefault(mode, []).append(timeit(fn))
_lib.tuning_set("VPR_GEMM256_STAGGER", None)
for k, v in res.items():
    print(json.dumps({"case": k, "us_by_mode": v}), flush=True)
