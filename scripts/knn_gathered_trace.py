"""Kernel-trace target: the local search ONE rank of an 8-GPU job runs per step (512 all-gathered queries x its shard),
bf16 100k-gallery shard (12.5k rows) and e4m3 1M-gallery shard (125k rows).
  rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/knn_gathered_trace.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
D, k = 8448, 10
g = torch.Generator(device=dev).manual_seed(0)
for (B, N, fp8) in [(512, 12500, False), (512, 125000, True)]:
    gal = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1)
    ws = ops.knn_workspace(B, N, D, k, dev)
    if fp8:
        G, gs = ops.quantize_fp8_rows(gal)
        Q, qs = ops.quantize_fp8_rows(q)
        fn = lambda: ops.knn_topk_fp8(Q, qs, G, gs, k, 0, ws)
    else:
        G, Q = gal.to(torch.bfloat16), q.to(torch.bfloat16)
        fn = lambda: ops.knn_topk(Q, G, k, 0, ws)
    del gal
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
