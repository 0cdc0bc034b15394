"""A/B of the score kernel's store path (VPR_KNN_VARIANT: 0 shipped policy = staged + nt above 131k rows, 5 direct dword
stores, 6 whole row segments through LDS, 7 the same with nt stores): score-stage time (events between the two stages of
the real call) and identical answers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
D, k, B = 8448, 10, 64
g = torch.Generator(device=dev).manual_seed(0)
VARIANTS = [int(v) for v in os.environ.get("AB_VARIANTS", "0,5,6,7").split(",")]


def score_us(call, n=8):
    ev = []
    for _ in range(3):
        call(ev)
    ev.clear()
    for _ in range(n):
        out = call(ev)
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return ts[len(ts) // 2], out


for (N, fp8) in [(100_000, False), (500_000, False), (125_000, True), (1_000_000, True), (20_000, False)]:
    if fp8:
        G = torch.empty((N, D), dtype=torch.uint8, device=dev)
        gs = torch.empty((N,), dtype=torch.float32, device=dev)
        for lo in range(0, N, 50000):
            n = min(50000, N - lo)
            G[lo:lo + n], gs[lo:lo + n] = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=1))
        Q, qs = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1))
    else:
        G = torch.empty((N, D), dtype=torch.bfloat16, device=dev)
        for lo in range(0, N, 50000):
            n = min(50000, N - lo)
            G[lo:lo + n] = torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
        Q = torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
    ws = ops.knn_workspace(B, N, D, k, dev)
    call = (lambda ev: ops.knn_topk_fp8(Q, qs, G, gs, k, 0, ws, score_events=ev)) if fp8 else \
           (lambda ev: ops.knn_topk(Q, G, k, 0, ws, score_events=ev))
    line = f"N={N:8d} {'e4m3' if fp8 else 'bf16'}:"
    gb = N * D * (1 if fp8 else 2) / 1e3
    ref = None
    for rep in range(2):
        for variant in VARIANTS:
            _lib.tuning_set("VPR_KNN_VARIANT", int(variant))
            ws.zero_()
            t, (v, i) = score_us(call)
            S = ops.knn_scores_view(ws, B, N, D, k).clone() if N <= 125_000 else None
            if ref is None:
                ref = (v.clone(), i.clone(), S)
            same = torch.equal(v, ref[0]) and torch.equal(i, ref[1]) and (S is None or torch.equal(S, ref[2]))
            line += f"  v{variant} {t:7.1f} us ({gb / t / 1e3:.2f} TB/s){'' if same else ' DIFF!'}"
    _lib.tuning_set("VPR_KNN_VARIANT", 0)
    print(line, flush=True)
    del G, ws
