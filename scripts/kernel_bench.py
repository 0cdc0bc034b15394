"""Per-kernel timing of the hand-written stages on one MI355X (HIP events on torch's stream).
Usage: python scripts/kernel_bench.py [--N 100000] [--B 64] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100000)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--D", type=int, default=8448)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--C", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--N8", type=int, default=125000)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    res = {}
    want = lambda n: not a.only or n in a.only.split(",")

    if want("knn"):
        gal = torch.nn.functional.normalize(torch.randn(a.N, a.D, device=dev, generator=g), dim=1).to(torch.bfloat16)
        q = torch.nn.functional.normalize(torch.randn(a.B, a.D, device=dev, generator=g), dim=1).to(torch.bfloat16)
        ws = ops.knn_workspace(a.B, a.N, a.D, a.k, dev)
        med, best = timeit(lambda: ops.knn_scores(q, gal, ws), a.iters)
        byt = a.N * a.D * 2 + a.B * a.D * 2 + a.B * a.N * 4
        res["knn_scores"] = dict(ms=med * 1e3, best_ms=best * 1e3, GBps=byt / med / 1e9,
                                 TFLOPs=2 * a.B * a.N * a.D / med / 1e12)
        med, best = timeit(lambda: ops.knn_select(q, gal, a.k, ws), a.iters)
        res["knn_select"] = dict(ms=med * 1e3, best_ms=best * 1e3)
        med, best = timeit(lambda: ops.knn_topk(q, gal, a.k, 0, ws), a.iters)
        res["knn_topk_total"] = dict(ms=med * 1e3, best_ms=best * 1e3,
                                     GBps_alg=(a.N * a.D * 2 + a.B * a.D * 2 + a.B * a.k * 8) / med / 1e9)
        del gal

    if want("knn8"):
        # fp8 gallery (BASELINE config 5 arithmetic): one GPU's share of a 1M-row gallery by default
        N8 = a.N8
        gal8 = torch.empty((N8, a.D), dtype=torch.uint8, device=dev)
        gs8 = torch.empty((N8,), dtype=torch.float32, device=dev)
        for lo in range(0, N8, 25000):
            n = min(25000, N8 - lo)
            x = torch.nn.functional.normalize(torch.randn(n, a.D, device=dev, generator=g), dim=1)
            gal8[lo:lo + n], gs8[lo:lo + n] = ops.quantize_fp8_rows(x)
        q8, qs8 = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(a.B, a.D, device=dev, generator=g), dim=1))
        ws = ops.knn_workspace(a.B, N8, a.D, a.k, dev)
        med, best = timeit(lambda: ops.knn_topk_fp8(q8, qs8, gal8, gs8, a.k, 0, ws), a.iters)
        byt = N8 * a.D + a.B * a.D + a.B * a.k * 8 + N8 * 4
        res["knn_topk_fp8_total"] = dict(N=N8, ms=med * 1e3, best_ms=best * 1e3, GBps_alg=byt / med / 1e9)
        from vpr_amd.gallery import GalleryShard, GraphedLocalTopK
        gr = GraphedLocalTopK(GalleryShard(gal8, gs8, None, N8, 0, "fp8_e4m3"), a.B, a.k)
        med, best = timeit(lambda: gr(q8, qs8), a.iters)
        res["knn_topk_fp8_graph_replay"] = dict(N=N8, ms=med * 1e3, best_ms=best * 1e3, GBps_alg=byt / med / 1e9)
        del gal8

    if want("salad"):
        B, C = a.B, a.C
        tokens = torch.randn(B, 257, C, device=dev, generator=g).to(torch.bfloat16)
        r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02)
        w = ops.SaladWeights(w1_sc=r(1024, C).bfloat16(), b1_sc=r(1024), w2_s=r(64, 512).bfloat16(), b2_s=r(64),
                             w2_c=r(128, 512).bfloat16(), b2_c=r(128), w1_t=r(512, C).bfloat16(), b1_t=r(512),
                             w2_t=r(256, 512).bfloat16(), b2_t=r(256), dustbin=1.0)
        med, best = timeit(lambda: ops.salad_aggregate(tokens, w), a.iters)
        flops = B * (2 * 256 * C * 1024 + 2 * 256 * 512 * 192 + 2 * C * 512 + 2 * 512 * 256 + 2 * 128 * 64 * 256)
        res["salad_total"] = dict(ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flops / med / 1e12)
        x = tokens[:, 1:, :].reshape(B * 256, C).contiguous()
        med, best = timeit(lambda: ops.gemm_nt_bf16(x, w.w1_sc, w.b1_sc, True, torch.bfloat16, tile256=True), a.iters)
        res["salad_gemm_l1"] = dict(ms=med * 1e3, TFLOPs=2 * B * 256 * C * 1024 / med / 1e12, kernel="gemm256_kernel")
        med, best = timeit(lambda: ops.gemm_nt_bf16(x, w.w1_sc, w.b1_sc, True, torch.bfloat16), a.iters)
        res["salad_gemm_l1_tile128"] = dict(ms=med * 1e3, TFLOPs=2 * B * 256 * C * 1024 / med / 1e12)
        sc = torch.randn(B, 256, 64, device=dev, generator=g)
        ft = torch.randn(B, 256, 128, device=dev, generator=g)
        tk = torch.randn(B, 256, device=dev, generator=g)
        med, best = timeit(lambda: ops.salad_sinkhorn_aggregate(sc, ft, tk, 1.0, 3, True), a.iters)
        res["salad_sinkhorn_aggregate"] = dict(ms=med * 1e3, best_ms=best * 1e3)

    if want("head"):
        x = torch.randn(a.B, a.D, device=dev, generator=g)
        W1 = torch.randn(512, a.D, device=dev, generator=g) * 0.01
        b1 = torch.zeros(512, device=dev)
        W2 = torch.randn(4, 512, device=dev, generator=g) * 0.05
        b2 = torch.zeros(4, device=dev)
        med, best = timeit(lambda: ops.pose_head(x, W1, b1, W2, b2, 2), a.iters)
        res["pose_head"] = dict(ms=med * 1e3, best_ms=best * 1e3, GBps=(W1.numel() * 4 + x.numel() * 4) / med / 1e9)
        xs = torch.randn(256, 49, 1024, device=dev, generator=g).to(torch.bfloat16)
        gm, bt = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
        Wh, bh = torch.randn(4, 1024, device=dev, generator=g) * 0.03, torch.zeros(4, device=dev)
        med, best = timeit(lambda: ops.ln_meanpool_head(xs, gm, bt, 1e-5, Wh, bh, 2), a.iters)
        res["ln_meanpool_head_256x49x1024"] = dict(ms=med * 1e3, GBps=xs.numel() * 2 / med / 1e9)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
