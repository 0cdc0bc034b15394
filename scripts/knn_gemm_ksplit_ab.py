"""A/B of the split-K 256 x 256-tile score GEMM (gathered query batches against small shards): whole vpr_knn_topk* call
with VPR_KNN_GEMM_KSPLIT=0 (round-2 routes: 128 x 128 kernel for bf16 below 256 tiles, unsplit gemm256 for e4m3) and the
default (K slices until the one-per-CU workgroups fill the chip; the level-0 select adds the slabs), same process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
D = 8448
g = torch.Generator(device=dev).manual_seed(0)


def rows(n):
    out = torch.empty((n, D), dtype=torch.float32, device=dev)
    for lo in range(0, n, 25000):
        m = min(25000, n - lo)
        out[lo:lo + m] = torch.nn.functional.normalize(torch.randn(m, D, device=dev, generator=g), dim=1)
    return out


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, N) in [(256, 25000), (512, 12500), (512, 6378), (256, 50000), (512, 25000), (512, 40000), (512, 125000)]:
    gal32 = rows(N)
    pos = torch.randint(0, N, (B,), device=dev, generator=g)
    q32 = torch.nn.functional.normalize(gal32[pos] + 0.1 * torch.randn(B, D, device=dev, generator=g), dim=1)
    for fp8 in (False, True):
        if fp8:
            G, gs = ops.quantize_fp8_rows(gal32)
            Q, qs = ops.quantize_fp8_rows(q32)
            call = lambda: ops.knn_topk_fp8(Q, qs, G, gs, 10, ws=ws)
        else:
            G, Q = gal32.to(torch.bfloat16), q32.to(torch.bfloat16)
            call = lambda: ops.knn_topk(Q, G, 10, ws=ws)
        ws = ops.knn_workspace(B, N, D, 10, dev)
        line = f"B={B:4d} N={N:6d} {'e4m3' if fp8 else 'bf16'}:"
        res = []
        for ks in ("0", "1"):
            _lib.tuning_set("VPR_KNN_GEMM_KSPLIT", int(ks))
            v, i = call()
            res.append((v.clone(), i.clone()))
            name = _lib.lib().vpr_knn_scores_kernel_name(int(fp8), B, N).decode().replace("vpr::", "")
            line += f"  ksplit={'on ' if ks == '1' else 'off'} {timeit(call):7.1f} us ({name})"
        same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0])
        print(line + f"  same={same} top1={bool(torch.equal(res[1][1][:, 0].long(), pos))}", flush=True)
    del gal32, G
