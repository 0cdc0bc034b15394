"""SALAD stage A/B in one process (B = 64, C = 1024 = BASELINE config 2): unfused route (VPR_SALAD_VARIANT=1: layer 1,
grouped second layers through HBM) vs fused route (second layers in the layer-1 tile epilogue), one-call vs staged with the
token MLP on a side stream; plus each stage on its own.  HIP events on the launch stream, median of --iters."""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops  # noqa: E402


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return round(ts[len(ts) // 2] * 1e3, 1), round(ts[0] * 1e3, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--C", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    B, C = a.B, a.C
    patch = torch.randn(B, 256, C, device=dev, generator=g).to(torch.bfloat16)
    cls = torch.randn(B, C, device=dev, generator=g).to(torch.bfloat16)
    r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02)
    w = ops.SaladWeights(w1_sc=r(1024, C).bfloat16(), b1_sc=r(1024), w2_s=r(64, 512).bfloat16(), b2_s=r(64),
                         w2_c=r(128, 512).bfloat16(), b2_c=r(128), w1_t=r(512, C).bfloat16(), b1_t=r(512),
                         w2_t=r(256, 512).bfloat16(), b2_t=r(256), dustbin=1.0)
    flops = B * (2 * 256 * C * 1024 + 2 * 256 * 512 * 192 + 2 * C * 512 + 2 * 512 * 256 + 2 * 128 * 64 * 256)
    res = {}
    outs = {}
    for name, variant, overlap in (("unfused_onecall", 1, False), ("fused_onecall", 0, False), ("fused_overlap", 0, True),
                                   ("unfused_onecall_again", 1, False), ("fused_overlap_again", 0, True)):
        with _lib.tuning(VPR_SALAD_VARIANT=variant):
            med, best = timeit(lambda: ops.salad_aggregate_split(patch, cls, w, 3, True, overlap=overlap), a.iters)
            outs[name] = ops.salad_aggregate_split(patch, cls, w, 3, True, overlap=overlap)[0].clone()
        res[name] = dict(us=med, best_us=best, TFLOPs=round(flops / med / 1e6, 1), frac_bf16_peak=round(flops / med / 1e6 / 2500, 3))
    res["max_abs_fused_vs_unfused"] = float((outs["fused_onecall"] - outs["unfused_onecall"]).abs().max())
    res["overlap_bit_identical"] = bool(torch.equal(outs["fused_onecall"], outs["fused_overlap"]))
    # the stages on their own (fused route)
    L = _lib.lib()
    m, l, t, hidden, n = 64, 128, 256, 512, 256
    ws = ops.workspace("salad", L.vpr_salad_workspace_bytes(B, n, C, m, l, t, hidden), dev)
    cw = w.c_struct()
    out = torch.empty((B, 8448), dtype=torch.float32, device=dev)
    out16 = torch.empty((B, 8448), dtype=torch.bfloat16, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    for variant in (0, 1):
        with _lib.tuning(VPR_SALAD_VARIANT=variant):
            tag = "fused" if variant == 0 else "unfused"
            res[f"stage_mlps_{tag}"] = timeit(lambda: _lib.check(L.vpr_salad_stage_mlps(p(patch), n * C, B, n, C, ctypes.byref(cw), m, l, t, hidden, p(ws), ws.numel(), ops._stream()), "mlps"), a.iters)
            res[f"stage_aggregate_{tag}"] = timeit(lambda: _lib.check(L.vpr_salad_stage_aggregate(B, n, C, 1.0, m, l, t, hidden, 3, p(out), p(out16), p(ws), ws.numel(), ops._stream()), "agg"), a.iters)
    res["stage_token"] = timeit(lambda: _lib.check(L.vpr_salad_stage_token(p(cls), C, B, n, C, ctypes.byref(cw), m, l, t, hidden, p(ws), ws.numel(), ops._stream()), "tok"), a.iters)
    x = patch.reshape(B * 256, C)
    res["gemm256_layer1_alone"] = timeit(lambda: ops.gemm_nt_bf16(x, w.w1_sc, w.b1_sc, True, torch.bfloat16, tile256=True), a.iters)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
