"""Phase clocks of sinkhorn_aggregate_kernel (timing-only library: `make -C .../csrc ablation`, VPR_AMD_LIBRARY pointing at
libvpr_amd_ablation.so).  Thread 0 of every workgroup stores s_memrealtime (100 MHz constant clock: 10 ns ticks) at
the phase boundaries; prints the median over workgroups of each phase, for one and two input slabs."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VPR_AMD_LIBRARY", os.path.join(ROOT, "visual-place-recognition-and-geopose-estimation_amd", "libvpr_amd_ablation.so"))
from vpr_amd import _lib, ops  # noqa: E402

NAMES = ["issue loads", "scores arrive + LDS transpose", "K = exp(M - rowmax)", "iterations", "P planes", "aggregation MFMAs (+ feature arrival)",
         "norms", "output stores"]


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    L.vpr_salad_sinkhorn_set_clocks.restype = ctypes.c_int
    L.vpr_salad_sinkhorn_set_clocks.argtypes = [ctypes.c_void_p]
    B = 64
    g = torch.Generator(device=dev).manual_seed(0)
    clocks = torch.zeros((B, 16), dtype=torch.int64, device=dev)
    assert L.vpr_salad_sinkhorn_set_clocks(ctypes.c_void_p(clocks.data_ptr())) == 0
    sc = torch.randn(2, B, 256, 64, device=dev, generator=g)
    ft = torch.randn(2, B, 256, 128, device=dev, generator=g)
    tk = torch.randn(B, 256, device=dev, generator=g)
    for _ in range(3):
        ops.salad_sinkhorn_aggregate(sc[0], ft[0], tk, 1.0, 3, True)
    torch.cuda.synchronize()
    c = clocks.cpu().double()
    d = (c[:, 1:9] - c[:, 0:8])
    tick_ns = 10.0
    print("one slab (vpr_salad_sinkhorn_aggregate), median over 64 workgroups, us:")
    for n, v in zip(NAMES, d.median(0).values.tolist()):
        print(f"  {n:42s} {v * tick_ns / 1e3:6.2f}")
    print(f"  {'kernel body (first to last clock)':42s} {((c[:, 8] - c[:, 0]).median().item()) * tick_ns / 1e3:6.2f}")
    print(f"  spread of workgroup start times: {(c[:, 0].max() - c[:, 0].min()).item() * tick_ns / 1e3:.2f} us")
    assert L.vpr_salad_sinkhorn_set_clocks(None) == 0          # off again: the buffer is about to be freed


def fuse2_phases():
    """Phase clocks of gemm256_fuse2_kernel (SALAD layer 1 + fused second layers), B = 64, C = 1024."""
    dev = torch.device("cuda:0")
    L = _lib.lib()
    L.vpr_gemm256_fuse2_set_clocks.restype = ctypes.c_int
    L.vpr_gemm256_fuse2_set_clocks.argtypes = [ctypes.c_void_p]
    B, C = 64, 1024
    g = torch.Generator(device=dev).manual_seed(0)
    clocks = torch.zeros((4 * B, 8), dtype=torch.int64, device=dev)
    assert L.vpr_gemm256_fuse2_set_clocks(ctypes.c_void_p(clocks.data_ptr())) == 0
    sk = torch.zeros((4 * B, 16), dtype=torch.int64, device=dev)  # the two-slab Sinkhorn kernel of the same call (4 workgroups per image)
    assert L.vpr_salad_sinkhorn_set_clocks(ctypes.c_void_p(sk.data_ptr())) == 0
    patch = torch.randn(B, 256, C, device=dev, generator=g).to(torch.bfloat16)
    cls = torch.randn(B, C, device=dev, generator=g).to(torch.bfloat16)
    r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02)
    w = ops.SaladWeights(w1_sc=r(1024, C).bfloat16(), b1_sc=r(1024), w2_s=r(64, 512).bfloat16(), b2_s=r(64),
                         w2_c=r(128, 512).bfloat16(), b2_c=r(128), w1_t=r(512, C).bfloat16(), b1_t=r(512),
                         w2_t=r(256, 512).bfloat16(), b2_t=r(256), dustbin=1.0)
    for _ in range(3):
        ops.salad_aggregate_split(patch, cls, w, 3, True, overlap=False)
    torch.cuda.synchronize()
    assert L.vpr_salad_sinkhorn_set_clocks(None) == 0
    c2 = sk.cpu().double()
    fin = c2[:, 8] > 0                                            # workgroups that went all the way (NQ = 4: the finishers)
    print(f"two slabs (inside vpr_salad_aggregate_split), {int(fin.sum())} finishing workgroups of {c2.shape[0]}, median, us:")
    names2 = NAMES[:6] + ["cross-wave V, partial norms, arrival", "finisher: norms + output stores"]
    d2 = (c2[:, 1:9] - c2[:, 0:8])[fin]
    for n, v in zip(names2, d2.median(0).values.tolist()):
        print(f"  {n:42s} {v * 10.0 / 1e3:6.2f}")
    print(f"  {'finisher body (first to last clock)':42s} {((c2[fin, 8] - c2[fin, 0]).median().item()) * 10.0 / 1e3:6.2f}")
    print(f"  {'non-finishers: start to arrival':42s} {((c2[~fin, 6] - c2[~fin, 0]).median().item()) * 10.0 / 1e3:6.2f}   (clock 6 = MFMAs issued)")
    c = clocks.cpu().double()
    names = ["prologue + K loop", "W2 requests, DMA tail, barrier", "bias / ReLU / bf16 -> LDS, W2 arrival", "second-layer MFMAs", "partial-sum stores"]
    d = c[:, 1:6] - c[:, 0:5]
    print("gemm256_fuse2_kernel, median over 256 tiles (thread 0 of each), us:")
    for n, v in zip(names, d.median(0).values.tolist()):
        print(f"  {n:42s} {v * 10.0 / 1e3:6.2f}")
    print(f"  {'kernel body':42s} {(c[:, 5] - c[:, 0]).median().item() * 10.0 / 1e3:6.2f}")
    print(f"  first start -> last end: {(c[:, 5].max() - c[:, 0].min()).item() * 10.0 / 1e3:.2f} us; start spread {(c[:, 0].max() - c[:, 0].min()).item() * 10.0 / 1e3:.2f} us")
    tn = torch.arange(4 * B) % 1                                   # (tile -> head mapping is rastered: report quantiles instead)
    for lo, hi, tag in ((0.0, 0.5, "faster half (score-head tiles: 64 outputs)"), (0.5, 1.0, "slower half (cluster-head tiles: 128 outputs)")):
        ep = (c[:, 5] - c[:, 1])
        order = torch.argsort(ep)
        sel = order[int(lo * len(order)):int(hi * len(order))]
        print(f"  {tag}: " + ", ".join(f"{n} {v * 10.0 / 1e3:.2f}" for n, v in zip(names, d[sel].median(0).values.tolist())))
    # cluster tiles (tn >= 2) do twice the second-layer work of score tiles; blockIdx -> tile mapping is rastered, so report quantiles
    q = torch.quantile(c[:, 5] - c[:, 1], torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64))
    print(f"  epilogue total (10/50/90 %): {[round(x * 10.0 / 1e3, 2) for x in q.tolist()]} us")


if __name__ == "__main__":
    main()
    fuse2_phases()
