"""What bounds gemm256_kernel on the gathered-batch score tile (512 queries x a gallery shard, K = 8448)?  Three probes:

 1. concurrent tiles: 512 x (128 t) rows for t = 32 .. 512 score tiles of 256 x 256 (one per CU up to 256): if a tile's
    time does not depend on how many other CUs are busy, the limit is inside the CU (operand feed = bytes in flight /
    latency), not a shared resource (L2 / fabric / HBM);
 2. prefetch distance (timing-only library, VPR_GEMM256_DEPTH = 10 / 6 / 2: the counted wait leaves 5 / 3 / 1 half-tiles
    of 16 KB in flight behind it): time per K-tile against bytes in flight;
 3. --pmc: just runs the 512 x 12.5k bf16 and 512 x 125k e4m3 calls a few times, for rocprofv3 --pmc passes
    (TCC_HIT / TCC_MISS, SQ_WAIT_*, MFMA busy).
usage: python scripts/gemm_feed_probe.py [--pmc]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PMC = "--pmc" in sys.argv
if not PMC:
    os.environ.setdefault("VPR_AMD_LIBRARY", os.path.join(ROOT, "visual-place-recognition-and-geopose-estimation_amd", "libvpr_amd_ablation.so"))
from vpr_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
D = 8448


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


def unit(n):
    return torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=g), dim=1).to(torch.bfloat16)


if PMC:
    q = unit(512)
    gal = unit(12500)
    ws = ops.knn_workspace(512, 12500, D, 10, dev)
    for _ in range(5):
        ops.knn_topk(q, gal, 10, 0, ws)
    N8 = 125000
    g8 = torch.empty((N8, D), dtype=torch.uint8, device=dev)
    gs = torch.empty((N8,), dtype=torch.float32, device=dev)
    for lo in range(0, N8, 25000):
        x = torch.nn.functional.normalize(torch.randn(25000, D, device=dev, generator=g), dim=1)
        g8[lo:lo + 25000], gs[lo:lo + 25000] = ops.quantize_fp8_rows(x)
    q8, qs = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(512, D, device=dev, generator=g), dim=1))
    ws8 = ops.knn_workspace(512, N8, D, 10, dev)
    for _ in range(5):
        ops.knn_topk_fp8(q8, qs, g8, gs, 10, 0, ws8)
    torch.cuda.synchronize()
    sys.exit(0)

q = unit(512)
print("probe 1: concurrent 256 x 256 tiles (M = 512 queries, K = 8448, bf16, gemm256_kernel, f32 out)")
for tiles in (32, 64, 128, 192, 256, 384, 512, 1024):
    rows = tiles * 128                       # 2 query tiles x (rows / 256) gallery tiles
    gal = unit(rows)
    us = timeit(lambda: ops.gemm_nt_bf16(q, gal, None, False, torch.float32, tile256=True))
    fl = 2.0 * 512 * rows * D
    print(f"  {tiles:5d} tiles ({rows:6d} gallery rows): {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  "
          f"{fl / us / 1e6 / min(tiles, 256):6.2f} TFLOP/s per busy CU  operand feed per CU {64 * 1024 * (D / 64) * max(1, tiles / 256) / us / 1e3 / max(1, tiles / 256):6.1f} GB/s")
    del gal
print("probe 2: prefetch distance (256 tiles, 32768 gallery rows)")
gal = unit(32768)
for depth, kb in ((10, 80), (6, 48), (2, 16)):
    _lib.tuning_set("VPR_GEMM256_DEPTH", depth)
    us = timeit(lambda: ops.gemm_nt_bf16(q, gal, None, False, torch.float32, tile256=True))
    print(f"  vmcnt({depth:2d}): {kb:3d} KB in flight behind every wait: {us:7.1f} us = {us / (D / 64):5.2f} us per K-tile, "
          f"{2.0 * 512 * 32768 * D / us / 1e6:7.1f} TFLOP/s")
_lib.tuning_set("VPR_GEMM256_DEPTH", None)
