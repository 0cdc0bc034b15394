"""Experiment: capture the HIP backbone path (with and without its cls side stream) into one HIP graph and replay it."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd.modules import DinoV2Salad
from vpr_amd.backbone import gemm_autotune
dev = torch.device("cuda:0")
torch.manual_seed(0)
ext = DinoV2Salad("vit_large").eval().to(dev).to(torch.bfloat16)
ext.backbone.fold_layerscale()
x = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16)
gemm_autotune(True, tuning=True)
for _ in range(3): ext.backbone(x, split=True)
torch.cuda.synchronize(); gemm_autotune(True, tuning=False)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
for side in (True, False):
    ext.backbone.cls_side_chain = side
    print(f"--- cls rows on a side stream: {side}")
    ref = ext.backbone(x, split=True)
    print("eager    enqueue %.2f ms  total %.2f ms" % timeit(lambda: ext.backbone(x, split=True)))
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): ext.backbone(x, split=True)          # warm this stream's buffers / side stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = ext.backbone(x, split=True)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", torch.equal(out.patch, ref.patch), torch.equal(out.cls, ref.cls))
    print("replay   enqueue %.2f ms  total %.2f ms" % timeit(lambda: g.replay()))
