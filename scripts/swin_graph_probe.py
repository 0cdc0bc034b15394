"""Probe: can the Hugging Face Swin forward (+ the fused LN / mean-pool / head kernel) be captured into a HIP graph, is the
replay bit-identical, and what does it buy on the launch-bound Swin-T path (BASELINE config 1 shape, batch 8)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import SwinConfig, SwinModel
from vpr_amd.graphed import GraphedForward
from vpr_amd.modules import SwinRegressionModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, cfg, B, size in (("swin-tiny", SwinConfig(), 8, 224), ("swin-tiny", SwinConfig(), 64, 224),
                           ("swin-base-384", SwinConfig(image_size=384, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12), 32, 384)):
    model = SwinRegressionModel(SwinModel(cfg)).to(dev).eval()
    x = torch.randn(B, 3, size, size, device=dev)
    with torch.no_grad():
        ref = model(x)
    fwd = GraphedForward(model)
    out = fwd(x)
    same = torch.equal(out, ref)
    def t(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        te = t(lambda: model(x))
    tg = t(lambda: fwd(x))
    print(f"{name} B={B}: eager {te:.2f} ms ({B / te * 1e3:.0f} images/s)  graph replay {tg:.2f} ms ({B / tg * 1e3:.0f} images/s)  identical={same}", flush=True)
