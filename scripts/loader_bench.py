"""Host side of the evaluation loops: serial PIL decode + .to(device) (what the loops did before) vs
loader.ImageBatchLoader (thread-pool decode into pinned buffers, async copy), alone and feeding the DINOv2 ViT-L/14 +
SALAD extractor (the 11 ms / 64-image GPU step of bench.py).  Synthetic JPEGs (quality 90, natural-ish content)."""
import os, sys, tempfile, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd.loader import ImageBatchLoader
from vpr_amd.modules import DinoV2Salad
from vpr_amd.preprocess import ResizeNormalize, HALF_MEAN, HALF_STD

dev = torch.device("cuda:0")
N, B = int(os.environ.get("LB_IMAGES", "1536")), 64
W, H = [int(v) for v in os.environ.get("LB_SIZE", "640x480").split("x")]
tmp = tempfile.mkdtemp(prefix="vpr_loader_")
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:H, 0:W]
names = []
for i in range(N):
    base = (np.sin(xx / (7 + i % 13)) + np.cos(yy / (5 + i % 11)))[..., None] * 60 + 128
    arr = np.clip(base + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
    names.append(f"{i:05d}.jpg")
    Image.fromarray(arr).save(os.path.join(tmp, names[-1]), quality=90)
print(f"{N} JPEGs {W}x{H}, {sum(os.path.getsize(os.path.join(tmp, f)) for f in names) / N / 1e3:.0f} KB each, host cores {os.cpu_count()}", flush=True)


def serial():
    for lo in range(0, N, B):
        part = names[lo:lo + B]
        yield torch.from_numpy(np.stack([np.asarray(Image.open(os.path.join(tmp, f)).convert("RGB")) for f in part])).to(dev)


def run(gen, consume=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u8 in gen:
        if consume is not None:
            consume(u8)
    torch.cuda.synchronize()
    return N / (time.perf_counter() - t0)


print(f"decode only: serial {run(serial()):8.0f} images/s", flush=True)
for export in ("raw", "arrow"):
    for w in (2, 4, 8, 15):
        print(f"decode only: loader export={export:5s} workers={w:2d} {run(u8 for _, _, u8 in ImageBatchLoader(tmp, names, B, dev, workers=w, export=export)):8.0f} images/s", flush=True)

ext = DinoV2Salad("vit_large").to(dev).to(torch.bfloat16).eval()
ext.backbone.fold_layerscale()
prep = ResizeNormalize(224, "bilinear", HALF_MEAN, HALF_STD, torch.bfloat16)
with torch.no_grad():
    step = lambda u8: ext(prep(u8))
    step(next(serial()))
    print(f"with ViT-L/14 + SALAD: serial {run(serial(), step):8.0f} images/s", flush=True)
    for export, w in (("raw", 8), ("arrow", 8), ("arrow", 15)):
        print(f"with ViT-L/14 + SALAD: loader export={export:5s} workers={w:2d} {run((u8 for _, _, u8 in ImageBatchLoader(tmp, names, B, dev, workers=w, export=export)), step):8.0f} images/s", flush=True)
    from vpr_amd.graphed import GraphedForward
    gext = GraphedForward(ext)
    gstep = lambda u8: gext(prep(u8))
    gstep(next(serial()))
    print(f"with ViT-L/14 + SALAD replayed from a HIP graph: serial {run(serial(), gstep):8.0f} images/s", flush=True)
    for export, w in (("raw", 8), ("arrow", 8), ("arrow", 12)):
        print(f"with ViT-L/14 + SALAD replayed from a HIP graph: loader export={export:5s} workers={w:2d} {run((u8 for _, _, u8 in ImageBatchLoader(tmp, names, B, dev, workers=w, export=export)), gstep):8.0f} images/s", flush=True)
for f in names:
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
