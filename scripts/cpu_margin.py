"""How far ahead of the GPU does the host run?  Enqueue time of K pipeline steps (no sync) vs their GPU time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from vpr_amd.modules import DinoV2Salad, FusedGeoPoseHead
from vpr_amd.pipeline import VPRGeoPosePipeline
from vpr_amd.retrieval import ShardedGallery
from vpr_amd.backbone import gemm_autotune
dev = torch.device("cuda:0")
torch.manual_seed(0)
ext = DinoV2Salad("vit_large").eval().to(dev).to(torch.bfloat16)
ext.backbone.fold_layerscale()
pos = nn.Sequential(nn.Linear(8448, 512), nn.ReLU(), nn.Linear(512, 2)).to(dev)
ang = nn.Sequential(nn.Linear(8448, 512), nn.ReLU(), nn.Linear(512, 2)).to(dev)
head = FusedGeoPoseHead(pos, ang, normalize=True)
shard = torch.nn.functional.normalize(torch.randn(20000, 8448, device=dev), dim=1).to(torch.bfloat16)
pipe = VPRGeoPosePipeline(ext, head, ShardedGallery(shard, 20000, 0, 1), 10)
images = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16)
gemm_autotune(True, tuning=True)
for _ in range(3): pipe.step(images)
torch.cuda.synchronize(); gemm_autotune(True, tuning=False)
for _ in range(2): pipe.step(images)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): pipe.step(images)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step   total {1e3*(t2-t0)/K:.2f} ms/step")
