"""cProfile of the host side of one pipeline step (enqueue only): where the ~6 ms of Python / launch time go."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from vpr_amd.modules import DinoV2Salad, FusedGeoPoseHead
from vpr_amd.pipeline import VPRGeoPosePipeline
from vpr_amd.retrieval import ShardedGallery
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
for _ in range(3): pipe.step(images)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): pipe.step(images)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr).sort_stats("tottime")
st.print_stats(18)
