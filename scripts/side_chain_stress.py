"""Race screen for the cls side stream: many forwards of the full-size backbone must equal the single-stream result."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd.modules import DinoV2Salad
dev = torch.device("cuda:0")
torch.manual_seed(0)
ext = DinoV2Salad("vit_large").to(dev).to(torch.bfloat16).eval()
ext.backbone.fold_layerscale()
x = torch.randn(64, 3, 224, 224, device=dev, dtype=torch.bfloat16)
ext.backbone.cls_side_chain = False
ref = ext(x).clone()
ext.backbone.cls_side_chain = True
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    out = ext(x)
    if not torch.equal(out, ref):
        bad += 1
print("iterations with a mismatch:", bad)
sys.exit(1 if bad else 0)
