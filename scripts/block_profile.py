import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd.backbone import Block
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
blk = Block(1024, 16).to(dev).to(torch.bfloat16).eval()
blk.fold_layerscale()
x = torch.randn(64, 257, 1024, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(3): blk(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): blk(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=90))
