"""A/B: second-slot workgroups of the attention kernel started n us late (VPR_ATTN_VARIANT = 20 + n), ViT-L/14 shapes
(B = 64, T = 257, 16 heads).  us per call (median of 30), bit-identity against the default."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
B, T, H = 64, 257, 16
qkv = (torch.randn(B * T, 3 * H * 64, device=dev) * 0.5).to(torch.bfloat16)
def run():
    return ops.attention_qkv_split_bf16(qkv, B, T, T - 1, H) if hasattr(ops, "attention_qkv_split_bf16") else None
def timeit(n=30):
    for _ in range(5): run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return round(ts[len(ts) // 2] * 1e3, 1)
ref = run().clone()
res = {}
for rep in range(2):
    for n in (0, 2, 4, 6, 8, 10, 13, 16):
        _lib.tuning_set("VPR_ATTN_VARIANT", 20 + n if n else 0)
        res.setdefault(n, []).append(timeit())
        assert torch.equal(run(), ref)
_lib.tuning_set("VPR_ATTN_VARIANT", None)
print(json.dumps({"us_by_delay_us": res}))
