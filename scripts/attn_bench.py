import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
B, T, H = 64, 257, 16
qkv = torch.randn(B, T, 3 * H * 64, device=dev, dtype=torch.bfloat16)
def sdpa():
    q = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    return F.scaled_dot_product_attention(q[0], q[1], q[2]).transpose(1, 2).reshape(B, T, H * 64)
def mine():
    return ops.attention_qkv_bf16(qkv, H)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for r in range(3):
    line = f"sdpa {timeit(sdpa):.1f} us"
    for v in (0, 13, 14, 12):      # 0 = default; 13 staging only, 14 + QK^T, 12 + softmax (ablations, wrong results)
        _lib.tuning_set("VPR_ATTN_VARIANT", int(v))
        line += f"   v{v} {timeit(mine):.1f} us"
    print(line)
