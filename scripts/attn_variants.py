import torch, torch.nn.functional as F, time
dev = torch.device("cuda:0")
B, T, C, H = 64, 257, 1024, 16
d = C // H
x = torch.randn(B, T, 3 * C, device=dev, dtype=torch.bfloat16)

def v_current(qkv):
    q = qkv.view(B, T, 3, H, d).permute(2, 0, 3, 1, 4)
    a = F.scaled_dot_product_attention(q[0], q[1], q[2])
    return a.transpose(1, 2).reshape(B, T, C)

def v_contig(qkv):
    q = qkv.view(B, T, 3, H, d).permute(2, 0, 3, 1, 4).contiguous()
    a = F.scaled_dot_product_attention(q[0], q[1], q[2])
    return a.transpose(1, 2).reshape(B, T, C)

def v_math_free(qkv):
    q, k, v = qkv.view(B, T, 3, H, d).unbind(2)
    a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return a.transpose(1, 2).reshape(B, T, C)

def timeit(fn, n=20):
    for _ in range(3): fn(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, fn in [("current", v_current), ("contig", v_contig), ("unbind", v_math_free)]:
    print(name, f"{timeit(fn):.1f} us")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    v_current(x); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60))
