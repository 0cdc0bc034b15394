import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
M, C, Hd = 64 * 257, 1024, 4096
x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
W1 = torch.randn(Hd, C, device=dev, dtype=torch.bfloat16) * 0.03
b1 = torch.randn(Hd, device=dev, dtype=torch.bfloat16) * 0.1
def exact(): return F.gelu(F.linear(x, W1, b1))
def fused(): return torch._addmm_activation(b1, x, W1.t(), use_gelu=True)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import torch.cuda.tunable as tun
for tune in (False, True):
    tun.enable(tune); tun.tuning_enable(tune)
    if tune: tun.set_max_tuning_duration(30)
    print("tuned" if tune else "default", f"exact {timeit(exact):.1f} us   fused {timeit(fused):.1f} us")
a, b = exact().float(), fused().float()
ref = F.gelu(F.linear(x.float(), W1.float(), b1.float()))
print("exact-vs-f32", (a - ref).abs().max().item(), "fused-vs-f32", (b - ref).abs().max().item(), "fused-vs-exact", (a - b).abs().max().item())
