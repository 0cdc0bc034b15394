import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
C, B = 384, 2
g = torch.Generator(device=dev).manual_seed(0)
tokens = torch.randn(B, 257, C, device=dev, generator=g).to(torch.bfloat16)
r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02)
w = ops.SaladWeights(w1_sc=r(1024, C).bfloat16(), b1_sc=r(1024), w2_s=r(64, 512).bfloat16(), b2_s=r(64),
                     w2_c=r(128, 512).bfloat16(), b2_c=r(128), w1_t=r(512, C).bfloat16(), b1_t=r(512),
                     w2_t=r(256, 512).bfloat16(), b2_t=r(256), dustbin=1.0)
x = tokens[:, 1:, :].reshape(B * 256, C).contiguous()
for name, fn in [
    ("gemm l1", lambda: ops.gemm_nt_bf16(x, w.w1_sc, w.b1_sc, True, torch.bfloat16)),
    ("gemm tok", lambda: ops.gemm_nt_bf16(tokens[:, 0, :].contiguous(), w.w1_t, w.b1_t, True, torch.bfloat16)),
    ("salad", lambda: ops.salad_aggregate(tokens, w)),
]:
    try:
        fn(); torch.cuda.synchronize(); print(name, "ok")
    except Exception as e:
        print(name, "FAILED", e)
# now after a torch SDPA call
import torch.nn.functional as F
q = torch.randn(2, 6, 257, 64, device=dev, dtype=torch.bfloat16)
F.scaled_dot_product_attention(q, q, q); torch.cuda.synchronize()
try:
    ops.salad_aggregate(tokens, w); torch.cuda.synchronize(); print("salad after sdpa ok")
except Exception as e:
    print("salad after sdpa FAILED", e)
