"""Workload for the two PMC passes on the head-only training step (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs, program
directly after `--`): 4 epoch-calls of 64 batches at the reference's shapes (D = 8448, hidden = 512, n_out = 2, B = 16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
D, hidden, n_out, B, N = 8448, 512, 2, 16, 1024
g = torch.Generator(device=dev).manual_seed(0)
X = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
Y = torch.randn(N, n_out, device=dev, generator=g)
W = [torch.randn(hidden, D, device=dev, generator=g) * 0.01, torch.zeros(hidden, device=dev),
     torch.randn(n_out, hidden, device=dev, generator=g) * 0.04, torch.zeros(n_out, device=dev)]
m, v = ops.head_train_state(W[0], W[2])
order = torch.randperm(N, device=dev, generator=g).to(torch.int32)
step = 1
for _ in range(4):
    ops.head_train_epoch(X, Y, order, B, *W, m, v, step)
    step += N // B
torch.cuda.synchronize()
