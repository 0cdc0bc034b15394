"""Stress of the arrival-counter hand-over inside vpr_head_train_step (head_mid_kernel's last workgroup reads what the
others wrote): the same 100-epoch run twice (39 900 steps each at the reference's shapes and dataset size), final parameters,
moments and every batch loss must agree bit for bit — a lost or early hand-over would show up as a difference."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops
dev = torch.device("cuda:0")
D, hidden, n_out, N, bs, epochs = 8448, 512, 2, 6378, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device=dev).manual_seed(0)
X = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
Y = torch.randn(N, n_out, device=dev, generator=g)
W0 = [torch.randn(hidden, D, device=dev, generator=g) * 0.01, torch.zeros(hidden, device=dev),
      torch.randn(n_out, hidden, device=dev, generator=g) * 0.04, torch.zeros(n_out, device=dev)]
orders = [torch.randperm(N, device=dev, generator=g).to(torch.int32) for _ in range(epochs)]


def run():
    W = [w.clone() for w in W0]
    m, v = ops.head_train_state(W[0], W[2])
    step, losses = 1, []
    t0 = time.perf_counter()
    for o in orders:
        l = ops.head_train_epoch(X, Y, o, bs, *W, m, v, step, lr=1e-4)
        step += l.numel()
        losses.append(l)
    torch.cuda.synchronize()
    return W, m, v, torch.cat(losses), time.perf_counter() - t0


a = run()
b = run()
same = all(torch.equal(p, q) for p, q in zip(a[0], b[0])) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
print(f"{a[3].numel()} steps per run, {a[4]:.2f} s / {b[4]:.2f} s ({a[4] / a[3].numel() * 1e6:.1f} us per step), first / last epoch loss "
      f"{a[3][:399].mean().item():.4f} / {a[3][-399:].mean().item():.4f}, finite: {bool(torch.isfinite(a[3]).all())}, bitwise identical: {same}")
assert same and torch.isfinite(a[3]).all()
