"""Head-only fine-tuning step on cached descriptors: vpr_head_train_step (HIP, three launches) beside
PyTorch autograd + torch.optim.AdamW on the same GPU (what the reference's loop runs per batch once the backbone is taken
out: dinov2salad_finetuning.py:119-125), us per step.  Shapes: the reference's head (8448 -> 512 -> 2, batch 16) and the
widest supported batch.  Also prints the roofline reading: algorithmic bytes = 7 * hidden * D * 4 + 2 * B * D * 4."""
import json, os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import ops

dev = torch.device("cuda:0")


def time_steps(fn, n):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(n):
        fn(10 + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, (time.perf_counter() - t0) * 1e6 / n


rows = []
for (D, hidden, n_out, B) in [(8448, 512, 2, 16), (8448, 512, 2, 64), (8448, 1024, 4, 64)]:
    torch.manual_seed(0)
    N = 4096
    X = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=1)
    Y = torch.randn(N, n_out, device=dev)
    perm = torch.randperm(N, device=dev).to(torch.int32)
    nb = N // B
    head = nn.Sequential(nn.Linear(D, hidden), nn.ReLU(), nn.Linear(hidden, n_out)).to(dev)
    W = [p.detach().clone() for p in (head[0].weight, head[0].bias, head[2].weight, head[2].bias)]
    m, v = ops.head_train_state(W[0], W[2])
    losses = torch.zeros(1, device=dev)
    cnt = [0]

    def hip_step(i):
        cnt[0] += 1
        b = i % nb
        ops.head_train_step(X, Y, perm[b * B:(b + 1) * B], *W, m, v, cnt[0], loss_out=losses)

    opt = torch.optim.AdamW(head.parameters(), lr=1e-5)
    permL = perm.long()

    def torch_step(i):
        b = i % nb
        idx = permL[b * B:(b + 1) * B]
        loss = nn.functional.mse_loss(head(X[idx]), Y[idx])
        opt.zero_grad()
        loss.backward()
        opt.step()

    opt_f = torch.optim.AdamW(head.parameters(), lr=1e-5, fused=True)

    def torch_fused_step(i):
        b = i % nb
        idx = permL[b * B:(b + 1) * B]
        loss = nn.functional.mse_loss(head(X[idx]), Y[idx])
        opt_f.zero_grad()
        loss.backward()
        opt_f.step()

    def hip_epoch(i):          # 64 batches per library call
        b = (i * 64) % (nb - 64 + 1) if nb > 64 else 0
        k = min(64, nb)
        ops.head_train_epoch(X, Y, perm[b * B:(b + k) * B], B, *W, m, v, cnt[0] + 1)
        cnt[0] += k

    g_hip, w_hip = time_steps(hip_step, 300)
    g_ep, w_ep = time_steps(hip_epoch, 12)
    g_ep, w_ep = g_ep / min(64, nb), w_ep / min(64, nb)
    g_t, w_t = time_steps(torch_step, 100)
    g_tf, w_tf = time_steps(torch_fused_step, 100)
    alg = 7 * hidden * D * 4 + 2 * B * D * 4
    row = {"shape": f"D={D} hidden={hidden} n_out={n_out} B={B}", "hip_us_per_step": round(g_hip, 2), "hip_host_us_per_step": round(w_hip, 2), "hip_epoch_call_us_per_step": round(g_ep, 2), "hip_epoch_call_host_us_per_step": round(w_ep, 2),
           "torch_autograd_adamw_us_per_step": round(g_t, 2), "torch_autograd_fused_adamw_us_per_step": round(g_tf, 2),
           "speedup_vs_torch": round(min(g_t, g_tf) / g_ep, 2), "algorithmic_bytes": alg,
           "achieved_GBps": round(alg / g_ep / 1e3, 1), "frac_of_8TBps": round(alg / g_ep / 1e3 / 8000, 3)}
    rows.append(row)
    print(json.dumps(row), flush=True)
