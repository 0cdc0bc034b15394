"""oracle/finetune.py pinned against the reference's own arithmetic: its training step is `loss_fn(model(x), y)`,
`loss.backward()`, `torch.optim.AdamW(lr=1e-5).step()` (dinov2salad/dinov2salad_finetuning.py:95-96,119-125) — torch calls
that run here on the CPU.  The hand-written backward pass and the AdamW restatement must reproduce them in float64 to
rounding, over several steps, ragged last batch included."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import finetune as oft


def _torch_head(D, hidden, n_out, seed, dtype):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(D, hidden), nn.ReLU(), nn.Linear(hidden, n_out)).to(dtype)     # the reference's regressor (:28-32)


def _state_from(head, dtype=np.float64):
    return oft.HeadState(head[0].weight.detach().numpy(), head[0].bias.detach().numpy(),
                         head[2].weight.detach().numpy(), head[2].bias.detach().numpy(), dtype=dtype)


@pytest.mark.parametrize("D,hidden,n_out,N,bs,lr", [(96, 32, 2, 37, 16, 1e-5), (64, 64, 4, 50, 7, 1e-2), (32, 32, 1, 9, 16, 1e-3)])
def test_oracle_step_equals_torch_autograd_and_adamw_f64(D, hidden, n_out, N, bs, lr):
    head = _torch_head(D, hidden, n_out, 0, torch.float64)
    st = _state_from(head)
    g = torch.Generator().manual_seed(1)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, dtype=torch.float64), dim=1)
    Y = torch.randn(N, n_out, generator=g, dtype=torch.float64)
    opt = torch.optim.AdamW(head.parameters(), lr=lr)
    loss_fn = nn.MSELoss()
    for epoch in range(3):
        order = torch.randperm(N, generator=g)
        for lo in range(0, N, bs):
            idx = order[lo:lo + bs]
            loss = loss_fn(head(X[idx]), Y[idx])
            opt.zero_grad()
            loss.backward()
            opt.step()
            lo_oracle = oft.train_step(st, X[idx].numpy(), Y[idx].numpy(), lr=lr)
            assert abs(lo_oracle - float(loss.detach())) <= 1e-12 * max(1.0, abs(float(loss.detach())))
    for p_t, p_o in zip((head[0].weight, head[0].bias, head[2].weight, head[2].bias), st.p):
        assert np.abs(p_t.detach().numpy() - p_o).max() <= 1e-13
    for i, p_t in enumerate((head[0].weight, head[0].bias, head[2].weight, head[2].bias)):
        assert np.abs(opt.state[p_t]["exp_avg"].numpy() - st.m[i]).max() <= 1e-14
        assert np.abs(opt.state[p_t]["exp_avg_sq"].numpy() - st.v[i]).max() <= 1e-14
    assert int(opt.state[head[0].weight]["step"]) == st.step


def test_oracle_gradients_against_finite_differences():
    head = _torch_head(24, 32, 3, 3, torch.float64)
    st = _state_from(head)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal((5, 24)), rng.standard_normal((5, 3))
    loss, grads = oft.loss_and_grads(st, x, y)
    for k, (p, g) in enumerate(zip(st.p, grads)):
        flat = p.reshape(-1)
        for e in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            old = flat[e]
            flat[e] = old + 1e-6
            lp, _ = oft.loss_and_grads(st, x, y)
            flat[e] = old - 1e-6
            lm, _ = oft.loss_and_grads(st, x, y)
            flat[e] = old
            assert abs((lp - lm) / 2e-6 - g.reshape(-1)[e]) <= 1e-7, (k, e)


def test_oracle_epoch_mean_loss_and_dead_units():
    """train_epoch = mean of the batch losses (:126-128); a hidden unit that never fires gets zero gradient, so AdamW only
    applies its decoupled decay to that unit's weights (m = v = 0 -> 0 / (0 + eps))."""
    head = _torch_head(16, 32, 2, 5, torch.float64)
    with torch.no_grad():
        head[0].bias[:4] = -100.0
    st = _state_from(head)
    w_dead = st.W1[:4].copy()
    rng = np.random.default_rng(1)
    X, Y = rng.standard_normal((20, 16)), rng.standard_normal((20, 2))
    losses = []
    st2 = _state_from(head)
    for lo in range(0, 20, 8):
        losses.append(oft.train_step(st2, X[lo:lo + 8], Y[lo:lo + 8], lr=1e-3))
    mean = oft.train_epoch(st, X, Y, np.arange(20), 8, lr=1e-3)
    assert abs(mean - np.mean(losses)) < 1e-15
    assert np.allclose(st.W1[:4], w_dead * (1 - 1e-3 * 1e-2) ** 3, rtol=1e-14, atol=0)
    assert np.all(st.m[0][:4] == 0) and np.all(st.v[0][:4] == 0)


@pytest.mark.parametrize("delta", [1.0, 0.25])
def test_oracle_huber_step_equals_torch_huberloss_f64(delta):
    """loss="huber" = nn.HuberLoss(delta) (dinov2salad_finetuning_2.py:154, swin_attempt_2.py:158), targets scaled so that
    residuals fall on both sides of delta."""
    D, hidden, n_out, N, bs, lr = 48, 32, 2, 30, 8, 1e-3
    head = _torch_head(D, hidden, n_out, 7, torch.float64)
    st = _state_from(head)
    g = torch.Generator().manual_seed(8)
    X = torch.randn(N, D, generator=g, dtype=torch.float64)
    Y = torch.randn(N, n_out, generator=g, dtype=torch.float64) * 1.5
    opt = torch.optim.AdamW(head.parameters(), lr=lr, weight_decay=0.05)
    loss_fn = nn.HuberLoss(delta=delta)
    outside = 0
    for lo in list(range(0, N, bs)) * 3:
        out = head(X[lo:lo + bs])
        outside += int(((out - Y[lo:lo + bs]).abs() >= delta).sum())
        loss = loss_fn(out, Y[lo:lo + bs])
        opt.zero_grad()
        loss.backward()
        opt.step()
        lo_oracle = oft.train_step(st, X[lo:lo + bs].numpy(), Y[lo:lo + bs].numpy(), loss="huber", huber_delta=delta, lr=lr, weight_decay=0.05)
        assert abs(lo_oracle - float(loss.detach())) <= 1e-12
    assert outside > 10                                   # both branches of the loss were exercised
    for p_t, p_o in zip((head[0].weight, head[0].bias, head[2].weight, head[2].bias), st.p):
        assert np.abs(p_t.detach().numpy() - p_o).max() <= 1e-13
