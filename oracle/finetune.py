"""CPU restatement of the head-only fine-tuning step — TEST ORACLE (numpy, float64 unless told otherwise).

What it restates, per batch, of dinov2salad/dinov2salad_finetuning.py:
  :37      preds = regressor(features)                  Linear(D,hidden) -> ReLU -> Linear(hidden,n_out)   (:28-32)
  :96,:121 loss = nn.MSELoss()(preds, targets)          mean over the B * n_out elements  (nn.HuberLoss(delta) of
           dinov2salad_finetuning_2.py:154 / swin_attempt_2.py:158 as the second loss kind)
  :123-125 optimizer.zero_grad(); loss.backward(); optimizer.step()
  :95      torch.optim.AdamW(model.parameters(), lr=1e-5)   defaults betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2
The backward pass is written out by hand (no autograd) and AdamW follows torch.optim's single-tensor update in its order of
operations: decoupled decay p *= 1 - lr*wd; m += (g - m)(1 - b1); v = b2 v + (1 - b2) g g; denom = sqrt(v)/sqrt(1 - b2^t) + eps;
p -= (lr / (1 - b1^t)) m / denom.
Pinned by tests/test_finetune_oracle_cpu.py against torch autograd + torch.optim.AdamW in float64 (the reference's own
arithmetic: its loop is these torch calls; the script itself trains at import and cannot be imported).
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np


class HeadState:
    """Parameters and AdamW moments of the two-layer head; `step` counts completed optimizer steps."""

    def __init__(self, W1, b1, W2, b2, dtype=np.float64):
        self.p = [np.array(a, dtype=dtype, copy=True) for a in (W1, b1, W2, b2)]
        self.m = [np.zeros_like(a) for a in self.p]
        self.v = [np.zeros_like(a) for a in self.p]
        self.step = 0

    @property
    def W1(self): return self.p[0]
    @property
    def b1(self): return self.p[1]
    @property
    def W2(self): return self.p[2]
    @property
    def b2(self): return self.p[3]


def forward(st: HeadState, x: np.ndarray):
    z = x @ st.W1.T + st.b1
    h = np.maximum(z, 0)
    return z, h, h @ st.W2.T + st.b2


def loss_and_grads(st: HeadState, x: np.ndarray, y: np.ndarray, loss: str = "mse", huber_delta: float = 1.0):
    """The loss (mean over every element) and its gradients w.r.t. (W1, b1, W2, b2).
    loss = "mse": nn.MSELoss (dinov2salad_finetuning.py:96); "huber": nn.HuberLoss(delta) (dinov2salad_finetuning_2.py:154,
    swin_transformer/swin_attempt_2.py:158): 0.5 d^2 where |d| < delta, delta (|d| - 0.5 delta) elsewhere."""
    x = np.asarray(x, dtype=st.W1.dtype)
    y = np.asarray(y, dtype=st.W1.dtype)
    z, h, o = forward(st, x)
    diff = o - y
    if loss == "mse":
        value = float(np.mean(diff * diff))
        do = 2.0 * diff / diff.size                  # d loss / d o
    elif loss == "huber":
        ad = np.abs(diff)
        inside = ad < huber_delta
        value = float(np.mean(np.where(inside, 0.5 * diff * diff, huber_delta * (ad - 0.5 * huber_delta))))
        do = np.where(inside, diff, huber_delta * np.sign(diff)) / diff.size
    else:
        raise ValueError(loss)
    loss = value
    gW2 = do.T @ h
    gb2 = do.sum(axis=0)
    dz = (do @ st.W2) * (h > 0)                      # ReLU backward: torch masks on the OUTPUT being > 0
    gW1 = dz.T @ x
    gb1 = dz.sum(axis=0)
    return loss, [gW1, gb1, gW2, gb2]


def adamw_update(st: HeadState, grads, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2) -> None:
    st.step += 1
    b1, b2 = betas
    bc1 = 1.0 - b1 ** st.step
    bc2 = 1.0 - b2 ** st.step
    for p, m, v, g in zip(st.p, st.m, st.v, grads):
        p *= 1.0 - lr * weight_decay
        m += (g - m) * (1.0 - b1)
        v *= b2
        v += (1.0 - b2) * g * g
        denom = np.sqrt(v) / np.sqrt(bc2) + eps
        p -= (lr / bc1) * (m / denom)


def train_step(st: HeadState, x, y, loss: str = "mse", huber_delta: float = 1.0, **opt) -> float:
    """One batch: forward, loss, backward, AdamW.  Returns the loss of the batch (before the update)."""
    value, grads = loss_and_grads(st, x, y, loss, huber_delta)
    adamw_update(st, grads, **opt)
    return value


def train_epoch(st: HeadState, X, Y, order, batch_size: int, **opt) -> float:
    """One pass over X[order] in batches of `batch_size` (the last one may be short, as DataLoader's default keeps it:
    dinov2salad_finetuning.py:89).  Returns the mean of the batch losses (:126-128)."""
    total, nb = 0.0, 0
    for lo in range(0, len(order), batch_size):
        idx = order[lo:lo + batch_size]
        total += train_step(st, X[idx], Y[idx], **opt)
        nb += 1
    return total / max(nb, 1)
