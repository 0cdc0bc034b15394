"""vpr_head_train_step (HIP: forward, MSELoss, backward, AdamW of the regression head on cached descriptors) against
oracle/finetune.py (numpy f64, pinned to torch autograd + torch.optim.AdamW by tests/test_finetune_oracle_cpu.py).
SURVEY.md §8f-4; reference loop dinov2salad/dinov2salad_finetuning.py:95-96,119-125.

Tolerance.  AdamW normalises the gradient, so a weight moves by about lr per step whatever the gradient's size, and an f32
rounding error e in a gradient g moves the update by at most lr * e / (|g| + eps-ish): bounded by a few per cent of lr per
step for the rare elements whose gradient cancels to almost nothing, ~1e-6 * lr for the rest.  PARAM_TOL = 0.05 * lr * steps
(absolute, per element); losses are compared to 2e-5 relative (f32 sums of B * n_out squares)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import finetune as oft
from vpr_amd import ops
from vpr_amd.finetune import finetune_head
from vpr_amd.modules import DINOv2RegressionModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _head(D, hidden, n_out, seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(D, hidden), nn.ReLU(), nn.Linear(hidden, n_out))     # the reference's init (:28-32)


def _data(N, D, n_out, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)           # unit-norm rows, like SALAD descriptors
    return X, torch.randn(N, n_out, generator=g)


def _gpu_params(head):
    return [p.detach().clone().to(DEV).contiguous() for p in (head[0].weight, head[0].bias, head[2].weight, head[2].bias)]


def _run_hip(head, X, Y, batches, lr, **hyper):
    W1, b1, W2, b2 = _gpu_params(head)
    m, v = ops.head_train_state(W1, W2)
    Xg, Yg = X.to(DEV), Y.to(DEV)
    losses = torch.zeros(len(batches), device=DEV)
    for i, idx in enumerate(batches):
        ops.head_train_step(Xg, Yg, None if idx is None else torch.as_tensor(idx, dtype=torch.int32, device=DEV), W1, b1, W2, b2,
                            m, v, i + 1, lr=lr, loss_out=losses[i:i + 1], **hyper)
    torch.cuda.synchronize()
    return [W1, b1, W2, b2], m, v, losses.cpu().numpy()


def _run_oracle(head, X, Y, batches, lr, **hyper):
    st = oft.HeadState(*(p.detach().numpy() for p in (head[0].weight, head[0].bias, head[2].weight, head[2].bias)))
    Xn, Yn = X.numpy().astype(np.float64), Y.numpy().astype(np.float64)
    losses = [oft.train_step(st, Xn[idx] if idx is not None else Xn, Yn[idx] if idx is not None else Yn, lr=lr, **hyper)
              for idx in batches]
    return st, np.array(losses)


def _batches(N, bs, steps, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < steps:
        perm = rng.permutation(N)
        out += [perm[lo:lo + bs] for lo in range(0, N, bs)]
    return out[:steps]


SHAPES = [
    # D, hidden, n_out, N, batch, steps, lr
    (8448, 512, 2, 80, 16, 12, 1e-5),       # the reference's head and batch size (:29-31, :89)
    (8448, 512, 2, 70, 16, 10, 1e-3),       # ragged last batch (70 = 4 * 16 + 6), larger steps
    (8448, 1024, 4, 64, 64, 4, 1e-4),       # fused (lat, lon, sin, cos) head width, largest batch
    (64, 32, 2, 23, 5, 12, 1e-3),
    (256, 64, 8, 40, 33, 6, 1e-2),
    (16, 32, 1, 7, 1, 9, 1e-3),             # one row per batch, one output
    (1024, 96, 3, 50, 17, 8, 1e-3),         # hidden not a power of two, n_out odd
]


@pytest.mark.parametrize("D,hidden,n_out,N,bs,steps,lr", SHAPES)
def test_head_train_step_matches_oracle(D, hidden, n_out, N, bs, steps, lr):
    head = _head(D, hidden, n_out, 0)
    X, Y = _data(N, D, n_out, 1)
    batches = _batches(N, bs, steps, 2)
    params, m, v, losses = _run_hip(head, X, Y, batches, lr)
    st, ref_losses = _run_oracle(head, X, Y, batches, lr)
    rel = np.abs(losses - ref_losses) / np.maximum(np.abs(ref_losses), 1e-12)
    assert rel.max() <= 2e-5, (rel.max(), losses, ref_losses)
    tol = 0.05 * lr * steps
    worst = 0.0
    for p, ref in zip(params, st.p):
        worst = max(worst, float(np.abs(p.cpu().numpy().astype(np.float64) - ref).max()))
    ms = ops.head_train_state_views(m, params[0], params[2])
    vs = ops.head_train_state_views(v, params[0], params[2])
    m_err = max(float(np.abs(a.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-30)) for a, r in zip(ms, st.m))
    v_err = max(float(np.abs(a.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-30)) for a, r in zip(vs, st.v))
    print(f"\n[head_train D={D} hidden={hidden} n_out={n_out} B={bs} steps={steps} lr={lr}] loss rel {rel.max():.1e}; "
          f"params |max| {worst:.2e} (= {worst / lr:.1e} lr, tol {tol:.1e}); moments rel m {m_err:.1e} v {v_err:.1e}")
    assert worst <= tol
    assert m_err <= 1e-4 and v_err <= 1e-4


def test_head_train_step_without_index_and_with_dead_units():
    """idx = NULL takes rows 0..B-1; hidden units that never fire get exactly zero gradient: their weights only decay and
    their moments stay exactly zero."""
    D, hidden, n_out, B = 128, 64, 2, 12
    head = _head(D, hidden, n_out, 3)
    with torch.no_grad():
        head[0].bias[:5] = -50.0
    X, Y = _data(B, D, n_out, 4)
    params, m, v, losses = _run_hip(head, X, Y, [None, None, None], 1e-3)
    st, ref_losses = _run_oracle(head, X, Y, [None, None, None], 1e-3)
    assert np.abs(losses - ref_losses).max() <= 2e-5 * np.abs(ref_losses).max()
    for p, ref in zip(params, st.p):
        assert np.abs(p.cpu().numpy() - ref).max() <= 0.05 * 1e-3 * 3
    mW1 = ops.head_train_state_views(m, params[0], params[2])[0]
    vW1 = ops.head_train_state_views(v, params[0], params[2])[0]
    assert torch.all(mW1[:5] == 0) and torch.all(vW1[:5] == 0)
    w0 = head[0].weight.detach()[:5].to(DEV)
    assert torch.allclose(params[0][:5], w0 * (1 - 1e-3 * 1e-2) ** 3, rtol=1e-6, atol=0)


def test_head_train_step_is_bitwise_reproducible_and_index_order_is_honoured():
    D, hidden, n_out, N, bs = 8448, 512, 2, 48, 16
    head = _head(D, hidden, n_out, 5)
    X, Y = _data(N, D, n_out, 6)
    batches = _batches(N, bs, 6, 7)
    a = _run_hip(head, X, Y, batches, 1e-4)
    b = _run_hip(head, X, Y, batches, 1e-4)
    for p, q in zip(a[0], b[0]):
        assert torch.equal(p, q)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    # gathering through idx == handing over the gathered rows (same rows, same order => same bits)
    W = _gpu_params(head)
    m, v = ops.head_train_state(W[0], W[2])
    idx = torch.as_tensor(batches[0], dtype=torch.long)
    ops.head_train_step(X[idx].contiguous().to(DEV), Y[idx].contiguous().to(DEV), None, *W, m, v, 1, lr=1e-4)
    W2_ = _gpu_params(head)
    m2, v2 = ops.head_train_state(W2_[0], W2_[2])
    ops.head_train_step(X.to(DEV), Y.to(DEV), idx.to(DEV, torch.int32), *W2_, m2, v2, 1, lr=1e-4)
    for p, q in zip(W, W2_):
        assert torch.equal(p, q)


def test_head_train_step_follows_torch_adamw_on_the_gpu():
    """Same batches through PyTorch autograd + torch.optim.AdamW in f32 on the GPU (what the reference runs)."""
    D, hidden, n_out, N, bs, steps, lr = 8448, 512, 2, 64, 16, 8, 1e-5
    head = _head(D, hidden, n_out, 8)
    X, Y = _data(N, D, n_out, 9)
    batches = _batches(N, bs, steps, 10)
    params, _, _, losses = _run_hip(head, X, Y, batches, lr)
    th = _head(D, hidden, n_out, 8).to(DEV)
    opt = torch.optim.AdamW(th.parameters(), lr=lr)
    Xg, Yg = X.to(DEV), Y.to(DEV)
    tl = []
    for idx in batches:
        i = torch.as_tensor(idx, device=DEV)
        loss = nn.functional.mse_loss(th(Xg[i]), Yg[i])
        opt.zero_grad()
        loss.backward()
        opt.step()
        tl.append(float(loss.detach()))
    assert np.abs(losses - np.array(tl)).max() <= 2e-5 * max(tl)
    worst = max(float((p - q.detach()).abs().max()) for p, q in zip(params, (th[0].weight, th[0].bias, th[2].weight, th[2].bias)))
    print(f"\n[head_train vs torch GPU f32] params |max| {worst:.2e} (= {worst / lr:.1e} lr)")
    assert worst <= 0.05 * lr * steps


def test_head_train_step_refuses_bad_arguments():
    head = _head(64, 32, 2, 0)
    W = _gpu_params(head)
    m, v = ops.head_train_state(W[0], W[2])
    X, Y = (t.to(DEV) for t in _data(70, 64, 2, 0))
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.head_train_step(X, Y, None, *W, m, v, 1)                        # B = 70 > 64
    with pytest.raises(RuntimeError, match="moment buffers"):
        ops.head_train_step(X[:8].contiguous(), Y[:8].contiguous(), None, *W, m[:-1], v[:-1], 1)
    with pytest.raises(RuntimeError, match="VPR_ERR_INVALID_ARG|invalid"):
        ops.head_train_step(X[:8].contiguous(), Y[:8].contiguous(), None, *W, m, v, 0)      # steps count from 1
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.head_train_step(X[:8].cpu(), Y[:8].contiguous(), None, *W, m, v, 1)


def test_finetune_head_hip_engine_equals_torch_engine_and_writes_the_reference_checkpoint(tmp_path):
    """finetune_head on cached descriptors: HIP engine vs PyTorch-autograd engine — same batches, same history, same
    weights to the AdamW tolerance; the checkpoint carries model + optimizer state in the reference's dict (:130-135) and
    the validation report (HIP pose head) sees the updated weights (version counters bumped by the training step)."""
    torch.manual_seed(0)
    N, epochs, bs, lr = 40, 3, 16, 1e-4
    X, _ = _data(N, 8448, 2, 11)
    labels = np.stack([219658.0 + 900 * np.random.default_rng(0).standard_normal(N), 143506.0 + 1100 * np.random.default_rng(1).standard_normal(N)], 1)
    Xg = X.to(DEV)
    runs = {}
    for engine in ("hip", "torch"):
        torch.manual_seed(1)
        model = DINOv2RegressionModel(nn.Identity()).to(DEV)
        out = finetune_head(model, Xg, labels, epochs=epochs, batch_size=bs, lr=lr, save_dir=str(tmp_path / engine),
                            val=(Xg[:8], labels[:8]), seed=3, log=lambda s: None, engine=engine)
        runs[engine] = (model, out)
    (mh, oh), (mt, ot) = runs["hip"], runs["torch"]
    assert oh["engine"] == "hip" and ot["engine"] == "torch"
    steps = epochs * ((N + bs - 1) // bs)
    for a, b in zip(mh.regressor.parameters(), mt.regressor.parameters()):
        assert float((a - b).abs().max()) <= 0.05 * lr * steps
    for ra, rb in zip(oh["history"], ot["history"]):
        assert abs(ra["train_loss"] - rb["train_loss"]) <= 2e-5 * abs(rb["train_loss"])
        assert abs(ra["val_mae"] - rb["val_mae"]) <= 1e-3 * abs(rb["val_mae"]) + 1e-2
    assert oh["history"][-1]["val_mae"] != oh["history"][0]["val_mae"]          # the report follows the in-place updates
    ck = torch.load(tmp_path / "hip" / f"checkpoint_{epochs - 1}_.pth", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    st = ck["optimizer_state_dict"]["state"]
    assert len(st) == 4 and all(float(s["step"]) == steps for s in st.values())
    ckt = torch.load(tmp_path / "torch" / f"checkpoint_{epochs - 1}_.pth", weights_only=True)
    for k in st:
        ref = ckt["optimizer_state_dict"]["state"][k]
        assert st[k]["exp_avg"].shape == ref["exp_avg"].shape
        scale = float(ref["exp_avg"].abs().max())
        assert float((st[k]["exp_avg"].cpu() - ref["exp_avg"].cpu()).abs().max()) <= 1e-4 * scale


def test_head_train_epoch_equals_the_same_steps_one_by_one():
    """vpr_head_train_epoch == vpr_head_train_step per batch (same launches): bitwise, ragged last batch included."""
    D, hidden, n_out, N, bs = 8448, 512, 2, 70, 16
    head = _head(D, hidden, n_out, 12)
    X, Y = _data(N, D, n_out, 13)
    order = np.random.default_rng(14).permutation(N)
    batches = [order[lo:lo + bs] for lo in range(0, N, bs)]
    params, m, v, losses = _run_hip(head, X, Y, batches, 1e-4)
    W = _gpu_params(head)
    m2, v2 = ops.head_train_state(W[0], W[2])
    ver = W[0]._version
    l2 = ops.head_train_epoch(X.to(DEV), Y.to(DEV), torch.as_tensor(order, dtype=torch.int32, device=DEV), bs, *W, m2, v2, 1, lr=1e-4)
    assert W[0]._version > ver
    for p, q in zip(params, W):
        assert torch.equal(p, q)
    assert torch.equal(m, m2) and torch.equal(v, v2) and np.array_equal(losses, l2.cpu().numpy())
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.head_train_epoch(X.to(DEV), Y.to(DEV), torch.as_tensor(order, dtype=torch.int32, device=DEV), 65, *W, m2, v2, 1)


@pytest.mark.parametrize("D,hidden,n_out,N,bs,steps,lr,delta", [(8448, 512, 2, 64, 16, 8, 1e-4, 1.0), (256, 64, 4, 40, 33, 6, 1e-3, 0.25)])
def test_head_train_step_huber_loss_matches_oracle(D, hidden, n_out, N, bs, steps, lr, delta):
    """loss="huber" (nn.HuberLoss(delta): dinov2salad_finetuning_2.py:154) with residuals on both sides of delta, and a weight
    decay other than the default."""
    head = _head(D, hidden, n_out, 20)
    X, Y = _data(N, D, n_out, 21)
    Y = Y * 1.5
    batches = _batches(N, bs, steps, 22)
    hyper = dict(loss="huber", huber_delta=delta, weight_decay=0.05)
    params, m, v, losses = _run_hip(head, X, Y, batches, lr, **hyper)
    st, ref_losses = _run_oracle(head, X, Y, batches, lr, **hyper)
    z, h, o = oft.forward(oft.HeadState(*(p.detach().numpy() for p in (head[0].weight, head[0].bias, head[2].weight, head[2].bias))),
                          X.numpy().astype(np.float64))
    frac_outside = float((np.abs(o - Y.numpy()) >= delta).mean())
    assert 0.05 < frac_outside < 0.95                     # both branches of the loss are in play
    rel = np.abs(losses - ref_losses) / np.abs(ref_losses)
    worst = max(float(np.abs(p.cpu().numpy().astype(np.float64) - r).max()) for p, r in zip(params, st.p))
    print(f"\n[head_train huber delta={delta} D={D}] outside {frac_outside:.2f}; loss rel {rel.max():.1e}; params {worst / lr:.1e} lr")
    assert rel.max() <= 2e-5 and worst <= 0.05 * lr * steps
    with pytest.raises(RuntimeError, match="mse.*huber|loss must be"):
        ops.head_train_step(X.to(DEV), Y.to(DEV), None, *params, m, v, 1, loss="l1")


def test_finetune_head_huber_with_a_learning_rate_schedule_hip_equals_torch():
    """finetune_head(loss="huber", weight_decay, lr_schedule): the HIP engine reads the optimizer's learning rate at every pass,
    so a host-side schedule (ReduceLROnPlateau-style, dinov2salad_finetuning_2.py:155,236) steers both engines alike."""
    N, epochs, bs, lr = 40, 3, 16, 2e-4
    X, _ = _data(N, 8448, 2, 31)
    rng = np.random.default_rng(5)
    labels = np.stack([219658.0 + 900 * rng.standard_normal(N), 143506.0 + 1100 * rng.standard_normal(N)], 1)
    Xg = X.to(DEV)
    sched = lambda epoch, history: lr * (0.5 ** epoch)
    runs = {}
    for engine in ("hip", "torch"):
        torch.manual_seed(2)
        model = DINOv2RegressionModel(nn.Identity()).to(DEV)
        out = finetune_head(model, Xg, labels, epochs=epochs, batch_size=bs, lr=lr, seed=4, log=lambda s: None, engine=engine,
                            loss="huber", huber_delta=0.5, weight_decay=0.05, lr_schedule=sched)
        runs[engine] = (model, out)
        assert out["optimizer"].param_groups[0]["lr"] == lr * 0.25
    steps = epochs * ((N + bs - 1) // bs)
    for a, b in zip(runs["hip"][0].regressor.parameters(), runs["torch"][0].regressor.parameters()):
        assert float((a - b).abs().max()) <= 0.05 * lr * steps
    for ra, rb in zip(runs["hip"][1]["history"], runs["torch"][1]["history"]):
        assert abs(ra["train_loss"] - rb["train_loss"]) <= 2e-5 * abs(rb["train_loss"])


def test_head_train_reproduces_the_reference_training_run():
    """vpr_head_train_epoch against tests/golden/head_finetune.json — losses and final weights of the reference's own model
    class trained by its own optimizer / loss calls (generated by tests/golden/make_golden.py from the imported reference)."""
    import json, os
    from test_oracle_golden import check_finetune_against_golden, regen_finetune
    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "head_finetune.json")))
    reg, x, y = regen_finetune(meta)
    for name, run in meta["runs"].items():
        W = _gpu_params(reg)
        m, v = ops.head_train_state(W[0], W[2])
        step, losses = 1, []
        for order in run["orders"]:
            l = ops.head_train_epoch(x.to(DEV), y.to(DEV), torch.as_tensor(order, dtype=torch.int32, device=DEV), meta["batch_size"],
                                     *W, m, v, step, lr=run["lr"])
            step += l.numel()
            losses += l.cpu().tolist()
        check_finetune_against_golden(run, losses, *(w.cpu().numpy() for w in W), steps=len(losses))
