"""One-off fuzz of vpr_head_train_step / vpr_head_train_epoch against oracle/finetune.py: random widths, hidden sizes,
output counts, batch sizes (incl. ragged passes and repeated rows in a batch), learning rates and hyper-parameters; shapes
the C ABI declares unsupported must come back as a Python exception, never as a fault.  Test infrastructure."""
import os, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import finetune as oft
from vpr_amd import ops
dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
worst, refused = 0.0, 0
for case in range(n):
    g = torch.Generator().manual_seed(case)
    D = rnd.choice([16, 64, 8448, 16 * rnd.randint(1, 160), 15 * rnd.randint(1, 10)])
    hidden = rnd.choice([32, 64, 512, 32 * rnd.randint(1, 20), 48])
    n_out = rnd.randint(1, 9)
    N = rnd.randint(1, 150)
    bs = rnd.choice([1, 5, 16, 17, 48, 64, 65, rnd.randint(1, 64)])
    lr = 10 ** rnd.uniform(-5, -2)
    hyper = dict(lr=lr, betas=(rnd.choice([0.9, 0.5, 0.0]), rnd.choice([0.999, 0.9])), eps=rnd.choice([1e-8, 1e-6]),
                 weight_decay=rnd.choice([1e-2, 0.0, 0.1]))
    epochs = rnd.randint(1, 2)
    W = [torch.randn(hidden, D, generator=g) / max(D, 1) ** 0.5, torch.randn(hidden, generator=g) * 0.1,
         torch.randn(n_out, hidden, generator=g) / hidden ** 0.5, torch.randn(n_out, generator=g) * 0.1]
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    Y = torch.randn(N, n_out, generator=g)
    with_repeats = rnd.random() < 0.3
    orders = [np.array([rnd.randrange(N) for _ in range(N)]) if with_repeats else np.random.default_rng(case + e).permutation(N)
              for e in range(epochs)]
    st = oft.HeadState(*(w.numpy() for w in W))
    try:
        Wg = [w.to(dev).contiguous() for w in W]
        m, v = ops.head_train_state(Wg[0], Wg[2])
        Xg, Yg = X.to(dev), Y.to(dev)
        step, losses = 1, []
        for o in orders:
            l = ops.head_train_epoch(Xg, Yg, torch.as_tensor(o, dtype=torch.int32, device=dev), bs, *Wg, m, v, step, **hyper)
            step += l.numel()
            losses.append(l)
        torch.cuda.synchronize()
    except RuntimeError as e:
        refused += 1
        ok = (D % 16) or (hidden % 32) or n_out > 8 or bs > 64 and N > 64
        print(f"case {case}: D={D} hidden={hidden} n_out={n_out} N={N} bs={bs}: refused ({str(e)[:70]}){'' if ok else '  UNEXPECTED'}", flush=True)
        assert ok
        continue
    ref_losses = []
    Xn, Yn = X.numpy().astype(np.float64), Y.numpy().astype(np.float64)
    for o in orders:
        for lo in range(0, N, bs):
            idx = o[lo:lo + bs]
            ref_losses.append(oft.train_step(st, Xn[idx], Yn[idx], **hyper))
    got = torch.cat(losses).cpu().numpy()
    steps = len(ref_losses)
    lerr = float(np.max(np.abs(got - np.array(ref_losses)) / np.maximum(np.abs(ref_losses), 1e-12)))
    perr = max(float(np.abs(w.cpu().numpy() - r).max()) for w, r in zip(Wg, st.p)) / (lr * steps)
    worst = max(worst, perr)
    flag = "" if (lerr <= 5e-5 and perr <= 0.05) else "  <-- OUT OF TOLERANCE"
    print(f"case {case}: D={D} hidden={hidden} n_out={n_out} N={N} bs={bs} steps={steps} lr={lr:.1e} {hyper['betas']} wd={hyper['weight_decay']}"
          f"{' repeats' if with_repeats else ''}: loss rel {lerr:.1e}, params {perr:.1e} lr*steps{flag}", flush=True)
    assert not flag
print(f"done: {n} cases, {refused} refused, worst parameter error {worst:.2e} of lr * steps (bound 0.05)")
