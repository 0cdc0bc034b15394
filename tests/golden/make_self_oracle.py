"""Self-oracle fixtures: outputs of the repo's OWN CPU restatement (oracle/salad.py, oracle/knn.py) on
seeded inputs, frozen so that a later edit of the oracle — the contract for the two stages the reference cannot pin —
cannot drift unnoticed.  NOT reference data: "parity unpinned" still applies to SALAD and kNN.
Run from the repo root:  python tests/golden/make_self_oracle.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import knn as oknn, salad as osalad  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def salad_inputs(seed=1234, B=2, C=256, hidden=512, m=64, l=128, t=256, std=0.05):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * std
    tokens = torch.randn(B, 257, C, generator=g).to(torch.bfloat16)
    w = dict(w1_sc=r(2 * hidden, C), b1_sc=r(2 * hidden), w2_s=r(m, hidden), b2_s=r(m), w2_c=r(l, hidden), b2_c=r(l),
             w1_t=r(hidden, C), b1_t=r(hidden), w2_t=r(t, hidden), b2_t=r(t))
    for k in list(w):
        if k.startswith("w"):
            w[k] = w[k].to(torch.bfloat16)
    return tokens, w


def knn_inputs(seed=4321, B=5, N=300, D=128):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
    gal = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    gal[17] = gal[3]                                     # an exact tie: lower index first
    return q, gal


def main():
    tokens, w = salad_inputs()
    desc = osalad.salad_aggregate(tokens, w, dustbin=0.75, iters=3)
    np.savez_compressed(os.path.join(OUT, "salad_cpu.npz"), descriptor=desc.numpy(), seed=1234, dustbin=0.75,
                        tokens_sum=float(tokens.float().sum()))
    q, gal = knn_inputs()
    v, i = oknn.knn_topk(q.to(torch.bfloat16), gal.to(torch.bfloat16), 7, 11)
    q8, qs = oknn.quantize_fp8_rows(q)
    g8, gs = oknn.quantize_fp8_rows(gal)
    v8, i8 = oknn.knn_topk_fp8(q8, qs, g8, gs, 7, 11)
    np.savez_compressed(os.path.join(OUT, "knn_cpu.npz"), vals=v.numpy(), idx=i.numpy(), vals_fp8=v8.numpy(), idx_fp8=i8.numpy(),
                        q_sum=float(q.sum()), seed=4321)
    print("wrote salad_cpu.npz, knn_cpu.npz")


if __name__ == "__main__":
    main()
