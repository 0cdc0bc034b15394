"""One-off fuzz of vpr_knn_topk / vpr_knn_topk_fp8 against oracle/knn.py over seeded random shapes (the generator of
tests/test_knn_gpu.py with other seeds), default routes and forced score-store variants.  Test infrastructure: imports
oracle/.   usage: python scripts/knn_fuzz.py [first_seed] [n_seeds]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import knn as oknn
from vpr_amd import _lib, ops
import test_knn_gpu as T
dev = torch.device("cuda:0")
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 6)
bad = total = 0
for seed in range(first, first + count):
    for fp8 in (False, True):
        for variant in ("0", "7"):
            _lib.tuning_set("VPR_KNN_VARIANT", int(variant))
            for (B, N, D, k, sd, base) in T._random_shapes(seed, 25, fp8):
                if fp8:
                    q, qs = T._fp8_rows(B, D, sd); g, gs = T._fp8_rows(N, D, sd + 1)
                    v_ref, i_ref = oknn.knn_topk_fp8(q, qs, g, gs, k, base)
                    v, i = ops.knn_topk_fp8(q.to(dev), qs.to(dev), g.to(dev), gs.to(dev), k, base)
                else:
                    q, g = T._unit_rows(B, D, sd), T._unit_rows(N, D, sd + 1)
                    v_ref, i_ref = oknn.knn_topk(q, g, k, base)
                    v, i = ops.knn_topk(q.to(dev), g.to(dev), k, base)
                total += 1
                if not (torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)):
                    bad += 1
                    print("MISMATCH", dict(seed=seed, fp8=fp8, variant=variant, B=B, N=N, D=D, k=k, sd=sd, base=base), flush=True)
    print(f"seed {seed}: {total} cases so far, {bad} mismatches", flush=True)
print("done:", total, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
