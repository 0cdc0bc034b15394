"""CPU: closed-form known answers that pin the SALAD and kNN oracles (SURVEY.md §8c — both are
'parity unpinned' by the reference, so these identities are what holds them in place)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import knn as oknn
from oracle import salad as osalad


def test_sinkhorn_column_marginals_exact():
    """(1) the v-update is the last half-step, so every column of P (dustbin included) sums to 1."""
    g = torch.Generator().manual_seed(0)
    S = torch.randn(3, 64, 256, generator=g, dtype=torch.float64) * 3
    P = osalad.matching_probs(S, dustbin=0.7, num_iters=3)
    assert P.shape == (3, 65, 256)
    assert (P.sum(1) - 1).abs().max().item() < 1e-12
    # row sums only approximately 1 / (n-m) after 3 iterations
    assert (P[:, :64].sum(2) - 1).abs().max().item() < 0.5
    assert (P[:, 64].sum(1) - 192).abs().max().item() < 40


def test_uniform_scores_closed_form():
    """(2) all scores == dustbin: P = 1/n for clusters, (n-m)/n for the dustbin, after 1 iteration."""
    S = torch.full((1, 64, 256), 1.5, dtype=torch.float64)
    for iters in (1, 3):
        P = osalad.matching_probs(S, dustbin=1.5, num_iters=iters)
        assert (P[:, :64] - 1 / 256).abs().max().item() < 1e-14
        assert (P[:, 64] - 192 / 256).abs().max().item() < 1e-14


def test_descriptor_norm_shares_and_invariances():
    g = torch.Generator().manual_seed(1)
    scores = torch.randn(2, 256, 64, generator=g)
    feats = torch.randn(2, 256, 128, generator=g)
    tok = torch.randn(2, 256, generator=g)
    out = osalad.sinkhorn_aggregate(scores, feats, tok, 1.0, 3)
    assert out.shape == (2, 8448)
    assert (out.pow(2).sum(1) - 1).abs().max().item() < 1e-12                      # (3)
    assert (out[:, :256].pow(2).sum(1) - 1 / 65).abs().max().item() < 1e-12
    assert (out[:, 256:].reshape(2, 128, 64).pow(2).sum(1) - 1 / 65).abs().max().item() < 1e-12
    perm = torch.randperm(256, generator=g)                                         # (4)
    assert (osalad.sinkhorn_aggregate(scores[:, perm], feats[:, perm], tok, 1.0, 3) - out).abs().max().item() < 1e-12
    assert (osalad.sinkhorn_aggregate(scores.double() + 2.5, feats, tok, 3.5, 3) - out).abs().max().item() < 1e-12   # (5)
    out32 = osalad.sinkhorn_aggregate(scores, feats, tok, 1.0, 3, dtype=torch.float32)                       # (6)
    assert (out32.double() - out).abs().max().item() < 2e-6
    # flatten order is l-major: index 256 + l*64 + m
    V = out[0, 256:].reshape(128, 64)
    assert torch.allclose(V[:, 5].norm(), torch.tensor(1 / math.sqrt(65.0), dtype=torch.float64))


def test_salad_mlps_quantisation_points():
    g = torch.Generator().manual_seed(2)
    tokens = torch.randn(1, 257, 128, generator=g).to(torch.bfloat16)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05
    w = dict(w1_sc=r(1024, 128).bfloat16(), b1_sc=r(1024), w2_s=r(64, 512).bfloat16(), b2_s=r(64),
             w2_c=r(128, 512).bfloat16(), b2_c=r(128), w1_t=r(512, 128).bfloat16(), b1_t=r(512),
             w2_t=r(256, 512).bfloat16(), b2_t=r(256))
    a = osalad.salad_aggregate(tokens, w, 1.0, 3, quantize=True)
    b = osalad.salad_aggregate(tokens, w, 1.0, 3, quantize=False)
    assert a.shape == (1, 8448) and 0 < (a - b).abs().max().item() < 1e-3


def test_knn_oracle_against_python_loop():
    g = torch.Generator().manual_seed(3)
    q = torch.randn(3, 64, generator=g).to(torch.bfloat16)
    gal = torch.randn(40, 64, generator=g).to(torch.bfloat16)
    gal[17] = gal[4]                                      # a tie
    v, i = oknn.knn_topk(q, gal, 6, index_base=100)
    for b in range(3):
        scores = []
        for n in range(40):
            s = sum(float(q[b, d]) * float(gal[n, d]) for d in range(64))   # python floats = fp64
            scores.append((-float(torch.tensor(s, dtype=torch.float64).to(torch.float32)), n))
        scores.sort()
        assert [100 + n for _, n in scores[:6]] == i[b].tolist()
        assert [-s for s, _ in scores[:6]] == v[b].tolist()


def test_knn_merge_equals_unsharded_and_pads():
    g = torch.Generator().manual_seed(4)
    q = torch.randn(5, 32, generator=g).to(torch.bfloat16)
    gal = torch.randn(90, 32, generator=g).to(torch.bfloat16)
    v_all, i_all = oknn.knn_topk(q, gal, 7)
    parts = [oknn.knn_topk(q, gal[lo:hi], 7, lo) for lo, hi in ((0, 30), (30, 33), (33, 90))]   # middle shard has < k rows
    vm, im = oknn.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(im, i_all) and torch.equal(vm, v_all)
    v, i = oknn.knn_topk(q, gal[:3], 5)
    assert (i[:, 3:] == -1).all() and torch.isinf(v[:, 3:]).all()
    assert oknn.recall_at_1(i_all[:, 0], i_all[:, 0]) == 1.0


def test_resize_tables_and_oracle_match_pil():
    """The preprocessing restatement is pinned by PIL itself (the resizer the reference's
    torchvision / HF transforms end in): identical bytes for both filters, down- and up-scaling."""
    import numpy as np
    from PIL import Image
    from oracle import preprocess as opre
    from vpr_amd.preprocess import resample_coeffs
    rng = np.random.default_rng(0)
    for (H, W) in [(480, 640), (224, 224), (200, 300), (777, 1033)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for filt, pf in (("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)):
            kx, xb, _ = resample_coeffs(W, 224, filt)
            ky, yb, _ = resample_coeffs(H, 224, filt)
            ref = np.asarray(Image.fromarray(img).resize((224, 224), pf))
            assert np.array_equal(opre.resize_u8(img, 224, kx, xb, ky, yb), ref), (H, W, filt)
    t = opre.to_tensor_normalize(np.array([[[0, 128, 255]]], dtype=np.uint8), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    assert t.shape == (3, 1, 1) and t[0, 0, 0] == -1.0 and t[2, 0, 0] == 1.0


# ------------------------------------------------------------------ frozen self-oracle fixtures
def _self_oracle():
    import importlib.util
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_self_oracle.py")
    spec = importlib.util.spec_from_file_location("make_self_oracle", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_reproduces_its_frozen_outputs():
    """The oracle is the contract of the two stages the reference cannot pin; these fixtures freeze it (self-oracle,
    not reference data): an edit that changes SALAD descriptors by > 1e-12 or any kNN index / value fails here."""
    mso = _self_oracle()
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = np.load(os.path.join(G, "salad_cpu.npz"))
    tokens, w = mso.salad_inputs(int(fx["seed"]))
    assert float(tokens.float().sum()) == float(fx["tokens_sum"])              # same inputs regenerated
    desc = osalad.salad_aggregate(tokens, w, dustbin=float(fx["dustbin"]), iters=3).numpy()
    assert np.abs(desc - fx["descriptor"]).max() < 1e-12
    fk = np.load(os.path.join(G, "knn_cpu.npz"))
    q, gal = mso.knn_inputs(int(fk["seed"]))
    assert float(q.sum()) == float(fk["q_sum"])
    v, i = oknn.knn_topk(q.to(torch.bfloat16), gal.to(torch.bfloat16), 7, 11)
    assert np.array_equal(i.numpy(), fk["idx"]) and np.array_equal(v.numpy(), fk["vals"])
    q8, qs = oknn.quantize_fp8_rows(q)
    g8, gs = oknn.quantize_fp8_rows(gal)
    v8, i8 = oknn.knn_topk_fp8(q8, qs, g8, gs, 7, 11)
    assert np.array_equal(i8.numpy(), fk["idx_fp8"]) and np.array_equal(v8.numpy(), fk["vals_fp8"])


@pytest.mark.gpu
def test_hip_path_matches_the_frozen_self_oracle(dev):
    from vpr_amd import ops
    from vpr_amd.ops import SaladWeights
    mso = _self_oracle()
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = np.load(os.path.join(G, "salad_cpu.npz"))
    tokens, w = mso.salad_inputs(int(fx["seed"]))
    out, _ = ops.salad_aggregate(tokens.to(dev), SaladWeights(**{k: v.to(dev) for k, v in w.items()}, dustbin=float(fx["dustbin"])), 3)
    assert np.abs(out.cpu().double().numpy() - fx["descriptor"]).max() < 1e-4
    fk = np.load(os.path.join(G, "knn_cpu.npz"))
    q, gal = mso.knn_inputs(int(fk["seed"]))
    v, i = ops.knn_topk(q.to(torch.bfloat16).to(dev), gal.to(torch.bfloat16).to(dev), 7, 11)
    assert np.array_equal(i.cpu().numpy(), fk["idx"]) and np.array_equal(v.cpu().numpy(), fk["vals"])
    q8, qs = oknn.quantize_fp8_rows(q)
    g8, gs = oknn.quantize_fp8_rows(gal)
    v8, i8 = ops.knn_topk_fp8(q8.to(dev), qs.to(dev), g8.to(dev), gs.to(dev), 7, 11)
    assert np.array_equal(i8.cpu().numpy(), fk["idx_fp8"]) and np.array_equal(v8.cpu().numpy(), fk["vals_fp8"])


def test_sinkhorn_solver_equals_hf_superglue_log_sinkhorn():
    """An INDEPENDENT implementation of the log-domain Sinkhorn iteration is importable: Hugging Face's SuperGlue port
    (transformers.models.superglue.modeling_superglue.log_sinkhorn_iterations) — the routine SALAD's solver descends
    from (SuperGlue's log_optimal_transport, arXiv:1911.11763 §3.3; SALAD arXiv:2311.15937 §3.2 reuses it with a
    dustbin ROW only).  With SALAD's marginals it must give the oracle's log-assignment: this pins the solver half of
    oracle/salad.py against a third party; the SALAD-specific wiring around it (marginals, dustbin row, normalisation
    order, MLP layout) stays pinned by the closed-form identities above only."""
    sg = pytest.importorskip("transformers.models.superglue.modeling_superglue")
    g = torch.Generator().manual_seed(12)
    B, m, n = 3, 64, 256
    S = torch.randn(B, m + 1, n, generator=g, dtype=torch.float64) * 2.5
    norm = -math.log(n + m)
    log_a = torch.full((B, m + 1), norm, dtype=torch.float64)
    log_a[:, -1] += math.log(n - m)
    log_b = torch.full((B, n), norm, dtype=torch.float64)
    for iters in (1, 3, 10):
        ours = osalad.log_otp_solver(log_a, log_b, S, iters)
        theirs = sg.log_sinkhorn_iterations(S, log_a, log_b, iters)
        assert (ours - theirs).abs().max().item() < 1e-12
    # and through the oracle's public entry (dustbin row appended, exp, dustbin dropped later)
    P = osalad.matching_probs(S[:, :m], dustbin=0.3, num_iters=3)
    S_aug = torch.cat([S[:, :m], torch.full((B, 1, n), 0.3, dtype=torch.float64)], 1)
    P_hf = torch.exp(sg.log_sinkhorn_iterations(S_aug, log_a, log_b, 3) - norm)
    assert (P - P_hf).abs().max().item() < 1e-12


def test_knn_oracle_agrees_with_numpy_brute_force():
    """The kNN contract written a second time with numpy only (f64 products of the bf16 values, f32 rounding, stable
    argsort = value desc / index asc) — guards the torch-based oracle against a sort / gather mistake; includes ties."""
    g = torch.Generator().manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(6, 96, generator=g), dim=1).to(torch.bfloat16)
    gal = torch.nn.functional.normalize(torch.randn(500, 96, generator=g), dim=1).to(torch.bfloat16)
    gal[40] = gal[7]; gal[300] = gal[7]
    q[0] = gal[7]
    v, i = oknn.knn_topk(q, gal, 9, index_base=100)
    s = (q.double().numpy() @ gal.double().numpy().T).astype(np.float32)
    order = np.argsort(-s, axis=1, kind="stable")[:, :9]
    assert np.array_equal(i.numpy(), order.astype(np.int32) + 100)
    assert np.array_equal(v.numpy(), np.take_along_axis(s, order, axis=1))
    assert i[0, :3].tolist() == [107, 140, 400]
