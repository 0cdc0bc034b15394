"""GPU parity of the kNN stage (through the C ABI) against oracle/knn.py.
Bit-exact indices; values = fp32 rounding of the exact dot product (exact equality asserted)."""
import pytest
import torch

from oracle import knn as oknn

pytestmark = pytest.mark.gpu


def _unit_rows(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g)
    return torch.nn.functional.normalize(x, dim=1).to(torch.bfloat16)


def _check(dev, B, N, D, k, seed=0, index_base=0):
    from vpr_amd import ops
    q, g = _unit_rows(B, D, seed), _unit_rows(N, D, seed + 1)
    v_ref, i_ref = oknn.knn_topk(q, g, k, index_base)
    v, i = ops.knn_topk(q.to(dev), g.to(dev), k, index_base)
    torch.cuda.synchronize()
    assert torch.equal(i.cpu(), i_ref), f"indices differ B={B} N={N} D={D} k={k}"
    assert torch.equal(v.cpu(), v_ref), f"values differ B={B} N={N} D={D} k={k}"


@pytest.mark.parametrize("B,N,D,k", [
    (64, 1000, 8448, 10),      # BASELINE config 2 gallery
    (1, 7, 64, 1),             # minimum everything
    (3, 17, 128, 5),           # ragged tiny
    (64, 4096, 256, 10),       # exactly one select chunk
    (64, 4097, 256, 10),       # one element into the second chunk
    (100, 5000, 512, 64),      # B not a multiple of 64, max k
    (17, 12500, 1024, 10),     # one 8-way shard of the 100k gallery (rows), reduced D
    (300, 3000, 512, 10),      # >= 256 queries: the GEMM-shaped score path (all-gathered multi-GPU batch)
    (512, 12500, 128, 10),     # 8 ranks x 64 queries against one shard's rows
    (3, 600, 16384, 5),        # rows wider than the fused final kernel's LDS budget: general rescore/order path
    (2, 300000, 64, 64),       # > 4096 level-0 candidates per query: register select level + general path
])
def test_knn_matches_oracle(dev, B, N, D, k):
    _check(dev, B, N, D, k)


def _random_shapes(seed, count, fp8):
    """Seeded shape sweep over every score route (stream / K-split stream / 128- and 256-tile GEMM, split-K or not),
    ragged everything; sized so the CPU oracle (f64 brute force) stays within ~1 s per case."""
    import random
    rnd = random.Random(seed)
    out = []
    step = 128 if fp8 else 64
    for _ in range(count):
        B = rnd.choice([rnd.randint(1, 64), rnd.randint(65, 191), rnd.randint(192, 256), rnd.randint(385, 600)])
        D = rnd.choice([step * rnd.randint(1, 2048 // step)] * 3 + [8448, 4096])
        N = rnd.choice([rnd.randint(1, 300), rnd.randint(301, 5000), rnd.randint(5001, 20000)])
        N = max(1, min(N, int(6e9 / (B * D * 8))))            # bound the oracle's B x N x D f64 work
        k = rnd.choice([1, 5, 10, rnd.randint(1, 64)])
        out.append((B, N, D, k, rnd.randint(0, 10**6), rnd.choice([0, 0, 12345])))
    return out


@pytest.mark.parametrize("B,N,D,k,seed,base", _random_shapes(2026, 40, False))
def test_knn_random_shapes_match_oracle(dev, B, N, D, k, seed, base):
    _check(dev, B, N, D, k, seed=seed, index_base=base)


@pytest.mark.parametrize("B,N,D,k,seed,base", _random_shapes(905, 28, True))
def test_knn_fp8_random_shapes_match_oracle(dev, B, N, D, k, seed, base):
    from vpr_amd import ops
    q, qs = _fp8_rows(B, D, seed)
    g, gs = _fp8_rows(N, D, seed + 1)
    v_ref, i_ref = oknn.knn_topk_fp8(q, qs, g, gs, k, base)
    v, i = ops.knn_topk_fp8(q.to(dev), qs.to(dev), g.to(dev), gs.to(dev), k, base)
    assert torch.equal(i.cpu(), i_ref), f"indices differ B={B} N={N} D={D} k={k}"
    assert torch.equal(v.cpu(), v_ref), f"values differ B={B} N={N} D={D} k={k}"


def test_knn_fewer_rows_than_k(dev):
    from vpr_amd import ops
    q, g = _unit_rows(4, 64, 3), _unit_rows(3, 64, 4)
    v_ref, i_ref = oknn.knn_topk(q, g, 8)
    v, i = ops.knn_topk(q.to(dev), g.to(dev), 8)
    assert torch.equal(i.cpu(), i_ref)
    assert torch.equal(v.cpu(), v_ref)


def test_knn_ties_prefer_lower_index(dev):
    """Duplicate gallery rows give exactly equal scores: lower index must win."""
    from vpr_amd import ops
    g = _unit_rows(500, 256, 5)
    g[100] = g[7]
    g[300] = g[7]
    q = g[[7, 20]].clone()
    v, i = ops.knn_topk(q.to(dev), g.to(dev), 4, index_base=1000)
    v_ref, i_ref = oknn.knn_topk(q, g, 4, index_base=1000)
    assert i.cpu()[0, :3].tolist() == [1007, 1100, 1300]
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)


def test_knn_all_equal_scores(dev):
    """Degenerate input: a zero query scores 0 against every row -> the k lowest indices win.
    (Worst case for the threshold select: every key ties on value.)"""
    from vpr_amd import ops
    N, D, k = 20000, 128, 10
    g = _unit_rows(N, D, 21)
    q = torch.zeros(2, D, dtype=torch.bfloat16)
    q[1] = g[5]
    v, i = ops.knn_topk(q.to(dev), g.to(dev), k)
    v_ref, i_ref = oknn.knn_topk(q, g, k)
    assert i.cpu()[0].tolist() == list(range(k)) and torch.equal(v.cpu()[0], torch.zeros(k))
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)


@pytest.mark.parametrize("descending", [True, False])
def test_knn_monotone_scores(dev, descending):
    """Scores sorted along the gallery index (all the winners inside one thread's stride / one
    chunk): the adversarial layout for the per-thread-max threshold bound."""
    from vpr_amd import ops
    N, D, k = 30000, 64, 64
    base = torch.nn.functional.normalize(torch.ones(1, D), dim=1)
    scale = torch.linspace(1.0, 0.01, N) if descending else torch.linspace(0.01, 1.0, N)
    g = (scale[:, None] * base).to(torch.bfloat16)
    q = base.to(torch.bfloat16).repeat(3, 1)
    v, i = ops.knn_topk(q.to(dev), g.to(dev), k)
    v_ref, i_ref = oknn.knn_topk(q, g, k)
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)


def test_knn_scores_stage_close_to_exact(dev):
    """The MFMA score matrix (approximate stage) is within 2e-5 of the exact fp64 scores."""
    from vpr_amd import ops
    B, N, D, k = 64, 3000, 8448, 10
    q, g = _unit_rows(B, D, 8), _unit_rows(N, D, 9)
    qd, gd = q.to(dev), g.to(dev)
    ws = ops.knn_workspace(B, N, D, k, dev)
    ops.knn_scores(qd, gd, ws)
    S = ops.knn_scores_view(ws, B, N, D, k).cpu().double()
    ref = oknn.knn_scores_f64(q, g)
    assert (S - ref).abs().max().item() < 2e-5


def test_knn_sharded_merge_equals_unsharded(dev):
    """Size-independent property: shard -> local top-k (global indices) -> merge == unsharded."""
    from vpr_amd import ops
    B, N, D, k, R = 32, 6000, 512, 10, 4
    q, g = _unit_rows(B, D, 11), _unit_rows(N, D, 12)
    qd, gd = q.to(dev), g.to(dev)
    v_all, i_all = ops.knn_topk(qd, gd, k)
    vs, is_ = [], []
    for r in range(R):
        lo, hi = r * N // R, (r + 1) * N // R
        v, i = ops.knn_topk(qd, gd[lo:hi].contiguous(), k, index_base=lo)
        vs.append(v), is_.append(i)
    vm, im = ops.topk_merge(torch.stack(vs), torch.stack(is_))
    assert torch.equal(im, i_all) and torch.equal(vm, v_all)
    vo, io = oknn.topk_merge(torch.stack(vs).cpu(), torch.stack(is_).cpu())
    assert torch.equal(im.cpu(), io) and torch.equal(vm.cpu(), vo)


def test_knn_full_size_planted_recall(dev):
    """BASELINE size (N=100k, D=8448, B=64): planted positives must come back as top-1 and the
    result must be sorted; checked without the CPU oracle (size-independent properties)."""
    from vpr_amd import ops
    N, D, B, k = 100_000, 8448, 64, 10
    g = torch.Generator(device=dev).manual_seed(1)
    gal = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
    pos = torch.randint(0, N, (B,), device=dev, generator=g)
    noise = torch.randn(B, D, device=dev, generator=g)
    q = torch.nn.functional.normalize(gal[pos].float() + 0.1 * noise, dim=1).to(torch.bfloat16)
    v, i = ops.knn_topk(q, gal, k)
    assert torch.equal(i[:, 0].long(), pos)
    assert bool((v[:, :-1] >= v[:, 1:]).all())
    # top-1 value equals the exact dot product of the planted pair
    exact = (q.double() * gal[pos].double()).sum(1).float()
    assert torch.equal(v[:, 0], exact)


# ------------------------------------------------------------------------------------------ fp8
def _fp8_rows(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)
    return oknn.quantize_fp8_rows(x)


def test_quantize_fp8_matches_torch(dev):
    from vpr_amd import ops
    g = torch.Generator().manual_seed(30)
    x = torch.randn(37, 8448, generator=g) * torch.logspace(-3, 2, 37)[:, None]
    x[5] = 0
    q_ref, s_ref = oknn.quantize_fp8_rows(x)
    q, s = ops.quantize_fp8_rows(x.to(dev))
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref)


@pytest.mark.parametrize("B", [150, 256, 300, 512])
def test_knn_fp8_gemm_path_scores_match_stream_path(dev, tune, B):
    """More than 64 queries: the block-scaled fp8 MFMA GEMMs — the 128 x 128-tile kernel (v_mfma_scale_f32_32x32x64_f8f6f4)
    when a 256-row query tile would be less than 3/4 full, else the 256 x 256-tile gemm256_kernel<true>
    (v_mfma_scale_f32_16x16x128_f8f6f4, LDS-DMA in flight across barriers) — must produce the score matrix the streaming kernel produces (same exact fp8 products,
    different f32 summation order)."""
    from vpr_amd import ops, _lib
    import ctypes
    N, D, k = 3001, 8448, 10
    assert _lib.lib().vpr_knn_scores_kernel_name(1, B, N).decode() == ("vpr::gemm256_kernel<true, 10>" if B in (256, 512) else "vpr::gemm_nt_fp8_kernel")
    q, qs = _fp8_rows(B, D, 41)
    g, gs = _fp8_rows(N, D, 42)
    q, qs, g, gs = q.to(dev), qs.to(dev), g.to(dev), gs.to(dev)
    outs = []
    for thr in ("100000", "65"):
        tune("VPR_KNN_GEMM_MIN_B", thr)
        ws = ops.knn_workspace(B, N, D, k, dev)
        ws.zero_()
        v, i = ops.knn_topk_fp8(q, qs, g, gs, k, 0, ws)
        outs.append((ops.knn_scores_view(ws, B, N, D, k).clone(), v, i))
    (s_stream, v0, i0), (s_gemm, v1, i1) = outs
    scale = s_stream.abs().max().item()
    assert (s_stream - s_gemm).abs().max().item() < 2e-6 * max(scale, 1.0)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)              # exact rescoring makes the final answer identical


@pytest.mark.parametrize("B,N,D,k", [(64, 3000, 8448, 10), (5, 300, 128, 3), (70, 9000, 1024, 20), (300, 4000, 8448, 10),
                                     (500, 2500, 1024, 10), (384, 700, 256, 5)])      # >= 384 queries: gemm256_kernel<true>
def test_knn_fp8_matches_oracle(dev, B, N, D, k):
    """BASELINE config 5 arithmetic (e4m3 descriptors, per-row scale) at oracle-sized N."""
    from vpr_amd import ops
    q, qs = _fp8_rows(B, D, 31)
    g, gs = _fp8_rows(N, D, 32)
    v_ref, i_ref = oknn.knn_topk_fp8(q, qs, g, gs, k, 7)
    v, i = ops.knn_topk_fp8(q.to(dev), qs.to(dev), g.to(dev), gs.to(dev), k, 7)
    assert torch.equal(i.cpu(), i_ref)
    assert torch.equal(v.cpu(), v_ref)


def test_knn_fp8_agrees_with_bf16_on_planted_positives(dev):
    """fp8 keeps the retrieval answer: planted positives are still top-1 (Recall@1 = 1)."""
    from vpr_amd import ops
    N, D, B = 20000, 8448, 32
    g = torch.Generator(device=dev).manual_seed(5)
    gal = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
    pos = torch.randint(0, N, (B,), device=dev, generator=g)
    qf = torch.nn.functional.normalize(gal[pos] + 0.1 * torch.randn(B, D, device=dev, generator=g), dim=1)
    g8, gs = ops.quantize_fp8_rows(gal)
    q8, qs = ops.quantize_fp8_rows(qf)
    _, i8 = ops.knn_topk_fp8(q8, qs, g8, gs, 5)
    _, i16 = ops.knn_topk(qf.to(torch.bfloat16), gal.to(torch.bfloat16), 5)
    assert torch.equal(i8[:, 0].long(), pos) and torch.equal(i16[:, 0].long(), pos)


def test_sharded_gallery_fp8_rows(dev):
    """ShardedGallery with e4m3 rows + per-row scales: same answer as vpr_knn_topk_fp8 on the quantised
    queries, and the planted neighbours are found (retrieval layer of BASELINE config 5)."""
    from vpr_amd import ops
    from vpr_amd.retrieval import ShardedGallery
    g = torch.Generator().manual_seed(21)
    N, B, D, k = 3000, 9, 8448, 5
    gal = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    pos = torch.randint(0, N, (B,), generator=g)
    q = torch.nn.functional.normalize(gal[pos] + 0.05 * torch.randn(B, D, generator=g), dim=1).to(torch.bfloat16).to(dev)
    g8, gs = ops.quantize_fp8_rows(gal.to(dev))
    sg = ShardedGallery(g8, N, 0, 1, scales=gs)
    v, i = sg.search(q, k)
    q8, qs = ops.quantize_fp8_rows(q.float())
    v2, i2 = ops.knn_topk_fp8(q8, qs, g8, gs, k)
    assert torch.equal(i, i2) and torch.equal(v, v2)
    assert torch.equal(i[:, 0].cpu().long(), pos)
    with pytest.raises(ValueError):
        ShardedGallery(g8, N, 0, 1)                      # fp8 rows without scales


# ------------------------------------------------------------- full-size shapes (BASELINE configs 3 and 5)
def _full_size_check(dev, B, N, k, fp8, seed, slab=65536, extra=16):
    """Exact parity at sizes the CPU oracle cannot brute-force (N x 8448 up to 1M rows), in three steps:
      1. gallery generated on the GPU slab by slab; queries = planted positives + noise (Recall@1 must be 1);
      2. an INDEPENDENT approximate search (torch matmul in f32 per slab + torch.topk, k + `extra` per query) gives a
         candidate set that contains the true top-k unless f32 GEMM error exceeds the gap to the (k+extra)-th score;
      3. oracle/knn.py (exact f64 products, stable order) on the union U of those candidates and of the rows the HIP
         path returned: every row of U is a real gallery row, U contains each query's true top-k, U is sorted by
         global index (ties keep their order) -> the oracle's top-k over U IS the global answer, compared bit for bit
         (indices and f32 values) with vpr_knn_topk / vpr_knn_topk_fp8."""
    from vpr_amd import ops
    D = 8448
    gen = torch.Generator(device=dev).manual_seed(seed)
    rows = torch.empty((N, D), dtype=torch.uint8 if fp8 else torch.bfloat16, device=dev)
    scales = torch.empty((N,), dtype=torch.float32, device=dev) if fp8 else None
    pos = torch.randint(0, N, (B,), device=dev, generator=gen)
    planted = torch.empty((B, D), dtype=torch.float32, device=dev)
    for lo in range(0, N, slab):
        n = min(slab, N - lo)
        x = torch.nn.functional.normalize(torch.randn(n, D, device=dev, generator=gen), dim=1)
        sel = (pos >= lo) & (pos < lo + n)
        planted[sel] = x[pos[sel] - lo]
        if fp8:
            rows[lo:lo + n], scales[lo:lo + n] = ops.quantize_fp8_rows(x)
        else:
            rows[lo:lo + n] = x.to(torch.bfloat16)
        del x
    qf = torch.nn.functional.normalize(planted + 0.1 * torch.randn(B, D, device=dev, generator=gen), dim=1)
    if fp8:
        q, qs = ops.quantize_fp8_rows(qf)
        v, i = ops.knn_topk_fp8(q, qs, rows, scales, k)
        q_deq = q.view(torch.float8_e4m3fn).float() * qs[:, None]
    else:
        q = qf.to(torch.bfloat16)
        v, i = ops.knn_topk(q, rows, k)
        q_deq = q.float()
    torch.cuda.synchronize()
    assert torch.equal(i[:, 0].long(), pos), "planted positives are not top-1"
    assert bool((v[:, :-1] >= v[:, 1:]).all())
    # 2. independent approximate candidates
    kk = k + extra
    best_v = torch.full((B, kk), float("-inf"), device=dev)
    best_i = torch.full((B, kk), -1, dtype=torch.int64, device=dev)
    for lo in range(0, N, slab):
        n = min(slab, N - lo)
        g = rows[lo:lo + n].view(torch.float8_e4m3fn).float() * scales[lo:lo + n, None] if fp8 else rows[lo:lo + n].float()
        sv, si = torch.topk(q_deq @ g.T, min(kk, n), dim=1)
        cv, ci = torch.cat([best_v, sv], 1), torch.cat([best_i, si + lo], 1)
        o = torch.topk(cv, kk, dim=1).indices
        best_v, best_i = torch.gather(cv, 1, o), torch.gather(ci, 1, o)
        del g
    # 3. exact oracle on the union
    U = torch.unique(torch.cat([best_i.flatten(), i.flatten().long()]))         # sorted ascending
    U = U[U >= 0]
    if fp8:
        v_ref, li = oknn.knn_topk_fp8(q.cpu(), qs.cpu(), rows[U].cpu(), scales[U].cpu(), k)
    else:
        v_ref, li = oknn.knn_topk(q.cpu(), rows[U].cpu(), k)
    i_ref = U.cpu()[li.long()].to(torch.int32)
    assert torch.equal(i.cpu(), i_ref), "indices differ from the exact answer"
    assert torch.equal(v.cpu(), v_ref), "values differ from the exact answer"


@pytest.mark.parametrize("B,N,k,fp8", [
    (64, 100_000, 10, False),     # config 3 on one GPU: bf16, one 196-row tile per workgroup
    (64, 150_000, 10, False),     # bf16 shard above 106k rows: the 256-row tile form, knn_scores_kernel<false, 256, 2, 4>
    (64, 125_000, 10, True),      # config 5, one 8-way shard of the 1M gallery: one 244-row tile per workgroup (256-row form)
    (512, 125_000, 10, True),     # config 5 on 8 GPUs: the all-gathered 512-query batch -> gemm256_kernel<true>
    (320, 60_000, 10, True),      # a 256-row tile would be 62 % full -> gemm_nt_fp8_kernel (128 x 128 tiles)
    (512, 12_500, 10, False),     # config 3 on 8 GPUs: 512 gathered queries x one shard: 98 tiles -> gemm256_kernel<false>, 2 K slices
    (256, 25_000, 10, False),     # config 3 on 4 GPUs: same 98 tiles, 2 K slices
    (512, 12_500, 10, True),      # e4m3 form of the split-K GEMM
    (512, 125_000, 10, False),    # a 1M-row bf16 gallery on 8 GPUs: enough 256 x 256 tiles -> gemm256_kernel<false>, f32 out
    (64, 1_000_000, 10, True),    # config 5 unsharded: 1M x 8448 e4m3 (8.4 GB) on one GPU: 8 tiles of 245 rows per workgroup
    (16, 2_000_000, 10, True),    # 2M rows (16.9 GB): > 4096 level-0 candidates per query -> register select level + fused final
    (64, 500_000, 10, False),     # 8.4 GB of bf16 rows: 4 tiles of 245 rows per workgroup, staged score stores
])
def test_knn_full_size_exact(dev, B, N, k, fp8):
    _full_size_check(dev, B, N, k, fp8, seed=B + N)


@pytest.mark.parametrize("B,N,fp8", [(64, 3000, False), (64, 30000, True), (300, 5000, False), (512, 6378, True)])
def test_knn_two_stage_call_equals_one_call(dev, B, N, fp8):
    """vpr_knn_topk_scores_stage + vpr_knn_topk_select_stage (what the pipeline runs when it times the score stage) ==
    vpr_knn_topk[_fp8]_checked, for the K-split stream form, the plain stream form and both GEMM routes."""
    from vpr_amd import ops
    D, k = 8448, 10
    g = torch.Generator(device=dev).manual_seed(B + N)
    gal = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1)
    ev = []
    if fp8:
        G, gs = ops.quantize_fp8_rows(gal)
        Q, qs = ops.quantize_fp8_rows(q)
        v1, i1 = ops.knn_topk_fp8(Q, qs, G, gs, k, 7)
        v2, i2 = ops.knn_topk_fp8(Q, qs, G, gs, k, 7, score_events=ev)
    else:
        G, Q = gal.to(torch.bfloat16), q.to(torch.bfloat16)
        v1, i1 = ops.knn_topk(Q, G, k, 7)
        v2, i2 = ops.knn_topk(Q, G, k, 7, score_events=ev)
    torch.cuda.synchronize()
    assert torch.equal(v1, v2) and torch.equal(i1, i2)
    assert len(ev) == 1 and ev[0][0].elapsed_time(ev[0][1]) > 0


@pytest.mark.parametrize("fp8", [False, True])
def test_knn_gemm_split_k_equals_unsplit(dev, tune, fp8):
    """Gathered batch against a shard with fewer than 256 score tiles: the 256 x 256-tile GEMM splits K into slabs the
    level-0 select adds.  Same final answer as the unsplit routes (exact rescoring), and the summed score matrix agrees
    with the unsplit one to f32 summation-order noise."""
    from vpr_amd import ops, _lib
    B, N, D, k = 512, 6378, 8448, 10
    assert _lib.lib().vpr_knn_scores_kernel_name(int(fp8), B, N).decode() == f"vpr::gemm256_kernel<{'true' if fp8 else 'false'}, 10>"
    g = torch.Generator(device=dev).manual_seed(77)
    gal = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
    pos = torch.randint(0, N, (B,), device=dev, generator=g)
    q = torch.nn.functional.normalize(gal[pos] + 0.1 * torch.randn(B, D, device=dev, generator=g), dim=1)
    if fp8:
        G, gs = ops.quantize_fp8_rows(gal)
        Q, qs = ops.quantize_fp8_rows(q)
        call = lambda ws: ops.knn_topk_fp8(Q, qs, G, gs, k, 0, ws)
    else:
        G, Q = gal.to(torch.bfloat16), q.to(torch.bfloat16)
        call = lambda ws: ops.knn_topk(Q, G, k, 0, ws)
    outs = []
    for ks in ("0", "1"):
        tune("VPR_KNN_GEMM_KSPLIT", ks)
        ws = ops.knn_workspace(B, N, D, k, dev)
        ws.zero_()
        v, i = call(ws)
        outs.append((ops.knn_scores_view(ws, B, N, D, k).clone(), v, i))
    (s0, v0, i0), (s1, v1, i1) = outs
    assert (s0 - s1).abs().max().item() < 2e-6 * max(s0.abs().max().item(), 1.0)
    assert torch.equal(i0, i1) and torch.equal(v0, v1) and torch.equal(i1[:, 0].long(), pos)


# ----------------------------------------------------------------- certification of the exactness contract
def _near_tie_problem(n_chunks, per_chunk, D=512, seed=50):
    """A gallery with 64 'band' rows whose EXACT scores against query 0 are 0.75 + c + r * 2^-24, r = 0..63 (one f32
    ulp apart), while their MFMA scores are scrambled: each band row carries a pair (+x_r, -x_r) on two coordinates
    where the query is equal (exactly zero contribution) but far apart in K, so the f32 accumulator passes through a
    different large intermediate value for every row.  More than KP - k = 14 rows inside the MFMA error of the k-th
    score: the approximate top-KP cut can drop true top-k rows."""
    g = torch.Generator().manual_seed(seed)
    N = 8192 * n_chunks
    gal = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(2, D, generator=g), dim=1) * 0.5
    q[0, 0], q[0, 1], q[0, 2], q[0, 500] = 2.0 ** -12, 0.75, 2.0 ** -3, 2.0 ** -3
    tail = torch.randn(D, generator=g) * 0.02
    tail[[0, 1, 2, 500]] = 0
    rows = []
    for r in range(64):
        chunk, j = (r % n_chunks, r // n_chunks) if per_chunk < 64 else (0, r)
        idx = chunk * 8192 + 97 * j + 5
        v = tail.clone()
        v[0], v[1] = r * 2.0 ** -12, 1.0
        x = 1.0 + (r * 37 % 64) / 64.0
        v[2], v[500] = x, -x
        gal[idx] = v
        rows.append(idx)
    return q.to(torch.bfloat16), gal.to(torch.bfloat16), rows


def test_knn_near_ties_are_widened_and_exact(dev):
    """64 rows within 64 ulps of each other around the k-th score, 8 per level-0 chunk: every chunk list holds its
    band rows, the top-KP cut does not -> the kernel must notice (status 1), rescore the whole band and return the
    exact answer; the unrelated query stays status 0."""
    from vpr_amd import ops
    q, gal, rows = _near_tie_problem(8, 8)
    k = 10
    v_ref, i_ref = oknn.knn_topk(q, gal, k)
    assert set(i_ref[0].tolist()) == set(rows[-k:])                   # the 10 highest r: the exact answer is known
    status = torch.full((2,), -1, dtype=torch.int32, device=dev)
    unc = torch.zeros(1, dtype=torch.int32, device=dev)
    bound = float(gal.float().norm(dim=1).max()) * 1.001
    v, i = ops.knn_topk(q.to(dev), gal.to(dev), k, norm_bound=bound, status=status, uncertified=unc)
    assert status.tolist() == [1, 0] and int(unc) == 0
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)


def test_knn_saturated_band_is_flagged_and_fallback_is_exact(dev):
    """All 64 band rows in ONE level-0 chunk: its list (KP = 24) is cut inside the band, the kernel cannot certify
    (status 2, counter incremented); exact_fallback re-runs the query exhaustively -> exact answer, status 3."""
    from vpr_amd import ops
    q, gal, rows = _near_tie_problem(3, 64)
    k = 10
    v_ref, i_ref = oknn.knn_topk(q, gal, k)
    bound = float(gal.float().norm(dim=1).max()) * 1.001
    status = torch.full((2,), -1, dtype=torch.int32, device=dev)
    unc = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.knn_topk(q.to(dev), gal.to(dev), k, norm_bound=bound, status=status, uncertified=unc)
    assert status.tolist() == [2, 0] and int(unc) == 1
    v, i = ops.knn_topk(q.to(dev), gal.to(dev), k, norm_bound=bound, status=status, exact_fallback=True)
    assert status.tolist() == [3, 0]
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)
    ve, ie = ops.knn_topk_exhaustive(q.to(dev), gal.to(dev), k, index_base=11)
    assert torch.equal(ie.cpu(), i_ref + 11) and torch.equal(ve.cpu(), v_ref)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("bound", [1e-9, 30.0, 1e6])
def test_knn_certification_paths_keep_the_answer(dev, fp8, bound):
    """The three outcomes forced through the norm bound on ordinary data: a tiny bound certifies at once (status 0),
    a loose one widens (1) or flags (2), an absurd one flags everything; with exact_fallback the answer is the
    oracle's in every case, bf16 and e4m3 alike."""
    from vpr_amd import ops
    B, N, D, k = 9, 20000, 1024, 10
    if fp8:
        q, qs = _fp8_rows(B, D, 61)
        g, gs = _fp8_rows(N, D, 62)
        v_ref, i_ref = oknn.knn_topk_fp8(q, qs, g, gs, k, 3)
        args = (q.to(dev), qs.to(dev), g.to(dev), gs.to(dev), k, 3)
        fn = ops.knn_topk_fp8
    else:
        q, g = _unit_rows(B, D, 61), _unit_rows(N, D, 62)
        v_ref, i_ref = oknn.knn_topk(q, g, k, 3)
        args = (q.to(dev), g.to(dev), k, 3)
        fn = ops.knn_topk
    status = torch.full((B,), -1, dtype=torch.int32, device=dev)
    v, i = fn(*args, norm_bound=bound, status=status, exact_fallback=True)
    st = status.tolist()
    if bound < 1:
        assert st == [0] * B
    elif bound > 1e3:
        assert st == [3] * B
    else:
        assert all(s in (0, 1, 3) for s in st)
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref)


def test_knn_default_bound_certifies_normalised_descriptors(dev):
    """BASELINE-like data (unit rows, D = 8448): every query certified without widening — the guard costs nothing
    on the benchmark path."""
    from vpr_amd import ops
    B, N, D, k = 64, 30000, 8448, 10
    q, g = _unit_rows(B, D, 71).to(dev), _unit_rows(N, D, 72).to(dev)
    status = torch.full((B,), -1, dtype=torch.int32, device=dev)
    unc = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.knn_topk(q, g, k, status=status, uncertified=unc)
    assert int(status.max()) == 0 and int(unc) == 0



# ----------------------------------------------------------------- score-store paths
@pytest.mark.parametrize("B,N,D,fp8", [(64, 20000, 8448, False), (64, 20000, 8448, True), (7, 1003, 256, False),
                                       (64, 131, 128, True), (33, 5, 64, False), (50, 150_001, 128, False)])
@pytest.mark.parametrize("variant", ["6", "7"])
def test_knn_staged_score_stores_equal_direct(dev, tune, B, N, D, fp8, variant):
    """The score kernel of multi-tile shards (N > 131k rows) writes each tile's scores as whole row segments staged
    through LDS (VPR_KNN_VARIANT 6 / 7 force that path, plain / nt stores, at any size): the score matrix and the answer
    are bit-identical to the direct-store kernel's (variant 5), ragged tiles, batches and row counts included."""
    from vpr_amd import ops
    k = 5
    if fp8:
        q, qs = _fp8_rows(B, D, 91)
        g, gs = _fp8_rows(N, D, 92)
        args = (q.to(dev), qs.to(dev), g.to(dev), gs.to(dev), k, 0)
        fn = ops.knn_topk_fp8
    else:
        args = (_unit_rows(B, D, 91).to(dev), _unit_rows(N, D, 92).to(dev), k, 0)
        fn = ops.knn_topk
    tune("VPR_KNN_GEMM_MIN_B", "100000")
    outs = []
    for var in ("5", variant):
        tune("VPR_KNN_VARIANT", var)
        ws = ops.knn_workspace(B, N, D, k, dev)
        ws.zero_()
        v, i = fn(*args, ws)
        outs.append((ops.knn_scores_view(ws, B, N, D, k).clone(), v, i))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


# ----------------------------------------------------------------- the error model behind the certificate
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("route,B,N", [("stream", 64, 120_000), ("stream K-split", 64, 3000), ("gemm128", 150, 3000),
                                       ("gemm256 K-split", 512, 6378), ("gemm256", 512, 66_000)])
@pytest.mark.parametrize("signs", ["random", "same"])
def test_mfma_scores_stay_inside_the_certificate_error_model(dev, fp8, route, B, N, signs):
    """every status-0 / 1 claim rests on |S_mfma - exact| <= eps = 1.1 * D * 2^-24 * |q| * G (f32 accumulation
    of exact products in any order).  Measured here per score route (streaming kernel, its K-split, the 128- and 256-tile
    GEMMs incl. split-K slabs summed by the level-0 select, the block-scaled 128-deep fp8 MFMAs) at D = 8448 against exact
    f64 dot products of the quantised operands, on unit rows with random signs and on all-positive rows (the worst case of
    the model: sum |q_i g_i| = |q . g|): the observed error must stay below HALF of eps.  The ratio is printed."""
    from vpr_amd import ops
    D, k = 8448, 10
    gen = torch.Generator(device=dev).manual_seed(B + N + int(fp8))
    mk = lambda n: torch.randn(n, D, device=dev, generator=gen)
    q32, g32 = mk(B), mk(N)
    if signs == "same":
        q32, g32 = q32.abs(), g32.abs()
    q32, g32 = torch.nn.functional.normalize(q32, dim=1), torch.nn.functional.normalize(g32, dim=1)
    ws = ops.knn_workspace(B, N, D, k, dev)
    if fp8:
        Q, qs = ops.quantize_fp8_rows(q32)
        G, gs = ops.quantize_fp8_rows(g32)
        ops.knn_topk_fp8(Q, qs, G, gs, k, 0, ws)
        qd = Q.view(torch.float8_e4m3fn).double() * qs.double()[:, None]
        bound = ops.NORM_BOUND_FP8
        deq = lambda lo, hi: G[lo:hi].view(torch.float8_e4m3fn).double() * gs[lo:hi].double()[:, None]
    else:
        Q, G = q32.to(torch.bfloat16), g32.to(torch.bfloat16)
        ops.knn_topk(Q, G, k, 0, ws)
        qd = Q.double()
        bound = ops.NORM_BOUND_BF16
        deq = lambda lo, hi: G[lo:hi].double()
    S = ops.knn_scores_view(ws, B, N, D, k)                   # MFMA scores (split-K slabs summed and written back by the select)
    eps = 1.1 * D * 2.0 ** -24 * qd.norm(dim=1) * bound       # per query, as knn_err_rel() * |q| in knn.hip
    worst = 0.0
    for lo in range(0, N, 8192):                              # exact f64 scores, a slab of gallery rows at a time
        hi = min(N, lo + 8192)
        exact = qd @ deq(lo, hi).T
        worst = max(worst, float(((S[:, lo:hi].double() - exact).abs() / eps[:, None]).max()))
    print(f"\n[certificate model] {route:16s} {'e4m3' if fp8 else 'bf16'} {signs:6s}: max |S_mfma - exact| / eps = {worst:.4f}")
    assert worst <= 0.5
