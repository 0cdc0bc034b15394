"""GPU: the sharded-retrieval code path on RCCL as far as one GPU can take it.

A one-rank `nccl` (= RCCL on ROCm) process group runs exactly the calls the N-rank job makes — `device_id=` init,
`all_gather_into_tensor` of the bf16 queries, of the packed (value, index) lists, on-device merge, teardown — so the
communicator set-up, dtype/layout handling and stream ordering have executed on the real backend; what one rank cannot
show is the xGMI transport and cross-rank ordering (covered by the 2-rank gloo tests in test_host_cpu.py /
test_bench_gpu.py and, when the driver has an 8-GPU node, by its scaling run).
Also here: {all-gather, local search, all-gather, merge} captured into ONE HIP graph (BASELINE config 5, SURVEY §7 step
5) and replayed == eager, for bf16 and e4m3 shards; the pipeline's head-on-a-side-stream schedule == the in-stream one."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_one_rank(dev):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def _data(dev, N, B, D=8448, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    gal = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
    pos = torch.randint(0, N, (B,), device=dev, generator=g)
    q = torch.nn.functional.normalize(gal[pos] + 0.1 * torch.randn(B, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
    return gal, q, pos


def test_sharded_search_through_rccl_one_rank(rccl_one_rank):
    from vpr_amd import ops
    from vpr_amd.retrieval import ShardedGallery
    dev = rccl_one_rank
    gal, q, pos = _data(dev, 5000, 16)
    rows = gal.to(torch.bfloat16)
    sg = ShardedGallery(rows, 5000, 0, 1, force_collectives=True)
    assert sg.collective
    v, i = sg.search_local_queries(q, 10)                     # all_gather (queries) -> search -> all_gather (top-k) -> merge
    v0, i0 = ops.knn_topk(q, rows, 10)
    assert torch.equal(i, i0) and torch.equal(v, v0)
    assert torch.equal(i[:, 0].long(), pos) and sg.uncertified_queries() == 0
    # e4m3 shard through the same collectives
    g8, gs = ops.quantize_fp8_rows(gal)
    sg8 = ShardedGallery(g8, 5000, 0, 1, scales=gs, force_collectives=True)
    v8, i8 = sg8.search_local_queries(q, 10)
    q8, qs = ops.quantize_fp8_rows(q.float())
    v8r, i8r = ops.knn_topk_fp8(q8, qs, g8, gs, 10)
    assert torch.equal(i8, i8r) and torch.equal(v8, v8r)


@pytest.mark.parametrize("fp8", [False, True])
def test_graphed_retrieval_with_collectives_equals_eager(rccl_one_rank, fp8):
    """One HIP graph holding the query all-gather, the shard search, the packed top-k all-gather and the merge."""
    from vpr_amd import ops
    from vpr_amd.retrieval import GraphedRetrieval, ShardedGallery
    dev = rccl_one_rank
    gal, q, pos = _data(dev, 20000, 32, seed=3 + fp8)
    if fp8:
        g8, gs = ops.quantize_fp8_rows(gal)
        sg = ShardedGallery(g8, 20000, 0, 1, scales=gs, force_collectives=True)
    else:
        sg = ShardedGallery(gal.to(torch.bfloat16), 20000, 0, 1, force_collectives=True)
    gr = GraphedRetrieval(sg, 32, 10)
    for trial in range(3):                                    # replay on fresh queries each time
        g = torch.Generator(device=dev).manual_seed(200 + trial)
        pos = torch.randint(0, 20000, (32,), device=dev, generator=g)
        q = torch.nn.functional.normalize(gal[pos] + 0.1 * torch.randn(32, 8448, device=dev, generator=g), dim=1).to(torch.bfloat16)
        v_g, i_g = gr(q)
        v_g, i_g = v_g.clone(), i_g.clone()
        v_e, i_e = sg.search_local_queries(q, 10)
        assert torch.equal(i_g, i_e) and torch.equal(v_g, v_e), trial
        assert torch.equal(i_g[:, 0].long(), pos)


def test_graphed_retrieval_local_only(dev):
    """Without a process group: the graph holds the local search (what one GPU of an unsharded deployment replays)."""
    from vpr_amd import ops
    from vpr_amd.retrieval import GraphedRetrieval, ShardedGallery
    gal, q, pos = _data(dev, 9000, 8, seed=9)
    rows = gal.to(torch.bfloat16)
    sg = ShardedGallery(rows, 9000)
    gr = GraphedRetrieval(sg, 8, 5)
    v_g, i_g = gr(q)
    v_e, i_e = ops.knn_topk(q, rows, 5)
    assert torch.equal(i_g, i_e) and torch.equal(v_g, v_e) and torch.equal(i_g[:, 0].long(), pos)


def test_pipeline_head_on_side_stream_equals_in_stream(rccl_one_rank):
    """overlap_head (default when the gallery is sharded): the pose head runs on a side stream beside the two
    collectives and the search; same kernels, same inputs -> bit-identical StepOutput."""
    import torch.nn as nn
    from vpr_amd.modules import DinoV2Salad, FusedGeoPoseHead
    from vpr_amd.pipeline import VPRGeoPosePipeline
    from vpr_amd.retrieval import ShardedGallery
    dev = rccl_one_rank
    torch.manual_seed(0)
    ext = DinoV2Salad("vit_small").to(dev).to(torch.bfloat16).eval()
    ext.backbone.fold_layerscale()
    pos = nn.Sequential(nn.Linear(8448, 64), nn.ReLU(), nn.Linear(64, 2)).to(dev)
    ang = nn.Sequential(nn.Linear(8448, 64), nn.ReLU(), nn.Linear(64, 2)).to(dev)
    head = FusedGeoPoseHead(pos, ang, normalize=True)
    gal = torch.nn.functional.normalize(torch.randn(3000, 8448, device=dev), dim=1).to(torch.bfloat16)
    images = torch.randn(4, 3, 224, 224, device=dev).to(torch.bfloat16)
    outs = []
    for overlap in (True, False, True):
        sg = ShardedGallery(gal, 3000, 0, 1, force_collectives=True)
        pipe = VPRGeoPosePipeline(ext, head, sg, 5, overlap_head=overlap)
        o = pipe.step(images)
        torch.cuda.synchronize()
        outs.append(o)
    for o in outs[1:]:
        assert torch.equal(o.pose, outs[0].pose) and torch.equal(o.topk_indices, outs[0].topk_indices)
        assert torch.equal(o.topk_scores, outs[0].topk_scores) and torch.equal(o.descriptors, outs[0].descriptors)


@pytest.mark.parametrize("fp8", [False, True])
def test_pipeline_graph_retrieval_equals_eager(rccl_one_rank, fp8):
    """VPRGeoPosePipeline(graph_retrieval=True): the retrieval leg of a step is one HIP-graph replay (captured at the
    first step, collectives inside) — bit-identical StepOutput to the eager pipeline, step after step."""
    import torch.nn as nn
    from vpr_amd import ops
    from vpr_amd.modules import DinoV2Salad, FusedGeoPoseHead
    from vpr_amd.pipeline import VPRGeoPosePipeline
    from vpr_amd.retrieval import ShardedGallery
    dev = rccl_one_rank
    torch.manual_seed(1)
    ext = DinoV2Salad("vit_small").to(dev).to(torch.bfloat16).eval()
    ext.backbone.fold_layerscale()
    pos = nn.Sequential(nn.Linear(8448, 64), nn.ReLU(), nn.Linear(64, 2)).to(dev)
    ang = nn.Sequential(nn.Linear(8448, 64), nn.ReLU(), nn.Linear(64, 2)).to(dev)
    head = FusedGeoPoseHead(pos, ang, normalize=True)
    gal = torch.nn.functional.normalize(torch.randn(7000, 8448, device=dev), dim=1)
    if fp8:
        g8, gs = ops.quantize_fp8_rows(gal)
        make = lambda: ShardedGallery(g8, 7000, 0, 1, scales=gs, force_collectives=True)
    else:
        rows = gal.to(torch.bfloat16)
        make = lambda: ShardedGallery(rows, 7000, 0, 1, force_collectives=True)
    eager = VPRGeoPosePipeline(ext, head, make(), 5)
    graphed = VPRGeoPosePipeline(ext, head, make(), 5, graph_retrieval=True)
    graphed.knn_events = []
    for trial in range(3):
        images = torch.randn(4, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(trial)).to(torch.bfloat16)
        a, b = eager.step(images), graphed.step(images)
        torch.cuda.synchronize()
        assert torch.equal(a.topk_indices, b.topk_indices) and torch.equal(a.topk_scores, b.topk_scores), trial
        assert torch.equal(a.pose, b.pose) and torch.equal(a.descriptors, b.descriptors)
    assert len(graphed.knn_events) == 3 and len(graphed._graphed) == 1
    assert graphed.gallery.uncertified_queries() == 0


def test_sharded_gallery_exact_fallback_fixes_flagged_queries(dev):
    """ShardedGallery(exact_fallback=True) — what evaluate.calculate_retrieval_scores uses: a query whose certificate
    fails (64 near-tied rows in one level-0 chunk, test_knn_gpu._near_tie_problem) comes back exact; the device counter
    still records that it was flagged; the graphed form refuses the option."""
    from oracle import knn as oknn
    from test_knn_gpu import _near_tie_problem
    from vpr_amd.retrieval import GraphedRetrieval, ShardedGallery
    q, gal, rows = _near_tie_problem(3, 64)
    k = 10
    v_ref, i_ref = oknn.knn_topk(q, gal, k)
    bound = float(gal.float().norm(dim=1).max()) * 1.001
    plain = ShardedGallery(gal.to(dev), gal.shape[0], norm_bound=bound)
    plain.search(q.to(dev), k)
    assert plain.uncertified_queries() == 1
    sg = ShardedGallery(gal.to(dev), gal.shape[0], norm_bound=bound, exact_fallback=True)
    v, i = sg.search(q.to(dev), k)
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.cpu(), v_ref) and sg.uncertified_queries() == 1
    with pytest.raises(RuntimeError, match="not capturable"):
        GraphedRetrieval(sg, 2, k)


def test_whole_pipeline_step_replays_from_one_hip_graph(dev):
    """The whole hot path — backbone, SALAD, kNN against the shard (certificate included), fused head — captured as ONE
    HIP graph per batch shape (graphed.GraphedForward around VPRGeoPosePipeline.step) and replayed: bit-identical
    StepOutput to the eager step, on fresh images each time."""
    import torch.nn as nn
    from vpr_amd.graphed import GraphedForward
    from vpr_amd.modules import DinoV2Salad, FusedGeoPoseHead
    from vpr_amd.pipeline import VPRGeoPosePipeline
    from vpr_amd.retrieval import ShardedGallery
    torch.manual_seed(4)
    ext = DinoV2Salad("vit_small").to(dev).to(torch.bfloat16).eval()
    ext.backbone.fold_layerscale()
    pos = nn.Sequential(nn.Linear(8448, 64), nn.ReLU(), nn.Linear(64, 2)).to(dev)
    ang = nn.Sequential(nn.Linear(8448, 64), nn.ReLU(), nn.Linear(64, 2)).to(dev)
    head = FusedGeoPoseHead(pos, ang, normalize=True)
    gal = torch.nn.functional.normalize(torch.randn(9000, 8448, device=dev), dim=1).to(torch.bfloat16)
    pipe = VPRGeoPosePipeline(ext, head, ShardedGallery(gal, 9000), 5)
    step = GraphedForward(pipe.step, module=ext)
    for trial in range(3):
        images = torch.randn(8, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(trial)).to(torch.bfloat16)
        ref = pipe.step(images)
        out = step(images)
        torch.cuda.synchronize()
        assert step.fallback_reason is None
        assert torch.equal(out.topk_indices, ref.topk_indices) and torch.equal(out.topk_scores, ref.topk_scores), trial
        assert torch.equal(out.pose, ref.pose) and torch.equal(out.descriptors, ref.descriptors), trial
    assert step.graphs() == 1 and pipe.gallery.uncertified_queries() == 0
