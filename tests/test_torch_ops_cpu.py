"""CPU: the torch.library operator layer (vpr_amd/torch_ops.py) — every op is registered under
torch.ops.vpr with a schema and a fake (meta) implementation, and the modules that mirror the reference's nn.Module seam
(dinov2salad_validation.py:49-52, swin_validation.py:43-46) trace with fake GPU tensors, no GPU and no kernel launch."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode


def test_every_op_is_registered_with_a_schema():
    from vpr_amd import torch_ops
    for name in torch_ops.OPS:
        op = getattr(torch.ops.vpr, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"vpr::{name}("), schema
    assert "!)? uncertified" in str(torch.ops.vpr.knn_topk.default._schema)                  # the one mutated argument
    assert "Tensor? W1" in str(torch.ops.vpr.pose_head.default._schema)
    sch = str(torch.ops.vpr.head_train_epoch.default._schema)                                # six tensors updated in place
    assert sch.count("!") == 6 and all(f"!) {n}" in sch for n in ("W1", "b1", "W2", "b2", "m", "v")), sch


def test_ops_refuse_cpu_tensors_through_the_dispatcher():
    """There is no CPU kernel behind the ops: a real CPU tensor reaches the wrapper's check and is refused loudly."""
    with pytest.raises(RuntimeError, match="GPU tensor"):
        torch.ops.vpr.pose_head(torch.zeros(1, 64), None, None, torch.zeros(2, 64), torch.zeros(2), -1)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        torch.ops.vpr.topk_merge(torch.zeros(2, 1, 3), torch.zeros(2, 1, 3, dtype=torch.int32))


def test_fake_implementations_give_shapes_and_dtypes():
    with FakeTensorMode():
        dev = "cuda"
        bf, f32 = torch.bfloat16, torch.float32
        C, h = 1024, 512
        mk = lambda *s, dtype=bf: torch.empty(*s, dtype=dtype, device=dev)
        w = [mk(2 * h, C), mk(2 * h, dtype=f32), mk(64, h), mk(64, dtype=f32), mk(128, h), mk(128, dtype=f32),
             mk(h, C), mk(h, dtype=f32), mk(256, h), mk(256, dtype=f32)]
        d, d16 = torch.ops.vpr.salad_aggregate(mk(4, 257, C), w, 1.0, 3)
        assert d.shape == (4, 8448) and d.dtype == f32 and d16.dtype == bf and d.device.type == "cuda"
        d, d16 = torch.ops.vpr.salad_aggregate_split(mk(4, 256, C), mk(4, C), w, 1.0, 3)
        assert d.shape == d16.shape == (4, 8448)
        w32 = [t.float() for t in w]
        d, _ = torch.ops.vpr.salad_aggregate_f32(mk(4, 256, C, dtype=f32), mk(4, C, dtype=f32), w32, 1.0, 3)
        assert d.shape == (4, 8448) and d.dtype == f32
        v, i, st = torch.ops.vpr.knn_topk(mk(64, 8448), mk(1000, 8448), 10, 5000, 1.002, None)
        assert v.shape == i.shape == (64, 10) and v.dtype == f32 and i.dtype == torch.int32 and st.shape == (64,)
        q8, qs = torch.ops.vpr.quantize_fp8_rows(mk(64, 8448, dtype=f32))
        assert q8.dtype == torch.uint8 and qs.shape == (64,)
        v, i, st = torch.ops.vpr.knn_topk_fp8(q8, qs, mk(1000, 8448, dtype=torch.uint8), mk(1000, dtype=f32), 10, 0, 1.0625,
                                              mk(1, dtype=torch.int32))
        assert v.shape == (64, 10)
        mv, mi = torch.ops.vpr.topk_merge(mk(8, 64, 10, dtype=f32), mk(8, 64, 10, dtype=torch.int32))
        assert mv.shape == mi.shape == (64, 10)
        out = torch.ops.vpr.pose_head(mk(64, 8448, dtype=f32), mk(1024, 8448, dtype=f32), mk(1024, dtype=f32), mk(4, 1024, dtype=f32),
                                      mk(4, dtype=f32), 2)
        assert out.shape == (64, 4)
        pooled, out = torch.ops.vpr.ln_meanpool_head(mk(256, 49, 1024), mk(1024, dtype=f32), mk(1024, dtype=f32), 1e-5,
                                                     mk(4, 1024, dtype=f32), mk(4, dtype=f32), 2)
        assert pooled.shape == (256, 1024) and out.shape == (256, 4)
        pooled, out = torch.ops.vpr.ln_meanpool_head(mk(8, 144, 1024), mk(1024, dtype=f32), mk(1024, dtype=f32), 1e-5, None, None, -1)
        assert pooled.shape == (8, 1024) and out.shape == (8, 0)
        W1, W2 = mk(512, 8448, dtype=f32), mk(2, 512, dtype=f32)
        n_state = 512 * 8448 + 512 + 2 * 512 + 2
        losses = torch.ops.vpr.head_train_epoch(mk(6378, 8448, dtype=f32), mk(6378, 2, dtype=f32), mk(6378, dtype=torch.int32), 16,
                                                W1, mk(512, dtype=f32), W2, mk(2, dtype=f32), mk(n_state, dtype=f32),
                                                mk(n_state, dtype=f32), 1, 1e-5, 0.9, 0.999, 1e-8, 1e-2)
        assert losses.shape == (399,) and losses.dtype == f32            # ceil(6378 / 16): the reference's batches per epoch


def test_modules_trace_with_fake_tensors():
    """The reference-shaped modules under FakeTensorMode: SALAD aggregator (both token layouts, both precisions), the
    DINOv2+SALAD regression model's head, the fused 4-wide head, the Swin linear / unit sin-cos heads.  Shapes flow
    through torch.ops.vpr.*; nothing touches a device."""
    from vpr_amd import modules
    from vpr_amd.backbone import SplitTokens
    agg = modules.SaladAggregator(768)
    agg.pack()                                                     # real CPU tensors (the dustbin read is a host value)
    agg.pack_f32()
    pos = torch.nn.Sequential(torch.nn.Linear(8448, 512), torch.nn.ReLU(), torch.nn.Linear(512, 2))
    ang = torch.nn.Sequential(torch.nn.Linear(8448, 512), torch.nn.ReLU(), torch.nn.Linear(512, 2))
    fused = modules.FusedGeoPoseHead(pos, ang)
    fused.pack()
    reg = modules.DINOv2RegressionModel(torch.nn.Identity())
    with FakeTensorMode(allow_non_fake_inputs=True):
        tok = torch.empty(5, 257, 768, dtype=torch.bfloat16, device="cuda")
        d, d16 = agg(tok, want_bf16=True)
        assert d.shape == (5, 8448) and d16.dtype == torch.bfloat16
        d2 = agg(SplitTokens(tok[:, 1:].contiguous(), tok[:, 0].contiguous()))
        assert d2.shape == (5, 8448)
        d3 = agg(tok.float())                                     # f32 tokens -> the f32-accurate aggregation
        assert d3.shape == (5, 8448) and d3.dtype == torch.float32
        assert fused(d).shape == (5, 4)
        assert reg(d).shape == (5, 2)                             # feature_extractor = Identity: the regressor seam alone


def test_modules_export_through_torch_export():
    """torch.export of the head module: the graph holds ONE call_function node, torch.ops.vpr.pose_head.default."""
    from vpr_amd import modules
    pos = torch.nn.Sequential(torch.nn.Linear(8448, 512), torch.nn.ReLU(), torch.nn.Linear(512, 2))
    ang = torch.nn.Sequential(torch.nn.Linear(8448, 512), torch.nn.ReLU(), torch.nn.Linear(512, 2))
    fused = modules.FusedGeoPoseHead(pos, ang).eval()
    fused.pack()
    W1, b1, W2, b2 = fused._packed

    class Head(torch.nn.Module):
        def forward(self, x):
            return torch.ops.vpr.pose_head(x, W1, b1, W2, b2, 2)

    ep = torch.export.export(Head(), (torch.empty(8, 8448, device="meta"),), strict=False)
    targets = [n.target for n in ep.graph.nodes if n.op == "call_function"]
    assert torch.ops.vpr.pose_head.default in targets


def test_bench_spawns_its_own_ranks(monkeypatch):
    """bench.spawn_ranks: `python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run with the
    driver's own flags (127.0.0.1 rendezvous, a free port, one process per GPU) and the same script arguments."""
    import os
    import sys
    import types
    import importlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5"])
    assert bench.spawn_ranks(8) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert "OMP_NUM_THREADS" in seen["env"]
