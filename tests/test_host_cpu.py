"""CPU: host-side mirror of the reference interface (state-dict keys, checkpoint forms, fused
head packing) and the sharded-retrieval plumbing over gloo with world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import heads as oheads
from oracle import knn as oknn


def test_state_dict_keys_match_reference():
    from vpr_amd import modules
    m = modules.DINOv2RegressionModel(nn.Identity())
    assert [k for k in m.state_dict()] == ["regressor.0.weight", "regressor.0.bias", "regressor.2.weight", "regressor.2.bias"]
    assert tuple(m.regressor[0].weight.shape) == (512, 8448) and tuple(m.regressor[2].weight.shape) == (2, 512)

    class FakeSwin(nn.Module):
        def __init__(self, h):
            super().__init__()
            self.config = type("C", (), {"hidden_size": h})()
            self.layernorm = nn.LayerNorm(h)
    s = modules.SwinRegressionModel(FakeSwin(768))
    assert {"regressor.weight", "regressor.bias"} <= set(s.state_dict())
    b = modules.SwinMLPRegressionModel(FakeSwin(1024))
    assert {"regressor.0.weight", "regressor.0.bias", "regressor.3.weight", "regressor.3.bias"} <= set(b.state_dict())
    assert tuple(b.regressor[0].weight.shape) == (512, 1024)
    agg = modules.SaladAggregator(768)
    keys = set(agg.state_dict())
    for k in ("score.0.weight", "score.3.weight", "cluster_features.0.weight", "cluster_features.3.bias",
              "token_features.0.weight", "token_features.2.bias", "dust_bin"):
        assert k in keys
    assert tuple(agg.score[0].weight.shape) == (512, 768, 1, 1) and tuple(agg.cluster_features[3].weight.shape) == (128, 512, 1, 1)


def test_checkpoint_forms(tmp_path):
    from vpr_amd import modules
    src = modules.DINOv2RegressionModel(nn.Identity())
    a, b = tmp_path / "wrapped.pth", tmp_path / "bare.pth"
    torch.save({"epoch": 3, "model_state_dict": src.state_dict(), "loss": 0.1}, a)    # dinov2salad_finetuning.py:130-135
    torch.save(src.state_dict(), b)                                                   # swin_attempt_2.py:255
    for path in (a, b):
        dst = modules.load_reference_checkpoint(modules.DINOv2RegressionModel(nn.Identity()), str(path))
        assert torch.equal(dst.regressor[0].weight, src.regressor[0].weight)


def test_fused_head_packing_is_block_diagonal():
    from vpr_amd import modules
    torch.manual_seed(0)
    pos = nn.Sequential(nn.Linear(64, 32), nn.ReLU(), nn.Linear(32, 2))
    ang = nn.Sequential(nn.Linear(64, 16), nn.ReLU(), nn.Dropout(0.0), nn.Linear(16, 2))
    W1, b1, W2, b2 = modules.FusedGeoPoseHead(pos, ang).pack()
    assert W1.shape == (48, 64) and W2.shape == (4, 48)
    assert torch.count_nonzero(W2[:2, 32:]) == 0 and torch.count_nonzero(W2[2:, :32]) == 0
    x = torch.randn(5, 64)
    fused = oheads.mlp_head(x, W1, b1, W2, b2, 2, dtype=torch.float32)
    with torch.no_grad():
        sep = torch.cat([pos(x), torch.nn.functional.normalize(ang(x), dim=1, eps=1e-6)], 1)
    assert torch.allclose(fused, sep, atol=1e-6)


def test_shard_bounds_cover_gallery():
    from vpr_amd.retrieval import shard_bounds
    for N, R in ((100000, 8), (12501, 7), (5, 8)):
        spans = [shard_bounds(N, r, R) for r in range(R)]
        assert spans[0][0] == 0 and spans[-1][1] == N
        assert all(spans[i][1] == spans[i + 1][0] for i in range(R - 1))


class OracleEngine:
    """Test double for the HIP engine: same contract, CPU oracle arithmetic."""
    def local_topk(self, q, gallery, k, index_base, scales=None):
        if gallery.dtype == torch.uint8:                       # e4m3 shard: quantise the queries like HipEngine does
            q8, qs = oknn.quantize_fp8_rows(q.float())
            return oknn.knn_topk_fp8(q8, qs, gallery, scales, k, index_base)
        return oknn.knn_topk(q, gallery, k, index_base)

    def merge(self, vals, idxs):
        return oknn.topk_merge(vals, idxs)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, D, B, k, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vpr_amd.retrieval import ShardedGallery, shard_bounds
    g = torch.Generator().manual_seed(0)
    gal = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1).to(torch.bfloat16)
    q_all = torch.nn.functional.normalize(torch.randn(world * B, D, generator=g), dim=1).to(torch.bfloat16)
    lo, hi = shard_bounds(N, rank, world)
    sg = ShardedGallery(gal[lo:hi].contiguous(), N, rank, world, engine=OracleEngine())
    v, i = sg.search_local_queries(q_all[rank * B:(rank + 1) * B].contiguous(), k)
    v_ref, i_ref = oknn.knn_topk(q_all, gal, k)
    ok = torch.equal(i, i_ref[rank * B:(rank + 1) * B]) and torch.equal(v, v_ref[rank * B:(rank + 1) * B])
    gathered = sg.gather_queries(q_all[rank * B:(rank + 1) * B].contiguous())
    ok = ok and torch.equal(gathered, q_all)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_retrieval_gloo_world2(world):
    """world = 3: uneven shards (301 rows -> 100 / 100 / 101) and a gathered batch that is not a power of two."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), 301, 64, 3, 5, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_recall_at_k():
    from vpr_amd import postproc
    idx = np.array([[3, 9], [4, 1], [7, 7]])
    assert postproc.recall_at_k(idx[:, :1], [3, 1, 7]) == pytest.approx(2 / 3)
    assert postproc.recall_at_k(idx, [3, 1, 7]) == 1.0
    assert postproc.recall_at_k(idx, [[5, 9], [0], [7]]) == pytest.approx(2 / 3)


# ------------------------------------------------------------------ gallery format / label transfer
def test_gallery_store_roundtrip_and_sharding(tmp_path):
    from vpr_amd import gallery
    g = torch.Generator().manual_seed(0)
    desc = torch.nn.functional.normalize(torch.randn(37, 128, generator=g), dim=1).to(torch.bfloat16)
    labels = np.stack([np.arange(37) * 10.0, np.arange(37) * -3.0, (np.arange(37) * 25.0) % 360, np.arange(37) % 4], 1)
    gallery.save_gallery(str(tmp_path / "gal"), desc, labels, filenames=[f"img_{i:04d}.jpg" for i in range(37)])
    parts = [gallery.load_gallery_shard(str(tmp_path / "gal"), torch.device("cpu"), r, 3) for r in range(3)]
    assert [p.index_base for p in parts] == [0, 12, 24] and sum(p.rows.shape[0] for p in parts) == 37
    assert torch.equal(torch.cat([p.rows for p in parts]), desc)
    assert np.array_equal(parts[1].labels, labels) and parts[0].dtype == "bf16" and parts[0].scales is None
    # fp8 form
    from oracle import knn as oknn_
    q8, sc = oknn_.quantize_fp8_rows(desc.float())
    gallery.save_gallery(str(tmp_path / "gal8"), q8, labels, scales=sc)
    p8 = gallery.load_gallery_shard(str(tmp_path / "gal8"), torch.device("cpu"), 1, 2)
    assert p8.dtype == "fp8_e4m3" and torch.equal(p8.rows, q8[18:]) and torch.equal(p8.scales, sc[18:])
    with pytest.raises(ValueError):
        gallery.save_gallery(str(tmp_path / "bad"), desc.float(), labels)


def test_labels_csv_and_positives():
    from vpr_amd import gallery, postproc
    here = os.path.dirname(os.path.abspath(__file__))
    labels, names = gallery.labels_from_csv(os.path.join(here, "golden", "labels_val_head48.csv"))
    assert labels.shape == (48, 4) and names[0].startswith("img_")
    pos_r = gallery.positives_by_region(labels[:5, 3], labels[:, 3])
    assert all(i in pos_r[i] for i in range(5))
    pos_d = gallery.positives_by_distance(labels[:5, :2], labels[:, :2], tau=50.0)
    assert all(i in pos_d[i] for i in range(5))
    # a query whose top-1 is itself is a hit under both rules; a far row is a miss under distance
    far = int(np.argmax(((labels[:, :2] - labels[0, :2]) ** 2).sum(1)))
    idx = np.array([[0], [far]])
    assert postproc.recall_at_k(idx, [pos_d[0], pos_d[0]]) == 0.5


def test_label_transfer_modes():
    from vpr_amd import gallery
    labels = np.array([[100.0, 10.0, 350.0, 1], [200.0, 20.0, 10.0, 1], [900.0, 90.0, 180.0, 2]])
    scores = torch.tensor([[0.9, 0.9, 0.1], [0.8, 0.2, -1.0]])
    idx = torch.tensor([[0, 1, 2], [2, 0, -1]], dtype=torch.int32)
    top1 = gallery.label_transfer(scores, idx, labels, "top1")
    assert np.allclose(top1, [[100, 10, 350], [900, 90, 180]])
    w = gallery.label_transfer(scores, idx, labels, "weighted", temperature=0.01)
    assert np.allclose(w[0, :2], [150.0, 15.0], atol=1e-6)        # equal weights on rows 0,1; row 2 negligible
    assert min(abs(w[0, 2] - 0.0), abs(w[0, 2] - 360.0)) < 1e-6   # circular mean of 350 and 10 deg
    assert np.allclose(w[1], [900, 90, 180], atol=1e-6)           # padding (-1) ignored


# ------------------------------------------------------------------ CSV writers
def test_csv_writers_reproduce_reference_files(tmp_path):
    """Round-trip the first rows of the reference's committed CSVs through our writers: the text
    must come back byte for byte (column order, sort order, float formatting)."""
    import pandas as pd
    from vpr_amd import reports
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    # validation_predictions.csv (val_and_test_swin_2.py:280-293)
    ref = open(os.path.join(here, "ref_validation_predictions_head.csv")).read()
    df = pd.read_csv(os.path.join(here, "ref_validation_predictions_head.csv"))
    out = tmp_path / "val.csv"
    reports.write_validation_csv(str(out), df["filename"], df[["true_latitude", "true_longitude"]].values,
                                 df[["predicted_latitude", "predicted_longitude"]].values)
    assert out.read_text() == ref
    # test_predictions_sorted.csv (:326-342) — feed shuffled rows, expect filename order back
    ref = open(os.path.join(here, "ref_test_predictions_sorted_head.csv")).read()
    df = pd.read_csv(os.path.join(here, "ref_test_predictions_sorted_head.csv")).sample(frac=1.0, random_state=0)
    out = tmp_path / "test.csv"
    reports.write_test_csv(str(out), df["filename"], df[["predicted_latitude", "predicted_longitude"]].values)
    assert out.read_text() == ref
    # preds.csv (swin_validation.py:121-134) — IDs from filenames, sorted by ID
    ref = open(os.path.join(here, "ref_preds_head.csv")).read()
    df = pd.read_csv(os.path.join(here, "ref_preds_head.csv")).sample(frac=1.0, random_state=1)
    names = [f"img_{i:04d}.jpg" for i in df["ID"]]
    out = tmp_path / "preds.csv"
    reports.write_id_sorted_preds(str(out), names, df[["latitude", "longitude"]].values)
    assert out.read_text() == ref
    # angle CSVs (angle_prediction/efficient_net/validation_script.py:213-220, test_script.py:269-276): rows of the
    # reference's committed final_csvs/validation_predictions.csv and test_pred.csv
    ref = open(os.path.join(here, "ref_angle_validation_head.csv")).read()
    df = pd.read_csv(os.path.join(here, "ref_angle_validation_head.csv"))
    out = tmp_path / "angle_val.csv"
    reports.write_angle_validation_csv(str(out), df["filename"], df["true_angle"].to_numpy(),
                                       df["predicted_angle"].to_numpy().astype(np.float32))
    assert out.read_text() == ref                                # incl. the recomputed angular_error column
    ref = open(os.path.join(here, "ref_angle_test_pred_head.csv")).read()
    df = pd.read_csv(os.path.join(here, "ref_angle_test_pred_head.csv")).sample(frac=1.0, random_state=2)
    out = tmp_path / "angle_test.csv"
    reports.write_angle_test_csv(str(out), df["filename"], df["predicted_angle_degrees"].to_numpy().astype(np.float32))
    assert out.read_text() == ref
    assert reports.extract_id("images_val/img_0042.jpg") == 42
    txt = reports.format_metrics(np.array([[1.0, 2.0], [3.0, 5.0]]), np.array([[1.5, 2.0], [2.0, 3.0]]))
    assert "MAE Latitude: 0.750000" in txt and "MAE Longitude: 1.000000" in txt


# ------------------------------------------------------------------ head-only fine-tuning (§8f-4)
def test_finetune_head_on_cached_descriptors(tmp_path):
    """Same optimiser / loss / batch order as a reference-style loop gives the same losses; the
    checkpoint it writes loads back through load_reference_checkpoint; the loss goes down."""
    from vpr_amd import finetune, modules
    torch.manual_seed(0)
    n = 96
    desc = torch.nn.functional.normalize(torch.randn(n, 8448), dim=1)
    w_true = torch.randn(8448, 2) * 3
    labels = (desc @ w_true).numpy() * np.array([900.0, 1200.0]) + np.array([219658.0, 143506.0])
    model = modules.DINOv2RegressionModel(nn.Identity())
    ref_head = nn.Sequential(nn.Linear(8448, 512), nn.ReLU(), nn.Linear(512, 2))
    ref_head.load_state_dict(model.regressor.state_dict())
    with pytest.raises(RuntimeError, match="no CPU fallback"):                 # the default engine is the HIP training step
        finetune.finetune_head(model, desc, labels, epochs=1, log=lambda s: None)
    out = finetune.finetune_head(model, desc, labels, epochs=3, batch_size=16, lr=1e-3, save_dir=str(tmp_path),
                                 val=(desc[:8], labels[:8]), seed=5, log=lambda s: None, engine="torch")
    # reference-style loop (dinov2salad_finetuning.py:95-128) on the same data and batch order
    mean, std = labels.mean(0), labels.std(0)
    y = torch.from_numpy(((labels - mean) / std).astype(np.float32))
    opt = torch.optim.AdamW(ref_head.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(5)
    losses = []
    for epoch in range(3):
        perm = torch.randperm(n, generator=g)
        tot = 0.0
        for lo in range(0, n, 16):
            idx = perm[lo:lo + 16]
            loss = nn.functional.mse_loss(ref_head(desc[idx]), y[idx])
            opt.zero_grad(); loss.backward(); opt.step()
            tot += float(loss)
        losses.append(tot / 6)
    got = [h["train_loss"] for h in out["history"]]
    assert np.allclose(got, losses, rtol=1e-5)
    assert got[-1] < got[0]
    assert np.allclose(out["scaler"].mean_, mean) and np.allclose(out["scaler"].scale_, std)
    re = modules.load_reference_checkpoint(modules.DINOv2RegressionModel(nn.Identity()), str(tmp_path / "checkpoint_2_.pth"))
    assert torch.equal(re.regressor[0].weight, model.regressor[0].weight)
    ck = torch.load(tmp_path / "checkpoint_2_.pth", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}


def test_sharded_gallery_fp8_shards_two_way_equals_unsharded():
    """Host logic for e4m3 shards (rows uint8 + per-row scales): two shards searched separately and merged give
    the unsharded answer; shape / argument validation."""
    from vpr_amd.retrieval import ShardedGallery, shard_bounds
    g = torch.Generator().manual_seed(5)
    N, B, D, k = 600, 7, 256, 4
    gal = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(torch.bfloat16)
    g8, gs = oknn.quantize_fp8_rows(gal)
    eng = OracleEngine()
    whole = ShardedGallery(g8, N, 0, 1, engine=eng, scales=gs)
    v, i = whole.search(q, k)
    parts_v, parts_i = [], []
    for r in range(2):
        lo, hi = shard_bounds(N, r, 2)
        pv, pi = eng.local_topk(q, g8[lo:hi], k, lo, gs[lo:hi])
        parts_v.append(pv); parts_i.append(pi)
    mv, mi = eng.merge(torch.stack(parts_v), torch.stack(parts_i))
    assert torch.equal(mi, i) and torch.equal(mv, v)
    with pytest.raises(ValueError):
        ShardedGallery(g8, N, 0, 1, engine=eng)                       # uint8 rows need scales
    with pytest.raises(ValueError):
        ShardedGallery(gal.to(torch.bfloat16), N, 0, 1, engine=eng, scales=gs)   # bf16 rows must not carry scales


def test_forced_collectives_single_rank_gloo_and_fp8_shard_helper(tmp_path):
    """One rank with force_collectives: both all-gathers and the merge still run (what bench.py --force-dist does on
    RCCL); sharded_gallery_from() carries an e4m3 shard's scales (it used to refuse fp8 shards)."""
    from vpr_amd.gallery import GalleryShard, sharded_gallery_from
    from vpr_amd.retrieval import ShardedGallery
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = torch.Generator().manual_seed(3)
        N, D, B, k = 300, 128, 5, 4
        gal = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
        q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(torch.bfloat16)
        sg = ShardedGallery(gal.to(torch.bfloat16), N, 0, 1, engine=OracleEngine(), force_collectives=True)
        assert sg.collective
        v, i = sg.search_local_queries(q, k)
        v_ref, i_ref = oknn.knn_topk(q, gal.to(torch.bfloat16), k)
        assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
        g8, gs = oknn.quantize_fp8_rows(gal)
        shard = GalleryShard(g8, gs, None, N, 0, "fp8_e4m3")
        sg8 = sharded_gallery_from(shard)
        sg8.engine = OracleEngine()
        v8, i8 = sg8.search(q, k)
        q8, qs = oknn.quantize_fp8_rows(q.float())
        v8r, i8r = oknn.knn_topk_fp8(q8, qs, g8, gs, k)
        assert torch.equal(i8, i8r) and torch.equal(v8, v8r)
        # collectives are only capturable on RCCL: the graphed form refuses a gloo group instead of failing mid-capture
        from vpr_amd.retrieval import GraphedRetrieval
        with pytest.raises(RuntimeError, match="RCCL"):
            GraphedRetrieval(sg, B, k)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("unit", [True, False])
def test_finetune_angle_head_matches_a_reference_style_loop(unit):
    """§8f-4 widened to the sin/cos heads: same loss (acos angular loss on unit vectors, swin_angle_finetuning_sin_cos.py
    :65-69, or MSE on the raw pair, swin_angle_finetuning_gemini.py:183), optimiser, scheduler and batch order as a
    loop written the reference's way -> same per-epoch losses; the error metric falls."""
    from vpr_amd import finetune
    torch.manual_seed(1)
    n, H = 144, 64
    feats = torch.randn(n, H)
    ang = (torch.atan2(feats[:, 0], feats[:, 1]) * 180 / np.pi) % 360                  # learnable from the features
    head = nn.Linear(H, 2)
    ref = nn.Linear(H, 2)
    ref.load_state_dict(head.state_dict())
    out = finetune.finetune_angle_head(head, feats, ang.numpy(), unit=unit, epochs=4, batch_size=48, lr=5e-2, seed=9,
                                       val=(feats[:32], ang[:32].numpy()), log=lambda s: None)
    a = torch.deg2rad(ang)
    tgt = torch.stack([torch.sin(a), torch.cos(a)], 1)                                  # [sin, cos] (:47)
    opt = torch.optim.AdamW(ref.parameters(), lr=5e-2)
    # the schedulers and clip norms of the two scripts (sin_cos.py:92-93,116; gemini.py:188,215)
    sched = (torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10) if unit else
             torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=1e-7))
    clip = 5.0 if unit else 1.0
    clipped = 0
    g = torch.Generator().manual_seed(9)
    losses = []
    for epoch in range(4):
        perm = torch.randperm(n, generator=g)
        tot = 0.0
        for lo in range(0, n, 48):
            idx = perm[lo:lo + 48]
            p = ref(feats[idx])
            if unit:
                p = torch.nn.functional.normalize(p, dim=1, p=2, eps=1e-6)
                cs = torch.clamp((p * tgt[idx]).sum(dim=1), -0.999999, 0.999999)
                loss = torch.mean(torch.rad2deg(torch.acos(cs)))
            else:
                loss = nn.functional.mse_loss(p, tgt[idx])
            if unit and torch.isnan(loss):
                continue
            opt.zero_grad()
            loss.backward()
            clipped += int(torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=clip) > clip)
            opt.step()
            tot += float(loss.detach())
        sched.step()
        losses.append(tot / 3)
    if unit:
        assert clipped > 0                       # the angular loss is in degrees: the clip at 5.0 is active, as in the reference
    got = [h["train_loss"] for h in out["history"]]
    assert np.allclose(got, losses, rtol=1e-5, atol=1e-6), (got, losses)
    assert out["history"][-1]["val_maae"] < out["history"][0]["val_maae"]
    assert torch.allclose(head.weight, ref.weight, atol=1e-6)
