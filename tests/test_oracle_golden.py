"""CPU: the oracle restatements against the golden vectors generated from the reference's own
classes / functions / committed CSVs (tests/golden/make_golden.py), and the product's host-side
post-processing against the same vectors."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import heads as oheads
from oracle import postproc as opost

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def regen_head(meta):
    """Re-create the reference-initialised head weights and inputs from the recorded seeds."""
    torch.manual_seed(meta["weight_seed"])
    reg = nn.Sequential(nn.Linear(8448, 512), nn.ReLU(), nn.Linear(512, 2))
    x = torch.nn.functional.normalize(torch.randn(8, 8448, generator=torch.Generator().manual_seed(meta["input_seed"])), dim=1)
    return reg, x


def test_head_golden_matches_reference_class():
    meta = json.load(open(os.path.join(G, "head_dinov2salad.json")))
    reg, x = regen_head(meta)
    sd = {"regressor." + k: v for k, v in reg.state_dict().items()}
    assert {k: list(v.shape) for k, v in sd.items()} == meta["state_dict_keys"]
    for k, v in sd.items():
        assert _sha(v) == meta["sha256"][k], f"regenerated {k} differs from the reference-initialised weights"
    assert _sha(x) == meta["x_sha256"]
    ref = torch.tensor(meta["outputs"], dtype=torch.float64)
    out = oheads.mlp_head(x, reg[0].weight.detach(), reg[0].bias.detach(), reg[2].weight.detach(), reg[2].bias.detach())
    assert (out - ref).abs().max().item() < 2e-6      # reference ran in fp32, oracle in fp64


def regen_finetune(meta):
    """Weights and data of tests/golden/head_finetune.json from its recipe (the weights are checked against the sha256 of the
    reference-constructed model)."""
    torch.manual_seed(meta["weight_seed"])
    reg = nn.Sequential(nn.Linear(8448, 512), nn.ReLU(), nn.Linear(512, 2))
    for k, v in reg.state_dict().items():
        assert _sha(v) == meta["initial_sha256"]["regressor." + k], f"regenerated {k} differs from the reference-initialised weights"
    g = torch.Generator().manual_seed(meta["data_seed"])
    x = torch.nn.functional.normalize(torch.randn(meta["n"], 8448, generator=g), dim=1)
    y = torch.randn(meta["n"], 2, generator=g)
    return reg, x, y


def check_finetune_against_golden(run, losses, W1, b1, W2, b2, steps):
    """Losses and final parameters of a training run against the reference-generated fixture: the reference ran in f32, so
    losses agree to 2e-5 relative and parameters to the AdamW bound 0.05 * lr * steps (tests/test_head_train_gpu.py)."""
    ref_l = np.array(run["losses"])
    assert np.abs(np.asarray(losses, dtype=np.float64) - ref_l).max() <= 2e-5 * np.abs(ref_l).max()
    tol = 0.05 * run["lr"] * steps
    W1 = np.asarray(W1, dtype=np.float64).reshape(-1)
    assert np.abs(W1[np.array(run["w1_sample_index"])] - np.array(run["w1_sample"])).max() <= tol
    assert np.abs(np.asarray(b1, dtype=np.float64)[:32] - np.array(run["b1_head"])).max() <= tol
    assert np.abs(np.asarray(W2, dtype=np.float64) - np.array(run["w2"])).max() <= tol
    assert np.abs(np.asarray(b2, dtype=np.float64) - np.array(run["b2"])).max() <= tol
    assert abs(np.abs(W1).mean() - run["w1_abs_mean"]) <= tol


def test_finetune_oracle_reproduces_the_reference_training_run():
    """oracle/finetune.py against tests/golden/head_finetune.json: the reference's DINOv2RegressionModel trained with its own
    optimizer / loss calls (tests/golden/make_golden.py head_finetune), 8 steps at lr 1e-5 (the script's) and 1e-3."""
    from oracle import finetune as oft
    meta = json.load(open(os.path.join(G, "head_finetune.json")))
    reg, x, y = regen_finetune(meta)
    for name, run in meta["runs"].items():
        st = oft.HeadState(*(p.detach().numpy() for p in (reg[0].weight, reg[0].bias, reg[2].weight, reg[2].bias)))
        losses = []
        for order in run["orders"]:
            order = np.array(order)
            for lo in range(0, meta["n"], meta["batch_size"]):
                idx = order[lo:lo + meta["batch_size"]]
                losses.append(oft.train_step(st, x.numpy()[idx], y.numpy()[idx], lr=run["lr"]))
        check_finetune_against_golden(run, losses, *st.p, steps=len(losses))


def test_maae_golden():
    d = json.load(open(os.path.join(G, "maae.json")))
    from vpr_amd import postproc
    for name, c in d["cases"].items():
        p, t = np.array(c["pred_deg"], dtype=np.float32), np.array(c["true_deg"], dtype=np.float32)
        assert abs(opost.maae_deg(p, t) - c["maae"]) < 1e-4, name
        assert abs(postproc.mean_absolute_angular_error(p, t) - c["maae"]) < 1e-4, name
    # sin/cos form agrees with the degree form (swin_angle_finetuning_gemini.py:131-146)
    c = d["cases"]["random64"]
    p, t = np.deg2rad(np.array(c["pred_deg"])), np.deg2rad(np.array(c["true_deg"]))
    psc, tsc = np.stack([np.sin(p), np.cos(p)], 1), np.stack([np.sin(t), np.cos(t)], 1)
    assert abs(opost.maae_sincos(psc, tsc) - c["maae"]) < 1e-3
    assert abs(opost.compute_angle_error(psc, tsc) - c["maae"]) < 1e-3
    assert np.allclose(postproc.sincos_to_degrees(psc), np.array(c["pred_deg"]) % 360, atol=1e-3)


def test_scaler_golden():
    d = json.load(open(os.path.join(G, "scaler_and_metrics.json")))
    from vpr_amd import postproc
    sc = postproc.LatLonScaler.campus()
    assert np.allclose(sc.mean_, d["mean_"], rtol=0, atol=1e-9) and np.allclose(sc.scale_, d["scale_"], rtol=0, atol=1e-9)
    z32 = np.array(d["inverse_f32"]["z"], dtype=np.float32)
    x32 = sc.inverse_transform(z32)
    assert x32.dtype == np.float32 and d["inverse_f32"]["dtype"] == "float32"
    assert np.array_equal(x32.astype(np.float64), np.array(d["inverse_f32"]["x"]))       # bit-exact fp32 semantics
    assert np.array_equal(opost.inverse_transform(z32, sc.mean_, sc.scale_), x32)
    z64 = np.array(d["inverse_f64"]["z"], dtype=np.float64)
    assert np.allclose(sc.inverse_transform(z64), np.array(d["inverse_f64"]["x"]), rtol=0, atol=1e-9)
    assert np.allclose(sc.transform(sc.inverse_transform(z64)), z64, atol=1e-12)


def test_scaler_fit_from_labels():
    d = json.load(open(os.path.join(G, "scaler_and_metrics.json")))
    from vpr_amd import postproc
    rng = np.random.default_rng(0)
    lab = rng.normal([219658.0, 143506.0], [900.0, 1200.0], size=(500, 2))
    m, s = opost.fit_scaler(lab)
    sc = postproc.LatLonScaler.fit(lab)
    assert np.allclose(sc.mean_, m) and np.allclose(sc.scale_, s)
    assert np.allclose(s, lab.std(axis=0, ddof=0))
    assert d["n_train"] == 6378


def test_metrics_of_committed_csvs():
    """final_loss / MSE / MAE formulas reproduce the numbers derived from the reference's CSVs."""
    d = json.load(open(os.path.join(G, "scaler_and_metrics.json")))
    from vpr_amd import postproc
    a = d["swin_tiny_preds_csv"]
    assert abs(a["final_loss"] - 154665.836) < 1e-2 and a["n"] == 362
    p, t = np.array(a["first_rows_pred"]), np.array(a["first_rows_true"])
    assert abs(postproc.final_loss(p, t) - opost.final_loss(p, t)) < 1e-9
    assert abs(postproc.final_loss(p, t) - 0.5 * np.sum((p - t) ** 2) / len(p)) < 1e-9
    b = d["swin_base_validation_predictions_csv"]
    assert abs(b["mse"] - 20833.2169) < 1e-3
    m = postproc.regression_metrics(np.array(b["first_rows_pred"]), np.array(b["first_rows_true"]))
    assert abs(m["rmse"] - np.sqrt(m["mse"])) < 1e-12 and m["mae"] == pytest.approx((m["mae_lat"] + m["mae_lon"]) / 2)


def test_swin_pooler_golden():
    z = np.load(os.path.join(G, "swin_pool_head.npz"))
    pre = torch.from_numpy(z["pre_ln"])
    gamma, beta = torch.from_numpy(z["gamma"]), torch.from_numpy(z["beta"])
    pooled, out = oheads.ln_meanpool_head(pre, gamma, beta, float(z["eps"]), torch.from_numpy(z["W"]), torch.from_numpy(z["b"]))
    assert (pooled - torch.from_numpy(z["pooled"]).double()).abs().max().item() < 5e-6
    assert (out - torch.from_numpy(z["out"]).double()).abs().max().item() < 5e-6
    _, out_u = oheads.ln_meanpool_head(pre, gamma, beta, float(z["eps"]), torch.from_numpy(z["W"]), torch.from_numpy(z["b"]), 0)
    assert (out_u - torch.from_numpy(z["out_unit"]).double()).abs().max().item() < 5e-6
