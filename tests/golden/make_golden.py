"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own classes and
functions in the build container (the reference cannot travel to the GPU box; only these small
data files do).  Run from the repo root:  python tests/golden/make_golden.py

What is imported (read-only, from /root/reference):
  dinov2salad/dinov2salad_validation.py   -> DINOv2RegressionModel (head 8448->512->2, :36-52)
  angle_prediction/swin/swin_angle_validation.py -> mean_absolute_angular_error (:48-50)
Both modules import packages that are absent here (icecream, torchvision); empty placeholder
modules are registered for those names only so that the `import` statements succeed — nothing
under test touches them.  Model NAME fetches (from_pretrained / torch.hub) are never executed.
Also derived here: StandardScaler constants from cleaned_dataset_files/labels_train.csv with
sklearn (the fit at dinov2salad_finetuning.py:79-81), metrics of the reference's committed CSVs,
and the HF Swin pooler on a random-init SwinModel(SwinConfig()) (architecture only).
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name, stubs):
    added = []
    for s in stubs:
        if s not in sys.modules:
            sys.modules[s] = types.ModuleType(s)
            added.append(s)
    if "icecream" in stubs:
        sys.modules["icecream"].ic = lambda *a, **k: None
    if "torchvision" in stubs:
        tv, tr = sys.modules["torchvision"], types.ModuleType("torchvision.transforms")
        for n in ("Compose", "Resize", "ToTensor", "Normalize"):
            setattr(tr, n, lambda *a, **k: None)
        tv.transforms = tr
        sys.modules["torchvision.transforms"] = tr
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    finally:                      # placeholders must not outlive the import (transformers probes torchvision)
        for s in added + (["torchvision.transforms"] if "torchvision" in added else []):
            sys.modules.pop(s, None)
    return mod


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def head_dinov2salad():
    """Reference head class on seeded weights/inputs.  W1 is 17 MB, so the file stores the seed and
    checksums; the test regenerates weights/inputs with the same torch RNG calls."""
    mod = _load(os.path.join(REF, "dinov2salad/dinov2salad_validation.py"), "ref_dinov2salad_validation",
                ["icecream", "torchvision"])

    class Stub(torch.nn.Module):        # stands where the hub model would be; identity on features
        def forward(self, x):
            return x

    torch.manual_seed(1234)
    model = mod.DINOv2RegressionModel(Stub()).eval()
    g = torch.Generator().manual_seed(4321)
    x = torch.nn.functional.normalize(torch.randn(8, 8448, generator=g), dim=1)
    with torch.no_grad():
        y = model(x)
    sd = model.state_dict()
    out = {
        "source": "dinov2salad/dinov2salad_validation.py:36-52 (DINOv2RegressionModel imported)",
        "weight_seed": 1234, "input_seed": 4321,
        "recipe": "torch.manual_seed(weight_seed); nn.Sequential(Linear(8448,512),ReLU,Linear(512,2)) "
                  "constructed exactly as the reference does; x = normalize(randn(8,8448, Generator(input_seed)))",
        "state_dict_keys": {k: list(v.shape) for k, v in sd.items()},
        "sha256": {k: sha(v) for k, v in sd.items()},
        "x_sha256": sha(x),
        "x_first_row_head": x[0, :8].tolist(),
        "outputs": y.tolist(),
    }
    with open(os.path.join(OUT, "head_dinov2salad.json"), "w") as f:
        json.dump(out, f, indent=1)


def maae():
    mod = _load(os.path.join(REF, "angle_prediction/swin/swin_angle_validation.py"), "ref_swin_angle_validation",
                ["icecream"])
    g = torch.Generator().manual_seed(7)
    pred = torch.rand(64, generator=g) * 360
    true = torch.rand(64, generator=g) * 360
    cases = {"random64": (pred, true),
             "wrap": (torch.tensor([10., 350., 180.]), torch.tensor([350., 10., 0.])),
             "edge": (torch.tensor([0., 359.999, 180., 90.]), torch.tensor([359.999, 0., 0., 270.]))}
    out = {"source": "angle_prediction/swin/swin_angle_validation.py:48-50 (function imported)", "cases": {}}
    for k, (p, t) in cases.items():
        out["cases"][k] = {"pred_deg": p.tolist(), "true_deg": t.tolist(),
                           "maae": float(mod.mean_absolute_angular_error(p, t))}
    with open(os.path.join(OUT, "maae.json"), "w") as f:
        json.dump(out, f, indent=1)


def scaler_and_csv_metrics():
    import pandas as pd
    from sklearn.preprocessing import StandardScaler
    tr = pd.read_csv(os.path.join(REF, "cleaned_dataset_files/labels_train.csv"))
    sc = StandardScaler().fit(tr[["latitude", "longitude"]].values)      # dinov2salad_finetuning.py:79-81
    z32 = np.array([[0.0, 0.0], [1.0, -1.0], [-2.5, 0.125], [0.3333333, 2.7182817]], dtype=np.float32)
    z64 = z32.astype(np.float64)
    val = pd.read_csv(os.path.join(REF, "cleaned_dataset_files/labels_val.csv"))
    preds = pd.read_csv(os.path.join(REF, "swin_transformer/results_csv/preds.csv"))
    val["ID"] = val["filename"].apply(lambda f: int(os.path.splitext(f)[0].split("_")[-1]))   # swin_validation.py:121-122
    j = preds.merge(val, on="ID", suffixes=("_p", "_t"))
    p = j[["latitude_p", "longitude_p"]].values
    t = j[["latitude_t", "longitude_t"]].values
    final_loss = 0.5 * (np.sum((p[:, 0] - t[:, 0]) ** 2) + np.sum((p[:, 1] - t[:, 1]) ** 2)) / len(p)   # swin_validation.py:100
    vp = pd.read_csv(os.path.join(REF, "swin_transformer/training_gemini_2_20250505_004059/validation_predictions.csv"))
    from sklearn.metrics import mean_absolute_error, mean_squared_error
    tt = vp[["true_latitude", "true_longitude"]].values
    pp = vp[["predicted_latitude", "predicted_longitude"]].values
    out = {
        "source": "sklearn StandardScaler on cleaned_dataset_files/labels_train.csv (fit as dinov2salad_finetuning.py:79-81); "
                  "metrics recomputed from the reference's committed CSVs",
        "n_train": int(len(tr)), "mean_": sc.mean_.tolist(), "scale_": sc.scale_.tolist(),
        "inverse_f32": {"z": z32.tolist(), "x": sc.inverse_transform(z32).astype(np.float64).tolist(),
                        "dtype": str(sc.inverse_transform(z32).dtype)},
        "inverse_f64": {"z": z64.tolist(), "x": sc.inverse_transform(z64).tolist()},
        "swin_tiny_preds_csv": {"n": int(len(j)), "final_loss": float(final_loss),
                                "mae_lat": float(np.mean(np.abs(p[:, 0] - t[:, 0]))),
                                "mae_lon": float(np.mean(np.abs(p[:, 1] - t[:, 1]))),
                                "first_rows_pred": p[:4].tolist(), "first_rows_true": t[:4].tolist()},
        "swin_base_validation_predictions_csv": {"n": int(len(vp)), "mse": float(mean_squared_error(tt, pp)),
                                                 "mae": float(mean_absolute_error(tt, pp)),
                                                 "mae_lat": float(mean_absolute_error(tt[:, 0], pp[:, 0])),
                                                 "mae_lon": float(mean_absolute_error(tt[:, 1], pp[:, 1])),
                                                 "first_rows_pred": pp[:4].tolist(), "first_rows_true": tt[:4].tolist()},
    }
    with open(os.path.join(OUT, "scaler_and_metrics.json"), "w") as f:
        json.dump(out, f, indent=1)


def swin_pool_head():
    """HF SwinModel pooler on a random-init Swin-T (architecture only; no NAME fetch): pre-LayerNorm
    last hidden state (hook on the final layernorm) -> pooler_output -> Linear(768,2) as in
    swin_validation.py:41-46 (+ the unit-normalised sin/cos variant, swin_angle_finetuning_sin_cos.py:58-62)."""
    from transformers import SwinConfig, SwinModel
    torch.manual_seed(0)
    bb = SwinModel(SwinConfig()).eval()
    reg = torch.nn.Linear(bb.config.hidden_size, 2)
    grabbed = {}
    bb.layernorm.register_forward_hook(lambda m, i, o: grabbed.__setitem__("pre", i[0].detach().clone()))
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        pooled = bb(pixel_values=x).pooler_output
        out = reg(pooled)
        out_unit = torch.nn.functional.normalize(out, dim=1, p=2, eps=1e-6)
    np.savez_compressed(os.path.join(OUT, "swin_pool_head.npz"),
                        pre_ln=grabbed["pre"].numpy().astype(np.float32),
                        gamma=bb.layernorm.weight.detach().numpy(), beta=bb.layernorm.bias.detach().numpy(),
                        eps=np.float64(bb.layernorm.eps), pooled=pooled.numpy(),
                        W=reg.weight.detach().numpy(), b=reg.bias.detach().numpy(),
                        out=out.numpy(), out_unit=out_unit.numpy())


def head_finetune():
    """The reference's head-only training step, run HERE with the reference's own model class: DINOv2RegressionModel
    (imported; same definition as dinov2salad_finetuning.py:21-37) around an identity extractor, `torch.optim.AdamW(
    model.parameters(), lr=...)` and `nn.MSELoss()` as at :95-96, the loop body of :119-125, batches of 16 (:89) from a
    seeded permutation, f32 as the reference runs it.  Stored: per-step losses and samples of the final parameters."""
    mod = _load(os.path.join(REF, "dinov2salad/dinov2salad_validation.py"), "ref_dinov2salad_validation_ft",
                ["icecream", "torchvision"])

    class Stub(torch.nn.Module):
        def forward(self, x):
            return x

    out = {"source": "DINOv2RegressionModel imported from dinov2salad/dinov2salad_validation.py:36-52 (= dinov2salad_finetuning.py:21-37); "
                     "optimizer / loss / loop body as dinov2salad_finetuning.py:95-96,119-125; torch " + torch.__version__ + " CPU f32",
           "weight_seed": 2468, "data_seed": 1357, "n": 64, "batch_size": 16, "epochs": 2,
           "recipe": "torch.manual_seed(weight_seed); model as the reference builds it; g = Generator(data_seed); x = normalize(randn(n, 8448, g)); "
                     "y = randn(n, 2, g); per epoch perm = randperm(n, g); batches perm[lo:lo+16]",
           "runs": {}}
    for name, lr in (("lr1e-5", 1e-5), ("lr1e-3", 1e-3)):
        torch.manual_seed(out["weight_seed"])
        model = mod.DINOv2RegressionModel(Stub())
        g = torch.Generator().manual_seed(out["data_seed"])
        x = torch.nn.functional.normalize(torch.randn(out["n"], 8448, generator=g), dim=1)
        y = torch.randn(out["n"], 2, generator=g)
        optimizer = torch.optim.AdamW(model.parameters(), lr=lr)
        loss_fn = torch.nn.MSELoss()
        model.train()
        losses, orders = [], []
        for epoch in range(out["epochs"]):
            perm = torch.randperm(out["n"], generator=g)
            orders.append(perm.tolist())
            for lo in range(0, out["n"], out["batch_size"]):
                idx = perm[lo:lo + out["batch_size"]]
                preds = model(x[idx])
                loss = loss_fn(preds, y[idx])
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                losses.append(float(loss.item()))
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        gi = torch.Generator().manual_seed(99)
        w1_idx = torch.randint(0, 512 * 8448, (64,), generator=gi)
        out["runs"][name] = {
            "lr": lr, "losses": losses, "orders": orders,
            "w1_sample_index": w1_idx.tolist(),
            "w1_sample": sd["regressor.0.weight"].reshape(-1)[w1_idx].double().tolist(),
            "b1_head": sd["regressor.0.bias"][:32].double().tolist(),
            "w2": sd["regressor.2.weight"].double().tolist(),
            "b2": sd["regressor.2.bias"].double().tolist(),
            "w1_abs_mean": float(sd["regressor.0.weight"].double().abs().mean()),
        }
    torch.manual_seed(out["weight_seed"])
    out["initial_sha256"] = {k: sha(v) for k, v in mod.DINOv2RegressionModel(Stub()).state_dict().items()}
    with open(os.path.join(OUT, "head_finetune.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "head_finetune":     # only this fixture (the others are unchanged)
        head_finetune()
        print("head_finetune.json written to", OUT)
        sys.exit(0)
    head_finetune()
    head_dinov2salad()
    maae()
    scaler_and_csv_metrics()
    swin_pool_head()
    print("golden files written to", OUT)
