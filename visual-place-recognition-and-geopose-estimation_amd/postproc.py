"""Host-side post-processing and metrics of the geopose path (numpy, as in the reference).

Mirrors, formula for formula:
  LatLonScaler.inverse_transform   scaler.inverse_transform(preds)
        dinov2salad/dinov2salad_validation.py:84, swin_transformer/swin_validation.py:82
        (sklearn StandardScaler fitted at dinov2salad/dinov2salad_finetuning.py:79-81; fp32 in ->
        fp32 out, so lat/lon carry 0.0156 ulp at 2.2e5 — pass float64 for exact de-normalisation)
  final_loss                       dinov2salad/dinov2salad_validation.py:101
  regression_metrics               MSE / RMSE / MAE(+lat, lon)  swin_transformer/val_and_test_swin_2.py:268-272
  sincos_to_degrees                angle_prediction/swin/swin_angle_finetuning_gemini.py:134-136
  mean_absolute_angular_error      angle_prediction/swin/swin_angle_validation.py:48-50
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass

import numpy as np

# StandardScaler constants implied by cleaned_dataset_files/labels_train.csv (N = 6378),
# the file the reference fits its scaler on (dinov2salad_finetuning.py:79-81).
CAMPUS_MEAN = (219658.4252116651, 143506.67654437126)
CAMPUS_SCALE = (918.58972058316, 1190.858018520488)


@dataclass
class LatLonScaler:
    mean_: np.ndarray
    scale_: np.ndarray

    @classmethod
    def campus(cls) -> "LatLonScaler":
        return cls(np.asarray(CAMPUS_MEAN, dtype=np.float64), np.asarray(CAMPUS_SCALE, dtype=np.float64))

    @classmethod
    def fit(cls, labels: np.ndarray) -> "LatLonScaler":
        labels = np.asarray(labels, dtype=np.float64)
        return cls(labels.mean(axis=0), labels.std(axis=0))

    @classmethod
    def load_json(cls, path: str) -> "LatLonScaler":
        with open(path) as f:
            d = json.load(f)
        return cls(np.asarray(d["mean_"], dtype=np.float64), np.asarray(d["scale_"], dtype=np.float64))

    def transform(self, x: np.ndarray) -> np.ndarray:
        x = np.asarray(x)
        return ((x - self.mean_) / self.scale_).astype(x.dtype if x.dtype.kind == "f" else np.float64)

    def inverse_transform(self, x: np.ndarray) -> np.ndarray:
        """sklearn semantics: X * scale_ + mean_, computed and returned in X's float dtype."""
        x = np.asarray(x)
        dt = x.dtype if x.dtype.kind == "f" else np.float64
        return (x.astype(dt) * self.scale_.astype(dt) + self.mean_.astype(dt)).astype(dt)


def final_loss(preds: np.ndarray, targets: np.ndarray) -> float:
    d = np.asarray(preds) - np.asarray(targets)
    return float(0.5 * (np.sum(d[:, 0] ** 2) + np.sum(d[:, 1] ** 2)) / len(d))


def regression_metrics(preds: np.ndarray, targets: np.ndarray) -> dict:
    p, t = np.asarray(preds, dtype=np.float64), np.asarray(targets, dtype=np.float64)
    mse = float(np.mean((p - t) ** 2))
    return {"mse": mse, "rmse": math.sqrt(mse), "mae": float(np.mean(np.abs(p - t))),
            "mae_lat": float(np.mean(np.abs(p[:, 0] - t[:, 0]))), "mae_lon": float(np.mean(np.abs(p[:, 1] - t[:, 1])))}


def sincos_to_degrees(sincos: np.ndarray) -> np.ndarray:
    """[sin, cos] -> degrees in [0, 360)."""
    sincos = np.asarray(sincos)
    return (np.rad2deg(np.arctan2(sincos[:, 0], sincos[:, 1])) + 360.0) % 360.0


def mean_absolute_angular_error(pred_deg: np.ndarray, true_deg: np.ndarray) -> float:
    d = np.abs(np.asarray(pred_deg) - np.asarray(true_deg))
    return float(np.mean(np.minimum(d, 360.0 - d)))


def recall_at_k(topk_idx: np.ndarray, positives) -> float:
    """Fraction of queries with at least one positive among their top-k gallery indices.
    `positives[i]` is an int or a collection of gallery indices that count as correct for query i."""
    hits = 0
    for row, pos in zip(np.asarray(topk_idx), positives):
        pos = {int(pos)} if np.isscalar(pos) else {int(p) for p in pos}
        hits += bool(pos.intersection(int(r) for r in row))
    return hits / max(1, len(topk_idx))
