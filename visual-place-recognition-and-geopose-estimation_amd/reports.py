"""CSV writers with the exact column layout / formatting of the reference's evaluation scripts
(SURVEY.md §8f-1), so downstream tooling that reads those files keeps working.

  write_id_sorted_preds      swin_transformer/swin_validation.py:121-134  (ID,latitude,longitude; sorted by ID)
  write_validation_csv       swin_transformer/val_and_test_swin_2.py:280-293  (float_format '%.6f')
  write_test_csv             swin_transformer/val_and_test_swin_2.py:326-342  (sorted by filename, '%.6f')
  format_metrics             the prints at val_and_test_swin_2.py:273-278
"""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np
import pandas as pd

from . import postproc


def extract_id(filename: str) -> int:
    """'img_0123.jpg' -> 123 (swin_validation.py:121-122)."""
    return int(os.path.splitext(filename)[0].split("_")[-1])


def write_id_sorted_preds(path: str, filenames: Sequence[str], preds: np.ndarray) -> pd.DataFrame:
    df = pd.DataFrame(np.asarray(preds), columns=["latitude", "longitude"])
    df["ID"] = [extract_id(f) for f in filenames]
    df = df[["ID", "latitude", "longitude"]].sort_values(by="ID")
    df.to_csv(path, index=False)
    return df


def write_validation_csv(path: str, filenames: Sequence[str], targets: np.ndarray, preds: np.ndarray) -> pd.DataFrame:
    t, p = np.asarray(targets), np.asarray(preds)
    df = pd.DataFrame({"filename": list(filenames), "true_latitude": t[:, 0], "true_longitude": t[:, 1],
                       "predicted_latitude": p[:, 0], "predicted_longitude": p[:, 1]})
    df["error_latitude"] = np.abs(df["true_latitude"] - df["predicted_latitude"])
    df["error_longitude"] = np.abs(df["true_longitude"] - df["predicted_longitude"])
    df.to_csv(path, index=False, float_format="%.6f")
    return df


def write_test_csv(path: str, filenames: Sequence[str], preds: np.ndarray) -> pd.DataFrame:
    p = np.asarray(preds)
    df = pd.DataFrame({"filename": list(filenames), "predicted_latitude": p[:, 0], "predicted_longitude": p[:, 1]})
    df = df.sort_values(by="filename")
    df.to_csv(path, index=False, float_format="%.6f")
    return df


def format_metrics(preds: np.ndarray, targets: np.ndarray) -> str:
    m = postproc.regression_metrics(preds, targets)
    return ("--- Evaluation Results on Validation Set (Original Scale) ---\n"
            f"  Mean Squared Error (MSE): {m['mse']:.6f}\n"
            f"  Root Mean Squared Error (RMSE): {m['rmse']:.6f}\n"
            f"  Mean Absolute Error (MAE): {m['mae']:.6f}\n"
            f"  MAE Latitude: {m['mae_lat']:.6f}\n"
            f"  MAE Longitude: {m['mae_lon']:.6f}")


def write_angle_validation_csv(path: str, filenames: Sequence[str], true_deg, pred_deg) -> pd.DataFrame:
    """filename,true_angle,predicted_angle,angular_error — angle_prediction/efficient_net/validation_script.py:213-220
    (the only angle-validation CSV layout in the reference; committed example:
    angle_prediction/efficient_net/final_csvs/validation_predictions.csv).  Predictions are f32 values widened to
    Python floats (the script's `.cpu().tolist()`), the error column is recomputed from the two columns in f64."""
    pred = [float(np.float32(p)) for p in np.asarray(pred_deg).ravel()]
    true = np.asarray(true_deg).ravel().tolist()
    df = pd.DataFrame({"filename": list(filenames), "true_angle": true, "predicted_angle": pred})
    df["angular_error"] = df.apply(lambda row: min(abs(row["predicted_angle"] - row["true_angle"]),
                                                   360 - abs(row["predicted_angle"] - row["true_angle"])), axis=1)
    df.to_csv(path, index=False)
    return df


def write_angle_test_csv(path: str, filenames: Sequence[str], pred_deg) -> pd.DataFrame:
    """filename,predicted_angle_degrees sorted by filename — angle_prediction/efficient_net/test_script.py:269-276."""
    df = pd.DataFrame({"filename": list(filenames),
                       "predicted_angle_degrees": [float(np.float32(p)) for p in np.asarray(pred_deg).ravel()]})
    df = df.sort_values(by="filename")
    df.to_csv(path, index=False)
    return df
