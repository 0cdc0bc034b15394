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
