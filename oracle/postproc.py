"""CPU restatement of the reference's host-side post-processing — TEST ORACLE.

  inverse_transform      scaler.inverse_transform(preds)   dinov2salad/dinov2salad_validation.py:84,
                                                           swin_transformer/swin_validation.py:82
                         (sklearn StandardScaler: X * scale_ + mean_, dtype of X is kept)
  fit_scaler             StandardScaler().fit(labels)      dinov2salad/dinov2salad_finetuning.py:79-81
  final_loss             0.5*(sum dlat^2 + sum dlon^2)/N   dinov2salad/dinov2salad_validation.py:101
  sincos_to_deg          atan2(sin, cos) -> deg -> (+360) % 360
                                                           angle_prediction/swin/swin_angle_finetuning_gemini.py:134-136
                         rad2deg(atan2) % 360              angle_prediction/dinov2salad/dino_v2_gemini.py:135-141
  maae_deg               mean(min(|d|, 360-|d|))           angle_prediction/swin/swin_angle_validation.py:48-50
  maae_sincos            the sin/cos form                  angle_prediction/swin/swin_angle_finetuning_gemini.py:131-146
  compute_angle_error    rad2deg(|atan2 diff|) % 360       angle_prediction/swin/swin_angle_finetuning_sin_cos.py:72-76
"""
import numpy as np


def fit_scaler(labels: np.ndarray):
    labels = np.asarray(labels, dtype=np.float64)
    return labels.mean(axis=0), labels.std(axis=0)       # population std (ddof=0), as sklearn


def inverse_transform(x: np.ndarray, mean_: np.ndarray, scale_: np.ndarray) -> np.ndarray:
    x = np.asarray(x)
    out = x * scale_.astype(x.dtype) if x.dtype == np.float32 else x * scale_
    return (out + (mean_.astype(x.dtype) if x.dtype == np.float32 else mean_)).astype(x.dtype)


def final_loss(preds: np.ndarray, targets: np.ndarray) -> float:
    return float(0.5 * (np.sum((preds[:, 0] - targets[:, 0]) ** 2) + np.sum((preds[:, 1] - targets[:, 1]) ** 2)) / len(preds))


def sincos_to_deg(sc: np.ndarray) -> np.ndarray:
    deg = np.rad2deg(np.arctan2(sc[:, 0], sc[:, 1]))
    return (deg + 360.0) % 360.0


def maae_deg(pred_deg: np.ndarray, true_deg: np.ndarray) -> float:
    d = np.abs(pred_deg - true_deg)
    return float(np.mean(np.minimum(d, 360.0 - d)))


def maae_sincos(pred_sc: np.ndarray, true_sc: np.ndarray) -> float:
    return maae_deg(sincos_to_deg(pred_sc), sincos_to_deg(true_sc))


def compute_angle_error(pred_sc: np.ndarray, true_sc: np.ndarray) -> float:
    pa = np.arctan2(pred_sc[:, 0], pred_sc[:, 1])
    ta = np.arctan2(true_sc[:, 0], true_sc[:, 1])
    diff = np.rad2deg(np.abs(pa - ta)) % 360.0
    return float(np.mean(np.minimum(diff, 360.0 - diff)))
