"""Head-only fine-tuning on frozen SALAD descriptors.

Mirrors dinov2salad/dinov2salad_finetuning.py:79-135: StandardScaler fitted on the training
labels (:79-81), `Linear(8448,512)-ReLU-Linear(512,2)` trained with AdamW(lr=1e-5) and MSELoss
(:95-96), batches of 16 shuffled (:89), one checkpoint per epoch in the reference's dict format
(:130-135) — so `load_reference_checkpoint` / the reference's own validation script read them.

MI355X-first difference: the extractor is frozen, so its descriptors are computed ONCE by the HIP
path (backbone -> vpr_salad_aggregate) and cached in HBM ([N,8448] f32, 215 MB for the 6378
training images) instead of re-running the backbone every epoch as the reference does; the head
itself (8.6 MFLOP/image) trains with PyTorch autograd on those cached descriptors.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Iterable, Optional

import numpy as np
import torch
import torch.nn as nn

from .modules import DINOv2RegressionModel
from .postproc import LatLonScaler


@torch.no_grad()
def cache_descriptors(extractor: nn.Module, image_batches: Iterable[torch.Tensor]) -> torch.Tensor:
    """Frozen feature_extractor over all batches -> [N, 8448] f32 on the GPU."""
    return torch.cat([extractor(x).float() for x in image_batches])


def finetune_head(model: DINOv2RegressionModel, descriptors: torch.Tensor, labels: np.ndarray,
                  epochs: int = 100, batch_size: int = 16, lr: float = 1e-5, save_dir: Optional[str] = None,
                  val: Optional[tuple] = None, seed: int = 0, log: Callable[[str], None] = print) -> dict:
    """Trains model.regressor on cached descriptors.  labels [N,2] raw (lat, lon); they are
    standardised with a scaler fitted here (returned and, if save_dir, dumped as JSON).
    val = (val_descriptors, val_labels_raw) for the per-epoch de-normalised report."""
    dev = descriptors.device
    scaler = LatLonScaler.fit(labels)
    y = torch.from_numpy(scaler.transform(np.asarray(labels, dtype=np.float64)).astype(np.float32)).to(dev)
    head = model.regressor.to(dev).float()
    for p in head.parameters():
        p.requires_grad_(True)
    opt = torch.optim.AdamW(head.parameters(), lr=lr)
    loss_fn = nn.MSELoss()
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = descriptors.shape[0]
    history = []
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "latlon_scaler.json"), "w") as f:
            json.dump({"mean_": scaler.mean_.tolist(), "scale_": scaler.scale_.tolist()}, f)
    for epoch in range(epochs):
        head.train()
        perm = torch.randperm(n, generator=g).to(dev)
        total, nb = 0.0, 0
        for lo in range(0, n, batch_size):
            idx = perm[lo:lo + batch_size]
            loss = loss_fn(head(descriptors[idx]), y[idx])
            opt.zero_grad()
            loss.backward()
            opt.step()
            total += float(loss)
            nb += 1
        rec = {"epoch": epoch, "train_loss": total / max(nb, 1)}
        if val is not None:
            head.eval()
            with torch.no_grad():
                pv = head(val[0]).cpu().numpy()
            pv = scaler.inverse_transform(pv)
            rec["val_mae"] = float(np.mean(np.abs(pv - np.asarray(val[1]))))
        history.append(rec)
        log(f"Epoch {epoch + 1} - Train Loss: {rec['train_loss']:.4f}" + (f" - Val MAE: {rec['val_mae']:.2f}" if val else ""))
        if save_dir:
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                        "optimizer_state_dict": opt.state_dict(), "loss": loss.detach()},
                       os.path.join(save_dir, f"checkpoint_{epoch}_.pth"))
    for p in head.parameters():
        p.requires_grad_(False)
    return {"scaler": scaler, "history": history}
