"""Head-only fine-tuning on frozen SALAD descriptors.

Mirrors dinov2salad/dinov2salad_finetuning.py:79-135: StandardScaler fitted on the training
labels (:79-81), `Linear(8448,512)-ReLU-Linear(512,2)` trained with AdamW(lr=1e-5) and MSELoss
(:95-96), batches of 16 shuffled (:89), one checkpoint per epoch in the reference's dict format
(:130-135) — so `load_reference_checkpoint` / the reference's own validation script read them.

MI355X-first difference: the extractor is frozen, so its descriptors are computed ONCE by the HIP
path (backbone -> vpr_salad_aggregate) and cached in HBM ([N,8448] f32, 215 MB for the 6378
training images) instead of re-running the backbone every epoch as the reference does; on a GPU the
head's training step itself (forward, MSELoss, backward, AdamW: dinov2salad_finetuning.py:119-125) is
the HIP entry point vpr_head_train_step — three launches per batch, no host synchronisation inside an
epoch, the 17 MB gradient of W1 never written to memory — checked against oracle/finetune.py (pinned to
torch autograd + torch.optim.AdamW).  engine="torch" keeps the PyTorch-autograd loop (CPU tensors, or an
explicit A/B on the GPU: scripts/head_train_bench.py) and is never chosen silently.
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


@torch.no_grad()
def cache_descriptors_from_images(extractor: nn.Module, image_dir: str, filenames, *, batch_size: int = 64,
                                  device: str = "cuda", prep=None) -> torch.Tensor:
    """The same cache straight from image files, in file-list order: decode-ahead loader (loader.ImageBatchLoader) ->
    GPU resize / normalise (the fine-tuning transform, dinov2salad_finetuning.py:45-50: 224 bilinear, mean = std = 0.5)
    -> extractor forward replayed from one HIP graph per batch shape (graphed.GraphedForward)."""
    from .graphed import GraphedForward
    from .loader import ImageBatchLoader
    from .preprocess import HALF_MEAN, HALF_STD, ResizeNormalize
    dev = torch.device(device)
    extractor = extractor.to(dev).eval()
    prep = prep or ResizeNormalize(224, "bilinear", HALF_MEAN, HALF_STD, torch.bfloat16)
    fwd = GraphedForward(extractor) if dev.type == "cuda" else extractor
    filenames = list(filenames)
    out = None
    for idxs, _, u8 in ImageBatchLoader(image_dir, filenames, batch_size, dev):
        d = fwd(prep(u8)).float()
        if out is None:
            out = torch.empty((len(filenames), d.shape[1]), dtype=torch.float32, device=dev)
        out[torch.tensor(idxs, device=dev)] = d
    return out


def finetune_head(model: DINOv2RegressionModel, descriptors: torch.Tensor, labels: np.ndarray,
                  epochs: int = 100, batch_size: int = 16, lr: float = 1e-5, save_dir: Optional[str] = None,
                  val: Optional[tuple] = None, seed: int = 0, log: Callable[[str], None] = print,
                  engine: str = "auto", loss: str = "mse", huber_delta: float = 1.0, weight_decay: float = 1e-2,
                  lr_schedule: Optional[Callable[[int, list], float]] = None) -> dict:
    """Trains model.regressor on cached descriptors.  labels [N,2] raw (lat, lon); they are
    standardised with a scaler fitted here (returned and, if save_dir, dumped as JSON).
    val = (val_descriptors, val_labels_raw) for the per-epoch de-normalised report.
    engine: "hip" (= "auto", the default) = vpr_head_train_epoch — needs the descriptors on the GPU and a Linear-ReLU-Linear head
    of a supported shape, and RAISES otherwise: there is no silent fallback; "torch" = the same loop as PyTorch autograd +
    torch.optim.AdamW, only when asked for by name (the CPU tests of the host logic, the A/B of scripts/head_train_bench.py).
    Both engines draw the same batches (same seeded permutations) and write the same checkpoint format.
    loss: "mse" (dinov2salad_finetuning.py:96) or "huber" with huber_delta (nn.HuberLoss(delta): dinov2salad_finetuning_2.py:154,
    swin_attempt_2.py:158); weight_decay: AdamW's (:95 default 0.01; _2.py:153 passes it explicitly).
    lr_schedule(epoch, history) -> lr for that epoch (host-side schedules such as the ReduceLROnPlateau of _2.py:155,236 are a
    few lines of Python on the validation history; the learning rate is an argument of every training call)."""
    if loss not in ("mse", "huber"):
        raise ValueError(f"finetune_head: loss must be 'mse' or 'huber', got {loss!r}")
    dev = descriptors.device
    if engine not in ("auto", "hip", "torch"):
        raise ValueError(f"finetune_head: unknown engine {engine!r}")
    if engine == "auto":
        engine = "hip"
    if engine == "hip" and not descriptors.is_cuda:
        raise RuntimeError("finetune_head: the training step is a HIP kernel and needs the descriptors on the GPU (there is no CPU "
                           "fallback); engine='torch' runs the PyTorch-autograd loop instead, on any device")
    scaler = LatLonScaler.fit(labels)
    y = torch.from_numpy(scaler.transform(np.asarray(labels, dtype=np.float64)).astype(np.float32)).to(dev)
    head = model.regressor.to(dev).float()
    for p in head.parameters():
        p.requires_grad_(True)
    opt = torch.optim.AdamW(head.parameters(), lr=lr, weight_decay=weight_decay)
    loss_fn = nn.MSELoss() if loss == "mse" else nn.HuberLoss(delta=huber_delta)
    hip = _HipHeadTrainer(head, descriptors.float().contiguous(), y, opt, loss, huber_delta) if engine == "hip" else None
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = descriptors.shape[0]
    history = []
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "latlon_scaler.json"), "w") as f:
            json.dump({"mean_": scaler.mean_.tolist(), "scale_": scaler.scale_.tolist()}, f)
    for epoch in range(epochs):
        head.train()
        if lr_schedule is not None:
            for grp in opt.param_groups:
                grp["lr"] = float(lr_schedule(epoch, history))
        perm = torch.randperm(n, generator=g).to(dev)
        total, nb = 0.0, 0
        if engine == "hip":
            loss = hip.epoch(perm, batch_size)
            total, nb = float(loss.sum()), loss.numel()        # the epoch's only host synchronisation
            loss = loss[-1]
        else:
            for lo in range(0, n, batch_size):
                idx = perm[lo:lo + batch_size]
                loss = loss_fn(head(descriptors[idx]), y[idx])
                opt.zero_grad()
                loss.backward()
                opt.step()
                total += float(loss.detach())
                nb += 1
        rec = {"epoch": epoch, "train_loss": total / max(nb, 1)}
        if val is not None:
            head.eval()
            with torch.no_grad():
                if val[0].is_cuda:           # the per-epoch report runs the head as inference runs it: the HIP pose-head kernel
                    from . import ops
                    lin = [m for m in head if isinstance(m, nn.Linear)]
                    f = lambda p: p.detach().float().contiguous()
                    pv = ops.pose_head(val[0].float().contiguous(), f(lin[0].weight), f(lin[0].bias), f(lin[1].weight),
                                       f(lin[1].bias)).cpu().numpy()
                else:
                    pv = head(val[0]).cpu().numpy()
            pv = scaler.inverse_transform(pv)
            rec["val_mae"] = float(np.mean(np.abs(pv - np.asarray(val[1]))))
        history.append(rec)
        log(f"Epoch {epoch + 1} - Train Loss: {rec['train_loss']:.4f}" + (f" - Val MAE: {rec['val_mae']:.2f}" if val else ""))
        if save_dir:
            if hip is not None:
                hip.export_optimizer_state()
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                        "optimizer_state_dict": opt.state_dict(), "loss": loss.detach()},
                       os.path.join(save_dir, f"checkpoint_{epoch}_.pth"))
    if hip is not None:
        hip.export_optimizer_state()
    for p in head.parameters():
        p.requires_grad_(False)
    return {"scaler": scaler, "history": history, "engine": engine, "optimizer": opt}


class _HipHeadTrainer:
    """The HIP training step driven over an epoch: parameters are the nn.Linear tensors themselves (updated in place by the
    kernels), AdamW moments live in two flat buffers ([W1 | b1 | W2 | b2]) and are copied into a torch.optim.AdamW's state
    only when a checkpoint wants `optimizer.state_dict()` (the reference's checkpoint dict, :130-135)."""

    def __init__(self, head: nn.Module, X: torch.Tensor, Y: torch.Tensor, opt: torch.optim.Optimizer, loss: str = "mse",
                 huber_delta: float = 1.0):
        from . import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.vpr.*)
        lin = [m for m in head if isinstance(m, nn.Linear)]
        rest = [m for m in head if not isinstance(m, (nn.Linear, nn.ReLU))]
        if len(lin) != 2 or rest or not isinstance(head[1], nn.ReLU):
            raise RuntimeError("finetune_head(engine='hip'): the head must be Linear -> ReLU -> Linear (dinov2salad_finetuning.py:28-32)")
        self.ops, self.opt, self.X, self.Y = ops, opt, X, Y.contiguous()
        self.params = [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias]
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                raise RuntimeError("finetune_head(engine='hip'): head parameters must be contiguous f32 GPU tensors")
        self.W1, self.b1, self.W2, self.b2 = (p.detach() for p in self.params)     # aliases that share the parameters' version counters
        self.m, self.v = ops.head_train_state(self.W1, self.W2)
        self.loss, self.huber_delta = loss, float(huber_delta)
        self.step = 0

    @property
    def hyper(self) -> dict:            # read at every pass: a schedule may have changed the optimizer's learning rate
        grp = self.opt.param_groups[0]
        return dict(lr=grp["lr"], betas=tuple(grp["betas"]), eps=grp["eps"], weight_decay=grp["weight_decay"])

    def epoch(self, perm: torch.Tensor, batch_size: int) -> torch.Tensor:
        """One pass in the order `perm` (a permutation of the cached rows, built by the caller; ragged last batch kept, as
        DataLoader's default does: :89).  Returns the batch losses (device tensor; nothing here waits for the GPU)."""
        perm32 = perm.to(device=self.X.device, dtype=torch.int32).contiguous()
        h = self.hyper              # through the operator layer: the dispatcher sees the six in-place updates (version counters)
        losses = torch.ops.vpr.head_train_epoch(self.X, self.Y, perm32, batch_size, self.W1, self.b1, self.W2, self.b2, self.m,
                                                self.v, self.step + 1, h["lr"], h["betas"][0], h["betas"][1], h["eps"],
                                                h["weight_decay"], self.loss, self.huber_delta)
        self.step += losses.numel()
        return losses

    def export_optimizer_state(self) -> None:
        """Fill the torch optimizer's per-parameter state (step, exp_avg, exp_avg_sq) from the flat moment buffers."""
        ms = self.ops.head_train_state_views(self.m, self.W1, self.W2)
        vs = self.ops.head_train_state_views(self.v, self.W1, self.W2)
        for p, mm, vv in zip(self.params, ms, vs):
            self.opt.state[p] = {"step": torch.tensor(float(self.step)), "exp_avg": mm.clone().view_as(p),
                                 "exp_avg_sq": vv.clone().view_as(p)}


# ------------------------------------------------------------------------------------- angle heads
def angle_targets(angles_deg) -> torch.Tensor:
    """[N] degrees -> [N, 2] = [sin, cos] (the Swin / DINOv2 angle scripts' order: swin_angle_finetuning_sin_cos.py:47)."""
    a = torch.deg2rad(torch.as_tensor(np.asarray(angles_deg), dtype=torch.float32))
    return torch.stack([torch.sin(a), torch.cos(a)], dim=1)


def angular_loss(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """swin_angle_finetuning_sin_cos.py:65-69: mean angle (degrees) between unit vectors, cosine clamped to +-0.999999."""
    cosine_sim = torch.clamp((preds * targets).sum(dim=1), -0.999999, 0.999999)
    return torch.mean(torch.rad2deg(torch.acos(cosine_sim)))


def angle_error_deg(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """compute_angle_error, :72-76: mean min(d, 360 - d) of the atan2-decoded angles."""
    diff = torch.rad2deg(torch.abs(torch.atan2(preds[:, 0], preds[:, 1]) - torch.atan2(targets[:, 0], targets[:, 1]))) % 360
    return torch.mean(torch.minimum(diff, 360 - diff))


def finetune_angle_head(head: nn.Module, features: torch.Tensor, angles_deg, *, unit: bool = True, epochs: int = 20,
                        batch_size: int = 48, lr: float = 1e-5, cosine_t_max: Optional[int] = 10,
                        grad_clip: Optional[float] = -1.0, warm_restarts: Optional[bool] = None,
                        val: Optional[tuple] = None, seed: int = 0, log: Callable[[str], None] = print) -> dict:
    """Head-only training of a sin/cos angle head on cached pooled features (Swin pooler output / DINOv2 CLS token,
    computed once by the HIP path: ops.ln_meanpool_head(..., want_pooled=True) / backbone(x, split=True).cls).
      unit=True   swin_angle_finetuning_sin_cos.py: F.normalize(head(x), eps=1e-6), angular_loss, AdamW(1e-5) with
                  CosineAnnealingLR(T_max=10) stepped per epoch (:92-93, :119), batches of 48 (:87), a batch whose loss
                  is NaN is skipped (:110-112), gradients clipped to norm 5.0 before every step (:116 — the loss is in
                  degrees, so the clip is active);
      unit=False  swin_angle_finetuning_gemini.py / dino_v2_gemini.py: raw (sin, cos), nn.MSELoss (:183), gradients
                  clipped to GRAD_CLIP_NORM = 1.0 (:215), CosineAnnealingWarmRestarts(T_0=10, T_mult=2, eta_min=1e-7) (:188).
    grad_clip: -1 = the script's value for the chosen form (5.0 / 1.0), None = no clipping; warm_restarts: None = the
    script's scheduler for the chosen form (`cosine_t_max` is T_max resp. T_0; None = no scheduler).
    The backbone stays frozen (the reference fine-tunes it as well: out of scope); `head` is the model's
    `regressor` / `head` sub-module, so the result saves under the reference's state-dict keys."""
    dev = features.device
    y = angle_targets(angles_deg).to(dev)
    head = head.to(dev).float()
    for p in head.parameters():
        p.requires_grad_(True)
    opt = torch.optim.AdamW(head.parameters(), lr=lr)
    if grad_clip is not None and grad_clip < 0:
        grad_clip = 5.0 if unit else 1.0
    if warm_restarts is None:
        warm_restarts = not unit
    sched = None
    if cosine_t_max:
        sched = (torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=cosine_t_max, T_mult=2, eta_min=1e-7)
                 if warm_restarts else torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=cosine_t_max))
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = features.shape[0]
    fwd = (lambda x: torch.nn.functional.normalize(head(x), dim=1, p=2, eps=1e-6)) if unit else head
    loss_fn = angular_loss if unit else nn.functional.mse_loss
    history = []
    for epoch in range(epochs):
        head.train()
        perm = torch.randperm(n, generator=g).to(dev)
        total, nb = 0.0, 0
        for lo in range(0, n, batch_size):
            idx = perm[lo:lo + batch_size]
            loss = loss_fn(fwd(features[idx]), y[idx])
            if unit and torch.isnan(loss):
                log("NaN loss detected! Skipping this batch.")
                continue
            opt.zero_grad()
            loss.backward()
            if grad_clip is not None:
                torch.nn.utils.clip_grad_norm_(head.parameters(), max_norm=grad_clip)
            opt.step()
            total += float(loss.detach())
        nb = max(1, -(-n // batch_size))                        # the scripts divide by len(train_loader), skipped batches included
        if sched is not None:
            sched.step()
        rec = {"epoch": epoch, "train_loss": total / max(nb, 1)}
        if val is not None:
            head.eval()
            with torch.no_grad():
                rec["val_maae"] = float(angle_error_deg(fwd(val[0]), angle_targets(val[1]).to(dev)))
        history.append(rec)
        log(f"Epoch {epoch + 1} | Train Loss: {rec['train_loss']:.2f}" + (f" | Val Error: {rec['val_maae']:.2f}" if val else ""))
    for p in head.parameters():
        p.requires_grad_(False)
    return {"history": history}
