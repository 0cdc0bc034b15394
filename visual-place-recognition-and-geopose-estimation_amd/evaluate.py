"""Drop-in evaluation entry points: same function name, arguments and printed/returned quantities as
the reference's validation scripts, running on the MI355X path.

  calculate_validation_scores(checkpoint_path, val_csv_path, image_dir)
      dinov2salad/dinov2salad_validation.py:55-116   (DINOv2+SALAD descriptor -> MLP head -> lat/lon)
  calculate_swin_validation_scores(checkpoint_path, val_csv_path, image_dir, preds_csv)
      swin_transformer/swin_validation.py:48-134     (Swin pooler + Linear head, ID-sorted preds.csv)

What differs from the reference loop: images are decoded on the host (PIL, by a thread
pool running a few batches ahead into pinned buffers: vpr_amd.loader) but resized / normalised on the
GPU (vpr_amd.preprocess, PIL-exact), batches stay on the device until
the end (one D2H copy instead of one per batch), the model objects are passed in or built from a
state dict instead of being fetched by name (torch.hub / from_pretrained need the network), and
the scaler is a LatLonScaler (JSON or the campus constants) instead of a joblib pickle.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import pandas as pd
import torch
from PIL import Image

from . import postproc, reports
from .graphed import GraphedForward
from .loader import ImageBatchLoader
from .modules import DINOv2RegressionModel, DinoV2Salad, load_reference_checkpoint
from .preprocess import HALF_MEAN, HALF_STD, IMAGENET_MEAN, IMAGENET_STD, ResizeNormalize


def _existing_rows(val_csv_path: str, image_dir: str) -> pd.DataFrame:
    val_df = pd.read_csv(val_csv_path)
    filtered = val_df[val_df["filename"].apply(lambda x: os.path.exists(os.path.join(image_dir, x)))]
    if len(val_df) != len(filtered):
        print(f"Warning: {len(val_df) - len(filtered)} images listed in the validation CSV were not found in the image directory.")
    return filtered.reset_index(drop=True)


def _graphed(model, dev: torch.device):
    """The model forward as one HIP-graph replay per batch (graphed.GraphedForward; a forward that cannot be captured
    falls back to the eager call by itself); the model itself on a CPU device."""
    return GraphedForward(model) if dev.type == "cuda" else model


def _batches(image_dir: str, filenames, batch_size: int, device):
    """(row indices, uint8 [B,H,W,3] device tensor) per batch, decode and host-to-device copy running ahead of the
    consumer (loader.ImageBatchLoader: same bytes, batches and order as the serial PIL loop it replaces)."""
    for idxs, _, u8 in ImageBatchLoader(image_dir, filenames, batch_size, device):
        yield idxs, u8


@torch.no_grad()
def calculate_validation_scores(checkpoint_path: str, val_csv_path: str, image_dir: str, *,
                                base_model: Optional[DinoV2Salad] = None, arch: str = "vit_base",
                                scaler: Optional[postproc.LatLonScaler] = None, batch_size: int = 16,
                                device: str = "cuda", verbose: bool = True, graph: bool = True,
                                dtype: torch.dtype = torch.bfloat16) -> dict:
    """dtype: torch.bfloat16 (default: the MI355X path — HIP backbone kernels, bf16-operand SALAD) or torch.float32 =
    the reference's own precision (dinov2salad_validation.py:65-66,80-81: `.cuda()`, no cast): f32 weights, the f32
    PyTorch block loop (erf GELU, f32 LayerNorm / attention), the f32-accurate SALAD aggregation
    (vpr_salad_aggregate_f32) and the f32 head — the yardstick the bf16 path is held to (tests/test_precision_gpu.py)."""
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("dtype must be torch.bfloat16 or torch.float32")
    df = _existing_rows(val_csv_path, image_dir)
    dev = torch.device(device)
    if base_model is None:
        base_model = DinoV2Salad(arch)
    base_model = base_model.to(dev).to(dtype).eval()
    model = DINOv2RegressionModel(base_model).to(dev)
    load_reference_checkpoint(model, checkpoint_path)            # checkpoint['model_state_dict'] or a bare state dict
    model.eval()
    if hasattr(getattr(base_model, "backbone", None), "fold_layerscale") and dtype == torch.bfloat16:
        base_model.backbone.gelu = "erf"                         # the checkpoint was trained behind nn.GELU(): exact form, not the GEMM epilogue's tanh
        base_model.backbone.fold_layerscale()                    # AFTER the load (a load un-folds): LayerScale into proj / fc2, enables the HIP backbone path
    if dtype == torch.bfloat16:
        base_model.aggregator.pack()
    else:
        base_model.aggregator.pack_f32()
    scaler = scaler or postproc.LatLonScaler.campus()
    prep = ResizeNormalize(224, "bilinear", HALF_MEAN, HALF_STD, dtype)            # validation.py:18-22

    filenames = df["filename"].tolist()
    preds_std = torch.empty((len(filenames), 2), dtype=torch.float32, device=dev)
    # one HIP-graph launch per batch instead of ~300 kernel launches: the host (and its GIL) belongs to the decode threads
    fwd = GraphedForward(model) if graph and dev.type == "cuda" else model
    for idxs, u8 in _batches(image_dir, filenames, batch_size, dev):
        x = prep(u8)
        preds_std[torch.tensor(idxs, device=dev)] = fwd(x)
    all_preds = scaler.inverse_transform(preds_std.cpu().numpy())                  # fp32 in -> fp32 out (:84)
    all_targets = df[["latitude", "longitude"]].to_numpy(dtype=np.float32)
    final_loss = postproc.final_loss(all_preds, all_targets)                        # :101
    if verbose:
        print(all_preds.shape, all_targets.shape)
        print(f"final_loss: {final_loss}")
        print("\nSample Predictions (Original Scale):")
        for i in range(min(5, len(all_preds))):
            (pl, po), (tl, to) = all_preds[i], all_targets[i]
            print(f"Prediction: (Lat: {pl:.6f}, Lon: {po:.6f}), True: (Lat: {tl:.6f}, Lon: {to:.6f}), "
                  f"Error: (Lat: {abs(pl - tl):.6f}, Lon: {abs(po - to):.6f})")
    return {"final_loss": final_loss, "preds": all_preds, "targets": all_targets, "preds_standardised": preds_std.cpu().numpy(),
            "filenames": filenames}


@torch.no_grad()
def calculate_swin_validation_scores(model, val_csv_path: str, image_dir: str, preds_csv: Optional[str] = None, *,
                                     checkpoint_path: Optional[str] = None, scaler: Optional[postproc.LatLonScaler] = None,
                                     batch_size: int = 16, device: str = "cuda", verbose: bool = True, graph: bool = True) -> dict:
    """`model`: a vpr_amd.modules.SwinRegressionModel (backbone object inside).  Preprocessing follows the
    HF Swin image processor: bicubic resize to 224, /255, ImageNet mean/std."""
    df = _existing_rows(val_csv_path, image_dir)
    dev = torch.device(device)
    model = model.to(dev).eval()
    if checkpoint_path:
        load_reference_checkpoint(model, checkpoint_path)
    scaler = scaler or postproc.LatLonScaler.campus()
    prep = ResizeNormalize(224, "bicubic", IMAGENET_MEAN, IMAGENET_STD, torch.float32)
    filenames = df["filename"].tolist()
    preds_std = torch.empty((len(filenames), 2), dtype=torch.float32, device=dev)
    fwd = _graphed(model, dev)             # Swin-T at batch 16 is launch-bound: the graph replay is 2x the eager forward
    for idxs, u8 in _batches(image_dir, filenames, batch_size, dev):
        preds_std[torch.tensor(idxs, device=dev)] = fwd(prep(u8))
    all_preds = scaler.inverse_transform(preds_std.cpu().numpy())
    all_targets = df[["latitude", "longitude"]].to_numpy(dtype=np.float32)
    final_loss = postproc.final_loss(all_preds, all_targets)                        # swin_validation.py:100
    if verbose:
        print(all_preds.shape)
        print(all_targets.shape)
        print(f"final_loss: {final_loss}")
    if preds_csv:
        reports.write_id_sorted_preds(preds_csv, filenames, all_preds)              # :121-134
        if verbose:
            print("Saved predictions to preds.csv")
    return {"final_loss": final_loss, "preds": all_preds, "targets": all_targets, "filenames": filenames}


# ---------------------------------------------------------------------------------------------------------------
# Swin-Base validation + test-set entry point: swin_transformer/val_and_test_swin_2.py
IMAGE_EXTENSIONS = ["*.jpg", "*.jpeg", "*.png", "*.bmp", "*.gif", "*.JPEG"]        # :44


def _loadable(path: str, decode: bool = False) -> bool:
    """The reference's per-file check (CampusDataset :73-90 `img.verify()`, TestImageDataset :150-155 returning None
    for the collate functions :179-195 to drop): a file that PIL cannot open is skipped with a message, not fatal.
    decode=True (test split): the full decode is attempted as well — the reference's TestImageDataset.__getitem__
    catches DECODE errors too (a truncated JPEG passes verify() and fails in convert('RGB')) and drops the item,
    where a verify()-only filter would let the file through and abort the run inside the batch loader."""
    try:
        with Image.open(path) as im:
            im.verify()
        if decode:
            with Image.open(path) as im:                     # verify() leaves the object unusable: reopen
                im.convert("RGB")
        return True
    except Exception as e:                                   # FileNotFoundError, UnidentifiedImageError, OSError, ...
        print(f"Warning: Skipping invalid/corrupt image file: {path} ({e})")
        return False


@torch.no_grad()
def _predict_files(model, image_dir: str, filenames, prep, scaler, batch_size: int, dev) -> np.ndarray:
    preds_std = torch.empty((len(filenames), 2), dtype=torch.float32, device=dev)
    fwd = _graphed(model, dev)
    for idxs, u8 in _batches(image_dir, filenames, batch_size, dev):
        preds_std[torch.tensor(idxs, device=dev)] = fwd(prep(u8))
    return scaler.inverse_transform(preds_std.cpu().numpy())


@torch.no_grad()
def validate_and_test_swin(model, val_csv_path: str, val_image_dir: str, test_image_dir: Optional[str] = None,
                           save_dir: Optional[str] = None, *, checkpoint_path: Optional[str] = None,
                           scaler: Optional[postproc.LatLonScaler] = None, image_size: int = 384, batch_size: int = 32,
                           device: str = "cuda", verbose: bool = True) -> dict:
    """`model`: a vpr_amd.modules.SwinMLPRegressionModel (reference class `SwinRegressionModel`, :164-177).
    Validation split (:247-293): rows whose image is missing or unreadable are skipped with a warning, predictions are
    de-normalised, MSE / RMSE / MAE / MAE-lat / MAE-lon are printed in the reference's format and
    `validation_predictions.csv` is written ('%.6f').  Test split (:296-342, when `test_image_dir` exists): every image
    file of IMAGE_EXTENSIONS in basename order, unreadable ones dropped, `test_predictions_sorted.csv` sorted by
    filename.  Preprocessing = the HF Swin image processor the script builds (:199): bicubic resize to `image_size`
    (384 for swin-base-patch4-window12-384, :39-40), /255, ImageNet mean/std — on the GPU, PIL-exact."""
    import glob
    dev = torch.device(device)
    model = model.to(dev).eval()
    if checkpoint_path:
        load_reference_checkpoint(model, checkpoint_path)                         # bare state dict, :231
    scaler = scaler or postproc.LatLonScaler.campus()
    prep = ResizeNormalize(image_size, "bicubic", IMAGENET_MEAN, IMAGENET_STD, torch.float32)
    out = {}

    val_df = pd.read_csv(val_csv_path)
    keep = []
    for i, f in enumerate(val_df["filename"]):
        p = os.path.join(val_image_dir, f)
        if not os.path.isfile(p):
            print(f"Warning: Image file not found and skipped: {p}")              # :92
        elif _loadable(p):
            keep.append(i)
    if not keep:
        raise ValueError("No valid image files found for the provided validation dataframe and image directory.")   # :95
    vdf = val_df.iloc[keep].reset_index(drop=True)
    names = vdf["filename"].tolist()
    preds = _predict_files(model, val_image_dir, names, prep, scaler, batch_size, dev)
    targets = vdf[["latitude", "longitude"]].to_numpy(dtype=np.float32)           # the script's float32 target tensors (:120)
    out.update(val_filenames=names, val_preds=preds, val_targets=targets, metrics=postproc.regression_metrics(preds, targets))
    if verbose:
        print("\n" + reports.format_metrics(preds, targets))
        print("-" * 30)
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, "validation_predictions.csv")              # :35
        reports.write_validation_csv(path, names, targets, preds)
        if verbose:
            print(f"Successfully saved validation prediction results to: {path}")

    if test_image_dir is None or not os.path.isdir(test_image_dir):
        if verbose:
            print(f"Warning: Test image directory not found: {test_image_dir}. Skipping test set prediction.")   # :23
        return out
    paths = []
    for ext in IMAGE_EXTENSIONS:
        paths.extend(glob.glob(os.path.join(test_image_dir, ext)))
    paths.sort(key=lambda p: os.path.basename(p))                                 # :133
    if not paths:
        print(f"No image files ({', '.join(IMAGE_EXTENSIONS)}) found in {test_image_dir}")                       # :136
        return out
    tnames = [os.path.basename(p) for p in paths if _loadable(p, decode=True)]    # None items dropped by the collate (:189-191)
    if not tnames:
        print("No test predictions were made. Check test data directory and image files.")                       # :321
        return out
    tpreds = _predict_files(model, test_image_dir, tnames, prep, scaler, batch_size, dev)
    out.update(test_filenames=tnames, test_preds=tpreds)
    if save_dir:
        path = os.path.join(save_dir, "test_predictions_sorted.csv")              # :36
        reports.write_test_csv(path, tnames, tpreds)
        if verbose:
            print(f"Successfully saved sorted test prediction results to: {path}")
    return out


# ---------------------------------------------------------------------------------------------------------------
# Angle (sin/cos head) validation: loop, metric, prints and CSV layout of angle_prediction/efficient_net/
# validation_script.py:162-221 / test_script.py:255-276 — the reference's only angle-validation callers.  Their
# backbone (EfficientNet-B0) is out of scope; the loop is backbone-agnostic and here serves the Swin / DINOv2 sin/cos
# models of vpr_amd.modules (output order [sin, cos]; order="cossin" decodes the EfficientNet convention).
@torch.no_grad()
def calculate_angle_validation_scores(model, val_csv_path: str, image_dir: str, results_csv: Optional[str] = None, *,
                                      test_image_dir: Optional[str] = None, test_csv: Optional[str] = None,
                                      checkpoint_path: Optional[str] = None, order: str = "sincos", image_size: int = 224,
                                      batch_size: int = 32, device: str = "cuda", verbose: bool = True, prep=None) -> dict:
    """`model`: pixel_values -> [B, 2] (SwinSinCosRegressionModel, SwinAngleRegressorSinCos, DinoV2AngleRegressorSinCos).
    Decoding as validation_script.py:179-182: atan2 -> degrees -> (+360) % 360; error min(d, 360 - d) (:188-189); MAAE =
    mean over the processed samples (:200-206).  Preprocessing default = the HF Swin processor (bicubic resize to
    `image_size`, /255, ImageNet statistics) on the GPU; pass `prep` for another one (dtype must suit the model)."""
    import glob
    if order not in ("sincos", "cossin"):
        raise ValueError("order must be 'sincos' or 'cossin'")
    dev = torch.device(device)
    model = model.to(dev).eval()
    if checkpoint_path:
        load_reference_checkpoint(model, checkpoint_path)
    prep = prep or ResizeNormalize(image_size, "bicubic", IMAGENET_MEAN, IMAGENET_STD, torch.float32)

    def predict_deg(directory, names):
        out = torch.empty((len(names), 2), dtype=torch.float32, device=dev)
        fwd = _graphed(model, dev)
        for idxs, u8 in _batches(directory, names, batch_size, dev):
            out[torch.tensor(idxs, device=dev)] = fwd(prep(u8)).float()
        s, c = (out[:, 0], out[:, 1]) if order == "sincos" else (out[:, 1], out[:, 0])
        return ((torch.rad2deg(torch.atan2(s, c)) + 360.0) % 360.0).cpu().numpy()            # f32, as the script's tensors

    df = _existing_rows(val_csv_path, image_dir)
    names = df["filename"].tolist()
    pred = predict_deg(image_dir, names)
    true = df["angle"].to_numpy()
    diff = np.abs(pred - true.astype(np.float32))
    err = np.minimum(diff, np.float32(360.0) - diff)
    maae = float(err.astype(np.float64).sum() / max(len(names), 1))
    if verbose:
        print("\n===================================")
        print("Prediction complete.")
        print(f"Total validation samples processed: {len(names)}")
        print(f"Mean Absolute Angular Error (MAAE): {maae:.4f} degrees")
        print("===================================")
    res = {"filenames": names, "pred_deg": pred, "true_deg": true, "angular_error": err, "maae": maae}
    if results_csv:
        reports.write_angle_validation_csv(results_csv, names, true, pred)
        if verbose:
            print(f"Prediction results saved to: {results_csv}")
    if test_image_dir and os.path.isdir(test_image_dir):
        paths = []
        for ext in IMAGE_EXTENSIONS:
            paths.extend(glob.glob(os.path.join(test_image_dir, ext)))
        tnames = sorted(os.path.basename(p) for p in paths if _loadable(p, decode=True))
        if tnames:
            tpred = predict_deg(test_image_dir, tnames)
            res.update(test_filenames=tnames, test_pred_deg=tpred)
            if test_csv:
                reports.write_angle_test_csv(test_csv, tnames, tpred)
    return res


# ---------------------------------------------------------------------------------------------------------------
# Retrieval-based geopose (north-star stage; the reference has no retrieval — SURVEY fact 3, §8f-2): the same CSV + image
# directory conventions as the validation scripts, with the gallery built from the training split.
@torch.no_grad()
def build_gallery_from_images(extractor: DinoV2Salad, csv_path: str, image_dir: str, out_dir: str, *, fp8: bool = False,
                              batch_size: int = 64, device: str = "cuda", image_size: int = 224, graph: bool = True) -> int:
    """labels CSV (`filename,timestamp,latitude,longitude,angle,Region_ID`, cleaned_dataset_files/labels_train.csv:1) +
    images -> on-disk gallery (gallery.save_gallery: bf16 rows, or e4m3 rows + per-row scales with fp8=True).
    Preprocessing = the DINOv2+SALAD validation transform (dinov2salad_validation.py:18-22).  Returns the row count."""
    from . import gallery as G, ops
    dev = torch.device(device)
    extractor = extractor.to(dev).to(torch.bfloat16).eval()
    df = _existing_rows(csv_path, image_dir)
    names = df["filename"].tolist()
    prep = ResizeNormalize(image_size, "bilinear", HALF_MEAN, HALF_STD, torch.bfloat16)
    desc = torch.empty((len(names), 8448), dtype=torch.float32, device=dev)
    fwd = GraphedForward(extractor) if graph and dev.type == "cuda" else extractor
    for idxs, u8 in _batches(image_dir, names, batch_size, dev):
        desc[torch.tensor(idxs, device=dev)] = fwd(prep(u8))
    labels = df[list(G.LABEL_COLUMNS)].to_numpy(dtype=np.float64)
    if fp8:
        rows, scales = ops.quantize_fp8_rows(desc)
        G.save_gallery(out_dir, rows, labels, scales=scales, filenames=names)
    else:
        G.save_gallery(out_dir, desc.to(torch.bfloat16), labels, filenames=names)
    return len(names)


@torch.no_grad()
def calculate_retrieval_scores(extractor: DinoV2Salad, gallery_dir: str, val_csv_path: str, image_dir: str, *, k: int = 10,
                               tau: float = 25.0, mode: str = "top1", batch_size: int = 64, device: str = "cuda", graph: bool = True,
                               image_size: int = 224, rank: int = 0, world: int = 1, group=None, verbose: bool = True) -> dict:
    """Validation split through descriptor -> sharded cosine top-k -> label transfer:
      pose      (lat, lon, angle) of the best match, or the softmax-weighted mean of the k matches (gallery.label_transfer);
      final_loss on lat/lon with the validation scripts' formula (dinov2salad_validation.py:101), MAAE on the angle
                (swin_angle_finetuning_gemini.py:131-146);
      Recall@1 / Recall@k with positives = gallery rows within `tau` label units of the query (Euclidean on the projected
                lat/lon) and, separately, rows of the same Region_ID.
    With world > 1 every rank runs this with its shard (load_gallery_shard) and its share of the queries is gathered by
    ShardedGallery.search_local_queries; the metrics are then those of this rank's queries."""
    from . import gallery as G
    from .retrieval import ShardedGallery
    dev = torch.device(device)
    extractor = extractor.to(dev).to(torch.bfloat16).eval()
    shard = G.load_gallery_shard(gallery_dir, dev, rank, world)
    # offline evaluation: unconditionally exact neighbours (queries the device certificate flags are re-run on f64 scores)
    sg = ShardedGallery(shard.rows, shard.n_total, rank, world, group=group, scales=shard.scales, exact_fallback=True)
    df = _existing_rows(val_csv_path, image_dir)
    names = df["filename"].tolist()
    prep = ResizeNormalize(image_size, "bilinear", HALF_MEAN, HALF_STD, torch.bfloat16)
    kk = min(k, shard.n_total)
    vals = torch.empty((len(names), kk), dtype=torch.float32, device=dev)
    idx = torch.empty((len(names), kk), dtype=torch.int32, device=dev)
    features = lambda x: extractor.features(x, want_bf16=True)
    fwd = GraphedForward(features, module=extractor) if graph and dev.type == "cuda" else features
    for idxs, u8 in _batches(image_dir, names, batch_size, dev):
        _, d16 = fwd(prep(u8))
        v, i = sg.search_local_queries(d16, kk)
        sel = torch.tensor(idxs, device=dev)
        vals[sel], idx[sel] = v, i
    labels = shard.labels
    pose = G.label_transfer(vals, idx, labels, mode=mode)
    targets = df[["latitude", "longitude"]].to_numpy(dtype=np.float64)
    top = idx.cpu().numpy()
    pos_d = G.positives_by_distance(targets, labels[:, :2], tau)
    pos_r = G.positives_by_region(df["Region_ID"].to_numpy(), labels[:, 3])
    res = {"filenames": names, "topk_scores": vals.cpu().numpy(), "topk_indices": top, "pose": pose,
           "final_loss": postproc.final_loss(pose[:, :2], targets),
           "maae": postproc.mean_absolute_angular_error(pose[:, 2], df["angle"].to_numpy(dtype=np.float64)),
           "recall_at_1_tau": postproc.recall_at_k(top[:, :1], pos_d), f"recall_at_{kk}_tau": postproc.recall_at_k(top, pos_d),
           "recall_at_1_region": postproc.recall_at_k(top[:, :1], pos_r),
           "uncertified_queries": sg.uncertified_queries()}        # flagged on the device (and re-run exactly), for the record
    if verbose:
        print(f"final_loss: {res['final_loss']}")
        print(f"Mean Absolute Angular Error (MAAE): {res['maae']:.4f} degrees")
        print(f"Recall@1 (tau={tau}): {res['recall_at_1_tau']:.4f}  Recall@{kk}: {res[f'recall_at_{kk}_tau']:.4f}  "
              f"Recall@1 (Region_ID): {res['recall_at_1_region']:.4f}")
    return res
