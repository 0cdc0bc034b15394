"""Gallery builder, on-disk format and retrieval-based geopose (the reference has
no retrieval code — its label files `cleaned_dataset_files/labels_{train,val}.csv`
(`filename,timestamp,latitude,longitude,angle,Region_ID`) are the side table this uses).

On-disk format (a directory; every array is a plain .npy so shards can be memory-mapped):
  meta.json            {"version":1, "n":N, "d":8448, "dtype":"bf16"|"fp8_e4m3", "label_columns":[...]}
  descriptors.npy      [N, D] uint16 (bf16 bit patterns) or uint8 (e4m3 bytes)
  scales.npy           [N] float32            (fp8 only: value = scale * fp8)
  labels.npy           [N, 4] float64         latitude, longitude, angle (deg), Region_ID
  filenames.txt        one per row (optional)
Rank r of R loads rows [N*r/R, N*(r+1)/R) only.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .retrieval import ShardedGallery, shard_bounds

LABEL_COLUMNS = ("latitude", "longitude", "angle", "Region_ID")


@dataclass
class GalleryShard:
    rows: torch.Tensor                    # [n_local, D] bf16 or uint8 on the GPU
    scales: Optional[torch.Tensor]        # [n_local] f32 (fp8) or None
    labels: np.ndarray                    # [N, 4] float64 — the FULL side table (small), host
    n_total: int
    index_base: int
    dtype: str


def save_gallery(path: str, descriptors: torch.Tensor, labels: np.ndarray, scales: Optional[torch.Tensor] = None,
                 filenames: Optional[Sequence[str]] = None) -> None:
    """descriptors: [N,D] torch.bfloat16 (bf16 gallery) or torch.uint8 (+ scales: fp8 gallery)."""
    os.makedirs(path, exist_ok=True)
    d = descriptors.detach().cpu().contiguous()
    if d.dtype == torch.bfloat16:
        arr, dtype = d.view(torch.uint16).numpy(), "bf16"
    elif d.dtype == torch.uint8:
        if scales is None:
            raise ValueError("fp8 gallery needs per-row scales")
        arr, dtype = d.numpy(), "fp8_e4m3"
        np.save(os.path.join(path, "scales.npy"), scales.detach().cpu().to(torch.float32).numpy())
    else:
        raise ValueError("descriptors must be bfloat16 or uint8 (e4m3 bytes)")
    labels = np.asarray(labels, dtype=np.float64)
    if labels.shape != (arr.shape[0], len(LABEL_COLUMNS)):
        raise ValueError(f"labels must be [N,{len(LABEL_COLUMNS)}] = {LABEL_COLUMNS}")
    np.save(os.path.join(path, "descriptors.npy"), arr)
    np.save(os.path.join(path, "labels.npy"), labels)
    if filenames is not None:
        with open(os.path.join(path, "filenames.txt"), "w") as f:
            f.write("\n".join(filenames))
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"version": 1, "n": int(arr.shape[0]), "d": int(arr.shape[1]), "dtype": dtype,
                   "label_columns": list(LABEL_COLUMNS)}, f)


def load_gallery_shard(path: str, device: torch.device, rank: int = 0, world: int = 1) -> GalleryShard:
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("version") != 1:
        raise ValueError("unknown gallery format version")
    lo, hi = shard_bounds(meta["n"], rank, world)
    desc = np.load(os.path.join(path, "descriptors.npy"), mmap_mode="r")
    rows = torch.from_numpy(np.array(desc[lo:hi])).to(device)          # copy of this rank's rows only
    scales = None
    if meta["dtype"] == "bf16":
        rows = rows.view(torch.bfloat16)
    else:
        sc = np.load(os.path.join(path, "scales.npy"), mmap_mode="r")
        scales = torch.from_numpy(np.array(sc[lo:hi])).to(device)
    labels = np.load(os.path.join(path, "labels.npy"))
    return GalleryShard(rows, scales, labels, meta["n"], lo, meta["dtype"])


def labels_from_csv(csv_path: str) -> Tuple[np.ndarray, List[str]]:
    """Reads the reference's label CSV layout -> ([N,4] float64, filenames)."""
    import pandas as pd
    df = pd.read_csv(csv_path)
    return df[list(LABEL_COLUMNS)].to_numpy(dtype=np.float64), df["filename"].tolist()


@torch.no_grad()
def build_descriptors(extractor, image_batches: Iterable[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Runs DinoV2Salad over batches of preprocessed images -> (f32 [N,8448], bf16 [N,8448]) on the GPU."""
    f32, b16 = [], []
    for images in image_batches:
        d, d16 = extractor.features(images, want_bf16=True)
        f32.append(d), b16.append(d16)
    return torch.cat(f32), torch.cat(b16)


# ---------------------------------------------------------------------------- retrieval -> pose
def label_transfer(topk_scores: torch.Tensor, topk_idx: torch.Tensor, labels: np.ndarray, mode: str = "top1",
                   temperature: float = 0.01) -> np.ndarray:
    """Geopose from retrieval: returns [B,3] = (lat, lon, angle_deg).
    mode "top1": the best match's labels.  mode "weighted": softmax(score / temperature) over the
    k matches — weighted mean for lat/lon, weighted circular mean for the angle."""
    idx = topk_idx.detach().cpu().numpy().astype(np.int64)
    sc = topk_scores.detach().cpu().numpy().astype(np.float64)
    valid = idx >= 0
    safe = np.where(valid, idx, 0)
    lat, lon, ang = labels[safe, 0], labels[safe, 1], np.deg2rad(labels[safe, 2])
    if mode == "top1":
        return np.stack([lat[:, 0], lon[:, 0], np.rad2deg(ang[:, 0]) % 360.0], 1)
    if mode != "weighted":
        raise ValueError("mode must be 'top1' or 'weighted'")
    w = np.where(valid, np.exp((sc - sc[:, :1]) / temperature), 0.0)
    w = w / w.sum(1, keepdims=True)
    a = np.rad2deg(np.arctan2((w * np.sin(ang)).sum(1), (w * np.cos(ang)).sum(1))) % 360.0
    return np.stack([(w * lat).sum(1), (w * lon).sum(1), a], 1)


def positives_by_distance(query_latlon: np.ndarray, gallery_latlon: np.ndarray, tau: float) -> List[np.ndarray]:
    """Gallery rows within Euclidean distance tau (label units) of each query."""
    q, g = np.asarray(query_latlon, dtype=np.float64), np.asarray(gallery_latlon, dtype=np.float64)
    d2 = ((q[:, None, :] - g[None, :, :]) ** 2).sum(-1)
    return [np.nonzero(row <= tau * tau)[0] for row in d2]


def positives_by_region(query_region: np.ndarray, gallery_region: np.ndarray) -> List[np.ndarray]:
    g = np.asarray(gallery_region)
    return [np.nonzero(g == r)[0] for r in np.asarray(query_region)]


# ------------------------------------------------------------------- hipGraph-captured retrieval
class GraphedLocalTopK:
    """Local shard search captured once into a HIP graph and replayed per batch (BASELINE config 5:
    'hipGraph-captured per-batch retrieval').  Static shapes: B queries, k, this shard.  The
    kernels take only stream-ordered arguments and a pre-allocated workspace, so the capture holds
    5 kernel nodes (scores, select x2, rescore, order) and replay costs one launch."""

    def __init__(self, shard: GalleryShard, batch: int, k: int):
        self.shard, self.k = shard, k
        dev = shard.rows.device
        D = shard.rows.shape[1]
        self.fp8 = shard.dtype != "bf16"
        self.q = torch.zeros((batch, D), dtype=shard.rows.dtype, device=dev)
        self.q_scale = torch.ones((batch,), dtype=torch.float32, device=dev) if self.fp8 else None
        self.ws = ops.knn_workspace(batch, shard.rows.shape[0], D, k, dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture (module load, allocator)
            self._run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.vals, self.idx = self._run()

    def _run(self):
        s = self.shard
        if self.fp8:
            return ops.knn_topk_fp8(self.q, self.q_scale, s.rows, s.scales, self.k, s.index_base, self.ws)
        return ops.knn_topk(self.q, s.rows, self.k, s.index_base, self.ws)

    def __call__(self, q: torch.Tensor, q_scale: Optional[torch.Tensor] = None):
        self.q.copy_(q)
        if self.fp8:
            self.q_scale.copy_(q_scale)
        self.graph.replay()
        return self.vals, self.idx


def sharded_gallery_from(shard: GalleryShard, rank: int = 0, world: int = 1, group=None) -> ShardedGallery:
    """A loaded shard (bf16 rows, or e4m3 bytes + per-row scales) as the distributed search object."""
    return ShardedGallery(shard.rows, shard.n_total, rank, world, group=group, scales=shard.scales)
