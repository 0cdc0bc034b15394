"""End-to-end inference step: images -> DINOv2 -> SALAD -> kNN vs. sharded gallery -> pose.

This is the loop body of the reference's evaluation scripts
(dinov2salad/dinov2salad_validation.py:78-88) with the north-star retrieval stage added
 and without the per-batch host sync (.cpu().numpy() at :81): everything stays
on the GPU, on one stream, until the caller asks for results.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .modules import DinoV2Salad, FusedGeoPoseHead
from .retrieval import ShardedGallery, all_gather_topk


@dataclass
class StepOutput:
    descriptors: torch.Tensor          # [B, 8448] f32
    topk_scores: torch.Tensor          # [B, k] f32
    topk_indices: torch.Tensor         # [B, k] int32 (global gallery rows)
    pose: torch.Tensor                 # [B, 4] f32: standardised (lat, lon), unit (sin, cos)


class VPRGeoPosePipeline:
    def __init__(self, extractor: DinoV2Salad, head: FusedGeoPoseHead, gallery: ShardedGallery, k: int = 10):
        self.extractor, self.head, self.gallery, self.k = extractor, head, gallery, k
        self.knn_events = None     # optional list collecting (start, end) events of the score kernel

    @torch.no_grad()
    def step(self, images: torch.Tensor) -> StepOutput:
        desc, desc16 = self.extractor.features(images, want_bf16=True)
        g = self.gallery
        q_all = g.gather_queries(desc16)
        if self.knn_events is not None and g.world >= 1 and getattr(g, "scales", None) is None:
            # same kernels as ShardedGallery.search, with HIP events around the score kernel
            B = q_all.shape[0]
            ws = ops.knn_workspace(B, g.rows.shape[0], q_all.shape[1], self.k, q_all.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.knn_scores(q_all, g.rows, ws)
            e1.record()
            self.knn_events.append((e0, e1))
            v, i = ops.knn_select(q_all, g.rows, self.k, ws, g.index_base)
            if g.world > 1:
                vs, is_ = all_gather_topk(v, i, g.world, g.group)
                v, i = ops.topk_merge(vs, is_)
        elif self.knn_events is not None:
            # fp8 shard: events around the whole local search (quantise queries + scores + select)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            v, i = g.engine.local_topk(q_all, g.rows, self.k, g.index_base, g.scales)
            e1.record()
            self.knn_events.append((e0, e1))
            if g.world > 1:
                vs, is_ = all_gather_topk(v, i, g.world, g.group)
                v, i = ops.topk_merge(vs, is_)
        else:
            v, i = g.search(q_all, self.k)
        b = desc.shape[0]
        v, i = v[g.rank * b:(g.rank + 1) * b], i[g.rank * b:(g.rank + 1) * b]
        pose = self.head(desc)
        return StepOutput(desc, v, i, pose)
