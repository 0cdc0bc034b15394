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
    SPLIT_TIMING_MIN_ROWS = 4096       # below: the score stage may be K-split (vpr_knn_topk only): time the whole call

    def __init__(self, extractor: DinoV2Salad, head: FusedGeoPoseHead, gallery: ShardedGallery, k: int = 10,
                 overlap_head: Optional[bool] = None):
        self.extractor, self.head, self.gallery, self.k = extractor, head, gallery, k
        self.knn_events = None     # optional list collecting (start, end) events of the score kernel
        # The pose head needs only the descriptor; the retrieval leg (two collectives with launch-latency gaps between
        # them when the gallery is sharded) runs beside it: head on a side stream, joined at the end of the step.
        self.overlap_head = gallery.collective if overlap_head is None else overlap_head
        self._side = {}

    def _side_stream(self, dev, main):
        key = (str(dev), main.cuda_stream)
        s = self._side.get(key)
        if s is None:
            s = self._side[key] = torch.cuda.Stream(device=dev)
        return s

    @torch.no_grad()
    def step(self, images: torch.Tensor) -> StepOutput:
        desc, desc16 = self.extractor.features(images, want_bf16=True)
        g = self.gallery
        pose = None
        if self.overlap_head:
            main = torch.cuda.current_stream(desc.device)
            side = self._side_stream(desc.device, main)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pose = self.head(desc)
            desc.record_stream(side)
        q_all = g.gather_queries(desc16)
        if self.knn_events is not None and getattr(g, "scales", None) is None and g.rows.shape[0] > self.SPLIT_TIMING_MIN_ROWS:
            # same kernels as ShardedGallery.search, with HIP events around the score kernel
            B = q_all.shape[0]
            ws = ops.knn_workspace(B, g.rows.shape[0], q_all.shape[1], self.k, q_all.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.knn_scores(q_all, g.rows, ws)
            e1.record()
            self.knn_events.append((e0, e1))
            v, i = ops.knn_select(q_all, g.rows, self.k, ws, g.index_base, norm_bound=g.norm_bound or ops.NORM_BOUND_BF16,
                                  uncertified=g.uncertified)
            if g.collective:
                vs, is_ = all_gather_topk(v, i, g.world, g.group)
                v, i = ops.topk_merge(vs, is_)
        elif self.knn_events is not None:
            # fp8 shard, or a shard small enough for the K-split form of the score stage (which only the whole
            # vpr_knn_topk call runs): events around the whole local search
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            v, i = g._local(q_all, self.k)
            e1.record()
            self.knn_events.append((e0, e1))
            if g.collective:
                vs, is_ = all_gather_topk(v, i, g.world, g.group)
                v, i = ops.topk_merge(vs, is_)
        else:
            v, i = g.search(q_all, self.k)
        b = desc.shape[0]
        if g.collective:
            v, i = v[g.rank * b:(g.rank + 1) * b], i[g.rank * b:(g.rank + 1) * b]
        if self.overlap_head:
            main.wait_stream(side)
            pose.record_stream(main)
        else:
            pose = self.head(desc)
        return StepOutput(desc, v, i, pose)
