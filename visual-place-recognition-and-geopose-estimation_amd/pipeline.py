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

from .modules import DinoV2Salad, FusedGeoPoseHead
from .retrieval import GraphedRetrieval, ShardedGallery


@dataclass
class StepOutput:
    descriptors: torch.Tensor          # [B, 8448] f32
    topk_scores: torch.Tensor          # [B, k] f32
    topk_indices: torch.Tensor         # [B, k] int32 (global gallery rows)
    pose: torch.Tensor                 # [B, 4] f32: standardised (lat, lon), unit (sin, cos)


class VPRGeoPosePipeline:
    def __init__(self, extractor: DinoV2Salad, head: FusedGeoPoseHead, gallery: ShardedGallery, k: int = 10,
                 overlap_head: Optional[bool] = None, graph_retrieval: bool = False):
        """graph_retrieval: the retrieval leg of every step — {query all-gather, local shard search, packed top-k
        all-gather, merge} — is ONE HIP-graph replay (retrieval.GraphedRetrieval, captured at the first step of a given
        batch size; every rank must step in lockstep, as with the eager collectives).  BASELINE config 5's "hipGraph-
        captured per-batch retrieval"."""
        self.extractor, self.head, self.gallery, self.k = extractor, head, gallery, k
        self.graph_retrieval = graph_retrieval
        self._graphed = {}
        self.knn_events = None     # optional list collecting (start, end) events of the score kernel (graph: of the replay)
        self.salad_events = None   # optional list collecting (start, end) events around the SALAD aggregation of every step
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
        desc, desc16 = self.extractor.features(images, want_bf16=True, events=self.salad_events)
        g = self.gallery
        pose = None
        if self.overlap_head:
            main = torch.cuda.current_stream(desc.device)
            side = self._side_stream(desc.device, main)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pose = self.head(desc)
            desc.record_stream(side)
        if self.graph_retrieval:
            key = (desc16.shape[0], torch.cuda.current_stream(desc.device).cuda_stream)   # one graph (and its buffers) per lane
            gr = self._graphed.get(key)
            if gr is None:
                gr = self._graphed[key] = GraphedRetrieval(g, desc16.shape[0], self.k)
            if self.knn_events is not None:                # a graph has no seam for events: the whole replay is timed
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            v, i = gr(desc16)
            if self.knn_events is not None:
                e1.record()
                self.knn_events.append((e0, e1))
            v, i = v.clone(), i.clone()                    # the graph's output buffers are rewritten by the next replay
        else:
            q_all = g.gather_queries(desc16)
            # knn_events: same kernels, run as the two stages of the call with HIP events around the score stage
            v, i = g.search(q_all, self.k, score_events=self.knn_events)
            if g.collective:
                b = desc.shape[0]
                v, i = v[g.rank * b:(g.rank + 1) * b], i[g.rank * b:(g.rank + 1) * b]
        if self.overlap_head:
            main.wait_stream(side)
            pose.record_stream(main)
        else:
            pose = self.head(desc)
        return StepOutput(desc, v, i, pose)
