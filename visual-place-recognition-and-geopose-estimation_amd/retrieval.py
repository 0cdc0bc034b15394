"""Gallery sharding + distributed top-k (no counterpart in the reference).

Rank r of R owns gallery rows [N*r/R, N*(r+1)/R) and searches them for EVERY query of the
global batch; the only exchange steps are two all-gathers per batch (RCCL over xGMI on GPUs,
gloo in the CPU tests): query descriptors [B_local, D] -> [R*B_local, D], and per-shard top-k
(value f32, global index int32) [B, k] -> [R, B, k], followed by an on-device merge with the same
(value desc, index asc) key, so the answer equals the unsharded search.

The search and merge kernels are injected (`engine`): the product engine is HipEngine (HIP
kernels, no fallback); tests pass an oracle-backed engine to exercise the sharding and the
collectives on CPU.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


def shard_bounds(N: int, rank: int, world: int) -> Tuple[int, int]:
    return N * rank // world, N * (rank + 1) // world


class HipEngine:
    """Local top-k and merge on the gfx950 kernels."""

    def local_topk(self, q, gallery, k, index_base, scales=None):
        """bf16 shard: q bf16.  fp8 shard (uint8 rows + per-row f32 `scales`): the gathered queries are
        quantised per row here (e4m3 + scale, vpr_quantize_fp8_rows) and searched by vpr_knn_topk_fp8."""
        if gallery.dtype == torch.uint8:
            if scales is None:
                raise ValueError("fp8 shard needs per-row scales")
            q8, qs = ops.quantize_fp8_rows(q.float())
            return ops.knn_topk_fp8(q8, qs, gallery, scales, k, index_base)
        return ops.knn_topk(q, gallery, k, index_base)

    def merge(self, vals, idxs):
        return ops.topk_merge(vals, idxs)


def all_gather_topk(v: torch.Tensor, i: torch.Tensor, world: int, group=None):
    """Per-shard (vals f32, idx i32) [B,k] -> [world, B, k] on every rank with ONE all-gather: the
    two arrays travel as one int32 buffer [B, 2k] (values bit-cast), since at these sizes
    (5 KB per rank) a collective costs its launch latency, not its bytes."""
    B, k = v.shape
    packed = torch.cat([v.contiguous().view(torch.int32), i.contiguous()], dim=1)          # [B, 2k] int32
    out = torch.empty((world * B, 2 * k), dtype=torch.int32, device=v.device)              # concatenated along dim 0:
    dist.all_gather_into_tensor(out, packed, group=group)                                   # the layout gloo and RCCL share
    out = out.view(world, B, 2 * k)
    return out[:, :, :k].contiguous().view(torch.float32), out[:, :, k:].contiguous()


class ShardedGallery:
    def __init__(self, local_rows: torch.Tensor, n_total: int, rank: int = 0, world: int = 1,
                 engine=None, group: Optional[dist.ProcessGroup] = None, scales: Optional[torch.Tensor] = None):
        """local_rows: [n_local, D] bf16, or uint8 e4m3 bytes with per-row f32 `scales` (value = scale * fp8)."""
        lo, hi = shard_bounds(n_total, rank, world)
        if local_rows.shape[0] != hi - lo:
            raise ValueError(f"rank {rank}: shard has {local_rows.shape[0]} rows, expected {hi - lo}")
        if (local_rows.dtype == torch.uint8) != (scales is not None):
            raise ValueError("fp8 shards (uint8 rows) come with per-row scales, bf16 shards without")
        if scales is not None and scales.numel() != hi - lo:
            raise ValueError("scales: one per local row")
        self.scales = scales
        self.rows, self.n_total, self.rank, self.world = local_rows, n_total, rank, world
        self.index_base = lo
        self.engine = engine if engine is not None else HipEngine()
        self.group = group

    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return q_local
        out = torch.empty((self.world * q_local.shape[0], q_local.shape[1]), dtype=q_local.dtype, device=q_local.device)
        dist.all_gather_into_tensor(out, q_local.contiguous(), group=self.group)
        return out

    def search(self, q_all: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """q_all [B, D]: the same on every rank.  Returns merged (vals [B,k], idx [B,k]) on every rank."""
        if self.scales is not None:
            v, i = self.engine.local_topk(q_all, self.rows, k, self.index_base, self.scales)
        else:
            v, i = self.engine.local_topk(q_all, self.rows, k, self.index_base)
        if self.world == 1:
            return v, i
        B = q_all.shape[0]
        vs, is_ = all_gather_topk(v, i, self.world, self.group)
        return self.engine.merge(vs, is_)

    def search_local_queries(self, q_local: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Data-parallel form: each rank contributes B_local queries and gets back its own rows."""
        q_all = self.gather_queries(q_local)
        v, i = self.search(q_all, k)
        b = q_local.shape[0]
        return v[self.rank * b:(self.rank + 1) * b], i[self.rank * b:(self.rank + 1) * b]
