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
from . import torch_ops  # noqa: F401  registers torch.ops.vpr.*


def shard_bounds(N: int, rank: int, world: int) -> Tuple[int, int]:
    return N * rank // world, N * (rank + 1) // world


class HipEngine:
    """Local top-k and merge on the gfx950 kernels."""

    def local_topk(self, q, gallery, k, index_base, scales=None, norm_bound=None, uncertified=None, ws=None,
                   score_events=None, exact_fallback=False):
        """bf16 shard: q bf16.  fp8 shard (uint8 rows + per-row f32 `scales`): the gathered queries are
        quantised per row here (e4m3 + scale, vpr_quantize_fp8_rows) and searched by vpr_knn_topk_fp8.
        `uncertified` (int32 [1] on the device) counts queries whose answer the kernels could not certify as the
        exact top-k (include/vpr_amd.h "Checked forms"); no host sync.  score_events: see ops.knn_topk.
        exact_fallback: read the per-query status back (one sync) and re-run flagged queries exhaustively."""
        plain = ws is None and score_events is None and not exact_fallback      # the dispatcher-visible ops (torch_ops.py)
        if gallery.dtype == torch.uint8:
            if scales is None:
                raise ValueError("fp8 shard needs per-row scales")
            if plain:
                q8, qs = torch.ops.vpr.quantize_fp8_rows(q.float())
                v, i, _ = torch.ops.vpr.knn_topk_fp8(q8, qs, gallery, scales, k, index_base, norm_bound or ops.NORM_BOUND_FP8,
                                                     uncertified)
                return v, i
            q8, qs = ops.quantize_fp8_rows(q.float())
            return ops.knn_topk_fp8(q8, qs, gallery, scales, k, index_base, ws,
                                    norm_bound=norm_bound or ops.NORM_BOUND_FP8, uncertified=uncertified,
                                    score_events=score_events, exact_fallback=exact_fallback)
        if plain:
            v, i, _ = torch.ops.vpr.knn_topk(q, gallery, k, index_base, norm_bound or ops.NORM_BOUND_BF16, uncertified)
            return v, i
        return ops.knn_topk(q, gallery, k, index_base, ws, norm_bound=norm_bound or ops.NORM_BOUND_BF16,
                            uncertified=uncertified, score_events=score_events, exact_fallback=exact_fallback)

    def merge(self, vals, idxs):
        return torch.ops.vpr.topk_merge(vals, idxs)


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
                 engine=None, group: Optional[dist.ProcessGroup] = None, scales: Optional[torch.Tensor] = None,
                 norm_bound: Optional[float] = None, force_collectives: bool = False, exact_fallback: bool = False):
        """local_rows: [n_local, D] bf16, or uint8 e4m3 bytes with per-row f32 `scales` (value = scale * fp8).
        norm_bound: upper bound of the (dequantised) row norms for the exactness certificate; None = L2-normalised
        descriptors (what SALAD emits); `measure_norm_bound()` computes it from the rows.
        force_collectives: run the two all-gathers and the merge even with one rank (exercises the RCCL path on a
        single GPU: bench.py --force-dist).
        exact_fallback: every local search reads its certificate back (one host sync per search) and re-runs the
        queries it could not certify on exact f64 scores — unconditionally exact answers, for offline evaluation; the
        serving path leaves it off and watches `uncertified_queries()` instead (not capturable in a HIP graph)."""
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
        self.norm_bound = norm_bound
        self.exact_fallback = exact_fallback
        self.collective = world > 1 or force_collectives
        # queries whose local answer was not certified exact, summed over every search of this object (device word)
        self.uncertified = torch.zeros(1, dtype=torch.int32, device=local_rows.device) if local_rows.is_cuda else None

    def measure_norm_bound(self, slab: int = 65536) -> float:
        """Largest L2 norm of the (dequantised) local rows, with 0.1 % slack; kept as this shard's norm bound."""
        best = 0.0
        for lo in range(0, self.rows.shape[0], slab):
            r = self.rows[lo:lo + slab]
            x = r.view(torch.float8_e4m3fn).float() * self.scales[lo:lo + slab, None] if self.scales is not None else r.float()
            best = max(best, float(x.norm(dim=1).max()))
        self.norm_bound = best * 1.001
        return self.norm_bound

    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        if not self.collective:
            return q_local
        out = torch.empty((self.world * q_local.shape[0], q_local.shape[1]), dtype=q_local.dtype, device=q_local.device)
        dist.all_gather_into_tensor(out, q_local.contiguous(), group=self.group)
        return out

    def _local(self, q_all: torch.Tensor, k: int, ws=None, score_events=None):
        if isinstance(self.engine, HipEngine):
            return self.engine.local_topk(q_all, self.rows, k, self.index_base, self.scales, self.norm_bound,
                                          self.uncertified, ws, score_events, self.exact_fallback)
        if self.scales is not None:
            return self.engine.local_topk(q_all, self.rows, k, self.index_base, self.scales)
        return self.engine.local_topk(q_all, self.rows, k, self.index_base)

    def search(self, q_all: torch.Tensor, k: int, ws=None, score_events=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """q_all [B, D]: the same on every rank.  Returns merged (vals [B,k], idx [B,k]) on every rank.
        score_events (HipEngine only): list collecting (start, end) timing events around the local score stage."""
        v, i = self._local(q_all, k, ws, score_events)
        if not self.collective:
            return v, i
        vs, is_ = all_gather_topk(v, i, self.world, self.group)
        return self.engine.merge(vs, is_)

    def search_local_queries(self, q_local: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Data-parallel form: each rank contributes B_local queries and gets back its own rows."""
        q_all = self.gather_queries(q_local)
        v, i = self.search(q_all, k)
        b = q_local.shape[0]
        return v[self.rank * b:(self.rank + 1) * b], i[self.rank * b:(self.rank + 1) * b]

    def uncertified_queries(self, reduce: bool = True) -> int:
        """Host read (one sync) of the counter: 0 means every answer so far was certified exact on the device.
        The device word counts THIS rank's shard searches only; a query flagged on another rank's shard makes the merged
        top-k just as unproven, so with a sharded gallery the count is summed over the group (reduce=True: a collective —
        every rank must call it; reduce=False returns the per-shard count)."""
        if self.uncertified is None:
            return 0
        if reduce and self.collective:
            total = self.uncertified.clone()
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
            return int(total)
        return int(self.uncertified)


class GraphedRetrieval:
    """One batch of retrieval — {query all-gather, local shard search, packed top-k all-gather, merge} — captured once
    into a HIP graph and replayed per batch (SURVEY §7 step 5 / BASELINE config 5: "hipGraph-captured per-batch
    retrieval").  Static shapes: B_local queries per rank, k, this shard.  The kernels take stream-ordered arguments
    and pre-allocated workspaces; RCCL's collectives are captured like any other stream work (every rank must
    capture and replay in lockstep).  With one rank and no forced collectives the graph holds the local search only.
    Replay: copy the queries into `self.q`, `replay()`, read `self.vals` / `self.idx` (this rank's B_local rows)."""

    def __init__(self, gallery: ShardedGallery, batch_local: int, k: int):
        self.g, self.k, self.b = gallery, k, batch_local
        if gallery.exact_fallback:
            raise RuntimeError("GraphedRetrieval: exact_fallback reads the certificate back on the host — not capturable")
        if gallery.collective and dist.get_backend(gallery.group) != "nccl":
            raise RuntimeError("GraphedRetrieval: collectives can only be captured on the RCCL ('nccl') backend; "
                               f"this process group is '{dist.get_backend(gallery.group)}' — use the eager search")
        rows = gallery.rows
        dev, D = rows.device, rows.shape[1]
        self.q = torch.zeros((batch_local, D), dtype=torch.bfloat16, device=dev)
        B = batch_local * (gallery.world if gallery.collective else 1)
        self.ws = ops.knn_workspace(B, rows.shape[0], D, k, dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        counted = gallery.uncertified.clone() if gallery.uncertified is not None else None
        with torch.cuda.stream(side):            # warm-up outside capture: module load, allocator, communicator set-up
            for _ in range(2):
                self._run()
            if counted is not None:              # the warm-up's all-zero queries (every score ties) are not searches
                gallery.uncertified.copy_(counted)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        self._stream = side
        with torch.cuda.graph(self.graph, stream=side):      # the warm-up's stream: stream-keyed workspaces are re-used, not re-allocated
            self.vals, self.idx = self._run()

    def close(self) -> None:
        """Destroy the graph and drop the stream-keyed workspaces its private stream left in the package caches."""
        self.graph = None
        if self._stream is not None:
            ops.drop_stream_caches(self._stream.cuda_stream)
            self._stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:                     # noqa: BLE001
            pass

    def _run(self):
        g = self.g
        q_all = g.gather_queries(self.q)
        v, i = g.search(q_all, self.k, self.ws)
        lo = g.rank * self.b if g.collective else 0
        return v[lo:lo + self.b], i[lo:lo + self.b]

    def __call__(self, q_local: torch.Tensor):
        self.q.copy_(q_local)
        self.graph.replay()
        return self.vals, self.idx
