"""`torch.library` operator layer over the C ABI.

The reference's seam is `nn.Module.forward` (dinov2salad/dinov2salad_validation.py:49-52: `feature_extractor(x)` then
`regressor(features)`; swin_transformer/swin_validation.py:43-46).  The modules of `vpr_amd.modules` call these ops at
that seam, so the HIP hot path is visible to PyTorch's dispatcher like any other operator:

    torch.ops.vpr.salad_aggregate / salad_aggregate_split / salad_aggregate_f32
    torch.ops.vpr.knn_topk / knn_topk_fp8 / topk_merge
    torch.ops.vpr.pose_head / ln_meanpool_head
    torch.ops.vpr.head_train_epoch                     (head-only fine-tuning pass; mutates parameters and AdamW moments)

Each op has
  * a CUDA(HIP) implementation = the ctypes wrapper of `vpr_amd.ops` (same validation, same stream, same workspaces;
    a non-zero C status raises RuntimeError; there is no CPU implementation — a CPU tensor is refused), and
  * a fake (meta) implementation that only computes output shapes / dtypes, so the modules can be traced with
    FakeTensorMode / `torch.export` / `torch.compile(fullgraph=True)` without a GPU (tests/test_torch_ops_cpu.py).
Inference only: no autograd formula is registered (the modules run under torch.no_grad, as the reference's
feature extractor does: dinov2salad_validation.py:50).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import ops

_W_NAMES = ("w1_sc", "b1_sc", "w2_s", "b2_s", "w2_c", "b2_c", "w1_t", "b1_t", "w2_t", "b2_t")


def _weights(ws: List[Tensor], dustbin: float, f32: bool = False):
    if len(ws) != 10:
        raise RuntimeError("SALAD weights: expected the 10 tensors (w1_sc, b1_sc, w2_s, b2_s, w2_c, b2_c, w1_t, b1_t, w2_t, b2_t)")
    cls = ops.SaladWeightsF32 if f32 else ops.SaladWeights
    return cls(**dict(zip(_W_NAMES, ws)), dustbin=dustbin)


def _desc_width(ws: List[Tensor]) -> int:
    return ws[8].shape[0] + ws[4].shape[0] * ws[2].shape[0]          # t + l * m


def weight_list(w: ops.SaladWeights) -> List[Tensor]:
    return [getattr(w, n) for n in _W_NAMES]


# ------------------------------------------------------------------------------------------------------------ SALAD
@torch.library.custom_op("vpr::salad_aggregate", mutates_args=())
def salad_aggregate(tokens: Tensor, weights: List[Tensor], dustbin: float, sinkhorn_iters: int) -> Tuple[Tensor, Tensor]:
    """tokens [B, 1+n, C] bf16 (cls first) -> (descriptor f32 [B, t+l*m], its bf16 copy).  vpr_salad_aggregate."""
    out, out16 = ops.salad_aggregate(tokens, _weights(weights, dustbin), sinkhorn_iters, True)
    return out, out16


@salad_aggregate.register_fake
def _(tokens, weights, dustbin, sinkhorn_iters):
    B, D = tokens.shape[0], _desc_width(weights)
    return tokens.new_empty((B, D), dtype=torch.float32), tokens.new_empty((B, D), dtype=torch.bfloat16)


@torch.library.custom_op("vpr::salad_aggregate_split", mutates_args=())
def salad_aggregate_split(patch: Tensor, cls: Tensor, weights: List[Tensor], dustbin: float,
                          sinkhorn_iters: int, token_done: bool = False) -> Tuple[Tensor, Tensor]:
    """patch [B, n, C] bf16 + cls [B, C] bf16 (the layout the HIP backbone computes in).  vpr_salad_aggregate_split, or —
    token_done: the token MLP of these cls rows already ran on the backbone's cls-row stream (ops.salad_stage_token) —
    vpr_salad_stage_mlps + vpr_salad_stage_aggregate."""
    out, out16 = ops.salad_aggregate_split(patch, cls, _weights(weights, dustbin), sinkhorn_iters, True, token_done=token_done)
    return out, out16


@salad_aggregate_split.register_fake
def _(patch, cls, weights, dustbin, sinkhorn_iters, token_done=False):
    B, D = patch.shape[0], _desc_width(weights)
    return patch.new_empty((B, D), dtype=torch.float32), patch.new_empty((B, D), dtype=torch.bfloat16)


@torch.library.custom_op("vpr::salad_aggregate_f32", mutates_args=())
def salad_aggregate_f32(patch: Tensor, cls: Tensor, weights: List[Tensor], dustbin: float,
                        sinkhorn_iters: int) -> Tuple[Tensor, Tensor]:
    """f32 patch [B, n, C] + cls [B, C], f32 weights: the aggregation at the reference's precision.  vpr_salad_aggregate_f32."""
    out, out16 = ops.salad_aggregate_f32((patch, cls), _weights(weights, dustbin, f32=True), sinkhorn_iters, True)
    return out, out16


@salad_aggregate_f32.register_fake
def _(patch, cls, weights, dustbin, sinkhorn_iters):
    B, D = patch.shape[0], _desc_width(weights)
    return patch.new_empty((B, D), dtype=torch.float32), patch.new_empty((B, D), dtype=torch.bfloat16)


# -------------------------------------------------------------------------------------------------------------- kNN
@torch.library.custom_op("vpr::knn_topk", mutates_args=("uncertified",))
def knn_topk(q: Tensor, gallery: Tensor, k: int, index_base: int, norm_bound: float,
             uncertified: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """q [B, D] bf16 x gallery [N, D] bf16 -> (scores f32 [B, k] descending, global indices int32 [B, k], certificate
    status int32 [B]: 0 / 1 = proven exact, 2 = not proven).  `uncertified` (int32 [1], optional) is incremented on the
    device by the number of status-2 queries.  vpr_knn_topk_checked."""
    status = torch.empty((q.shape[0],), dtype=torch.int32, device=q.device)
    v, i = ops.knn_topk(q, gallery, k, index_base, norm_bound=norm_bound, status=status, uncertified=uncertified)
    return v, i, status


@knn_topk.register_fake
def _(q, gallery, k, index_base, norm_bound, uncertified):
    B = q.shape[0]
    return (q.new_empty((B, k), dtype=torch.float32), q.new_empty((B, k), dtype=torch.int32),
            q.new_empty((B,), dtype=torch.int32))


@torch.library.custom_op("vpr::knn_topk_fp8", mutates_args=("uncertified",))
def knn_topk_fp8(q: Tensor, q_scale: Tensor, gallery: Tensor, gallery_scale: Tensor, k: int, index_base: int,
                 norm_bound: float, uncertified: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """e4m3 bytes + per-row f32 scales on both sides (BASELINE config 5).  vpr_knn_topk_fp8_checked."""
    status = torch.empty((q.shape[0],), dtype=torch.int32, device=q.device)
    v, i = ops.knn_topk_fp8(q, q_scale, gallery, gallery_scale, k, index_base, norm_bound=norm_bound, status=status,
                            uncertified=uncertified)
    return v, i, status


@knn_topk_fp8.register_fake
def _(q, q_scale, gallery, gallery_scale, k, index_base, norm_bound, uncertified):
    B = q.shape[0]
    return (q.new_empty((B, k), dtype=torch.float32), q.new_empty((B, k), dtype=torch.int32),
            q.new_empty((B,), dtype=torch.int32))


@torch.library.custom_op("vpr::quantize_fp8_rows", mutates_args=())
def quantize_fp8_rows(x: Tensor) -> Tuple[Tensor, Tensor]:
    """x [rows, D] f32 -> (e4m3 bytes uint8 [rows, D], per-row scale f32 [rows]).  vpr_quantize_fp8_rows."""
    return ops.quantize_fp8_rows(x)


@quantize_fp8_rows.register_fake
def _(x):
    return x.new_empty(x.shape, dtype=torch.uint8), x.new_empty((x.shape[0],), dtype=torch.float32)


@torch.library.custom_op("vpr::topk_merge", mutates_args=())
def topk_merge(vals: Tensor, idxs: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-shard top-k lists [shards, B, k] (global indices) -> merged [B, k], (value desc, index asc).  vpr_topk_merge."""
    return ops.topk_merge(vals, idxs)


@topk_merge.register_fake
def _(vals, idxs):
    _, B, k = vals.shape
    return vals.new_empty((B, k), dtype=torch.float32), vals.new_empty((B, k), dtype=torch.int32)


# ------------------------------------------------------------------------------------------------------------ heads
@torch.library.custom_op("vpr::pose_head", mutates_args=())
def pose_head(x: Tensor, W1: Optional[Tensor], b1: Optional[Tensor], W2: Tensor, b2: Tensor, sincos_offset: int) -> Tensor:
    """W2 relu(W1 x + b1) + b2 in f32 (W1 None: a single Linear), optional unit-normalised (sin, cos) pair at
    `sincos_offset`.  DINOv2RegressionModel.regressor, dinov2salad_validation.py:43-47,52.  vpr_pose_head[_split]."""
    return ops.pose_head(x, W1, b1, W2, b2, sincos_offset)


@pose_head.register_fake
def _(x, W1, b1, W2, b2, sincos_offset):
    return x.new_empty((x.shape[0], W2.shape[0]), dtype=torch.float32)


@torch.library.custom_op("vpr::ln_meanpool_head", mutates_args=())
def ln_meanpool_head(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, Wh: Optional[Tensor], bh: Optional[Tensor],
                     sincos_offset: int) -> Tuple[Tensor, Tensor]:
    """x [B, T, H] (bf16 | f32): pooled = mean_t LayerNorm(x) [B, H] f32; out = Wh pooled + bh [B, n_out] f32 ([B, 0]
    without a head).  HF Swin pooler + Linear, swin_validation.py:43-46.  vpr_ln_meanpool_head."""
    pooled, out = ops.ln_meanpool_head(x, gamma, beta, eps, Wh, bh, sincos_offset, want_pooled=True)
    if out is None:
        out = torch.empty((x.shape[0], 0), dtype=torch.float32, device=x.device)
    return pooled, out


@ln_meanpool_head.register_fake
def _(x, gamma, beta, eps, Wh, bh, sincos_offset):
    B, _, H = x.shape
    return x.new_empty((B, H), dtype=torch.float32), x.new_empty((B, 0 if Wh is None else Wh.shape[0]), dtype=torch.float32)


# ------------------------------------------------------------------------------------------------- head fine-tuning
@torch.library.custom_op("vpr::head_train_epoch", mutates_args=("W1", "b1", "W2", "b2", "m", "v"))
def head_train_epoch(X: Tensor, Y: Tensor, order: Tensor, batch_size: int, W1: Tensor, b1: Tensor, W2: Tensor, b2: Tensor,
                     m: Tensor, v: Tensor, first_step: int, lr: float, beta1: float, beta2: float, eps: float,
                     weight_decay: float, loss: str = "mse", huber_delta: float = 1.0) -> Tensor:
    """One pass of head-only fine-tuning over the cached descriptor rows listed in `order` (int32), batches of `batch_size`:
    forward, MSELoss (or HuberLoss), backward and AdamW of Linear(D,hidden)-ReLU-Linear(hidden,n_out) per batch, IN PLACE on the
    parameters and on the moment buffers (ops.head_train_state).  Returns the batch losses.  dinov2salad_finetuning.py:113-128
    (one epoch of the loop) on descriptors computed once.  vpr_head_train_epoch."""
    return ops.head_train_epoch(X, Y, order, batch_size, W1, b1, W2, b2, m, v, first_step, lr, (beta1, beta2), eps, weight_decay,
                                loss, huber_delta)


@head_train_epoch.register_fake
def _(X, Y, order, batch_size, W1, b1, W2, b2, m, v, first_step, lr, beta1, beta2, eps, weight_decay, loss="mse", huber_delta=1.0):
    return X.new_empty(((order.shape[0] + batch_size - 1) // batch_size,), dtype=torch.float32)


OPS = ("head_train_epoch", "salad_aggregate", "salad_aggregate_split", "salad_aggregate_f32", "knn_topk", "knn_topk_fp8", "quantize_fp8_rows",
       "topk_merge", "pose_head", "ln_meanpool_head")
