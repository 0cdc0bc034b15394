"""CPU brute-force kNN — TEST ORACLE, "parity unpinned" (the reference has no retrieval code;
SURVEY.md §8a-8 defines the stage: cosine top-k of L2-normalised 8448-d descriptors).

Contract (shared with include/vpr_amd.h): operands are bf16; score = fp32 rounding of the exact
dot product (accumulated in fp64: every bf16*bf16 product is exact in fp64); order = score
descending, ties -> lower index; a sharded search merged by the same key gives the same answer.
"""
import torch


def knn_scores_f64(q_bf16: torch.Tensor, g_bf16: torch.Tensor) -> torch.Tensor:
    return q_bf16.to(torch.float64) @ g_bf16.to(torch.float64).T


def knn_topk(q_bf16: torch.Tensor, g_bf16: torch.Tensor, k: int, index_base: int = 0):
    """-> (vals f32 [B,k], idx int32 [B,k]); pads with (-inf, -1) when N < k."""
    s32 = knn_scores_f64(q_bf16, g_bf16).to(torch.float32)
    B, N = s32.shape
    order = torch.sort(-s32, dim=1, stable=True).indices[:, :k]     # stable: lower index first on ties
    vals = torch.gather(s32, 1, order)
    idx = order.to(torch.int32) + index_base
    if N < k:
        pad_v = torch.full((B, k - N), float("-inf"), dtype=torch.float32)
        pad_i = torch.full((B, k - N), -1, dtype=torch.int32)
        vals, idx = torch.cat([vals, pad_v], 1), torch.cat([idx, pad_i], 1)
    return vals, idx


def quantize_fp8_rows(x: torch.Tensor):
    """Per-row symmetric e4m3 quantisation, as vpr_quantize_fp8_rows: scale = max|x|/448 (1 if the
    row is zero), q = fp8_rne(x / scale).  Returns (bytes uint8 [rows,D], scale f32 [rows])."""
    x = x.to(torch.float32)
    m = x.abs().amax(dim=1)
    scale = torch.where(m > 0, m / 448.0, torch.ones_like(m))
    q = (x / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def _order_topk(s32: torch.Tensor, k: int, index_base: int):
    B, N = s32.shape
    order = torch.sort(-s32, dim=1, stable=True).indices[:, :k]
    vals = torch.gather(s32, 1, order)
    idx = order.to(torch.int32) + index_base
    if N < k:
        vals = torch.cat([vals, torch.full((B, k - N), float("-inf"), dtype=torch.float32)], 1)
        idx = torch.cat([idx, torch.full((B, k - N), -1, dtype=torch.int32)], 1)
    return vals, idx


def knn_topk_fp8(q_u8, q_scale, g_u8, g_scale, k: int, index_base: int = 0):
    """fp8 contract: score = f32( ((sum_i q_i g_i exact in f64) * q_scale) * g_scale ), f64 products."""
    qf = q_u8.view(torch.float8_e4m3fn).to(torch.float64)
    gf = g_u8.view(torch.float8_e4m3fn).to(torch.float64)
    s = (qf @ gf.T) * q_scale.to(torch.float64)[:, None]
    s = s * g_scale.to(torch.float64)[None, :]
    return _order_topk(s.to(torch.float32), k, index_base)


def topk_merge(vals: torch.Tensor, idxs: torch.Tensor):
    """vals/idxs [shards,B,k] -> [B,k] by (value desc, index asc); idx < 0 entries are padding."""
    R, B, k = vals.shape
    v = vals.permute(1, 0, 2).reshape(B, R * k).clone()
    i = idxs.permute(1, 0, 2).reshape(B, R * k).to(torch.int64)
    v[i < 0] = float("-inf")
    key_i = torch.where(i < 0, torch.full_like(i, 2**40), i)
    # sort by index first (stable), then by value (stable) -> (value desc, index asc)
    o1 = torch.sort(key_i, dim=1, stable=True).indices
    v1, i1 = torch.gather(v, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.sort(-v1, dim=1, stable=True).indices[:, :k]
    ov, oi = torch.gather(v1, 1, o2), torch.gather(i1, 1, o2)
    return ov, oi.to(torch.int32)


def recall_at_1(top1_idx: torch.Tensor, positives: torch.Tensor) -> float:
    """Fraction of queries whose top-1 equals the planted positive index."""
    return float((top1_idx.to(torch.int64) == positives.to(torch.int64)).double().mean())
