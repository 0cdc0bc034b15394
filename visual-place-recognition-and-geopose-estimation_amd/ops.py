"""Tensor-level wrappers over the C ABI (include/vpr_amd.h).

Each function takes torch tensors that already live on the GPU, launches the HIP kernels on
torch's current stream and returns torch tensors; PyTorch is used for device memory and streams
only.  Any non-zero status raises RuntimeError — there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import NamedTuple, Optional, Tuple

import torch

from . import _lib

_WORKSPACES = {}


def _raw_stream(device_index: int = -1) -> int:
    """Current HIP stream handle of a device (torch.cuda.current_stream() costs ~8 us of Python per call; this is
    the C entry point underneath it, ~0.3 us — there are ~300 launches per pipeline step)."""
    if device_index < 0:
        device_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(device_index)


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(_raw_stream())


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _need(t: torch.Tensor, dtype: torch.dtype, name: str, ndim: Optional[int] = None) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dims, got {t.dim()}")


def capturing() -> bool:
    """True while the current stream records into a HIP graph: whatever a cache hands out now is baked into the graph
    as a raw address and must stay allocated for as long as the graph may be replayed."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


_GRAPH_PINNED: list = []       # buffers a captured graph addresses, kept alive when their cache entry is replaced


def workspace(name: str, nbytes: int, device: torch.device, stream_key: Optional[int] = None, zero: bool = False) -> torch.Tensor:
    """Cached byte buffer per (device, stream, name); grows, never shrinks.  Keyed by the current stream so that
    two streams driving the library concurrently (e.g. two batches in flight) never share scratch memory.  A buffer
    handed out during graph capture is never freed (a later, larger request gets a new one; the old stays pinned).
    stream_key: raw handle of the stream that OWNS the buffer when the caller is working for it from a helper stream.
    zero: a new buffer is zero-filled (kernels with arrival counters at the head of their workspace need zeros on first use
    and leave zeros behind: vpr_pose_head_fused)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _raw_stream(idx) if stream_key is None else stream_key, name)
    ent = _WORKSPACES.get(key)
    if ent is None or ent[0].numel() < nbytes:
        if ent is not None and ent[1]:
            _GRAPH_PINNED.append(ent[0])
        alloc = torch.zeros if zero else torch.empty
        ent = _WORKSPACES[key] = [alloc(max(int(nbytes), 256), dtype=torch.uint8, device=device), False]
    if not ent[1] and capturing():
        ent[1] = True
    return ent[0]


def drop_stream_caches(raw_stream: int) -> None:
    """Forget every workspace keyed by this stream handle (graphed.GraphedForward.close(): its private stream is gone,
    and so are the graphs that addressed the buffers)."""
    for key in [k for k in _WORKSPACES if k[1] == raw_stream]:
        del _WORKSPACES[key]


# ------------------------------------------------------------------------------------------ SALAD
@dataclass
class SaladWeights:
    """Kernel-format SALAD weights (bf16 [out,in] matrices, f32 biases), all on one GPU.

    w1_sc = cat(score.0.weight, cluster_features.0.weight) [2*hidden, C]; see include/vpr_amd.h.
    """
    w1_sc: torch.Tensor
    b1_sc: torch.Tensor
    w2_s: torch.Tensor
    b2_s: torch.Tensor
    w2_c: torch.Tensor
    b2_c: torch.Tensor
    w1_t: torch.Tensor
    b1_t: torch.Tensor
    w2_t: torch.Tensor
    b2_t: torch.Tensor
    dustbin: float = 1.0

    def validate(self) -> Tuple[int, int, int, int, int]:
        for n in ("w1_sc", "w2_s", "w2_c", "w1_t", "w2_t"):
            _need(getattr(self, n), torch.bfloat16, n, 2)
        for n in ("b1_sc", "b2_s", "b2_c", "b1_t", "b2_t"):
            _need(getattr(self, n), torch.float32, n, 1)
        return self._shapes()

    def _shapes(self) -> Tuple[int, int, int, int, int]:
        hidden2, C = self.w1_sc.shape
        hidden = hidden2 // 2
        m, l, t = self.w2_s.shape[0], self.w2_c.shape[0], self.w2_t.shape[0]
        ok = (self.w2_s.shape[1] == hidden and self.w2_c.shape[1] == hidden and
              tuple(self.w1_t.shape) == (hidden, C) and self.w2_t.shape[1] == hidden and
              self.b1_sc.numel() == 2 * hidden and self.b2_s.numel() == m and
              self.b2_c.numel() == l and self.b1_t.numel() == hidden and self.b2_t.numel() == t)
        if not ok:
            raise RuntimeError("SaladWeights: inconsistent shapes")
        return C, hidden, m, l, t

    def c_struct(self) -> _lib.SaladWeightsC:
        fs, fc = _salad_w2_fragments(self)
        ptrs = [getattr(self, n).data_ptr() for n, _ in _lib.SaladWeightsC._fields_[:10]]
        return _lib.SaladWeightsC(*ptrs, fs.data_ptr() if fs is not None else None, fc.data_ptr() if fc is not None else None)


_SALAD_FRAGS: dict = {}
salad_use_fragments = True      # False: hand the C ABI null *_frag pointers (the kernel then reads W2 row-major; tests / A/B)


def _salad_w2_fragments(w: "SaladWeights"):
    """(w2_s, w2_c) in MFMA fragment order (vpr_salad_pack_w2_fragments), packed once per (storage, version) like the pose
    head's planes: entries keep their source tensors alive (no recycled-address hits) and are pinned once a HIP graph
    has been captured on them."""
    out = []
    if not salad_use_fragments:
        return None, None
    for src in (w.w2_s, w.w2_c):
        n_out, hidden = src.shape
        if src.dtype != torch.bfloat16 or not src.is_cuda or n_out % 16 or hidden % 256:
            out.append(None)
            continue
        key = (src.data_ptr(), src._version, n_out, hidden, str(src.device))
        hit = _SALAD_FRAGS.get(key)
        if hit is None:
            while len(_SALAD_FRAGS) >= 16:
                old = _SALAD_FRAGS.pop(next(iter(_SALAD_FRAGS)))
                if old[2]:
                    _GRAPH_PINNED.append(old)
            frag = torch.empty_like(src)
            st = _lib.lib().vpr_salad_pack_w2_fragments(_ptr(src), n_out, hidden, _ptr(frag), _stream())
            _lib.check(st, "vpr_salad_pack_w2_fragments")
            hit = _SALAD_FRAGS[key] = [frag, src, False]
        if not hit[2] and capturing():
            hit[2] = True
        out.append(hit[0])
    return out[0], out[1]


def salad_aggregate(tokens: torch.Tensor, w: SaladWeights, sinkhorn_iters: int = 3,
                    want_bf16: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """tokens [B, 1+n, C] bf16 (cls first) -> (descriptor f32 [B, t+l*m], bf16 copy or None)."""
    _need(tokens, torch.bfloat16, "tokens", 3)
    C, hidden, m, l, t = w.validate()
    B, tpi, Ct = tokens.shape
    if Ct != C:
        raise RuntimeError(f"tokens have C={Ct}, weights expect {C}")
    n = tpi - 1
    L = _lib.lib()
    nbytes = L.vpr_salad_workspace_bytes(B, n, C, m, l, t, hidden)
    ws = workspace("salad", nbytes, tokens.device, zero=True)
    out = torch.empty((B, t + l * m), dtype=torch.float32, device=tokens.device)
    out16 = torch.empty((B, t + l * m), dtype=torch.bfloat16, device=tokens.device) if want_bf16 else None
    cw = w.c_struct()
    st = L.vpr_salad_aggregate(_ptr(tokens), B, tpi, C, ctypes.byref(cw), float(w.dustbin), m, l, t, hidden,
                               int(sinkhorn_iters), _ptr(out), _ptr(out16), _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_salad_aggregate")
    return out, out16


_SALAD_SIDE: dict = {}


def _salad_side_stream(device: torch.device, main_raw: int) -> torch.cuda.Stream:
    key = (device.index if device.index is not None else torch.cuda.current_device(), main_raw)
    s = _SALAD_SIDE.get(key)
    if s is None:
        s = _SALAD_SIDE[key] = torch.cuda.Stream(device=device)
    return s


def salad_stage_token(cls: torch.Tensor, w: SaladWeights, n: int, owner_raw_stream: int) -> None:
    """Stage T of the aggregation on its own (vpr_salad_stage_token): token MLP of the B cls rows on the CURRENT stream,
    result left in the SALAD workspace that belongs to stream `owner_raw_stream` (the stream that will run
    salad_aggregate_split(..., token_done=True) for the same batch, ordered after this call by the caller).  The DINOv2
    backbone calls it from its cls-row side stream, where the cls tokens are final ~0.3 ms before the patch tokens."""
    _need(cls, torch.bfloat16, "cls", 2)
    C, hidden, m, l, t = w.validate()
    B, Ct = cls.shape
    if Ct != C:
        raise RuntimeError(f"cls {tuple(cls.shape)} does not match weights with C={C}")
    L = _lib.lib()
    ws = workspace("salad", L.vpr_salad_workspace_bytes(B, n, C, m, l, t, hidden), cls.device, stream_key=owner_raw_stream, zero=True)
    cw = w.c_struct()
    st = L.vpr_salad_stage_token(_ptr(cls), C, B, n, C, ctypes.byref(cw), m, l, t, hidden, _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_salad_stage_token")


def salad_aggregate_split(patch: torch.Tensor, cls: torch.Tensor, w: SaladWeights, sinkhorn_iters: int = 3,
                          want_bf16: bool = True, overlap: Optional[bool] = None,
                          token_done: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """patch [B, n, C] bf16 + cls [B, C] bf16 -> (descriptor f32 [B, t+l*m], bf16 copy or None).
    token_done: salad_stage_token already ran for these cls rows (and is ordered before this call): stages M and A only.
    overlap=True: the aggregation runs as its three stages with the token MLP (64 cls rows: two 4 us weight streams) on a
    side stream beside the all-CU score / cluster GEMM, joined before the Sinkhorn stage — same kernels, same workspace,
    bit-identical result to the one-call form.  Off by default: one fork + join between two HIP streams costs ~25 us here;
    the pipeline hides the token MLP for free by running it on the backbone's cls-row stream (salad_stage_token)."""
    _need(patch, torch.bfloat16, "patch", 3)
    _need(cls, torch.bfloat16, "cls", 2)
    C, hidden, m, l, t = w.validate()
    B, n, Ct = patch.shape
    if Ct != C or tuple(cls.shape) != (B, C):
        raise RuntimeError(f"patch {tuple(patch.shape)} / cls {tuple(cls.shape)} do not match weights with C={C}")
    L = _lib.lib()
    ws = workspace("salad", L.vpr_salad_workspace_bytes(B, n, C, m, l, t, hidden), patch.device, zero=True)
    out = torch.empty((B, t + l * m), dtype=torch.float32, device=patch.device)
    out16 = torch.empty((B, t + l * m), dtype=torch.bfloat16, device=patch.device) if want_bf16 else None
    cw = w.c_struct()
    if token_done:
        raw = _stream()
        st = L.vpr_salad_stage_mlps(_ptr(patch), n * C, B, n, C, ctypes.byref(cw), m, l, t, hidden, _ptr(ws), ws.numel(), raw)
        _lib.check(st, "vpr_salad_stage_mlps")
        st = L.vpr_salad_stage_aggregate(B, n, C, float(w.dustbin), m, l, t, hidden, int(sinkhorn_iters), _ptr(out), _ptr(out16),
                                         _ptr(ws), ws.numel(), raw)
        _lib.check(st, "vpr_salad_stage_aggregate")
        return out, out16
    if overlap is None:
        overlap = False     # measured (scripts/salad_ab.py): the fork + join of a side stream costs more than the 10 us it hides
    if not overlap:
        st = L.vpr_salad_aggregate_split(_ptr(patch), _ptr(cls), B, n, C, ctypes.byref(cw), float(w.dustbin), m, l, t,
                                         hidden, int(sinkhorn_iters), _ptr(out), _ptr(out16), _ptr(ws), ws.numel(), _stream())
        _lib.check(st, "vpr_salad_aggregate_split")
        return out, out16
    dev_idx = patch.device.index if patch.device.index is not None else torch.cuda.current_device()
    main_raw = _raw_stream(dev_idx)
    main = torch.cuda.current_stream(patch.device)
    side = _salad_side_stream(patch.device, main_raw)
    side.wait_stream(main)                                   # cls (and the workspace's previous consumer) are ready
    st = L.vpr_salad_stage_token(_ptr(cls), C, B, n, C, ctypes.byref(cw), m, l, t, hidden, _ptr(ws), ws.numel(),
                                 ctypes.c_void_p(side.cuda_stream))
    _lib.check(st, "vpr_salad_stage_token")
    st = L.vpr_salad_stage_mlps(_ptr(patch), n * C, B, n, C, ctypes.byref(cw), m, l, t, hidden, _ptr(ws), ws.numel(),
                                ctypes.c_void_p(main_raw))
    _lib.check(st, "vpr_salad_stage_mlps")
    main.wait_stream(side)
    st = L.vpr_salad_stage_aggregate(B, n, C, float(w.dustbin), m, l, t, hidden, int(sinkhorn_iters), _ptr(out), _ptr(out16),
                                     _ptr(ws), ws.numel(), ctypes.c_void_p(main_raw))
    _lib.check(st, "vpr_salad_stage_aggregate")
    return out, out16


@dataclass
class SaladWeightsF32(SaladWeights):
    """The same ten tensors, all f32 (vpr_salad_weights_f32): operands of the f32-accurate aggregation."""

    def validate(self) -> Tuple[int, int, int, int, int]:
        for n in ("w1_sc", "w2_s", "w2_c", "w1_t", "w2_t"):
            _need(getattr(self, n), torch.float32, n, 2)
        for n in ("b1_sc", "b2_s", "b2_c", "b1_t", "b2_t"):
            _need(getattr(self, n), torch.float32, n, 1)
        return self._shapes()

    def c_struct(self) -> _lib.SaladWeightsF32C:
        return _lib.SaladWeightsF32C(*[getattr(self, n).data_ptr() for n, _ in _lib.SaladWeightsF32C._fields_])


def salad_aggregate_f32(tokens, w: SaladWeightsF32, sinkhorn_iters: int = 3,
                        want_bf16: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """The aggregation at the reference's precision: `tokens` = [B, 1+n, C] f32 (cls first) or a (patch [B,n,C], cls [B,C])
    pair of f32 tensors, f32 weights -> (descriptor f32 [B, t+l*m], bf16 copy or None).  vpr_salad_aggregate_f32."""
    if isinstance(tokens, torch.Tensor):
        _need(tokens, torch.float32, "tokens", 3)
        B, tpi, Ct = tokens.shape
        n = tpi - 1
        patch_ptr, patch_stride = tokens.data_ptr() + 4 * Ct, tpi * Ct
        cls_ptr, cls_stride = tokens.data_ptr(), tpi * Ct
        device = tokens.device
    else:
        patch, cls = tokens
        _need(patch, torch.float32, "patch", 3)
        _need(cls, torch.float32, "cls", 2)
        B, n, Ct = patch.shape
        if tuple(cls.shape) != (B, Ct):
            raise RuntimeError("salad_aggregate_f32: cls must be [B, C]")
        patch_ptr, patch_stride, cls_ptr, cls_stride = patch.data_ptr(), n * Ct, cls.data_ptr(), Ct
        device = patch.device
    C, hidden, m, l, t = w.validate()
    if Ct != C:
        raise RuntimeError(f"tokens have C={Ct}, weights expect {C}")
    L = _lib.lib()
    ws = workspace("salad_f32", L.vpr_salad_f32_workspace_bytes(B, n, C, m, l, t, hidden), device)
    out = torch.empty((B, t + l * m), dtype=torch.float32, device=device)
    out16 = torch.empty((B, t + l * m), dtype=torch.bfloat16, device=device) if want_bf16 else None
    cw = w.c_struct()
    st = L.vpr_salad_aggregate_f32(ctypes.c_void_p(patch_ptr), patch_stride, ctypes.c_void_p(cls_ptr), cls_stride, B, n, C,
                                   ctypes.byref(cw), float(w.dustbin), m, l, t, hidden, int(sinkhorn_iters),
                                   _ptr(out), _ptr(out16), _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_salad_aggregate_f32")
    return out, out16


def salad_sinkhorn_aggregate(scores: torch.Tensor, feats: torch.Tensor, tokfeat: torch.Tensor,
                             dustbin: float, sinkhorn_iters: int = 3,
                             want_bf16: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """scores [B,n,m] f32, feats [B,n,l] f32, tokfeat [B,t] f32 -> descriptor (Sinkhorn stage only)."""
    _need(scores, torch.float32, "scores", 3)
    _need(feats, torch.float32, "feats", 3)
    _need(tokfeat, torch.float32, "tokfeat", 2)
    B, n, m = scores.shape
    l, t = feats.shape[2], tokfeat.shape[1]
    if feats.shape[:2] != (B, n) or tokfeat.shape[0] != B:
        raise RuntimeError("salad_sinkhorn_aggregate: inconsistent shapes")
    out = torch.empty((B, t + l * m), dtype=torch.float32, device=scores.device)
    out16 = torch.empty_like(out, dtype=torch.bfloat16) if want_bf16 else None
    st = _lib.lib().vpr_salad_sinkhorn_aggregate(_ptr(scores), _ptr(feats), _ptr(tokfeat), B, n, m, l, t,
                                                 float(dustbin), int(sinkhorn_iters), _ptr(out), _ptr(out16),
                                                 _stream())
    _lib.check(st, "vpr_salad_sinkhorn_aggregate")
    return out, out16


def gemm_nt_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                 out_dtype: torch.dtype = torch.float32, tile256: bool = False) -> torch.Tensor:
    """act(a @ w.T + bias): a [M,K] bf16, w [N,K] bf16 -> [M,N] f32 or bf16 (MFMA, f32 accumulate)."""
    _need(a, torch.bfloat16, "a", 2)
    _need(w, torch.bfloat16, "w", 2)
    if bias is not None:
        _need(bias, torch.float32, "bias", 1)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or (bias is not None and bias.numel() != N):
        raise RuntimeError("gemm_nt_bf16: inconsistent shapes")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("gemm_nt_bf16: out_dtype must be float32 or bfloat16")
    out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    fn = _lib.lib().vpr_gemm256_nt_bf16 if tile256 else _lib.lib().vpr_gemm_nt_bf16
    st = fn(_ptr(a), K, 0, 0, _ptr(w), K, _ptr(bias), int(relu), _ptr(out), N,
            int(out_dtype == torch.bfloat16), M, N, K, _stream())
    _lib.check(st, "vpr_gemm256_nt_bf16" if tile256 else "vpr_gemm_nt_bf16")
    return out


# -------------------------------------------------------------------------------------------- kNN
def knn_workspace(B: int, N: int, D: int, k: int, device: torch.device) -> torch.Tensor:
    nbytes = _lib.lib().vpr_knn_workspace_bytes(B, N, D, k)
    if nbytes == 0:
        raise RuntimeError(f"vpr_knn: unsupported shape B={B} N={N} D={D} k={k} (need D % 64 == 0, 1 <= k <= 64)")
    return workspace("knn", nbytes, device)


NORM_BOUND_BF16 = 1.002     # L2-normalised rows rounded to bf16 (the unchecked C entry points assume the same)
NORM_BOUND_FP8 = 1.0625     # ... quantised to e4m3 with a per-row scale


def _check_args(B: int, device, status: Optional[torch.Tensor], uncertified: Optional[torch.Tensor]) -> None:
    if status is not None:
        _need(status, torch.int32, "status", 1)
        if status.numel() != B:
            raise RuntimeError("status: one int32 per query")
    if uncertified is not None:
        _need(uncertified, torch.int32, "uncertified", 1)


def knn_topk(q: torch.Tensor, gallery: torch.Tensor, k: int, index_base: int = 0,
             ws: Optional[torch.Tensor] = None, *, norm_bound: float = NORM_BOUND_BF16,
             status: Optional[torch.Tensor] = None, uncertified: Optional[torch.Tensor] = None,
             exact_fallback: bool = False, score_events: Optional[list] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """q [B,D] bf16, gallery [N,D] bf16 -> (scores f32 [B,k] descending, indices int32 [B,k]).
    score_events: a list -> the call runs as its two stages (vpr_knn_topk_scores_stage / _select_stage: same kernels,
    same result) and appends a (start, end) pair of timing events around the score stage.
    The kernels certify each query's answer as the exact top-k (include/vpr_amd.h, "Checked forms"):
    `status` (int32 [B]) receives 0 / 1 (certified) or 2 (not certified), `uncertified` (int32 [1]) counts the 2s
    without a host sync; `norm_bound` = upper bound of the gallery row norms.  exact_fallback=True reads the status
    back (one sync) and re-runs the flagged queries exhaustively, so the result is exact unconditionally."""
    _need(q, torch.bfloat16, "q", 2)
    _need(gallery, torch.bfloat16, "gallery", 2)
    B, D = q.shape
    N = gallery.shape[0]
    if gallery.shape[1] != D:
        raise RuntimeError("knn_topk: q and gallery disagree on D")
    if ws is None:
        ws = knn_workspace(B, N, D, k, q.device)
    if exact_fallback and status is None:
        status = torch.empty((B,), dtype=torch.int32, device=q.device)
    _check_args(B, q.device, status, uncertified)
    vals = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=q.device)
    if score_events is not None:
        _topk_two_stage(q, None, gallery, None, B, N, D, k, index_base, vals, idx, ws, norm_bound, status, uncertified,
                        score_events)
    else:
        st = _lib.lib().vpr_knn_topk_checked(_ptr(q), _ptr(gallery), B, N, D, int(k), int(index_base), _ptr(vals), _ptr(idx),
                                             _ptr(ws), ws.numel(), float(norm_bound), _ptr(status), _ptr(uncertified), _stream())
        _lib.check(st, "vpr_knn_topk_checked")
    if exact_fallback:
        _exhaustive_fixup(q, None, gallery, None, k, index_base, status, vals, idx)
    return vals, idx


def _topk_two_stage(q, q_scale, gallery, gallery_scale, B, N, D, k, index_base, vals, idx, ws, norm_bound, status,
                    uncertified, score_events) -> None:
    fp8 = q_scale is not None
    args = (_ptr(q), _ptr(q_scale) if fp8 else None, _ptr(gallery), _ptr(gallery_scale) if fp8 else None, int(fp8),
            B, N, D, int(k))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st = _lib.lib().vpr_knn_topk_scores_stage(*args, _ptr(ws), ws.numel(), _stream())
    e1.record()
    _lib.check(st, "vpr_knn_topk_scores_stage")
    score_events.append((e0, e1))
    st = _lib.lib().vpr_knn_topk_select_stage(*args, int(index_base), _ptr(vals), _ptr(idx), _ptr(ws), ws.numel(),
                                              float(norm_bound), _ptr(status), _ptr(uncertified), _stream())
    _lib.check(st, "vpr_knn_topk_select_stage")


def knn_topk_exhaustive(q: torch.Tensor, gallery: torch.Tensor, k: int, index_base: int = 0,
                        q_scale: Optional[torch.Tensor] = None, gallery_scale: Optional[torch.Tensor] = None,
                        ws: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Every score computed exactly (f64), then the same selection: exact by construction, slow (fallback for the
    queries the certification flags).  bf16 operands, or uint8 e4m3 operands with both per-row scales."""
    fp8 = q.dtype == torch.uint8
    _need(q, torch.uint8 if fp8 else torch.bfloat16, "q", 2)
    _need(gallery, q.dtype, "gallery", 2)
    if fp8:
        _need(q_scale, torch.float32, "q_scale", 1)
        _need(gallery_scale, torch.float32, "gallery_scale", 1)
    B, D = q.shape
    N = gallery.shape[0]
    if gallery.shape[1] != D:
        raise RuntimeError("knn_topk_exhaustive: q and gallery disagree on D")
    if ws is None:
        ws = knn_workspace(B, N, D, k, q.device)
    vals = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=q.device)
    st = _lib.lib().vpr_knn_topk_exhaustive(_ptr(q), _ptr(q_scale) if fp8 else None, _ptr(gallery),
                                            _ptr(gallery_scale) if fp8 else None, int(fp8), B, N, D, int(k),
                                            int(index_base), _ptr(vals), _ptr(idx), _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_knn_topk_exhaustive")
    return vals, idx


def _exhaustive_fixup(q, q_scale, gallery, gallery_scale, k, index_base, status, vals, idx) -> int:
    """Host side of exact_fallback: one D2H read of `status`, exhaustive re-run of the queries marked 2."""
    bad = torch.nonzero(status == 2).flatten()           # syncs
    if bad.numel() == 0:
        return 0
    qb = q[bad].contiguous()
    v2, i2 = knn_topk_exhaustive(qb, gallery, k, index_base, q_scale[bad].contiguous() if q_scale is not None else None,
                                 gallery_scale)
    vals[bad], idx[bad] = v2, i2
    status[bad] = 3                                       # 3 = exact through the exhaustive path
    return int(bad.numel())


def quantize_fp8_rows(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [rows, D] f32 -> (e4m3 bytes [rows, D] uint8, per-row scale [rows] f32); value = scale * fp8."""
    _need(x, torch.float32, "x", 2)
    rows, D = x.shape
    q = torch.empty((rows, D), dtype=torch.uint8, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    st = _lib.lib().vpr_quantize_fp8_rows(_ptr(x), rows, D, _ptr(q), _ptr(scale), _stream())
    _lib.check(st, "vpr_quantize_fp8_rows")
    return q, scale


def knn_topk_fp8(q: torch.Tensor, q_scale: torch.Tensor, gallery: torch.Tensor, gallery_scale: torch.Tensor,
                 k: int, index_base: int = 0, ws: Optional[torch.Tensor] = None, *, norm_bound: float = NORM_BOUND_FP8,
                 status: Optional[torch.Tensor] = None, uncertified: Optional[torch.Tensor] = None,
                 exact_fallback: bool = False, score_events: Optional[list] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp8 (e4m3 bytes + per-row f32 scale) variant of knn_topk; D % 128 == 0.  norm_bound bounds the norms of the
    DEQUANTISED gallery rows."""
    _need(q, torch.uint8, "q", 2)
    _need(gallery, torch.uint8, "gallery", 2)
    _need(q_scale, torch.float32, "q_scale", 1)
    _need(gallery_scale, torch.float32, "gallery_scale", 1)
    B, D = q.shape
    N = gallery.shape[0]
    if gallery.shape[1] != D or q_scale.numel() != B or gallery_scale.numel() != N:
        raise RuntimeError("knn_topk_fp8: inconsistent shapes")
    if ws is None:
        ws = knn_workspace(B, N, D, k, q.device)
    if exact_fallback and status is None:
        status = torch.empty((B,), dtype=torch.int32, device=q.device)
    _check_args(B, q.device, status, uncertified)
    vals = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=q.device)
    if score_events is not None:
        _topk_two_stage(q, q_scale, gallery, gallery_scale, B, N, D, k, index_base, vals, idx, ws, norm_bound, status,
                        uncertified, score_events)
    else:
        st = _lib.lib().vpr_knn_topk_fp8_checked(_ptr(q), _ptr(q_scale), _ptr(gallery), _ptr(gallery_scale), B, N, D, int(k),
                                                 int(index_base), _ptr(vals), _ptr(idx), _ptr(ws), ws.numel(),
                                                 float(norm_bound), _ptr(status), _ptr(uncertified), _stream())
        _lib.check(st, "vpr_knn_topk_fp8_checked")
    if exact_fallback:
        _exhaustive_fixup(q, q_scale, gallery, gallery_scale, k, index_base, status, vals, idx)
    return vals, idx


def knn_scores(q: torch.Tensor, gallery: torch.Tensor, ws: torch.Tensor) -> None:
    """Stage 1 only (the HBM-bound score kernel); results stay in `ws`."""
    _need(q, torch.bfloat16, "q", 2)
    _need(gallery, torch.bfloat16, "gallery", 2)
    B, D = q.shape
    st = _lib.lib().vpr_knn_scores(_ptr(q), _ptr(gallery), B, gallery.shape[0], D, _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_knn_scores")


def knn_select(q: torch.Tensor, gallery: torch.Tensor, k: int, ws: torch.Tensor,
               index_base: int = 0, *, norm_bound: float = NORM_BOUND_BF16, status: Optional[torch.Tensor] = None,
               uncertified: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stage 2+3 (candidate selection, exact rescoring, ordering, certification) on scores already in `ws`."""
    B, D = q.shape
    _check_args(B, q.device, status, uncertified)
    vals = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=q.device)
    st = _lib.lib().vpr_knn_select_checked(_ptr(q), _ptr(gallery), B, gallery.shape[0], D, int(k), int(index_base),
                                           _ptr(vals), _ptr(idx), _ptr(ws), ws.numel(), float(norm_bound),
                                           _ptr(status), _ptr(uncertified), _stream())
    _lib.check(st, "vpr_knn_select_checked")
    return vals, idx


def knn_scores_view(ws: torch.Tensor, B: int, N: int, D: int, k: int) -> torch.Tensor:
    """View of the score matrix S[B, N] inside a kNN workspace (tests)."""
    ld = ctypes.c_int(0)
    p = _lib.lib().vpr_knn_scores_ptr(_ptr(ws), B, N, D, k, ctypes.byref(ld))
    if not p:
        raise RuntimeError("vpr_knn_scores_ptr: unsupported shape")
    off = p - ws.data_ptr()
    flat = ws[off: off + B * ld.value * 4].view(torch.float32)
    return flat.view(B, ld.value)[:, :N]


def topk_merge(vals: torch.Tensor, idxs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """vals/idxs [shards, B, k] (per-shard top-k with global indices) -> merged [B, k]."""
    _need(vals, torch.float32, "vals", 3)
    _need(idxs, torch.int32, "idxs", 3)
    if vals.shape != idxs.shape:
        raise RuntimeError("topk_merge: vals and idxs shapes differ")
    R, B, k = vals.shape
    ov = torch.empty((B, k), dtype=torch.float32, device=vals.device)
    oi = torch.empty((B, k), dtype=torch.int32, device=vals.device)
    st = _lib.lib().vpr_topk_merge(_ptr(vals), _ptr(idxs), R, B, k, _ptr(ov), _ptr(oi), _stream())
    _lib.check(st, "vpr_topk_merge")
    return ov, oi


# ------------------------------------------------------------------------------------------ heads
_POSE_PLANES: dict = {}


def _pose_w1_planes(W1: torch.Tensor, frag: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(hi, lo) bf16 planes of a first-layer weight, packed once per (storage, version) by vpr_pose_head_pack_w1 (row-major)
    or vpr_pose_head_pack_w1_frag (MFMA fragment order, for the single-launch kernel).
    The entry keeps a reference to W1: while it is cached its storage cannot be freed and handed to another weight
    of the same shape (a recycled address with version 0 would otherwise hit the stale planes)."""
    key = (W1.data_ptr(), W1._version, tuple(W1.shape), str(W1.device), bool(frag))
    hit = _POSE_PLANES.get(key)
    if hit is None:
        while len(_POSE_PLANES) >= 12:
            old = _POSE_PLANES.pop(next(iter(_POSE_PLANES)))  # oldest first (dicts keep insertion order)
            if old[3]:
                _GRAPH_PINNED.append(old)                     # a captured graph reads these planes: never freed
        hi = torch.empty(W1.shape, dtype=torch.bfloat16, device=W1.device)
        lo = torch.empty(W1.shape, dtype=torch.bfloat16, device=W1.device)
        if frag:
            st = _lib.lib().vpr_pose_head_pack_w1_frag(_ptr(W1), W1.shape[0], W1.shape[1], _ptr(hi), _ptr(lo), _stream())
            _lib.check(st, "vpr_pose_head_pack_w1_frag")
        else:
            st = _lib.lib().vpr_pose_head_pack_w1(_ptr(W1), W1.numel(), _ptr(hi), _ptr(lo), _stream())
            _lib.check(st, "vpr_pose_head_pack_w1")
        hit = _POSE_PLANES[key] = [hi, lo, W1, False]
    if not hit[3] and capturing():
        hit[3] = True
    return hit[0], hit[1]


def pose_head(x: torch.Tensor, W1: Optional[torch.Tensor], b1: Optional[torch.Tensor], W2: torch.Tensor,
              b2: torch.Tensor, sincos_offset: int = -1, split: bool = True, fused: bool = False) -> torch.Tensor:
    """W2 relu(W1 x + b1) + b2 in f32 (W1 None -> single Linear); optional unit-normalised pair.
    split: first layer as four bf16 MFMAs on (hi, lo) planes of x and W1 (f32 accuracy, ~3x faster than the
    exact-f32 MFMA path, which split=False keeps).  fused=True (with split): vpr_pose_head_fused — fragment-order weight
    planes and, with VPR_POSE_VARIANT=1, ONE launch whose split-K slabs are finished by arrival counters.  Measured
    (scripts/pose_ab.py, B = 64, D = 8448, hidden 1024): one launch 31.8 us, fragment planes + epilogue launch 22.6 us,
    row-major planes + epilogue launch (vpr_pose_head_split, the default) 21.8 us — the serial finisher tail of the
    counter form costs more than the second launch it removes.  All forms are bitwise reproducible."""
    _need(x, torch.float32, "x", 2)
    _need(W2, torch.float32, "W2", 2)
    _need(b2, torch.float32, "b2", 1)
    B, D = x.shape
    n_out = W2.shape[0]
    hidden = 0
    if W1 is not None:
        _need(W1, torch.float32, "W1", 2)
        _need(b1, torch.float32, "b1", 1)
        hidden = W1.shape[0]
        if W1.shape[1] != D or b1.numel() != hidden or W2.shape[1] != hidden:
            raise RuntimeError("pose_head: inconsistent MLP shapes")
    elif W2.shape[1] != D:
        raise RuntimeError("pose_head: W2 must be [n_out, D] for the linear head")
    if b2.numel() != n_out:
        raise RuntimeError("pose_head: b2 size")
    L = _lib.lib()
    out = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    fused_bytes = L.vpr_pose_head_fused_workspace_bytes(B, D, hidden) if (hidden > 0 and split and fused) else 0
    if fused_bytes > 0:
        hi, lo = _pose_w1_planes(W1, frag=True)
        ws = workspace("pose_fused", fused_bytes, x.device, zero=True)
        st = L.vpr_pose_head_fused(_ptr(x), _ptr(hi), _ptr(lo), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(out), B, D, hidden,
                                   n_out, int(sincos_offset), _ptr(ws), ws.numel(), _stream())
        _lib.check(st, "vpr_pose_head_fused")
        return out
    if hidden > 0 and split and D % 32 == 0 and hidden % 16 == 0:
        hi, lo = _pose_w1_planes(W1)
        ws = workspace("pose", L.vpr_pose_head_split_workspace_bytes(B, D, hidden), x.device)
        st = L.vpr_pose_head_split(_ptr(x), _ptr(hi), _ptr(lo), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(out), B, D, hidden,
                                   n_out, int(sincos_offset), _ptr(ws), ws.numel(), _stream())
        _lib.check(st, "vpr_pose_head_split")
        return out
    ws = workspace("pose", L.vpr_pose_head_workspace_bytes(B, D, hidden, n_out), x.device)
    st = L.vpr_pose_head(_ptr(x), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(out), B, D, hidden, n_out,
                         int(sincos_offset), _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_pose_head")
    return out


def head_train_state(W1: torch.Tensor, W2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Zeroed AdamW moment buffers (m, v) for vpr_head_train_step, laid out [W1 | b1 | W2 | b2]."""
    hidden, D = W1.shape
    n = _lib.lib().vpr_head_train_state_floats(D, hidden, W2.shape[0])
    return (torch.zeros(n, dtype=torch.float32, device=W1.device), torch.zeros(n, dtype=torch.float32, device=W1.device))


def head_train_state_views(buf: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor):
    """The four per-parameter views (W1, b1, W2, b2 shaped) of one moment buffer."""
    hidden, D = W1.shape
    n_out = W2.shape[0]
    o1 = hidden * D
    o2 = o1 + hidden
    o3 = o2 + n_out * hidden
    return buf[:o1].view(hidden, D), buf[o1:o2], buf[o2:o3].view(n_out, hidden), buf[o3:o3 + n_out]


def _loss_kind(loss: str) -> int:
    """VPR_LOSS_MSE / VPR_LOSS_HUBER of include/vpr_amd.h."""
    if loss not in ("mse", "huber"):
        raise RuntimeError(f"head_train: loss must be 'mse' (nn.MSELoss) or 'huber' (nn.HuberLoss), got {loss!r}")
    return 1 if loss == "huber" else 0


def _head_train_check(X, Y, W1, b1, W2, b2, m, v):
    _need(X, torch.float32, "X", 2)
    _need(Y, torch.float32, "Y", 2)
    _need(W1, torch.float32, "W1", 2)
    _need(b1, torch.float32, "b1", 1)
    _need(W2, torch.float32, "W2", 2)
    _need(b2, torch.float32, "b2", 1)
    _need(m, torch.float32, "m", 1)
    _need(v, torch.float32, "v", 1)
    hidden, D = W1.shape
    n_out = W2.shape[0]
    if X.shape[1] != D or Y.shape[0] != X.shape[0] or Y.shape[1] != n_out or b1.numel() != hidden or W2.shape[1] != hidden \
            or b2.numel() != n_out:
        raise RuntimeError("head_train: inconsistent shapes")
    if m.numel() != _lib.lib().vpr_head_train_state_floats(D, hidden, n_out) or v.numel() != m.numel():
        raise RuntimeError("head_train: moment buffers must hold vpr_head_train_state_floats() floats (ops.head_train_state)")
    return D, hidden, n_out


def head_train_epoch(X: torch.Tensor, Y: torch.Tensor, order: torch.Tensor, batch_size: int, W1: torch.Tensor, b1: torch.Tensor,
                     W2: torch.Tensor, b2: torch.Tensor, m: torch.Tensor, v: torch.Tensor, first_step: int, lr: float = 1e-5,
                     betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, loss: str = "mse",
                     huber_delta: float = 1.0) -> torch.Tensor:
    """One pass over the rows listed in `order` (int32, device) in batches of `batch_size` (vpr_head_train_epoch: the whole
    launch sequence enqueued by ONE library call).  Returns the batch losses [ceil(n / batch_size)] (device tensor; nothing
    waits for the GPU).  The caller guarantees 0 <= order < X.shape[0]: the kernels gather rows by these indices unchecked."""
    D, hidden, n_out = _head_train_check(X, Y, W1, b1, W2, b2, m, v)
    _need(order, torch.int32, "order", 1)
    n = order.numel()
    if n < 1 or batch_size < 1:
        raise RuntimeError("head_train_epoch: empty pass")
    L = _lib.lib()
    nbytes = L.vpr_head_train_workspace_bytes(min(batch_size, n), D, hidden, n_out)
    if nbytes == 0:
        raise RuntimeError(f"head_train_epoch: unsupported shape B={min(batch_size, n)} D={D} hidden={hidden} n_out={n_out} "
                           "(need 1 <= B <= 64, D % 16 == 0, hidden % 32 == 0, n_out <= 8)")
    ws = workspace("head_train", nbytes, X.device)
    losses = torch.empty((n + batch_size - 1) // batch_size, dtype=torch.float32, device=X.device)
    st = L.vpr_head_train_epoch(_ptr(X), X.stride(0), _ptr(order), n, int(batch_size), _ptr(Y), Y.stride(0), D, hidden, n_out,
                                _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(m), _ptr(v), int(first_step), float(lr),
                                float(betas[0]), float(betas[1]), float(eps), float(weight_decay), _loss_kind(loss),
                                float(huber_delta), _ptr(losses), _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_head_train_epoch")
    for t in (W1, b1, W2, b2, m, v):
        torch.autograd.graph.increment_version(t)
    return losses


def head_train_step(X: torch.Tensor, Y: torch.Tensor, idx: Optional[torch.Tensor], W1: torch.Tensor, b1: torch.Tensor,
                    W2: torch.Tensor, b2: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float = 1e-5,
                    betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                    loss_out: Optional[torch.Tensor] = None, loss: str = "mse", huber_delta: float = 1.0) -> None:
    """One batch of head-only fine-tuning on cached descriptors (vpr_head_train_step): forward, MSELoss, backward and
    AdamW update of Linear(D,hidden)-ReLU-Linear(hidden,n_out), in place on W1 / b1 / W2 / b2 / m / v.
    X [rows, D] f32, Y [rows, n_out] f32, idx [B] int32 (rows of the batch; None = all rows of X in order).  loss_out: a
    one-element f32 tensor (e.g. losses[i:i+1]) that receives the batch loss.  loss: "mse" (nn.MSELoss) or "huber"
    (nn.HuberLoss(delta=huber_delta)).  No host synchronisation."""
    D, hidden, n_out = _head_train_check(X, Y, W1, b1, W2, b2, m, v)
    if idx is not None:
        _need(idx, torch.int32, "idx", 1)
        B = idx.numel()
    else:
        B = X.shape[0]
    L = _lib.lib()
    if loss_out is not None:
        _need(loss_out, torch.float32, "loss_out")
        if loss_out.numel() != 1:
            raise RuntimeError("head_train_step: loss_out must have one element")
    nbytes = L.vpr_head_train_workspace_bytes(B, D, hidden, n_out)
    if nbytes == 0:
        raise RuntimeError(f"head_train_step: unsupported shape B={B} D={D} hidden={hidden} n_out={n_out} "
                           "(need 1 <= B <= 64, D % 16 == 0, hidden % 32 == 0, n_out <= 8)")
    ws = workspace("head_train", nbytes, X.device)
    st = L.vpr_head_train_step(_ptr(X), X.stride(0), _ptr(idx), _ptr(Y), Y.stride(0), B, D, hidden, n_out,
                               _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(m), _ptr(v), int(step), float(lr),
                               float(betas[0]), float(betas[1]), float(eps), float(weight_decay), _loss_kind(loss),
                               float(huber_delta), _ptr(loss_out), _ptr(ws), ws.numel(), _stream())
    _lib.check(st, "vpr_head_train_step")
    for t in (W1, b1, W2, b2, m, v):          # written behind PyTorch's back: version-keyed caches (pose-head weight planes) must see it
        torch.autograd.graph.increment_version(t)


def ln_meanpool_head(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                     Wh: Optional[torch.Tensor] = None, bh: Optional[torch.Tensor] = None,
                     sincos_offset: int = -1, want_pooled: bool = True
                     ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """x [B,T,H] (bf16|f32): pooled = mean_t LayerNorm(x); out = Wh pooled + bh.  -> (pooled, out)."""
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError("ln_meanpool_head: x must be bf16 or f32")
    _need(x, x.dtype, "x", 3)
    _need(gamma, torch.float32, "gamma", 1)
    _need(beta, torch.float32, "beta", 1)
    B, T, H = x.shape
    n_out = 0
    if Wh is not None:
        _need(Wh, torch.float32, "Wh", 2)
        _need(bh, torch.float32, "bh", 1)
        n_out = Wh.shape[0]
        if Wh.shape[1] != H or bh.numel() != n_out:
            raise RuntimeError("ln_meanpool_head: head shapes")
    pooled = torch.empty((B, H), dtype=torch.float32, device=x.device) if (want_pooled or Wh is None) else None
    out = torch.empty((B, n_out), dtype=torch.float32, device=x.device) if Wh is not None else None
    st = _lib.lib().vpr_ln_meanpool_head(_ptr(x), int(x.dtype == torch.bfloat16), B, T, H, _ptr(gamma), _ptr(beta),
                                         float(eps), _ptr(pooled), _ptr(Wh), _ptr(bh), n_out, int(sincos_offset),
                                         _ptr(out), _stream())
    _lib.check(st, "vpr_ln_meanpool_head")
    return pooled, out


def f32_to_bf16(src: torch.Tensor) -> torch.Tensor:
    _need(src, torch.float32, "src")
    dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    st = _lib.lib().vpr_f32_to_bf16(_ptr(src), _ptr(dst), src.numel(), _stream())
    _lib.check(st, "vpr_f32_to_bf16")
    return dst


def layernorm_bf16(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm over the last dim of a contiguous bf16 tensor (gamma/beta bf16 or f32) -> bf16."""
    _need(x, torch.bfloat16, "x")
    if gamma.dtype not in (torch.bfloat16, torch.float32) or beta.dtype != gamma.dtype:
        raise RuntimeError("layernorm_bf16: gamma/beta must both be bf16 or both f32")
    _need(gamma, gamma.dtype, "gamma", 1)
    _need(beta, beta.dtype, "beta", 1)
    C = x.shape[-1]
    if gamma.numel() != C or beta.numel() != C:
        raise RuntimeError("layernorm_bf16: parameter size")
    y = torch.empty_like(x)
    st = _lib.lib().vpr_layernorm_bf16(_ptr(x), _ptr(gamma), _ptr(beta), int(gamma.dtype == torch.bfloat16), float(eps),
                                       _ptr(y), x.numel() // C, C, _stream())
    _lib.check(st, "vpr_layernorm_bf16")
    return y


def patchify_bf16(images: torch.Tensor, patch: int, kpad: int, lead_rows: int = 1) -> torch.Tensor:
    """images [B, Cin, H, W] bf16 -> flattened patches [B * (lead_rows + n), kpad] bf16 in token order
    (lead_rows zero rows per image for the cls slot; K zero-padded to kpad)."""
    _need(images, torch.bfloat16, "images", 4)
    B, Cin, H, W = images.shape
    n = (H // patch) * (W // patch)
    out = torch.empty((B * (lead_rows + n), kpad), dtype=torch.bfloat16, device=images.device)
    st = _lib.lib().vpr_patchify_bf16(_ptr(images), B, Cin, H, W, int(patch), int(kpad), int(lead_rows), _ptr(out), _stream())
    _lib.check(st, "vpr_patchify_bf16")
    return out


def bias_layernorm_bf16(x: torch.Tensor, pre_bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                        eps: float) -> torch.Tensor:
    """LayerNorm(f32(x) + pre_bias) -> bf16; pre_bias [C] f32 is added before the statistics."""
    _need(x, torch.bfloat16, "x")
    _need(pre_bias, torch.float32, "pre_bias", 1)
    if gamma.dtype not in (torch.bfloat16, torch.float32) or beta.dtype != gamma.dtype:
        raise RuntimeError("bias_layernorm_bf16: gamma/beta must both be bf16 or both f32")
    _need(gamma, gamma.dtype, "gamma", 1)
    _need(beta, beta.dtype, "beta", 1)
    C = x.shape[-1]
    if gamma.numel() != C or beta.numel() != C or pre_bias.numel() != C:
        raise RuntimeError("bias_layernorm_bf16: parameter size")
    y = torch.empty_like(x)
    st = _lib.lib().vpr_bias_layernorm_bf16(_ptr(x), _ptr(pre_bias), _ptr(gamma), _ptr(beta),
                                            int(gamma.dtype == torch.bfloat16), float(eps), _ptr(y),
                                            x.numel() // C, C, _stream())
    _lib.check(st, "vpr_bias_layernorm_bf16")
    return y


def add_layernorm_bf16(x: torch.Tensor, res: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                       eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(x + res rounded to bf16, LayerNorm of that sum): the residual add fused into the next norm."""
    _need(x, torch.bfloat16, "x")
    _need(res, torch.bfloat16, "res")
    if res.shape != x.shape:
        raise RuntimeError("add_layernorm_bf16: x and res shapes differ")
    if gamma.dtype not in (torch.bfloat16, torch.float32) or beta.dtype != gamma.dtype:
        raise RuntimeError("add_layernorm_bf16: gamma/beta must both be bf16 or both f32")
    C = x.shape[-1]
    if gamma.numel() != C or beta.numel() != C:
        raise RuntimeError("add_layernorm_bf16: parameter size")
    s, y = torch.empty_like(x), torch.empty_like(x)
    st = _lib.lib().vpr_add_layernorm_bf16(_ptr(x), _ptr(res), _ptr(s), _ptr(gamma), _ptr(beta),
                                           int(gamma.dtype == torch.bfloat16), float(eps), _ptr(y),
                                           x.numel() // C, C, _stream())
    _lib.check(st, "vpr_add_layernorm_bf16")
    return s, y


def attention_qkv_bf16(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """qkv [B, T, 3*H*64] bf16 (fused projection output) -> softmax(q k^T / 8) v as [B, T, H*64] bf16."""
    _need(qkv, torch.bfloat16, "qkv", 3)
    B, T, C3 = qkv.shape
    C = C3 // 3
    if C3 != 3 * C or C % heads or C // heads != 64:
        raise RuntimeError("attention_qkv_bf16: needs head_dim 64 and a [B,T,3*H*64] input")
    out = torch.empty((B, T, C), dtype=torch.bfloat16, device=qkv.device)
    st = _lib.lib().vpr_attention_qkv_bf16(_ptr(qkv), _ptr(out), B, T, heads, 64, 0.125, _stream())
    _lib.check(st, "vpr_attention_qkv_bf16")
    return out


def attention_qkv_split_bf16(qkv: torch.Tensor, B: int, T: int, body_tokens: int, heads: int) -> torch.Tensor:
    """Same attention on the backbone's split row layout: qkv [B*T, 3*H*64] with token t of image b in
    row b*body_tokens + t (t < body_tokens) or B*body_tokens + b*(T-body_tokens) + (t-body_tokens)."""
    _need(qkv, torch.bfloat16, "qkv", 2)
    rows, C3 = qkv.shape
    C = C3 // 3
    if C3 != 3 * C or C % heads or C // heads != 64 or rows != B * T or not (0 <= body_tokens <= T):
        raise RuntimeError("attention_qkv_split_bf16: needs head_dim 64 and a [B*T, 3*H*64] input")
    out = torch.empty((rows, C), dtype=torch.bfloat16, device=qkv.device)
    st = _lib.lib().vpr_attention_qkv_split_bf16(_ptr(qkv), _ptr(out), B, T, int(body_tokens), B * int(body_tokens),
                                                 heads, 64, 0.125, _stream())
    _lib.check(st, "vpr_attention_qkv_split_bf16")
    return out


def skinny_linear_bf16(inp: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor,
                       mode: int = 0, stats_bias: Optional[torch.Tensor] = None,
                       row_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Linear layer on a few rows, written into `out` (a row slice of a larger buffer is fine):
    mode 0 out = inp W^T + b; 1 gelu_tanh(inp W^T + b); 2 out += inp W^T; 3 relu; 4 erf-GELU.  With row_stats [N/16, M, 2] f32 it
    also leaves the per-16-column (mean, M2) of bf16(out) + stats_bias there (LayerNorm statistics partials)."""
    for t, name in ((inp, "inp"), (weight, "weight"), (out, "out")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"skinny_linear_bf16: {name} must be a GPU bf16 matrix with unit column stride")
    M, K = inp.shape
    N = weight.shape[0]
    if weight.shape[1] != K or tuple(out.shape) != (M, N):
        raise RuntimeError("skinny_linear_bf16: shape mismatch")
    if mode != 2:
        if bias is None or not bias.is_cuda or bias.numel() != N or bias.dtype not in (torch.bfloat16, torch.float32) \
                or not bias.is_contiguous():
            raise RuntimeError("skinny_linear_bf16: bias [N] bf16/f32 required")
    if row_stats is not None:
        _need(row_stats, torch.float32, "row_stats", 3)
        if tuple(row_stats.shape) != (N // 16, M, 2) or N % 16:
            raise RuntimeError("skinny_linear_bf16: row_stats must be [N/16, M, 2] with N % 16 == 0")
        if stats_bias is not None:
            _need(stats_bias, torch.float32, "stats_bias", 1)
        st = _lib.lib().vpr_skinny_linear_stats_bf16(_ptr(inp), inp.stride(0), _ptr(weight), weight.stride(0),
                                                     _ptr(bias) if mode != 2 else None,
                                                     int(bias is not None and bias.dtype == torch.bfloat16), int(mode),
                                                     _ptr(out), out.stride(0), M, N, K, _ptr(stats_bias), _ptr(row_stats),
                                                     _stream())
        _lib.check(st, "vpr_skinny_linear_stats_bf16")
        return out
    st = _lib.lib().vpr_skinny_linear_bf16(_ptr(inp), inp.stride(0), _ptr(weight), weight.stride(0), _ptr(bias) if mode != 2 else None,
                                           int(bias is not None and bias.dtype == torch.bfloat16), int(mode),
                                           _ptr(out), out.stride(0), M, N, K, _stream())
    _lib.check(st, "vpr_skinny_linear_bf16")
    return out


class ClsLinearConsts(NamedTuple):
    """Static operands of the LayerNorm-fused cls-row linear (see vpr_bias_layernorm_cls_linear_bf16)."""
    w_scaled: torch.Tensor     # [N, C] bf16 = W * gamma
    colsum: torch.Tensor       # [N] f32
    cprime: torch.Tensor       # [N] f32 = W' pre_bias
    bprime: torch.Tensor       # [N] f32 = b + W beta

    @staticmethod
    def build(weight: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
              pre_bias: Optional[torch.Tensor]) -> "ClsLinearConsts":
        w32 = weight.detach().float()
        ws = (w32 * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
        ws32 = ws.float()
        cprime = ws32 @ pre_bias.float() if pre_bias is not None else torch.zeros(weight.shape[0], device=weight.device)
        return ClsLinearConsts(ws, ws32.sum(1).contiguous(), cprime.contiguous(),
                               (bias.detach().float() + w32 @ beta.detach().float()).contiguous())


def bias_layernorm_cls_linear_bf16(x: torch.Tensor, pre_bias: Optional[torch.Tensor], gamma: torch.Tensor,
                                   beta: torch.Tensor, eps: float, cls_row0: int, row_stats: torch.Tensor,
                                   consts: ClsLinearConsts, out: torch.Tensor, gelu: bool = False) -> torch.Tensor:
    """y = LayerNorm(x + pre_bias) for every row of x [M, C] (returned), and in the same launch
    out[:] = act(y[cls_row0 : cls_row0 + out.shape[0]] W^T + b) (act = tanh-GELU if gelu), from the raw rows and
    `consts` = ClsLinearConsts.build(W, b, gamma, beta, pre_bias).  row_stats: the statistics partials of those
    rows, left by skinny_linear_bf16(..., row_stats=...) when it wrote them."""
    _need(x, torch.bfloat16, "x", 2)
    for t, name in ((gamma, "gamma"), (beta, "beta")):
        _need(t, torch.bfloat16, name, 1)
    if pre_bias is not None:
        _need(pre_bias, torch.float32, "pre_bias", 1)
    M, C = x.shape
    n_cls, N = out.shape
    _need(consts.w_scaled, torch.bfloat16, "w_scaled", 2)
    for t, name in ((consts.colsum, "colsum"), (consts.cprime, "cprime"), (consts.bprime, "bprime")):
        _need(t, torch.float32, name, 1)
        if t.numel() != N:
            raise RuntimeError("bias_layernorm_cls_linear_bf16: constant vector size")
    if not out.is_cuda or out.dtype != torch.bfloat16 or out.stride(1) != 1:
        raise RuntimeError("bias_layernorm_cls_linear_bf16: out must be a GPU bf16 matrix with unit column stride")
    if consts.w_scaled.shape != (N, C) or gamma.numel() != C or beta.numel() != C:
        raise RuntimeError("bias_layernorm_cls_linear_bf16: shape mismatch")
    _need(row_stats, torch.float32, "row_stats", 3)
    if tuple(row_stats.shape) != (C // 16, n_cls, 2):
        raise RuntimeError("bias_layernorm_cls_linear_bf16: row_stats must be [C/16, n_cls, 2]")
    y = torch.empty_like(x)
    st = _lib.lib().vpr_bias_layernorm_cls_linear_bf16(_ptr(x), _ptr(pre_bias), _ptr(gamma), _ptr(beta), float(eps), _ptr(y),
                                                       M, C, int(cls_row0), n_cls, _ptr(row_stats), _ptr(consts.w_scaled),
                                                       consts.w_scaled.stride(0), _ptr(consts.colsum), _ptr(consts.cprime),
                                                       _ptr(consts.bprime), int(gelu), _ptr(out), out.stride(0), N, _stream())
    _lib.check(st, "vpr_bias_layernorm_cls_linear_bf16")
    return y
