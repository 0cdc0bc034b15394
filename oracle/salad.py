"""CPU restatement of SALAD aggregation (arXiv:2311.15937) — TEST ORACLE, "parity unpinned".

Follows the call site dinov2salad/dinov2salad_validation.py:49-51 (feature_extractor(x) ->
[B, 8448], width pinned at :44); the body comes from torch.hub "serizba/salad" (:65), which is
not in the reference tree and cannot be fetched offline, so the published algorithm is restated
(SURVEY.md §8a-2):

  S = score(x) [B,m,n], F = cluster_features(x) [B,l,n], g = token_features(cls) [B,t]
  append dustbin row -> [B,m+1,n]; log-domain Sinkhorn, `iters` iterations, reg = 1, with
  log a_i = -log(n+m) (i<m), log a_m = log(n-m) - log(n+m), log b_j = -log(n+m);
  each iteration u = log a - LSE_j(S + v), then v = log b - LSE_i(S + u);
  P = exp(S + u + v + log(n+m)); drop the dustbin row;
  V[l,m] = sum_j F[l,j] P[m,j]; L2-normalise V over l; L2-normalise g;
  out = L2-normalise(concat[g, V.flatten() (index l*m + m_idx)]).

The HIP path keeps its operands in bf16 and its hidden activations in bf16; `quantize=True`
mirrors exactly those roundings (and nothing else) so the comparison isolates kernel arithmetic.
"""
import math

import torch
import torch.nn.functional as F


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16, returned in x's dtype."""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


def log_otp_solver(log_a, log_b, M, num_iters: int, reg: float = 1.0):
    M = M / reg
    u = torch.zeros_like(log_a)
    v = torch.zeros_like(log_b)
    for _ in range(num_iters):
        u = log_a - torch.logsumexp(M + v.unsqueeze(1), dim=2)
        v = log_b - torch.logsumexp(M + u.unsqueeze(2), dim=1)
    return M + u.unsqueeze(2) + v.unsqueeze(1)


def matching_probs(S, dustbin: float, num_iters: int = 3, reg: float = 1.0):
    """S [B, m, n] -> P [B, m+1, n] (dustbin row last), scaled so that every column sums to 1."""
    B, m, n = S.shape
    S_aug = torch.empty(B, m + 1, n, dtype=S.dtype)
    S_aug[:, :m, :] = S
    S_aug[:, m, :] = dustbin
    norm = -math.log(n + m)
    log_a = torch.full((B, m + 1), norm, dtype=S.dtype)
    log_a[:, -1] = log_a[:, -1] + math.log(n - m)
    log_b = torch.full((B, n), norm, dtype=S.dtype)
    log_P = log_otp_solver(log_a, log_b, S_aug, num_iters, reg)
    return torch.exp(log_P - norm)


def sinkhorn_aggregate(scores, feats, tokfeat, dustbin: float, iters: int = 3, dtype=torch.float64):
    """scores [B,n,m], feats [B,n,l], tokfeat [B,t]  (token-major, as the HIP stage takes them)."""
    S = scores.to(dtype).transpose(1, 2).contiguous()      # [B, m, n]
    Fm = feats.to(dtype).transpose(1, 2).contiguous()      # [B, l, n]
    g = tokfeat.to(dtype)
    P = matching_probs(S, dustbin, iters)[:, :-1, :]       # [B, m, n]
    V = torch.einsum("bln,bmn->blm", Fm, P)                # [B, l, m]
    V = F.normalize(V, p=2, dim=1)
    out = torch.cat([F.normalize(g, p=2, dim=-1), V.flatten(1)], dim=-1)
    return F.normalize(out, p=2, dim=-1)


def salad_mlps(tokens, w, dtype=torch.float64, quantize: bool = True):
    """tokens [B,1+n,C] (bf16 values); w = dict of the kernel-format weights (CPU tensors).
    Returns scores [B,n,m], feats [B,n,l], tokfeat [B,t]."""
    x = tokens[:, 1:, :].to(dtype)
    cls = tokens[:, 0, :].to(dtype)
    W = {k: v.to(dtype) for k, v in w.items() if torch.is_tensor(v)}
    hidden = W["w1_sc"].shape[0] // 2
    H = torch.relu(x @ W["w1_sc"].T + W["b1_sc"])
    if quantize:
        H = bf16_round(H)
    scores = H[..., :hidden] @ W["w2_s"].T + W["b2_s"]
    feats = H[..., hidden:] @ W["w2_c"].T + W["b2_c"]
    Ht = torch.relu(cls @ W["w1_t"].T + W["b1_t"])
    if quantize:
        Ht = bf16_round(Ht)
    tok = Ht @ W["w2_t"].T + W["b2_t"]
    return scores, feats, tok


def salad_aggregate(tokens, w, dustbin: float, iters: int = 3, dtype=torch.float64, quantize: bool = True):
    scores, feats, tok = salad_mlps(tokens, w, dtype, quantize)
    return sinkhorn_aggregate(scores, feats, tok, dustbin, iters, dtype)
