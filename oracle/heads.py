"""CPU restatement of the regression / angle heads — TEST ORACLE.

  mlp_head            DINOv2RegressionModel.regressor  dinov2salad/dinov2salad_validation.py:43-47,52
                      Swin-Base MLP head (Dropout = identity in eval)
                                                       swin_transformer/val_and_test_swin_2.py:168-177
                      sin/cos MLP head   angle_prediction/swin/swin_angle_finetuning_gemini.py:101-106
  linear head         swin_transformer/swin_validation.py:41,46
  unit-normalised [sin, cos]  F.normalize(out, dim=1, p=2, eps=1e-6)
                      angle_prediction/swin/swin_angle_finetuning_sin_cos.py:56-62
  swin_pooler         HF SwinModel: LayerNorm(last hidden) -> mean over tokens
                      (call site swin_transformer/swin_validation.py:43-45)
Pinned by tests/golden/head_dinov2salad.json (reference class imported) and swin_pool_head.npz.
"""
import torch
import torch.nn.functional as F


def _normalize_pair(out: torch.Tensor, off: int) -> torch.Tensor:
    if off is None or off < 0:
        return out
    out = out.clone()
    out[:, off:off + 2] = F.normalize(out[:, off:off + 2], dim=1, p=2, eps=1e-6)
    return out


def mlp_head(x, W1, b1, W2, b2, sincos_offset: int = -1, dtype=torch.float64):
    x, W2, b2 = x.to(dtype), W2.to(dtype), b2.to(dtype)
    if W1 is not None:
        x = torch.relu(x @ W1.to(dtype).T + b1.to(dtype))
    return _normalize_pair(x @ W2.T + b2, sincos_offset)


def swin_pooler(last_hidden, gamma, beta, eps: float, dtype=torch.float64):
    """[B,T,H] -> [B,H]: LayerNorm over H then mean over T."""
    h = last_hidden.to(dtype)
    y = F.layer_norm(h, (h.shape[-1],), gamma.to(dtype), beta.to(dtype), eps)
    return y.mean(dim=1)


def ln_meanpool_head(last_hidden, gamma, beta, eps, Wh=None, bh=None, sincos_offset: int = -1,
                     dtype=torch.float64):
    pooled = swin_pooler(last_hidden, gamma, beta, eps, dtype)
    if Wh is None:
        return pooled, None
    return pooled, _normalize_pair(pooled @ Wh.to(dtype).T + bh.to(dtype), sincos_offset)
