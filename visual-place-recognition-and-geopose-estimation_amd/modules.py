"""Host-side mirror of the reference's model interface, with the hot ops running as HIP kernels.

Same class names, constructor arguments, sub-module names and state-dict keys as the reference,
so its checkpoints load and its evaluation loops run unchanged:

  DINOv2RegressionModel(base_model)     dinov2salad/dinov2salad_validation.py:36-52
        .feature_extractor (frozen)  -> [B, 8448]; .regressor = Linear(8448,512)-ReLU-Linear(512,2)
        keys regressor.0.weight/bias, regressor.2.weight/bias
  SwinRegressionModel(backbone)         swin_transformer/swin_validation.py:37-46
        .backbone, .regressor = Linear(hidden, 2); key regressor.weight/bias
  SwinMLPRegressionModel(backbone)      swin_transformer/val_and_test_swin_2.py:164-177
        .regressor = Linear(hidden,512)-ReLU-Dropout-Linear(512,2); keys regressor.0.*, regressor.3.*
  SwinSinCosRegressionModel(backbone)   angle_prediction/swin/swin_angle_finetuning_sin_cos.py:52-62
        Linear(hidden,2) + F.normalize(eps=1e-6) -> unit [sin, cos]
  SwinAngleRegressorSinCos(backbone)    angle_prediction/swin/swin_angle_finetuning_gemini.py:92-128
        Linear(H,H/2)-ReLU-Dropout-Linear(H/2,2) -> raw [sin, cos]; keys regressor.0.*, regressor.3.*
  DinoV2AngleRegressorSinCos(backbone)  angle_prediction/dinov2salad/dino_v2_gemini.py:99-114
        DINOv2 CLS -> Dropout -> head = Linear(hidden,2) -> raw [sin, cos]; keys backbone.* (HF layout accepted), head.*
The reference builds its backbones with from_pretrained(NAME) inside __init__ (a network fetch);
here the backbone object is passed in.  `DinoV2Salad` is the `feature_extractor`:
DINOv2 (PyTorch) + SaladAggregator (HIP), keys `aggregator.*` as in serizba/salad.
All forwards are inference-only (torch.no_grad) and require GPU tensors.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from . import torch_ops  # noqa: F401  registers torch.ops.vpr.* (torch.library custom ops over the C ABI)
from .backbone import DinoV2, SplitTokens

_vpr = torch.ops.vpr      # the forwards below call the dispatcher-visible ops (CUDA impl = ctypes -> C ABI; fake impl for tracing)


class SaladAggregator(nn.Module):
    """Parameters laid out like serizba/salad's SALAD module (1x1 convs for score / cluster
    features, Linears for the token features, scalar dust_bin); forward = vpr_salad_aggregate."""

    def __init__(self, num_channels: int = 1024, num_clusters: int = 64, cluster_dim: int = 128,
                 token_dim: int = 256, hidden: int = 512):
        super().__init__()
        self.num_channels, self.num_clusters = num_channels, num_clusters
        self.cluster_dim, self.token_dim, self.hidden = cluster_dim, token_dim, hidden
        self.token_features = nn.Sequential(nn.Linear(num_channels, hidden), nn.ReLU(), nn.Linear(hidden, token_dim))
        self.cluster_features = nn.Sequential(nn.Conv2d(num_channels, hidden, 1), nn.Dropout(0.3), nn.ReLU(),
                                              nn.Conv2d(hidden, cluster_dim, 1))
        self.score = nn.Sequential(nn.Conv2d(num_channels, hidden, 1), nn.Dropout(0.3), nn.ReLU(),
                                   nn.Conv2d(hidden, num_clusters, 1))
        self.dust_bin = nn.Parameter(torch.tensor(1.0))
        self._packed: Optional[ops.SaladWeights] = None
        self._packed_f32: Optional[ops.SaladWeightsF32] = None

    def pack(self) -> ops.SaladWeights:
        """Kernel-format weights (bf16 matrices, f32 biases); call again after loading a state dict."""
        m2 = lambda w: w.detach().reshape(w.shape[0], -1)
        bf = lambda w: w.to(torch.bfloat16).contiguous()
        f32 = lambda b: b.detach().to(torch.float32).contiguous()
        s, c, t = self.score, self.cluster_features, self.token_features
        self._packed = ops.SaladWeights(
            w1_sc=bf(torch.cat([m2(s[0].weight), m2(c[0].weight)], 0)), b1_sc=f32(torch.cat([s[0].bias, c[0].bias], 0)),
            w2_s=bf(m2(s[3].weight)), b2_s=f32(s[3].bias), w2_c=bf(m2(c[3].weight)), b2_c=f32(c[3].bias),
            w1_t=bf(m2(t[0].weight)), b1_t=f32(t[0].bias), w2_t=bf(m2(t[2].weight)), b2_t=f32(t[2].bias),
            dustbin=float(self.dust_bin.detach().cpu()))
        return self._packed

    def pack_f32(self) -> ops.SaladWeightsF32:
        """f32 kernel-format weights for the f32-accurate aggregation (vpr_salad_aggregate_f32): the parameters as they
        are, no bf16 rounding (the reference keeps the aggregator in fp32: dinov2salad_validation.py:65-66)."""
        m2 = lambda w: w.detach().reshape(w.shape[0], -1).to(torch.float32).contiguous()
        f32 = lambda b: b.detach().to(torch.float32).contiguous()
        s, c, t = self.score, self.cluster_features, self.token_features
        self._packed_f32 = ops.SaladWeightsF32(
            w1_sc=torch.cat([m2(s[0].weight), m2(c[0].weight)], 0).contiguous(), b1_sc=f32(torch.cat([s[0].bias, c[0].bias], 0)),
            w2_s=m2(s[3].weight), b2_s=f32(s[3].bias), w2_c=m2(c[3].weight), b2_c=f32(c[3].bias),
            w1_t=m2(t[0].weight), b1_t=f32(t[0].bias), w2_t=m2(t[2].weight), b2_t=f32(t[2].bias),
            dustbin=float(self.dust_bin.detach().cpu()))
        return self._packed_f32

    def token_stage(self, cls_rows: torch.Tensor, owner_raw_stream: int) -> None:
        """The token MLP of these cls rows on the current stream, into the workspace of `owner_raw_stream`
        (backbone.DinoV2.cls_tail_hook: the backbone's cls-row stream has the cls tokens ~0.3 ms before the main stream has
        the patch tokens, so the stage costs the step nothing)."""
        w = self._packed or self.pack()
        ops.salad_stage_token(cls_rows, w, 256, owner_raw_stream)

    @torch.no_grad()
    def forward(self, tokens, want_bf16: bool = False):
        """tokens [B, 1+n, C] (cls first) or a backbone.SplitTokens pair -> descriptor [B, 8448] f32 (and a bf16 copy).
        bf16 tokens: the bf16-operand kernels (benchmark path); f32 tokens: the f32-accurate aggregation with f32
        weights — the reference's precision (its extractor runs in fp32)."""
        first = tokens.patch if isinstance(tokens, SplitTokens) else tokens
        if first.dtype == torch.float32:
            w32 = self._packed_f32 or self.pack_f32()
            if isinstance(tokens, SplitTokens):
                patch, cls = tokens.patch.contiguous(), tokens.cls.contiguous()
            else:
                patch, cls = tokens[:, 1:].contiguous(), tokens[:, 0].contiguous()
            desc, desc16 = _vpr.salad_aggregate_f32(patch, cls, torch_ops.weight_list(w32), w32.dustbin, 3)
            return (desc, desc16) if want_bf16 else desc
        w = self._packed or self.pack()
        if isinstance(tokens, SplitTokens):
            desc, desc16 = _vpr.salad_aggregate_split(tokens.patch, tokens.cls, torch_ops.weight_list(w), w.dustbin, 3,
                                                      bool(tokens.token_ready))
        else:
            desc, desc16 = _vpr.salad_aggregate(tokens, torch_ops.weight_list(w), w.dustbin, 3)
        return (desc, desc16) if want_bf16 else desc


class DinoV2Salad(nn.Module):
    """`feature_extractor`: images [B,3,224,224] -> L2-normalised descriptor [B, 8448]."""

    def __init__(self, arch: str = "vit_large"):
        super().__init__()
        self.backbone = DinoV2(arch)
        self.aggregator = SaladAggregator(self.backbone.embed_dim)

    token_on_cls_stream = True      # SALAD's token MLP rides on the backbone's cls-row stream (bit-identical; A/B switch)

    @torch.no_grad()
    def tokens(self, x: torch.Tensor) -> torch.Tensor:
        """Final-norm tokens [B, 1+n, C] bf16, cls first (the hub model's layout)."""
        t = self.backbone(x)
        return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)

    @torch.no_grad()
    def features(self, x: torch.Tensor, want_bf16: bool = False, events: Optional[list] = None):
        """images -> descriptor (and its bf16 copy): backbone tokens stay in the layout the backbone
        computed them in (SplitTokens on the HIP path), no re-layout copy before SALAD.
        events: a list -> a (start, end) pair of timing events around the aggregation on the current stream is appended
        (bench.py: the SALAD stage as the step runs it, token MLP on the backbone's cls-row stream)."""
        self.backbone.cls_tail_hook = self.aggregator.token_stage if self.token_on_cls_stream else None
        try:
            t = self.backbone(x, split=True)
        finally:
            self.backbone.cls_tail_hook = None
        if t.patch.dtype not in (torch.bfloat16, torch.float32):
            t = SplitTokens(t.patch.to(torch.bfloat16), t.cls.to(torch.bfloat16))
        if events is None:
            return self.aggregator(t, want_bf16=want_bf16)      # f32 tokens (an f32 model) take the f32-accurate aggregation
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self.aggregator(t, want_bf16=want_bf16)
        e1.record()
        events.append((e0, e1))
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.features(x)


def _mlp_head_args(seq: nn.Sequential):
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    if len(lin) != 2:
        raise RuntimeError("expected Linear -> ReLU -> [Dropout] -> Linear")
    f = lambda p: p.detach().to(torch.float32).contiguous()
    return f(lin[0].weight), f(lin[0].bias), f(lin[1].weight), f(lin[1].bias)


class DINOv2RegressionModel(nn.Module):
    def __init__(self, base_model: nn.Module):
        super().__init__()
        self.feature_extractor = base_model
        for p in self.feature_extractor.parameters():
            p.requires_grad = False
        self.regressor = nn.Sequential(nn.Linear(8448, 512), nn.ReLU(), nn.Linear(512, 2))

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        features = self.feature_extractor(x)
        return _vpr.pose_head(features.float().contiguous(), *_mlp_head_args(self.regressor), -1)


def _swin_hidden(backbone: nn.Module, pixel_values: torch.Tensor):
    """Pre-LayerNorm last hidden state [B,T,H] of an HF SwinModel, plus its final LayerNorm."""
    emb, dims = backbone.embeddings(pixel_values)
    enc = backbone.encoder(emb, dims)
    return enc[0].contiguous(), backbone.layernorm


class SwinRegressionModel(nn.Module):
    """backbone: an HF `SwinModel` (or anything with .embeddings/.encoder/.layernorm/.config)."""

    def __init__(self, backbone: nn.Module):
        super().__init__()
        self.backbone = backbone
        self.regressor = nn.Linear(self.backbone.config.hidden_size, 2)
        self.normalize_output = False

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        h, ln = _swin_hidden(self.backbone, pixel_values)
        _, out = _vpr.ln_meanpool_head(h, ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), ln.eps,
                                       self.regressor.weight.detach().float().contiguous(),
                                       self.regressor.bias.detach().float().contiguous(),
                                       0 if self.normalize_output else -1)
        return out


class SwinSinCosRegressionModel(SwinRegressionModel):
    """Unit-normalised [sin, cos] output (F.normalize(out, dim=1, p=2, eps=1e-6))."""

    def __init__(self, backbone: nn.Module):
        super().__init__(backbone)
        self.normalize_output = True


class SwinMLPRegressionModel(nn.Module):
    def __init__(self, backbone: nn.Module, dropout_prob: float = 0.3, hidden: int = 512):
        super().__init__()
        self.backbone = backbone
        self.regressor = nn.Sequential(nn.Linear(self.backbone.config.hidden_size, hidden), nn.ReLU(),
                                       nn.Dropout(dropout_prob), nn.Linear(hidden, 2))

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        h, ln = _swin_hidden(self.backbone, pixel_values)
        pooled, _ = _vpr.ln_meanpool_head(h, ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), ln.eps,
                                          None, None, -1)
        return _vpr.pose_head(pooled, *_mlp_head_args(self.regressor), -1)


class SwinAngleRegressorSinCos(SwinMLPRegressionModel):
    """angle_prediction/swin/swin_angle_finetuning_gemini.py:92-128 (there also called `SwinRegressionModel`; renamed
    because swin_validation.py's class owns that name here): Swin pooler -> Linear(H, H/2) -> ReLU -> Dropout ->
    Linear(H/2, 2) = raw (sin, cos), NOT unit-normalised (its loss is an MSE on the pair); decoded with
    postproc.sincos_to_degrees (:131-136).  Keys regressor.0.*, regressor.3.*."""

    def __init__(self, backbone: nn.Module, dropout_rate: float = 0.0):
        super().__init__(backbone, dropout_rate, hidden=backbone.config.hidden_size // 2)


class DinoV2AngleRegressorSinCos(nn.Module):
    """angle_prediction/dinov2salad/dino_v2_gemini.py:99-114: DINOv2 CLS token -> Dropout -> `head` = Linear(hidden, 2) =
    raw (sin, cos); decoded by prediction_to_angle_deg (:135-141) = postproc.sincos_to_degrees.  The reference builds
    the backbone with AutoModel.from_pretrained("facebook/dinov2-base") (a fetch); here a vpr_amd.backbone.DinoV2 is
    passed in, and its load hook takes the Hugging Face key layout the reference's checkpoints carry
    (`backbone.embeddings.*`, `backbone.encoder.layer.N.*`), so `load_state_dict(torch.load(best_model.pth))` works
    unchanged.  Keys: backbone.*, head.weight [2, hidden], head.bias [2]."""

    def __init__(self, backbone: DinoV2, dropout_rate: float = 0.1):
        super().__init__()
        self.backbone = backbone
        self.dropout = nn.Dropout(dropout_rate)                # eval: identity
        self.head = nn.Linear(backbone.embed_dim, 2)

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        t = self.backbone(pixel_values, split=True)            # final-norm tokens; the cls rows are a contiguous [B, C]
        pooled = t.cls.float().contiguous()
        return _vpr.pose_head(pooled, None, None, self.head.weight.detach().float().contiguous(),
                              self.head.bias.detach().float().contiguous(), -1)


class FusedGeoPoseHead(nn.Module):
    """(lat, lon, sin, cos) in ONE kernel call from two independent MLP heads on the same features:
    hidden layers are row-concatenated and the output layer is block-diagonal, so each half is its
    own head (the zero blocks add exact zeros; only the f32 summation order of the split-K
    partition can differ from a separate call, ~1e-7).  pos: DINOv2RegressionModel
    .regressor; ang: Linear-ReLU-Linear(…,2) giving [sin, cos], unit-normalised if `normalize`."""

    def __init__(self, pos: nn.Sequential, ang: nn.Sequential, normalize: bool = True):
        super().__init__()
        self.pos, self.ang, self.normalize = pos, ang, normalize
        self._packed = None

    def pack(self):
        W1p, b1p, W2p, b2p = _mlp_head_args(self.pos)
        W1a, b1a, W2a, b2a = _mlp_head_args(self.ang)
        hp, ha = W1p.shape[0], W1a.shape[0]
        W1 = torch.cat([W1p, W1a], 0).contiguous()
        b1 = torch.cat([b1p, b1a], 0).contiguous()
        W2 = torch.zeros(4, hp + ha, dtype=torch.float32, device=W1.device)
        W2[:2, :hp] = W2p
        W2[2:, hp:] = W2a
        self._packed = (W1, b1, W2, torch.cat([b2p, b2a], 0).contiguous())
        return self._packed

    @torch.no_grad()
    def forward(self, features: torch.Tensor) -> torch.Tensor:
        W1, b1, W2, b2 = self._packed or self.pack()
        return _vpr.pose_head(features, W1, b1, W2, b2, 2 if self.normalize else -1)


def load_reference_checkpoint(model: nn.Module, path: str) -> nn.Module:
    """Accepts both checkpoint forms the reference writes: {'model_state_dict': ...}
    (dinov2salad_finetuning.py:130-135, loaded at dinov2salad_validation.py:68-69) and bare
    state dicts (swin_attempt_2.py:255, loaded at val_and_test_swin_2.py:231).  The load is strict,
    as the reference's: the `feature_extractor.*` keys of the hub model (`backbone.model.blocks.N.attn.qkv`,
    `ls1.gamma`, `patch_embed.proj`, `mask_token`, 37x37 `pos_embed`) are mapped onto DinoV2's by its
    load hook (backbone.DinoV2._adopt_foreign_keys / checkpoint.py).  A load un-folds LayerScale: call
    `backbone.fold_layerscale()` and `aggregator.pack()` afterwards (evaluate.py does)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    state = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    model.load_state_dict(state)
    for m in model.modules():
        if hasattr(m, "_packed"):
            m._packed = None
        if hasattr(m, "_packed_f32"):
            m._packed_f32 = None
    return model
