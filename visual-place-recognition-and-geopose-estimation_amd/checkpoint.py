"""State-dict key maps for the `feature_extractor.*` half of the reference's checkpoints.

The reference saves `model.state_dict()` of `DINOv2RegressionModel(torch.hub.load("serizba/salad",
"dinov2_salad"))` (dinov2salad/dinov2salad_finetuning.py:130-135; loaded with a strict
`load_state_dict` at dinov2salad_validation.py:65-69).  Neither third-party module is in the
reference tree, so the layouts below are restated from their published sources:

* serizba/salad `VPRModel`: `backbone` = a wrapper whose `.model` is the facebookresearch/dinov2
  `DinoVisionTransformer`, `aggregator` = `SALAD` -> keys `backbone.model.<dinov2 key>` and
  `aggregator.{score,cluster_features}.{0,3}.*`, `aggregator.token_features.{0,2}.*`,
  `aggregator.dust_bin`.
* facebookresearch/dinov2 `DinoVisionTransformer` (dinov2/models/vision_transformer.py): `cls_token`
  [1,1,C], `pos_embed` [1, 1+37*37, C] (518-px pre-training grid), `mask_token` [1,C] (unused at
  inference), `patch_embed.proj.{weight,bias}`, `blocks.N.{norm1,norm2}.*`, `blocks.N.attn.{qkv,proj}.*`,
  `blocks.N.mlp.{fc1,fc2}.*`, `blocks.N.{ls1,ls2}.gamma`, `norm.*`; chunked checkpoints spell the
  blocks `blocks.<chunk>.<N>.`.
* Hugging Face `transformers.Dinov2Model` (importable here; the architecture pin of
  tests/test_backbone_hf.py): `embeddings.*`, `encoder.layer.N.*` with separate query / key / value.

`vpr_amd.backbone.DinoV2` keeps its own short names (`blocks.N.qkv`, `blocks.N.ls1`,
`patch_embed.weight`, `pos_embed` sized for the 224-px grid); `convert_state_dict` maps any of the
three layouts (with any prefix, e.g. `feature_extractor.backbone.model.`) onto them, drops
`mask_token`, and resamples `pos_embed` to the model's grid the way the source model does at run
time (`interpolate_pos_encoding`: bicubic, no antialias; dinov2's hub models pass
`scale_factor = (side + 0.1) / 37`, Hugging Face passes `size=`).
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict
from typing import Dict, Mapping, Optional

import torch
import torch.nn.functional as F

# facebookresearch/dinov2: `interpolate_offset = 0.1` in every hub entry point (vits14 ... vitg14).
HUB_INTERPOLATE_OFFSET = 0.1

_BLOCK_RENAMES = (
    (".attn.qkv.", ".qkv."), (".attn.proj.", ".proj."), (".mlp.fc1.", ".fc1."), (".mlp.fc2.", ".fc2."),
    (".ls1.gamma", ".ls1"), (".ls2.gamma", ".ls2"),
)
_HF_BLOCK_RENAMES = (
    (".attention.output.dense.", ".proj."), (".mlp.fc1.", ".fc1."), (".mlp.fc2.", ".fc2."),
    (".layer_scale1.lambda1", ".ls1"), (".layer_scale2.lambda1", ".ls2"),
)


def interpolate_pos_embed(pos_embed: torch.Tensor, side: int, offset: float = HUB_INTERPOLATE_OFFSET) -> torch.Tensor:
    """[1, 1+M*M, C] -> [1, 1+side*side, C]: the cls position is kept, the M x M patch grid is
    resampled bicubically (f32, align_corners=False, no antialias).  offset > 0 reproduces dinov2's
    `scale_factor=(side + offset) / M` call (the output grid is still side x side, but the sample
    coordinates use that factor); offset == 0 is the `size=(side, side)` call of Hugging Face."""
    n_old = pos_embed.shape[1] - 1
    M = int(round(math.sqrt(n_old)))
    if M * M != n_old:
        raise ValueError(f"pos_embed has {n_old} patch positions: not a square grid")
    if M == side:
        return pos_embed
    C = pos_embed.shape[-1]
    cls_pos, grid = pos_embed[:, :1], pos_embed[:, 1:]
    grid = grid.reshape(1, M, M, C).permute(0, 3, 1, 2).float()
    if offset:
        sf = float(side + offset) / M
        grid = F.interpolate(grid, scale_factor=(sf, sf), mode="bicubic", align_corners=False, antialias=False)
    else:
        grid = F.interpolate(grid, size=(side, side), mode="bicubic", align_corners=False, antialias=False)
    if tuple(grid.shape[-2:]) != (side, side):
        raise RuntimeError(f"pos_embed resample gave {tuple(grid.shape[-2:])}, wanted {(side, side)}")
    grid = grid.permute(0, 2, 3, 1).reshape(1, side * side, C).to(pos_embed.dtype)
    return torch.cat([cls_pos, grid], dim=1)


def _split_prefix(key: str):
    """(prefix up to and including the DinoV2 module, rest) for a key that belongs to the backbone,
    recognised by its first backbone-level token; None for other keys."""
    m = re.search(r"(^|\.)(cls_token|pos_embed|mask_token|register_tokens|patch_embed\.|blocks\.|norm\.|"
                  r"embeddings\.|encoder\.layer\.|layernorm\.)", key)
    if m is None:
        return None
    cut = m.start(2)
    return key[:cut], key[cut:]


def convert_state_dict(state: Mapping[str, torch.Tensor], num_patches: Optional[int] = None,
                       interpolate_offset: Optional[float] = None) -> "OrderedDict[str, torch.Tensor]":
    """Maps hub / serizba-salad / Hugging-Face DINOv2 keys (any prefix) to vpr_amd's; keys that are
    already native, and every non-backbone key (regressor.*, aggregator.*), pass through unchanged.
    num_patches: the target model's patch count (pos_embed is resampled to it when it differs).
    interpolate_offset: None = 0.1 for hub-style keys, 0 for Hugging-Face keys."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    qkv_parts: Dict[str, Dict[str, torch.Tensor]] = {}
    for key, val in state.items():
        # the SALAD wrapper's `.model` level (serizba/salad DINOv2 wrapper)
        key = key.replace("backbone.model.", "backbone.")
        sp = _split_prefix(key)
        if sp is None or ".aggregator." in "." + key or key.startswith("aggregator."):
            out[key] = val
            continue
        prefix, rest = sp
        hf = rest.startswith(("embeddings.", "encoder.layer.", "layernorm."))
        if rest.startswith(("mask_token", "embeddings.mask_token")):
            continue                                     # training-only parameter
        if rest.startswith("register_tokens"):
            raise ValueError("DINOv2 checkpoints with register tokens are not supported (the reference's hub entry has none)")
        if hf:
            rest = (rest.replace("embeddings.cls_token", "cls_token")
                        .replace("embeddings.position_embeddings", "pos_embed")
                        .replace("embeddings.patch_embeddings.projection.", "patch_embed.")
                        .replace("encoder.layer.", "blocks."))
            if rest.startswith("layernorm."):
                rest = "norm." + rest[len("layernorm."):]
            m = re.match(r"(blocks\.\d+)\.attention\.attention\.(query|key|value)\.(weight|bias)$", rest)
            if m:
                qkv_parts.setdefault(f"{prefix}{m.group(1)}.qkv.{m.group(3)}", {})[m.group(2)] = val
                continue
            for a, b in _HF_BLOCK_RENAMES:
                rest = rest.replace(a, b)
        else:
            rest = re.sub(r"^blocks\.\d+\.(\d+)\.", r"blocks.\1.", rest)       # chunked: blocks.<chunk>.<N>.
            rest = rest.replace("patch_embed.proj.", "patch_embed.")
            for a, b in _BLOCK_RENAMES:
                rest = rest.replace(a, b)
        if rest == "pos_embed" and num_patches is not None and val.shape[1] != 1 + num_patches:
            side = int(round(math.sqrt(num_patches)))
            if side * side != num_patches:
                raise ValueError("pos_embed resampling needs a square target grid")
            off = interpolate_offset if interpolate_offset is not None else (0.0 if hf else HUB_INTERPOLATE_OFFSET)
            val = interpolate_pos_embed(val, side, off)
        out[prefix + rest] = val
    for key, parts in qkv_parts.items():
        if set(parts) != {"query", "key", "value"}:
            raise ValueError(f"{key}: incomplete query/key/value set")
        out[key] = torch.cat([parts["query"], parts["key"], parts["value"]], dim=0)     # dinov2's fused qkv row order
    return out
