"""Architecture + checkpoint pins of the DINOv2 backbone (SURVEY.md §8 a-1 / b).

The reference obtains its backbone from torch.hub (dinov2salad/dinov2salad_validation.py:65 — a network
fetch, absent offline).  The independent implementation that IS importable, here and on the GPU box, is
`transformers.Dinov2Model`: same architecture (patch-14 embed, cls token, learned position embedding
resampled bicubically to the input grid, pre-norm blocks with LayerScale and erf-GELU, final LayerNorm).
Random-init HF weights -> key map (vpr_amd/checkpoint.py) -> same images -> same tokens:
  * CPU, f32: the block loop against HF, <= 1e-5 (observed 0 — same torch kernels in the same order);
  * GPU, bf16 HIP path: error against HF-f32 no larger than HF's own bf16 run (+15 %), for both GELU forms,
    and the descriptor-level size of the tanh-vs-erf GELU difference is measured and bounded.
"""
import copy
import math

import numpy as np

import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf_model(hidden, layers, heads, seed=0, layerscale=0.3):
    from transformers import Dinov2Config, Dinov2Model
    torch.manual_seed(seed)
    cfg = Dinov2Config(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, image_size=518,
                       patch_size=14, layerscale_value=layerscale)
    hf = Dinov2Model(cfg).eval()
    with torch.no_grad():                       # biases / norms / LayerScale away from their constant init values
        for p in hf.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        hf.embeddings.cls_token.normal_(std=0.5)
    return hf


def _ours(hidden, layers, heads):
    import vpr_amd.backbone as bb
    name = f"test_{hidden}_{layers}_{heads}"
    bb.CONFIGS[name] = (hidden, layers, heads)
    return bb.DinoV2(name).eval()


def test_block_loop_equals_hf_dinov2model_f32_cpu():
    hf = _hf_model(128, 3, 2)
    m = _ours(128, 3, 2)
    assert hf.embeddings.position_embeddings.shape[1] == 1 + 37 * 37          # the 518-px grid DINOv2 ships
    res = m.load_state_dict(hf.state_dict())                                  # strict; q/k/v fused, 37x37 -> 16x16
    assert not res.missing_keys and not res.unexpected_keys
    assert m.pos_embed.shape[1] == 257
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state                            # HF resamples pos_embed per call (size=)
        out = m(x)
    assert out.shape == ref.shape == (2, 257, 128)
    assert (out - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_fold_and_load_in_any_order_give_the_same_function():
    """fold -> load used to drop LayerScale, save(folded) -> load applied it twice."""
    hf = _hf_model(128, 2, 2, seed=1)
    x = torch.randn(1, 3, 224, 224)
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state
    tol = 2e-5 * ref.abs().max().item()

    a = _ours(128, 2, 2); a.load_state_dict(hf.state_dict())                  # unfolded
    b = _ours(128, 2, 2); b.load_state_dict(hf.state_dict()); b.fold_layerscale()   # load -> fold
    c = _ours(128, 2, 2); c.fold_layerscale(); c.load_state_dict(hf.state_dict())   # fold -> load (INTEGRATION.md's old order)
    assert not c.blocks[0].folded                                             # the load ended the fold licence
    d = _ours(128, 2, 2); d.load_state_dict(b.state_dict())                   # state dict of a folded model, fresh module
    e = copy.deepcopy(c); e.fold_layerscale()                                 # fold -> load -> fold
    assert all(blk.folded for blk in b.blocks) and all(blk.folded for blk in e.blocks)
    assert torch.all(b.blocks[0].ls1 == 1) and torch.all(d.blocks[1].ls2 == 1)
    with torch.no_grad():
        for name, model in (("unfolded", a), ("load-fold", b), ("fold-load", c), ("saved-folded", d), ("fold-load-fold", e)):
            err = (model(x) - ref).abs().max().item()
            assert err <= tol, (name, err, tol)


def _hub_named(state_native, prefix, side_old=37, chunked=False):
    """A state dict in the hub model's key layout (facebookresearch/dinov2 names below serizba/salad's
    `backbone.model.`), synthesised from a native DinoV2 state dict: same tensors, a `mask_token`, and a
    side_old x side_old position grid (random: its resampled form is checked against F.interpolate)."""
    out = {}
    g = torch.Generator().manual_seed(5)
    for k, v in state_native.items():
        hk = (k.replace(".qkv.", ".attn.qkv.").replace(".proj.", ".attn.proj.").replace(".fc1.", ".mlp.fc1.")
               .replace(".fc2.", ".mlp.fc2."))
        if hk.endswith((".ls1", ".ls2")):
            hk += ".gamma"
        if hk.startswith("patch_embed."):
            hk = hk.replace("patch_embed.", "patch_embed.proj.")
        if chunked and hk.startswith("blocks."):
            n = int(hk.split(".")[1])
            hk = f"blocks.{n // 2}." + hk[len("blocks."):]
        if k == "pos_embed":
            v = 0.02 * torch.randn(1, 1 + side_old * side_old, v.shape[-1], generator=g)
        out[prefix + hk] = v
    out[prefix + "mask_token"] = torch.zeros(1, state_native["cls_token"].shape[-1])
    return out


@pytest.mark.parametrize("chunked", [False, True])
def test_reference_checkpoint_with_hub_key_names_loads_strict(tmp_path, chunked):
    """dinov2salad_validation.py:65-69: hub model inside DINOv2RegressionModel, strict load of
    checkpoint['model_state_dict'] whose feature_extractor.* keys carry the hub names."""
    import torch.nn.functional as F
    import vpr_amd.backbone as bb
    from vpr_amd.modules import DINOv2RegressionModel, DinoV2Salad, load_reference_checkpoint
    bb.CONFIGS["test_tiny"] = (128, 4, 2)
    torch.manual_seed(2)
    src = DinoV2Salad("test_tiny")
    for blk in src.backbone.blocks:
        torch.nn.init.normal_(blk.ls1, std=0.3)
        torch.nn.init.normal_(blk.ls2, std=0.3)
    full = DINOv2RegressionModel(src)
    state = {}
    native_bb = {k: v.clone() for k, v in src.backbone.state_dict().items()}
    state.update(_hub_named(native_bb, "feature_extractor.backbone.model.", chunked=chunked))
    state.update({"feature_extractor.aggregator." + k: v.clone() for k, v in src.aggregator.state_dict().items()})
    state.update({"regressor." + k: v.clone() for k, v in full.regressor.state_dict().items()})
    path = str(tmp_path / "ckpt.pth")
    torch.save({"epoch": 49, "model_state_dict": state, "loss": 0.0}, path)

    dst = DINOv2RegressionModel(DinoV2Salad("test_tiny"))
    load_reference_checkpoint(dst, path)                                       # strict: raises on any unmapped key
    got = dst.feature_extractor.backbone.state_dict()
    for k, v in native_bb.items():
        if k != "pos_embed":
            assert torch.equal(got[k], v), k
    # position grid: cls row kept, 37x37 -> 16x16 exactly as dinov2's interpolate_pos_encoding computes it for a
    # 224-px input (bicubic, scale_factor = (16 + 0.1) / 37 on both axes, no antialias)
    pe = state["feature_extractor.backbone.model.pos_embed"]
    grid = pe[:, 1:].reshape(1, 37, 37, -1).permute(0, 3, 1, 2)
    want = F.interpolate(grid, scale_factor=(16.1 / 37, 16.1 / 37), mode="bicubic")
    assert want.shape[-2:] == (16, 16)
    assert torch.equal(got["pos_embed"][:, 0], pe[:, 0])
    assert torch.allclose(got["pos_embed"][:, 1:], want.permute(0, 2, 3, 1).reshape(1, 256, -1), atol=0, rtol=0)
    # the size= variant (Hugging Face) samples at different coordinates: the offset must matter, or this test pins nothing
    alt = F.interpolate(grid, size=(16, 16), mode="bicubic").permute(0, 2, 3, 1).reshape(1, 256, -1)
    assert (alt - got["pos_embed"][:, 1:]).abs().max().item() > 1e-5
    assert torch.equal(dst.regressor[0].weight, full.regressor[0].weight)
    assert torch.equal(dst.feature_extractor.aggregator.dust_bin, src.aggregator.dust_bin)


def test_unknown_or_register_token_keys_are_refused():
    m = _ours(128, 2, 2)
    sd = {k: v for k, v in m.state_dict().items()}
    sd["register_tokens"] = torch.zeros(1, 4, 128)
    with pytest.raises(ValueError):
        m.load_state_dict(sd)
    sd.pop("register_tokens")
    sd["blocks.0.attn.rope.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):                                          # strict load: unexpected key
        m.load_state_dict(sd)


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("shape,gelu", [((384, 12, 6), "tanh"), ((384, 12, 6), "erf"),
                                        ((1024, 24, 16), "tanh")])     # ViT-L/14: the benchmark's architecture
def test_hip_backbone_matches_hf_dinov2model_on_gpu(dev, shape, gelu):
    """bf16 HIP path (split rows, side-stream cls chain, patchify embedding, fused kernels) against HF's f32
    forward of the same weights; yardstick = HF's own bf16 forward against its f32 forward."""
    hf = _hf_model(*shape, seed=3, layerscale=0.3 if shape[1] == 12 else 0.15).to(dev)
    m = _ours(*shape)
    m.load_state_dict(hf.state_dict())
    m = m.to(dev).to(torch.bfloat16).eval()
    m.gelu = gelu
    m.fold_layerscale()
    x = torch.randn(4, 3, 224, 224, device=dev)
    xb = x.to(torch.bfloat16)
    assert m._hip_split_ok(xb)
    with torch.no_grad():
        ref = hf(pixel_values=xb.float()).last_hidden_state                    # f32 math on the bf16-rounded images
        hf16 = copy.deepcopy(hf).to(torch.bfloat16)
        yard = hf16(pixel_values=xb).last_hidden_state.float()
        out = m(xb).float()
    rms = lambda d: d.pow(2).mean().sqrt().item()
    e_out, e_yard, scale = rms(out - ref), rms(yard - ref), rms(ref)
    print(f"\n[hf-pin] {shape} gelu={gelu}: HIP path rms err {e_out:.3e}, HF bf16 rms err {e_yard:.3e}, token rms {scale:.3f}")
    assert e_out < 1.15 * e_yard and e_out < 0.02 * scale, (e_out, e_yard, scale)
    assert (out - ref).abs().max().item() < 2.0 * (yard - ref).abs().max().item()


@pytest.mark.gpu
def test_tanh_gelu_deviation_at_descriptor_level(dev):
    """Puts a number on the HIP path's default GELU form (hipBLASLt epilogue = tanh) against the exact erf
    form DINOv2 uses, where it matters: on the 8448-d SALAD descriptor.  Yardstick: the distance of either
    bf16 descriptor from the descriptor of an f32 backbone run (bf16 rounding noise)."""
    from vpr_amd.backbone import SplitTokens
    from vpr_amd.modules import DinoV2Salad
    import vpr_amd.backbone as bb
    bb.CONFIGS["test_384"] = (384, 12, 6)
    hf = _hf_model(384, 12, 6, seed=4)
    ext = DinoV2Salad("test_384")
    ext.backbone.load_state_dict(hf.state_dict())
    for p in ext.aggregator.parameters():
        if p.dim() > 0:
            torch.nn.init.normal_(p, std=0.02)
    ext32 = copy.deepcopy(ext).to(dev).eval()                                   # f32 block loop (erf), PyTorch ops
    ext = ext.to(dev).to(torch.bfloat16).eval()
    ext.backbone.fold_layerscale()
    ext.aggregator.pack()
    x = torch.randn(8, 3, 224, 224, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        d = {}
        for g in ("tanh", "erf"):
            ext.backbone.gelu = g
            d[g] = ext.features(x)
        t32 = ext32.backbone(x.float())
        d32 = ext.aggregator(t32.to(torch.bfloat16).contiguous())
    dev_gelu = (d["tanh"] - d["erf"]).abs().max().item()
    noise = max((d["erf"] - d32).abs().max().item(), (d["tanh"] - d32).abs().max().item())
    cos = torch.nn.functional.cosine_similarity(d["tanh"], d["erf"], dim=1).min().item()
    print(f"\n[gelu] descriptor max|tanh - erf| = {dev_gelu:.3e}; bf16-vs-f32 backbone max diff = {noise:.3e}; min cosine {cos:.6f}")
    assert dev_gelu <= 1.5 * noise                     # the GELU form is inside the bf16 noise of the backbone
    assert cos > 0.999


def test_angle_head_variants_keep_the_reference_state_dict_keys():
    """§8 a-5: the gemini sin/cos heads.  Swin MLP variant: keys regressor.0/.3 with hidden = H/2
    (swin_angle_finetuning_gemini.py:101-106); DINOv2 CLS variant: backbone.* + head.* (dino_v2_gemini.py:99-106), and a
    checkpoint in the Hugging Face key layout (what AutoModel.from_pretrained produces) loads strictly."""
    from transformers import SwinConfig, SwinModel
    from vpr_amd import modules
    sw = modules.SwinAngleRegressorSinCos(SwinModel(SwinConfig()))
    keys = {k for k in sw.state_dict() if k.startswith("regressor.")}
    assert keys == {"regressor.0.weight", "regressor.0.bias", "regressor.3.weight", "regressor.3.bias"}
    assert tuple(sw.regressor[0].weight.shape) == (384, 768) and tuple(sw.regressor[3].weight.shape) == (2, 384)
    hf = _hf_model(128, 2, 2, seed=7)
    m = modules.DinoV2AngleRegressorSinCos(_ours(128, 2, 2))
    ref_sd = {"backbone." + k: v for k, v in hf.state_dict().items()}
    ref_sd.update({"head.weight": torch.randn(2, 128), "head.bias": torch.randn(2)})
    res = m.load_state_dict(ref_sd)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m.head.weight, ref_sd["head.weight"])
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        want = hf(pixel_values=x).last_hidden_state[:, 0]
        got = m.backbone(x)[:, 0]
    assert (want - got).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


@pytest.mark.gpu
def test_dinov2_cls_sincos_model_on_gpu(dev):
    """dino_v2_gemini.py:108-114 on the HIP path: CLS of the final-norm tokens -> head (vpr_pose_head, hidden = 0) ->
    raw (sin, cos) -> degrees; against the same arithmetic in f32 torch on HF's Dinov2Model."""
    from vpr_amd import modules, postproc
    hf = _hf_model(384, 12, 6, seed=8).to(dev)
    head = torch.nn.Linear(384, 2).to(dev)
    m = modules.DinoV2AngleRegressorSinCos(_ours(384, 12, 6))
    sd = {"backbone." + k: v for k, v in hf.state_dict().items()}
    sd.update({"head." + k: v for k, v in head.state_dict().items()})
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.backbone.to(torch.bfloat16)
    m.backbone.gelu = "erf"
    m.backbone.fold_layerscale()
    x = torch.randn(6, 3, 224, 224, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        ref = head(hf(pixel_values=x.float()).last_hidden_state[:, 0])
        hf16 = copy.deepcopy(hf).to(torch.bfloat16)
        yard = head(hf16(pixel_values=x).last_hidden_state[:, 0].float())
        out = m(x)
    e_out, e_yard = (out - ref).abs().max().item(), (yard - ref).abs().max().item()
    print(f"\n[cls-sincos] HIP path max err {e_out:.3e}, HF bf16 max err {e_yard:.3e}")
    assert out.shape == (6, 2) and e_out < 2.0 * e_yard + 1e-3
    deg = postproc.sincos_to_degrees(out.cpu().numpy())
    want = (torch.rad2deg(torch.atan2(out[:, 0], out[:, 1])) % 360.0).cpu().numpy()      # prediction_to_angle_deg :135-141
    assert np.allclose(deg, want, atol=1e-3)


@pytest.mark.gpu
def test_forward_after_a_load_refolds_and_takes_the_hip_path(dev):
    """A load un-folds LayerScale; the next bf16 GPU forward folds again by itself (auto_fold), so a freshly loaded
    checkpoint runs the HIP backbone path — and computes the loaded function (against HF f32)."""
    hf = _hf_model(384, 12, 6, seed=11).to(dev)
    m = _ours(384, 12, 6).to(dev).to(torch.bfloat16).eval()
    m.fold_layerscale()                                        # folded with its random init ...
    m.load_state_dict(hf.state_dict())                         # ... then a checkpoint arrives
    assert not any(b.folded for b in m.blocks)
    x = torch.randn(2, 3, 224, 224, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        out = m(x).float()
        ref = hf(pixel_values=x.float()).last_hidden_state
    assert all(b.folded for b in m.blocks) and m._hip_split_ok(x)
    rms = lambda d: d.pow(2).mean().sqrt().item()
    assert rms(out - ref) < 0.02 * rms(ref)
