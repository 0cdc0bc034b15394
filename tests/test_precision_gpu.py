"""Distance of the MI355X path from the reference's own precision.

The reference runs the extractor, the aggregator and the head in fp32 (dinov2salad/dinov2salad_validation.py:65-66,
80-81: `.cuda()` with no cast).  The benchmark path keeps tokens, SALAD weights and SALAD hidden activations in bf16.
These tests put numbers — and gates — on that difference, at the two shapes that matter (BASELINE config 2: C = 1024,
B = 64; the hub model's: C = 768, B = 16):

  A. SALAD stage, same bf16-valued inputs, against the f64 oracle WITHOUT the bf16 rounding of the hidden activations
     (`quantize=False`): isolates the one rounding point inside the HIP stage.
  B. SALAD stage on genuinely f32 tokens / weights: the bf16 path (operands rounded on the way in) and the f32-accurate
     path (`vpr_salad_aggregate_f32`: three bf16 planes per operand, exact products, f32 accumulation, f32 hidden
     activations) against the f64 oracle on the unrounded inputs.
  C. End to end through `evaluate.calculate_validation_scores` on the same images + checkpoint: `dtype=torch.float32`
     (f32 block loop, f32-accurate SALAD, f32 head) against the same stages in torch f32 + the f64 oracle — the north
     star's "lat/lon within 1e-4" on the standardised head output — and the bf16 path against that f32 path.

Bounds asserted below are the numbers DESIGN.md §4 quotes; the observed values are printed (`-s`).
"""
import numpy as np
import pytest
import torch

from oracle import heads as oheads, salad as osalad

pytestmark = pytest.mark.gpu

# asserted bounds (max abs over the 8448-d unit descriptor, entries ~1e-2)
HIDDEN_ROUNDING_BOUND = 1e-4      # A: bf16 hidden activations vs none, same operands (north-star tolerance)
BF16_OPERANDS_BOUND = 2.5e-4      # B: bf16 tokens + weights + hidden vs f64 on the f32 operands
F32_PATH_BOUND = 5e-7             # B: vpr_salad_aggregate_f32 vs f64 on the f32 operands
# standardised (lat, lon), head outputs O(1)
E2E_F32_BOUND = 1e-4              # C: f32 path vs torch-f32 backbone + f64 oracle  (north star: within 1e-4)
E2E_BF16_VS_F32_BOUND = 5e-2      # C: bf16 benchmark path vs the f32 path


def _f32_weights(C, seed, hidden=512, m=64, l=128, t=256, std=0.02):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * std
    return dict(w1_sc=r(2 * hidden, C), b1_sc=r(2 * hidden), w2_s=r(m, hidden), b2_s=r(m), w2_c=r(l, hidden), b2_c=r(l),
                w1_t=r(hidden, C), b1_t=r(hidden), w2_t=r(t, hidden), b2_t=r(t))


def _bf16_weights(w):
    return {k: (v.to(torch.bfloat16) if k.startswith("w") else v) for k, v in w.items()}


@pytest.mark.parametrize("C,B", [(1024, 64), (768, 16)])
def test_salad_hidden_rounding_against_unquantised_oracle(dev, C, B):
    """A.  Same bf16-valued tokens and weights on both sides; the oracle keeps its hidden activations in f64."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(1000 + C + B)
    tokens = torch.randn(B, 257, C, generator=g).to(torch.bfloat16)
    w = _bf16_weights(_f32_weights(C, seed=C))
    ref_q = osalad.salad_aggregate(tokens, w, 1.0, 3, quantize=True)
    ref = osalad.salad_aggregate(tokens, w, 1.0, 3, quantize=False)
    out, _ = ops.salad_aggregate(tokens.to(dev), ops.SaladWeights(**{k: v.to(dev) for k, v in w.items()}, dustbin=1.0), 3)
    out = out.cpu().double()
    e_kernel, e_hidden, e_total = (out - ref_q).abs().max().item(), (ref_q - ref).abs().max().item(), (out - ref).abs().max().item()
    print(f"\n[precision A] C={C} B={B}: HIP vs quantised oracle {e_kernel:.2e}; quantised vs unquantised oracle {e_hidden:.2e}; "
          f"HIP vs unquantised f64 oracle {e_total:.2e}  (entries ~{ref.abs().mean().item():.1e})")
    assert e_total < HIDDEN_ROUNDING_BOUND


@pytest.mark.parametrize("C,B", [(1024, 64), (768, 16)])
def test_salad_f32_path_and_bf16_path_against_f64_on_f32_operands(dev, C, B):
    """B.  Tokens and weights that are NOT bf16-representable: what the reference's fp32 aggregator sees."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(2000 + C + B)
    tokens = torch.randn(B, 257, C, generator=g)
    w = _f32_weights(C, seed=C + 1)
    ref = osalad.salad_aggregate(tokens, w, 1.0, 3, quantize=False)                              # f64 on the f32 operands
    td = tokens.to(dev)
    w32 = ops.SaladWeightsF32(**{k: v.to(dev) for k, v in w.items()}, dustbin=1.0)
    out32, out32_16 = ops.salad_aggregate_f32(td, w32, 3, want_bf16=True)
    e32 = (out32.cpu().double() - ref).abs().max().item()
    # the split-pair form (what an f32 backbone hands over) sees the same numbers through other addresses
    out32s, _ = ops.salad_aggregate_f32((td[:, 1:].contiguous(), td[:, 0].contiguous()), w32, 3)
    assert torch.equal(out32s, out32) and torch.equal(out32_16.cpu(), out32.cpu().to(torch.bfloat16))
    wb = _bf16_weights(w)
    out16, _ = ops.salad_aggregate(td.to(torch.bfloat16), ops.SaladWeights(**{k: v.to(dev) for k, v in wb.items()}, dustbin=1.0), 3)
    e16 = (out16.cpu().double() - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(out16.cpu().double(), ref, dim=1).min().item()
    print(f"\n[precision B] C={C} B={B}: f32-accurate path vs f64 {e32:.2e}; bf16 path (bf16 tokens, weights, hidden) vs f64 {e16:.2e}, "
          f"min cosine {cos:.7f}")
    assert e32 < F32_PATH_BOUND
    assert e16 < BF16_OPERANDS_BOUND and cos > 0.9999


def test_salad_f32_path_larger_scores_and_edge_shapes(dev):
    """The f32 path on a wide score range (max-subtraction in both LSE passes), one image, ViT-S width."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(5)
    for C, B, std, dust in ((384, 1, 0.08, -2.0), (768, 3, 0.05, 0.5), (64, 2, 0.3, 1.0)):
        tokens = torch.randn(B, 257, C, generator=g) * 2
        w = _f32_weights(C, seed=C, std=std)
        ref = osalad.salad_aggregate(tokens, w, dust, 3, quantize=False)
        out, _ = ops.salad_aggregate_f32(tokens.to(dev), ops.SaladWeightsF32(**{k: v.to(dev) for k, v in w.items()}, dustbin=dust), 3)
        assert torch.isfinite(out).all()
        err = (out.cpu().double() - ref).abs().max().item()
        print(f"\n[precision B'] C={C} B={B} std={std}: f32-accurate path vs f64 {err:.2e}")
        assert err < 1e-5


def _write_dataset(tmp_path, n_img, seed):
    from PIL import Image
    import pandas as pd
    rng = np.random.default_rng(seed)
    img_dir = tmp_path / "images_val"
    img_dir.mkdir()
    names = [f"img_{i:04d}.png" for i in range(n_img)]
    for n in names:
        # smooth-ish content (blocks + noise): closer to photographs than white noise, still deterministic
        base = rng.integers(0, 256, (8, 8, 3)).repeat(28, 0).repeat(28, 1)
        img = np.clip(base + rng.integers(-20, 21, (224, 224, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(img_dir / n)
    df = pd.DataFrame({"filename": names, "timestamp": "12:00", "latitude": rng.normal(219658, 900, n_img).round(),
                       "longitude": rng.normal(143506, 1100, n_img).round(), "angle": 0, "Region_ID": 1})
    csv = tmp_path / "labels_val.csv"
    df.to_csv(csv, index=False)
    return img_dir, csv, names


@pytest.mark.parametrize("shape,n_img", [((768, 12, 12), 16), ((1024, 24, 16), 8)])      # hub model (ViT-B/14), benchmark (ViT-L/14)
def test_end_to_end_f32_path_and_bf16_path_on_standardised_latlon(dev, tmp_path, shape, n_img):
    """C.  Same images, same checkpoint, through the drop-in entry point in both precisions.
    Weights: HF-style random init with LayerScale 0.3 / 0.15 (every block contributes, as in a trained model), SALAD
    weights N(0, 0.02), and a head scaled so that the standardised (lat, lon) outputs are O(1) like a trained
    regressor's (Linear default init x100 on layer 1 and x10 on layer 2 — default-init outputs would be ~1e-2 and hide
    any descriptor error behind the biases)."""
    from PIL import Image
    from test_backbone_hf import _hf_model
    import vpr_amd.backbone as bb
    from vpr_amd import evaluate, modules
    from vpr_amd.preprocess import ResizeNormalize
    hidden, layers, heads = shape
    arch = f"prec_{hidden}_{layers}"
    bb.CONFIGS[arch] = shape
    img_dir, csv, names = _write_dataset(tmp_path, n_img, seed=hidden)
    hf = _hf_model(hidden, layers, heads, seed=11, layerscale=0.3 if layers == 12 else 0.15)
    torch.manual_seed(12)
    src = modules.DinoV2Salad(arch)
    src.backbone.load_state_dict(hf.state_dict())
    for p in src.aggregator.parameters():
        if p.dim() > 0:
            torch.nn.init.normal_(p, std=0.02)
    model = modules.DINOv2RegressionModel(src)
    with torch.no_grad():
        model.regressor[0].weight.mul_(100.0)
        model.regressor[2].weight.mul_(10.0)
    ck = tmp_path / "checkpoint.pth"
    torch.save({"epoch": 1, "model_state_dict": model.state_dict()}, ck)

    run = lambda dt: evaluate.calculate_validation_scores(str(ck), str(csv), str(img_dir), base_model=modules.DinoV2Salad(arch),
                                                          batch_size=8, verbose=False, dtype=dt)
    res32 = run(torch.float32)
    res16 = run(torch.bfloat16)
    p32, p16 = res32["preds_standardised"].astype(np.float64), res16["preds_standardised"].astype(np.float64)

    # the same stages in reference arithmetic: torch f32 backbone on the same preprocessed images, f64 oracle SALAD on its
    # f32 tokens with the f32 weights, f64 head
    prep = ResizeNormalize(224, "bilinear", (0.5,) * 3, (0.5,) * 3, torch.float32)
    u8 = torch.stack([torch.from_numpy(np.array(Image.open(img_dir / n).convert("RGB"))) for n in names]).to(dev)
    with torch.no_grad():
        tokens = torch.cat([src.backbone.to(dev).float()(prep(u8[i:i + 8])) for i in range(0, n_img, 8)]).cpu()
    agg = src.aggregator
    m2 = lambda w: w.detach().reshape(w.shape[0], -1).float().cpu()
    f = lambda b: b.detach().float().cpu()
    w = dict(w1_sc=torch.cat([m2(agg.score[0].weight), m2(agg.cluster_features[0].weight)], 0),
             b1_sc=torch.cat([f(agg.score[0].bias), f(agg.cluster_features[0].bias)], 0),
             w2_s=m2(agg.score[3].weight), b2_s=f(agg.score[3].bias), w2_c=m2(agg.cluster_features[3].weight),
             b2_c=f(agg.cluster_features[3].bias), w1_t=m2(agg.token_features[0].weight), b1_t=f(agg.token_features[0].bias),
             w2_t=m2(agg.token_features[2].weight), b2_t=f(agg.token_features[2].bias))
    desc = osalad.salad_aggregate(tokens, w, float(agg.dust_bin), 3, quantize=False)
    r = model.regressor
    ref = oheads.mlp_head(desc, f(r[0].weight), f(r[0].bias), f(r[2].weight), f(r[2].bias)).numpy()

    e32 = np.abs(p32 - ref).max()
    e16 = np.abs(p16 - p32).max()
    scale = np.abs(ref).max()
    units = e16 * np.array([918.58972058316, 1190.858018520488]).max()
    print(f"\n[precision C] {shape}: standardised (lat, lon) |max| {scale:.2f}; f32 path vs torch-f32 + f64 oracle {e32:.2e}; "
          f"bf16 path vs f32 path {e16:.2e} (= {units:.1f} label units at the campus scaler)")
    assert scale > 0.3                                                   # the head is sensitive: outputs are O(1)
    assert e32 < E2E_F32_BOUND
    assert e16 < E2E_BF16_VS_F32_BOUND
