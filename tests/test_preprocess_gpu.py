"""GPU: vpr_preprocess_resize_normalize against PIL (bytes, exact) and ToTensor+Normalize (f32)."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,filt,mean,std", [
    (480, 640, "bilinear", (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),           # dinov2salad_validation.py:18-22
    (1080, 1920, "bilinear", (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),   # dinov2salad_finetuning.py:45-50
    (600, 800, "bicubic", (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),      # HF Swin processor
    (224, 224, "bilinear", (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),
    (150, 199, "bicubic", (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),            # upscaling
])
def test_resize_normalize_matches_pil(dev, H, W, filt, mean, std):
    from PIL import Image
    from vpr_amd.preprocess import ResizeNormalize
    rng = np.random.default_rng(H + W)
    imgs = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    pf = Image.BILINEAR if filt == "bilinear" else Image.BICUBIC
    ref_u8 = np.stack([np.asarray(Image.fromarray(im).resize((224, 224), pf)) for im in imgs])
    ref_f = np.stack([opre.to_tensor_normalize(u, mean, std) for u in ref_u8])
    pp = ResizeNormalize(224, filt, mean, std, torch.float32)
    out, out_u8 = pp(torch.from_numpy(imgs).to(dev), return_bytes=True)
    assert np.array_equal(out_u8.cpu().numpy(), ref_u8)
    assert np.abs(out.cpu().numpy() - ref_f).max() < 1e-6
    out16 = ResizeNormalize(224, filt, mean, std, torch.bfloat16)(torch.from_numpy(imgs).to(dev))
    assert torch.equal(out16.cpu(), torch.from_numpy(ref_f).to(torch.bfloat16))
