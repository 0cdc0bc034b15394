"""GPU image preprocessing: the reference resizes and normalises every image on
the host, on the main thread (DataLoader num_workers=0: dinov2salad/dinov2salad_validation.py:63).
Here a batch of decoded RGB uint8 images goes to the GPU once and comes back as the normalised
model input, bit-identical to PIL's resize followed by ToTensor + Normalize:

  torchvision Resize((224,224)) on a PIL image  ->  Image.resize(BILINEAR)   (validation.py:18-22)
  HF AutoImageProcessor for Swin                ->  Image.resize(BICUBIC), /255, ImageNet mean/std

The coefficient tables follow Pillow's precompute_coeffs / normalize_coeffs_8bpc (Resample.c):
support = filter_support * max(scale, 1), triangle or Keys(a=-0.5) cubic weights normalised to 1,
quantised to 22 fractional bits; the kernels do the integer accumulation (csrc/preprocess.hip).
"""
from __future__ import annotations

import ctypes
import math
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .ops import _need, _ptr, _stream, workspace

PRECISION_BITS = 22
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
HALF_MEAN, HALF_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)


def _bilinear(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


@lru_cache(maxsize=64)
def resample_coeffs(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for a full-image box.
    Returns (kk int32 [out, ksize], bounds int32 [out, 2] = (xmin, count), ksize)."""
    fn, fsupport = _FILTERS[filt]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


class ResizeNormalize:
    """uint8 [B,H,W,3] on the GPU -> [B,3,out,out] (bf16 | f32), PIL-exact."""

    def __init__(self, out_size: int = 224, filt: str = "bilinear", mean: Sequence[float] = HALF_MEAN,
                 std: Sequence[float] = HALF_STD, out_dtype: torch.dtype = torch.bfloat16):
        if filt not in _FILTERS:
            raise ValueError("filt must be 'bilinear' or 'bicubic'")
        if out_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("out_dtype must be bfloat16 or float32")
        self.out_size, self.filt, self.out_dtype = out_size, filt, out_dtype
        self._mean = (ctypes.c_float * 3)(*[float(m) for m in mean])
        self._std = (ctypes.c_float * 3)(*[float(s) for s in std])
        self._tables = {}

    def _device_tables(self, H: int, W: int, device: torch.device):
        key = (H, W, str(device))
        if key not in self._tables:
            kx, xb, ksx = resample_coeffs(W, self.out_size, self.filt)
            ky, yb, ksy = resample_coeffs(H, self.out_size, self.filt)
            t = lambda a: torch.from_numpy(a.copy()).to(device)
            self._tables[key] = (t(kx), t(xb), ksx, t(ky), t(yb), ksy)
        return self._tables[key]

    @torch.no_grad()
    def __call__(self, images_u8: torch.Tensor, return_bytes: bool = False):
        _need(images_u8, torch.uint8, "images", 4)
        B, H, W, C = images_u8.shape
        if C != 3:
            raise RuntimeError("images must be [B,H,W,3] RGB")
        dev = images_u8.device
        kx, xb, ksx, ky, yb, ksy = self._device_tables(H, W, dev)
        O = self.out_size
        out = torch.empty((B, 3, O, O), dtype=self.out_dtype, device=dev)
        out_u8 = torch.empty((B, O, O, 3), dtype=torch.uint8, device=dev) if return_bytes else None
        L = _lib.lib()
        ws = workspace("preprocess", L.vpr_preprocess_workspace_bytes(B, H, O), dev)
        st = L.vpr_preprocess_resize_normalize(_ptr(images_u8), B, H, W, O, O, _ptr(kx), _ptr(xb), ksx, _ptr(ky),
                                               _ptr(yb), ksy, self._mean, self._std, _ptr(out),
                                               int(self.out_dtype == torch.bfloat16), _ptr(out_u8), _ptr(ws),
                                               ws.numel(), _stream())
        _lib.check(st, "vpr_preprocess_resize_normalize")
        return (out, out_u8) if return_bytes else out
