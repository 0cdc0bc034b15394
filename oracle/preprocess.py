"""CPU restatement of the reference's image preprocessing — TEST ORACLE.

  Resize((224,224)) -> ToTensor -> Normalize(mean, std)   dinov2salad/dinov2salad_validation.py:18-22
torchvision.Resize on a PIL image calls PIL's Image.resize(BILINEAR) (antialiased, 8-bit integer
path); ToTensor = u8/255 in f32, CHW; Normalize = (x - mean) / std.  Pinned against PIL itself in
tests/test_oracle_selfchecks.py (PIL is the library the reference runs; torchvision is absent here).
The coefficient tables come from vpr_amd.preprocess.resample_coeffs (shared host-side table code,
itself verified against PIL); the resampling arithmetic below is an independent numpy version.
"""
import numpy as np


def resize_u8(img: np.ndarray, out: int, kx, xb, ky, yb) -> np.ndarray:
    """img [H,W,3] u8 -> [out,out,3] u8 with Pillow's two-pass fixed-point arithmetic."""
    H, W, _ = img.shape
    tmp = np.zeros((H, out, 3), np.uint8)
    for xx in range(out):
        x0, c = xb[xx]
        acc = (img[:, x0:x0 + c, :].astype(np.int64) * kx[xx, :c][None, :, None]).sum(1) + (1 << 21)
        tmp[:, xx, :] = np.clip(acc >> 22, 0, 255)
    o = np.zeros((out, out, 3), np.uint8)
    for yy in range(out):
        y0, c = yb[yy]
        acc = (tmp[y0:y0 + c].astype(np.int64) * ky[yy, :c][:, None, None]).sum(0) + (1 << 21)
        o[yy] = np.clip(acc >> 22, 0, 255)
    return o


def to_tensor_normalize(u8: np.ndarray, mean, std) -> np.ndarray:
    """[H,W,3] u8 -> [3,H,W] f32: (u8/255 - mean)/std in f32, as ToTensor + Normalize."""
    x = u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
