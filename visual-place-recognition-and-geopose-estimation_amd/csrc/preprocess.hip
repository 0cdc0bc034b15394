// preprocess.hip — GPU image preprocessing: antialiased resize (PIL's two-pass 8-bit integer
// resampler, bit for bit) + ToTensor + Normalize, fused into the second pass.
//
// Replaces the host-side, main-thread transforms of the reference's datasets:
//   Resize((224,224)) -> ToTensor -> Normalize(0.5, 0.5)   dinov2salad/dinov2salad_validation.py:18-22
//   ... with ImageNet mean/std                              dinov2salad/dinov2salad_finetuning.py:45-50
//   HF AutoImageProcessor (bicubic resize, rescale, normalize)  swin_transformer/swin_validation.py:30
// torchvision's Resize on a PIL image and HF's processor both end in PIL's Image.resize, whose
// 8-bit path is integer arithmetic: coefficients are quantised to 22 fractional bits on the host
// (vpr_amd/preprocess.py reproduces Pillow's precompute_coeffs), each pass accumulates
// sum(pixel * k) + 2^21 in int32, shifts by 22 and clamps to [0,255] — so the GPU result equals
// PIL's bytes exactly, and the float stage is (u8/255 - mean) / std in f32 like ToTensor+Normalize.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int PP_PRECISION_BITS = 22;   // Pillow: 32 - 8 - 2

__device__ __forceinline__ int clip8(int v) {
  v >>= PP_PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// pass 1 (horizontal): in [B,H,W,3] u8 -> tmp [B,H,OW,3] u8.  One thread per (row, out column).
__global__ __launch_bounds__(256) void resize_h_kernel(
    const uint8_t* __restrict__ in, int H, int W, int OW, const int32_t* __restrict__ kx,
    const int32_t* __restrict__ xb, int ksize, uint8_t* __restrict__ tmp) {
  const long long b = blockIdx.z;
  const int y = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= OW) return;
  const int xmin = xb[2 * xx], cnt = xb[2 * xx + 1];
  const uint8_t* row = in + ((b * H + y) * W + xmin) * 3;
  const int32_t* k = kx + (long long)xx * ksize;
  int s0 = 1 << (PP_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < cnt; ++x) {
    const int w = k[x];
    s0 += row[3 * x] * w;
    s1 += row[3 * x + 1] * w;
    s2 += row[3 * x + 2] * w;
  }
  uint8_t* o = tmp + ((b * H + y) * OW + xx) * 3;
  o[0] = (uint8_t)clip8(s0);
  o[1] = (uint8_t)clip8(s1);
  o[2] = (uint8_t)clip8(s2);
}

// pass 2 (vertical) + ToTensor + Normalize: tmp [B,H,OW,3] u8 -> out [B,3,OH,OW] (f32 | bf16),
// optionally also the resized bytes [B,OH,OW,3] (for parity tests against PIL).
template <bool BF16>
__global__ __launch_bounds__(256) void resize_v_normalize_kernel(
    const uint8_t* __restrict__ tmp, int H, int OH, int OW, const int32_t* __restrict__ ky,
    const int32_t* __restrict__ yb, int ksize, float m0, float m1, float m2, float sd0, float sd1,
    float sd2, void* __restrict__ out, uint8_t* __restrict__ out_u8) {
  const long long b = blockIdx.z;
  const int yy = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= OW) return;
  const int ymin = yb[2 * yy], cnt = yb[2 * yy + 1];
  const int32_t* k = ky + (long long)yy * ksize;
  const uint8_t* col = tmp + ((b * H + ymin) * OW + xx) * 3;
  int s0 = 1 << (PP_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < cnt; ++y) {
    const int w = k[y];
    const uint8_t* p = col + (long long)y * OW * 3;
    s0 += p[0] * w;
    s1 += p[1] * w;
    s2 += p[2] * w;
  }
  const int c0 = clip8(s0), c1 = clip8(s1), c2 = clip8(s2);
  if (out_u8) {
    uint8_t* o = out_u8 + ((b * OH + yy) * OW + xx) * 3;
    o[0] = (uint8_t)c0; o[1] = (uint8_t)c1; o[2] = (uint8_t)c2;
  }
  // ToTensor: u8 / 255 in f32; Normalize: (x - mean) / std  (division, as torch does)
  const float f0 = ((float)c0 / 255.0f - m0) / sd0;
  const float f1 = ((float)c1 / 255.0f - m1) / sd1;
  const float f2 = ((float)c2 / 255.0f - m2) / sd2;
  const long long plane = (long long)OH * OW;
  const long long o = b * 3 * plane + (long long)yy * OW + xx;
  if (BF16) {
    uint16_t* ob = reinterpret_cast<uint16_t*>(out);
    ob[o] = f32_to_bf16_bits(f0); ob[o + plane] = f32_to_bf16_bits(f1); ob[o + 2 * plane] = f32_to_bf16_bits(f2);
  } else {
    float* of = reinterpret_cast<float*>(out);
    of[o] = f0; of[o + plane] = f1; of[o + 2 * plane] = f2;
  }
}

}  // namespace vpr

using namespace vpr;

extern "C" size_t vpr_preprocess_workspace_bytes(int B, int H, int OW) {
  if (B <= 0 || H <= 0 || OW <= 0) return 0;
  return align_up((size_t)B * H * OW * 3, 256);
}

extern "C" int vpr_preprocess_resize_normalize(
    const uint8_t* in, int B, int H, int W, int OH, int OW,
    const int32_t* kx, const int32_t* xbounds, int ksize_x,
    const int32_t* ky, const int32_t* ybounds, int ksize_y,
    const float* mean3, const float* std3,          /* HOST pointers: 3 floats each */
    void* out, int out_is_bf16, uint8_t* out_u8,
    void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!in || !kx || !xbounds || !ky || !ybounds || !mean3 || !std3 || !out || !workspace) return VPR_ERR_INVALID_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || ksize_x <= 0 || ksize_y <= 0) return VPR_ERR_INVALID_ARG;
  if (B > 65535 || H > 65535 || OH > 65535) return VPR_ERR_UNSUPPORTED;
  if (workspace_bytes < (size_t)B * H * OW * 3) return VPR_ERR_WORKSPACE;
  uint8_t* tmp = static_cast<uint8_t*>(workspace);
  const int bx = (OW + 255) / 256;
  VPR_TRY_LAUNCH(launch_kernel(resize_h_kernel, dim3(bx, H, B), dim3(256), 0, stream, in, H, W, OW, kx, xbounds,
                               ksize_x, tmp));
  const float m0 = mean3[0], m1 = mean3[1], m2 = mean3[2];
  const float s0 = std3[0], s1 = std3[1], s2 = std3[2];   // the kernel divides, as Normalize does
  if (out_is_bf16)
    VPR_TRY_LAUNCH(launch_kernel(resize_v_normalize_kernel<true>, dim3(bx, OH, B), dim3(256), 0, stream, tmp, H, OH,
                                 OW, ky, ybounds, ksize_y, m0, m1, m2, s0, s1, s2, out, out_u8));
  else
    VPR_TRY_LAUNCH(launch_kernel(resize_v_normalize_kernel<false>, dim3(bx, OH, B), dim3(256), 0, stream, tmp, H, OH,
                                 OW, ky, ybounds, ksize_y, m0, m1, m2, s0, s1, s2, out, out_u8));
  return VPR_OK;
}
