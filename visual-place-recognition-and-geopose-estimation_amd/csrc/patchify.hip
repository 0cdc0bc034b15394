// patchify.hip — non-overlapping patch extraction for the ViT patch embedding (backbone helper).
// The stride-P PxP convolution is a GEMM over flattened patches; MIOpen runs it as an implicit GEMM
// plus layout transposes (~150 us at B = 64) and PyTorch then needs a cat (cls row) and an add
// (position embedding).  This kernel writes the GEMM's A operand directly in the token layout:
//   out[b*(lead + G*G) + lead + py*G + px][c*P*P + i*P + j] = img[b][c][py*P + i][px*P + j]
// with `lead` zero rows per image (the cls slot) and the K dimension zero-padded to kpad, so one
// library GEMM produces [B, 1 + n, C] in place and the cls / position / bias terms become a static
// additive matrix consumed by the first (add+)LayerNorm.
// HBM-bound: algorithmic bytes = B*3*H*W*2 in + B*(lead+n)*kpad*2 out (19.3 + 21.1 MB at B = 64).
// One workgroup per (image, patch row): the 3 x P image rows (contiguous 2*W-byte lines) are
// staged in LDS with 16-byte loads, then the G patches of that row leave as 16-byte chunks.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int PF_MAX_LDS = 48 * 1024;

__global__ __launch_bounds__(256) void patchify_kernel(const uint16_t* __restrict__ img, uint16_t* __restrict__ out,
                                                       int Cin, int H, int W, int P, int kpad, int lead) {
  extern __shared__ __attribute__((aligned(16))) uint16_t rows[];   // [Cin][P][W]
  const int G = W / P, GH = H / P;
  const int b = blockIdx.x / GH, py = blockIdx.x % GH;
  const int chunks_per_line = W >> 3;
  const int nlines = Cin * P;
  for (int t = threadIdx.x; t < nlines * chunks_per_line; t += blockDim.x) {
    const int line = t / chunks_per_line, ch = t - line * chunks_per_line;
    const int c = line / P, i = line - c * P;
    const uint16_t* src = img + (((long long)b * Cin + c) * H + (py * P + i)) * W + ch * 8;
    *reinterpret_cast<s16x8*>(rows + line * W + ch * 8) = *reinterpret_cast<const s16x8*>(src);
  }
  __syncthreads();
  const int K = Cin * P * P;
  const int kchunks = kpad >> 3;
  const long long row0 = (long long)b * (lead + GH * G) + lead + (long long)py * G;
  const bool pairs = ((P | W) & 1) == 0;      // even P, W: k even <=> j even, so elements (k, k+1) share a 4-byte LDS word
  for (int t = threadIdx.x; t < G * kchunks; t += blockDim.x) {
    const int px = t / kchunks, kc = t - px * kchunks;
    s16x8 o;
    if (pairs) {
      const uint32_t* rows32 = reinterpret_cast<const uint32_t*>(rows);
      uint32_t w4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = kc * 8 + 2 * q;
        uint32_t v = 0;
        if (k < K) {
          const int c = k / (P * P), r = k - c * P * P;
          const int i = r / P, j = r - i * P;
          v = rows32[((c * P + i) * W + px * P + j) >> 1];
        }
        w4[q] = v;
      }
      const uint4 u4 = make_uint4(w4[0], w4[1], w4[2], w4[3]);
      o = __builtin_bit_cast(s16x8, u4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kc * 8 + e;
        short v = 0;
        if (k < K) {
          const int c = k / (P * P), r = k - c * P * P;
          const int i = r / P, j = r - i * P;
          v = (short)rows[(c * P + i) * W + px * P + j];
        }
        o[e] = v;
      }
    }
    *reinterpret_cast<s16x8*>(out + (row0 + px) * kpad + kc * 8) = o;
  }
  if (py == 0) {   // the image's leading (cls) rows: zeros
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = threadIdx.x; t < lead * kchunks; t += blockDim.x)
      *reinterpret_cast<s16x8*>(out + ((long long)b * (lead + GH * G)) * kpad + t * 8) = z;
  }
}

}  // namespace vpr

using namespace vpr;

extern "C" int vpr_patchify_bf16(const uint16_t* images, int B, int Cin, int H, int W, int patch, int kpad,
                                 int lead_rows, uint16_t* out, void* stream) {
  if (!images || !out || B < 0 || Cin <= 0 || H <= 0 || W <= 0 || patch <= 0 || lead_rows < 0) return VPR_ERR_INVALID_ARG;
  if (B == 0) return VPR_OK;
  if ((H % patch) || (W % patch) || (W % 8) || (kpad % 8) || kpad < Cin * patch * patch) return VPR_ERR_UNSUPPORTED;
  const size_t lds = (size_t)Cin * patch * W * sizeof(uint16_t);
  if (lds > PF_MAX_LDS) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(images) | reinterpret_cast<uintptr_t>(out)) & 15) return VPR_ERR_UNSUPPORTED;
  if ((long long)B * (H / patch) > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  VPR_TRY_LAUNCH(launch_kernel(patchify_kernel, dim3((unsigned)(B * (H / patch))), dim3(256), lds,
                               static_cast<hipStream_t>(stream), images, out, Cin, H, W, patch, kpad, lead_rows));
  return VPR_OK;
}
