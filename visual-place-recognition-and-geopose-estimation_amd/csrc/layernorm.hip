// layernorm.hip — row LayerNorm for bf16 activations (the DINOv2 backbone calls it 49 times per
// forward: 2 per block + the final norm; PyTorch's kernel reaches ~2 TB/s on [16448, 1024]).
// HBM-bound: algorithmic bytes = 2 * M * C * 2.  One wave per row, 4 rows per workgroup, every
// lane holds its 16-B chunks in registers (chunk = lane + 64*i), two-pass statistics in f32
// (mean, then centred sum of squares) like torch, output rounded once to bf16.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

// With `res` != nullptr the kernel first forms s = bf16(x + res) (the residual add PyTorch would do
// as a separate pass), stores it to `sum_out`, and normalises s — one read of each operand, two
// writes, instead of add (r2 w1) + LayerNorm (r1 w1).
template <int NCH, typename ParamT, int RW>
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ res, uint16_t* __restrict__ sum_out,
    const float* __restrict__ pre_bias, const ParamT* __restrict__ gamma, const ParamT* __restrict__ beta,
    float eps, uint16_t* __restrict__ y, long long M, int C) {
  // RW rows per wave, all their 16-byte chunks requested before the first reduction (one round of
  // workgroups for [16448, 1024] instead of two).  It did not pay: see launch_ln.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * RW;
  if (row0 >= M) return;
  const int nchunks = C >> 3;
  float v[RW][NCH][8];
  float s[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const bool valid = row0 + rr < M;
    const long long row = valid ? row0 + rr : M - 1;              // a clamped duplicate re-reads the last row, stores nothing
    const uint16_t* xr = x + row * C;
    s[rr] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nchunks) {
        const s16x8 q = *reinterpret_cast<const s16x8*>(xr + ch * 8);
        if (res != nullptr) {   // uniform
          const s16x8 r = *reinterpret_cast<const s16x8*>(res + row * C + ch * 8);
          s16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint16_t sb = f32_to_bf16_bits(bf16_bits_to_f32((uint16_t)q[e]) + bf16_bits_to_f32((uint16_t)r[e]));
            o[e] = (short)sb;
            v[rr][i][e] = bf16_bits_to_f32(sb);
            s[rr] += v[rr][i][e];
          }
          if (valid) *reinterpret_cast<s16x8*>(sum_out + row * C + ch * 8) = o;
        } else if (pre_bias != nullptr) {   // uniform: a per-column f32 offset carried outside the bf16 stream
          const float4 p0 = *reinterpret_cast<const float4*>(pre_bias + ch * 8);
          const float4 p1 = *reinterpret_cast<const float4*>(pre_bias + ch * 8 + 4);
          const float pb[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[rr][i][e] = bf16_bits_to_f32((uint16_t)q[e]) + pb[e]; s[rr] += v[rr][i][e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[rr][i][e] = bf16_bits_to_f32((uint16_t)q[e]); s[rr] += v[rr][i][e]; }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[rr][i][e] = 0.f;
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const long long row = row0 + rr;
    if (row >= M) break;
    const float mean = wave_sum(s[rr]) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < nchunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[rr][i][e] - mean; ss = fmaf(d, d, ss); }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
    uint16_t* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nchunks) {
        s16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float g, b;
          if constexpr (sizeof(ParamT) == 2) {
            g = bf16_bits_to_f32((uint16_t)gamma[ch * 8 + e]);
            b = bf16_bits_to_f32((uint16_t)beta[ch * 8 + e]);
          } else {
            g = gamma[ch * 8 + e];
            b = beta[ch * 8 + e];
          }
          o[e] = (short)f32_to_bf16_bits((v[rr][i][e] - mean) * rstd * g + b);
        }
        *reinterpret_cast<s16x8*>(yr + ch * 8) = o;
      }
    }
  }
}

template <typename ParamT>
static int launch_ln(const uint16_t* x, const uint16_t* res, uint16_t* sum_out, const float* pb, const ParamT* g, const ParamT* b,
                     float eps, uint16_t* y, long long M, int C, hipStream_t stream) {
  const int nch = (C / 8 + 63) / 64;
  // A/B switch: rows per wave.  Measured in bench.py on one box:
  const int rw = tune_or(TUNE_LN_ROWS, 1);               // 2 rows 11.80/11.80 ms per step, 1 row 11.75/11.72 -> default 1
#define VPR_LN_LAUNCH(NCHV, RWV)                                                                                   \
  VPR_TRY_LAUNCH(launch_kernel(layernorm_bf16_kernel<NCHV, ParamT, RWV>, dim3((unsigned)((M + 4 * RWV - 1) / (4 * RWV))), \
                               dim3(256), 0, stream, x, res, sum_out, pb, g, b, eps, y, M, C))
  if (rw == 2 && nch <= 2) {
    if (nch == 1) VPR_LN_LAUNCH(1, 2); else VPR_LN_LAUNCH(2, 2);
    return VPR_OK;
  }
  switch (nch) {
    case 1: VPR_LN_LAUNCH(1, 1); break;
    case 2: VPR_LN_LAUNCH(2, 1); break;
    case 3: VPR_LN_LAUNCH(3, 1); break;
    case 4: VPR_LN_LAUNCH(4, 1); break;
    default: return VPR_ERR_UNSUPPORTED;
  }
#undef VPR_LN_LAUNCH
  return VPR_OK;
}

}  // namespace vpr

using namespace vpr;

extern "C" int vpr_layernorm_bf16(const uint16_t* x, const void* gamma, const void* beta, int params_are_bf16,
                                  float eps, uint16_t* y, long long M, int C, void* stream) {
  if (!x || !gamma || !beta || !y || M < 0 || C <= 0) return VPR_ERR_INVALID_ARG;
  if (M == 0) return VPR_OK;
  if ((C % 8) || C > 2048 || M > 0x1fffffffcLL) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return VPR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (params_are_bf16)
    return launch_ln<uint16_t>(x, nullptr, nullptr, nullptr, static_cast<const uint16_t*>(gamma), static_cast<const uint16_t*>(beta), eps, y, M, C, s);
  return launch_ln<float>(x, nullptr, nullptr, nullptr, static_cast<const float*>(gamma), static_cast<const float*>(beta), eps, y, M, C, s);
}

extern "C" int vpr_add_layernorm_bf16(const uint16_t* x, const uint16_t* res, uint16_t* sum_out, const void* gamma,
                                      const void* beta, int params_are_bf16, float eps, uint16_t* y, long long M,
                                      int C, void* stream) {
  if (!x || !res || !sum_out || !gamma || !beta || !y || M < 0 || C <= 0) return VPR_ERR_INVALID_ARG;
  if (M == 0) return VPR_OK;
  if ((C % 8) || C > 2048 || M > 0x1fffffffcLL) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) |
       reinterpret_cast<uintptr_t>(sum_out)) & 15)
    return VPR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (params_are_bf16)
    return launch_ln<uint16_t>(x, res, sum_out, nullptr, static_cast<const uint16_t*>(gamma), static_cast<const uint16_t*>(beta), eps, y, M, C, s);
  return launch_ln<float>(x, res, sum_out, nullptr, static_cast<const float*>(gamma), static_cast<const float*>(beta), eps, y, M, C, s);
}

extern "C" int vpr_bias_layernorm_bf16(const uint16_t* x, const float* pre_bias, const void* gamma, const void* beta,
                                       int params_are_bf16, float eps, uint16_t* y, long long M, int C, void* stream) {
  if (!x || !pre_bias || !gamma || !beta || !y || M < 0 || C <= 0) return VPR_ERR_INVALID_ARG;
  if (M == 0) return VPR_OK;
  if ((C % 8) || C > 2048 || M > 0x1fffffffcLL) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(pre_bias)) & 15)
    return VPR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (params_are_bf16)
    return launch_ln<uint16_t>(x, nullptr, nullptr, pre_bias, static_cast<const uint16_t*>(gamma), static_cast<const uint16_t*>(beta), eps, y, M, C, s);
  return launch_ln<float>(x, nullptr, nullptr, pre_bias, static_cast<const float*>(gamma), static_cast<const float*>(beta), eps, y, M, C, s);
}
