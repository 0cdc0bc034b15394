// skinny.hip — out[M, N] = epilogue(in[M, K] * W[N, K]^T) for a handful of rows (M <= 64 per launch
// row-group): the cls-token rows of the ViT backbone, which the split row layout keeps out of the
// big GEMMs so that those see an exact number of 256-row tiles.  A library GEMM spends 9-14 us on
// each of these (four per block, 1.0 ms per step); the work is a weight stream — N*K*2 bytes
// (2-8 MB, L2/Infinity-Cache resident between steps) against 2*M*N*K <= 0.5 GFLOP.
// Workgroup = 16 output columns (16 W rows, 32-128 KB of weights) x up to 64 input rows; its 4 / 8 /
// 16 waves split K so that a wave has one or two 128-deep slices (the kernel is one or two memory
// round trips long, not a streaming loop); a wave issues all 20 fragment loads of a slice before
// the MFMAs (v_mfma_f32_16x16x32_bf16, W rows as the A operand so a lane ends up with 4
// consecutive output columns), partial sums meet in LDS, wave w finishes row block w.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

enum { SK_BIAS = 0, SK_BIAS_GELU = 1, SK_ACCUMULATE = 2, SK_BIAS_RELU = 3, SK_BIAS_GELU_ERF = 4,
       SK_BIAS_F32 = 5 /* out is float*, ldo in floats: SALAD token features */ };

__device__ __forceinline__ float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x * sigmoid(2 u): the form hipBLASLt's epilogue uses
  const float u = 0.7978845608028654f * x * fmaf(0.044715f * x, x, 1.0f);
  return x / (1.0f + __expf(-2.0f * u));
}

template <typename BiasT, int NW, int MBW>
__global__ __launch_bounds__(NW * 64) void skinny_linear_kernel(
    const uint16_t* __restrict__ in, int ldi, const uint16_t* __restrict__ W, int ldw,
    const BiasT* __restrict__ bias, int mode, uint16_t* __restrict__ out, int ldo, int M, int N, int K,
    const float* __restrict__ stats_bias, float* __restrict__ row_stats) {
  constexpr int UNR = MBW == 4 ? 4 : 8;          // K-steps per batch: (1 + MBW) * UNR fragment loads in flight
  __shared__ float red[NW][MBW][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MBW);
  const int ksteps = K >> 5;
  const int kbeg = (int)((long long)ksteps * wave / NW), kend = (int)((long long)ksteps * (wave + 1) / NW);
  const uint16_t* wp = W + (long long)min(n0 + r, N - 1) * ldw + 8 * g;
  const uint16_t* ip[MBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) ip[mb] = in + (long long)min(m0 + mb * 16 + r, M - 1) * ldi + 8 * g;
  f32x4 acc[MBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int ks = kbeg;
  for (; ks + UNR <= kend; ks += UNR) {
    bf16x8 wf[UNR], xf[UNR][MBW];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      wf[u] = *reinterpret_cast<const bf16x8*>(wp + (ks + u) * 32);
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) xf[u][mb] = *reinterpret_cast<const bf16x8*>(ip[mb] + (ks + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u][mb], acc[mb], 0, 0, 0);
  }
  for (; ks < kend; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, *reinterpret_cast<const bf16x8*>(ip[mb] + ks * 32), acc[mb], 0, 0, 0);
  }
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) *reinterpret_cast<f32x4*>(&red[wave][mb][lane][0]) = acc[mb];
  __syncthreads();
  // wave w < MBW finishes row block w: C/D col = row m (lane & 15), rows 4g+e = 4 consecutive output columns
  if (wave >= MBW) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][wave][lane][0]);
#pragma unroll
  for (int p = 1; p < NW; ++p) {                // fixed order: deterministic
    const f32x4 t = *reinterpret_cast<const f32x4*>(&red[p][wave][lane][0]);
    s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
  }
  const int m = m0 + wave * 16 + r, n = n0 + 4 * g;
  const bool live = m < M && n < N;
  if (mode == SK_BIAS_F32) {                    // f32 output, bias only
    if (live) {
      float* of = reinterpret_cast<float*>(out) + (long long)m * ldo + n;
      for (int e = 0; e < 4 && n + e < N; ++e) {
        float b;
        if constexpr (sizeof(BiasT) == 2) b = bf16_bits_to_f32((uint16_t)bias[n + e]); else b = bias[n + e];
        of[e] = s[e] + b;
      }
    }
    return;
  }
  uint16_t* op = out + (long long)min(m, M - 1) * ldo + min(n, N - 1);
  float v[4] = {s[0], s[1], s[2], s[3]};
  uint16_t ob[4] = {0, 0, 0, 0};
  if (live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e >= N) break;
      if (mode == SK_ACCUMULATE) {
        v[e] += bf16_bits_to_f32(op[e]);
      } else {
        float b;
        if constexpr (sizeof(BiasT) == 2) b = bf16_bits_to_f32((uint16_t)bias[n + e]); else b = bias[n + e];
        v[e] += b;
        if (mode == SK_BIAS_GELU) v[e] = gelu_tanh(v[e]);
        if (mode == SK_BIAS_GELU_ERF) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));   // nn.GELU() of DINOv2
        if (mode == SK_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
      }
      ob[e] = f32_to_bf16_bits(v[e]);
    }
    if (n + 3 < N && (ldo & 3) == 0) {
      ushort4 o;
      o.x = ob[0]; o.y = ob[1]; o.z = ob[2]; o.w = ob[3];
      *reinterpret_cast<ushort4*>(op) = o;
    } else {
      for (int e = 0; e < 4 && n + e < N; ++e) op[e] = ob[e];
    }
  }
  if (row_stats != nullptr) {
    // (mean, centred sum of squares) of this row over the workgroup's 16 columns, of the values the next
    // LayerNorm will read: bf16(out) + stats_bias.  The consumer merges the N/16 partials in a fixed
    // order (Chan's formula), so the LayerNorm statistics cost it no pass over the rows.  N % 16 == 0.
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      t[e] = bf16_bits_to_f32(ob[e]) + (stats_bias != nullptr && n + e < N ? stats_bias[n + e] : 0.f);
    float sm = (t[0] + t[1]) + (t[2] + t[3]);
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / 16.0f);
    float m2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = t[e] - mean; m2 = fmaf(d, d, m2); }
    m2 += __shfl_xor(m2, 16, 64);
    m2 += __shfl_xor(m2, 32, 64);
    if (g == 0 && m < M) {
      float* dst = row_stats + ((long long)blockIdx.x * M + m) * 2;
      dst[0] = mean; dst[1] = m2;
    }
  }
}

}  // namespace vpr

namespace vpr {
int launch_skinny_linear(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias,
                         int bias_is_bf16, int mode, uint16_t* out, int ldo, int M, int N, int K,
                         const float* stats_bias, float* row_stats, void* stream) {
  if (row_stats != nullptr && (N % 16)) return VPR_ERR_UNSUPPORTED;
  if (!in || !W || !out || M < 0 || N <= 0 || K <= 0 || mode < 0 || mode > 5) return VPR_ERR_INVALID_ARG;
  if (mode != SK_ACCUMULATE && !bias) return VPR_ERR_INVALID_ARG;
  if (M == 0) return VPR_OK;
  if ((K % 32) || (ldi % 8) || (ldw % 8) || ldi < K || ldw < K || ldo < N) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(W)) & 15) return VPR_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(out) & 7) return VPR_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ksteps = K / 32;
  const int nw_auto = ksteps >= 64 ? 16 : (ksteps >= 32 ? 8 : 4);
  const int nw_env = tune_or(TUNE_SKINNY_NW, 0);
  const int nw = (nw_env == 4 || nw_env == 8 || nw_env == 16) ? nw_env : nw_auto;
  // few column blocks (N <= 2048: at most 128 workgroups): one 16-row block per workgroup so the
  // launch still covers the chip and a workgroup pulls 1/4 of the input rows through its L1
  const int mbw_env = tune_or(TUNE_SKINNY_MBW, 0);              // A/B switch
  const int mbw = (mbw_env == 1 || mbw_env == 4) ? mbw_env : ((N + 15) / 16 <= 128 && M > 16 ? 1 : 4);
  const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + 16 * mbw - 1) / (16 * mbw)));
  const bool bf = bias_is_bf16 || !bias;
#define VPR_SKINNY_LAUNCH(T, NWV, MBV)                                                                              \
  VPR_TRY_LAUNCH(launch_kernel(skinny_linear_kernel<T, NWV, MBV>, grid, dim3(NWV * 64), 0, st, in, ldi, W, ldw,     \
                               static_cast<const T*>(bias), mode, out, ldo, M, N, K, stats_bias, row_stats))
#define VPR_SKINNY_NW(T, MBV)                                                                                       \
  do {                                                                                                              \
    if (nw == 16) VPR_SKINNY_LAUNCH(T, 16, MBV); else if (nw == 8) VPR_SKINNY_LAUNCH(T, 8, MBV); else VPR_SKINNY_LAUNCH(T, 4, MBV); \
  } while (0)
  if (bf) { if (mbw == 1) VPR_SKINNY_NW(uint16_t, 1); else VPR_SKINNY_NW(uint16_t, 4); }
  else    { if (mbw == 1) VPR_SKINNY_NW(float, 1); else VPR_SKINNY_NW(float, 4); }
#undef VPR_SKINNY_NW
#undef VPR_SKINNY_LAUNCH
  return VPR_OK;
}
}  // namespace vpr

using namespace vpr;

extern "C" int vpr_skinny_linear_bf16(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias,
                                      int bias_is_bf16, int mode, uint16_t* out, int ldo, int M, int N, int K,
                                      void* stream) {
  return launch_skinny_linear(in, ldi, W, ldw, bias, bias_is_bf16, mode, out, ldo, M, N, K, nullptr, nullptr, stream);
}

extern "C" int vpr_skinny_linear_stats_bf16(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias,
                                            int bias_is_bf16, int mode, uint16_t* out, int ldo, int M, int N, int K,
                                            const float* stats_bias, float* row_stats, void* stream) {
  if (!row_stats) return VPR_ERR_INVALID_ARG;
  return launch_skinny_linear(in, ldi, W, ldw, bias, bias_is_bf16, mode, out, ldo, M, N, K, stats_bias, row_stats, stream);
}
