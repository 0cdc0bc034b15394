// ln_cls_linear.hip — one launch for "LayerNorm of every row" + "linear layer on the LayerNorm of the
// few cls rows".  In the split row layout the cls rows' linear layers are 64-row launches that cost
// ~7 us each (dispatch latency + two memory round trips, measured by doubling them: +0.33 ms per 48
// launches) for ~3 us of work.  The two that directly follow a LayerNorm (qkv, fc1) need nothing
// but the residual rows the LayerNorm itself reads, so their workgroups ride in the LayerNorm's
// grid: blocks [0, sk_blocks) take the statistics of the cls rows from the per-16-column partials
// the producer of those rows left behind (vpr_skinny_linear_stats_bf16; merged in a fixed order,
// no pass over the rows) and run the skinny GEMM of skinny.hip on the RAW rows, the normalisation
// being applied to the 64 x 16 result instead of the 64 x C input (identity in the comment below;
// normalising the operand fragments in every workgroup repeats the LayerNorm's VALU work N/16
// times and made the launch 2x longer); the remaining blocks are the LayerNorm of layernorm.hip.  The skinny
// blocks come first in the grid so they are resident while the LayerNorm blocks stream.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

struct LnClsArgs {
  const uint16_t* x; const float* pre_bias; const uint16_t* gamma; const uint16_t* beta; float eps;
  uint16_t* y; long long M; int C;
  long long cls_row0; int n_cls; const float* row_stats; int stat_parts;
  const uint16_t* W; int ldw; const float* colsum; const float* cprime; const float* bprime; int gelu; uint16_t* out; int ldo; int N;
  int sk_blocks_x, sk_blocks;
};

__device__ __forceinline__ float lcl_gelu_tanh(float x) {
  const float u = 0.7978845608028654f * x * fmaf(0.044715f * x, x, 1.0f);
  return x / (1.0f + __expf(-2.0f * u));
}

__device__ __forceinline__ void lcl_load_chunk(const uint16_t* xr, const float* pb, int ch, float (&v)[8]) {
  const s16x8 q = *reinterpret_cast<const s16x8*>(xr + ch * 8);
  if (pb != nullptr) {
    const float4 p0 = *reinterpret_cast<const float4*>(pb + ch * 8);
    const float4 p1 = *reinterpret_cast<const float4*>(pb + ch * 8 + 4);
    const float p[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf16_bits_to_f32((uint16_t)q[e]) + p[e];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf16_bits_to_f32((uint16_t)q[e]);
  }
}

// ---- the cls-row linear of LayerNorm(x + pre_bias), without normalising anything element-wise:
//   LN(v) W^T + b = rstd * (x W'^T + c' - mean * colsum(W')) + b',   v = x + pre_bias,
//   W' = W * diag(gamma), c' = W' pre_bias, b' = b + W beta   (W', c', colsum, b' are static per layer)
// so the raw bf16 rows are the MFMA operand as they are (4 waves split K, 64 rows x 16 columns).
__device__ __forceinline__ void lcl_skinny_block(const LnClsArgs& a, int bx, int by, float (*red)[4][64][4], float (*stats)[2]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C;
  const int m0 = by * 64, n0 = bx * 16;
  const int r = lane & 15, g = lane >> 4;
  const uint16_t* wp = a.W + (long long)min(n0 + r, a.N - 1) * a.ldw + 8 * g;
  const uint16_t* ip[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) ip[mb] = a.x + (a.cls_row0 + min(m0 + mb * 16 + r, a.n_cls - 1)) * C + 8 * g;
  const int ksteps = C >> 5;
  const int kbeg = ksteps * wave / 4, kend = ksteps * (wave + 1) / 4;
  f32x4 acc[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int ks = kbeg;
  // two K-steps per round trip (10 fragment loads in flight per lane): the kernel must stay within 64 VGPRs
  // so that the LayerNorm blocks keep 8 waves per SIMD — at 5 the HBM stream of the LayerNorm part slows down
  for (; ks + 2 <= kend; ks += 2) {
    bf16x8 wf[2], xf[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      wf[u] = *reinterpret_cast<const bf16x8*>(wp + (ks + u) * 32);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) xf[u][mb] = *reinterpret_cast<const bf16x8*>(ip[mb] + (ks + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u][mb], acc[mb], 0, 0, 0);
  }
  for (; ks < kend; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, *reinterpret_cast<const bf16x8*>(ip[mb] + ks * 32), acc[mb], 0, 0, 0);
  }
  {   // statistics of rows m0 .. m0+63 from the producer's per-16-column partials (mean_p, M2_p), 4 threads
      // per row, equal counts, fixed order: mean = avg(mean_p), M2 = sum(M2_p) + 16 * sum((mean_p - mean)^2)
    const int row = min(m0 + (tid >> 2), a.n_cls - 1), part = tid & 3;
    const float* ps = a.row_stats + (long long)row * 2;
    const long long pstride = (long long)a.n_cls * 2;
    float sm = 0.f;
#pragma unroll 8
    for (int p = part; p < a.stat_parts; p += 4) sm += ps[p * pstride];
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    const float mean = sm / (float)a.stat_parts;
    float m2 = 0.f;
#pragma unroll 8
    for (int p = part; p < a.stat_parts; p += 4) {
      const float2 q = *reinterpret_cast<const float2*>(ps + p * pstride);
      const float d = q.x - mean;
      m2 += q.y + 16.0f * d * d;
    }
    m2 += __shfl_xor(m2, 1, 64);
    m2 += __shfl_xor(m2, 2, 64);
    if (part == 0) { stats[tid >> 2][0] = mean; stats[tid >> 2][1] = 1.0f / sqrtf(m2 / (float)C + a.eps); }
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<f32x4*>(&red[wave][mb][lane][0]) = acc[mb];
  __syncthreads();
  f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][wave][lane][0]);
#pragma unroll
  for (int p = 1; p < 4; ++p) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(&red[p][wave][lane][0]);
    s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
  }
  const int m = m0 + wave * 16 + r, n = n0 + 4 * g;
  if (m >= a.n_cls || n >= a.N) return;
  const float mean = stats[wave * 16 + r][0], rstd = stats[wave * 16 + r][1];
  uint16_t* op = a.out + (long long)m * a.ldo + n;
  float v[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e >= a.N) break;
    v[e] = fmaf(rstd, v[e] + a.cprime[n + e] - mean * a.colsum[n + e], a.bprime[n + e]);
    if (a.gelu) v[e] = lcl_gelu_tanh(v[e]);
  }
  if (n + 3 < a.N && (a.ldo & 3) == 0) {
    ushort4 o;
    o.x = f32_to_bf16_bits(v[0]); o.y = f32_to_bf16_bits(v[1]); o.z = f32_to_bf16_bits(v[2]); o.w = f32_to_bf16_bits(v[3]);
    *reinterpret_cast<ushort4*>(op) = o;
  } else {
    for (int e = 0; e < 4 && n + e < a.N; ++e) op[e] = f32_to_bf16_bits(v[e]);
  }
}

template <int NCH>
__global__ __launch_bounds__(256, 8) void ln_cls_linear_kernel(LnClsArgs a) {
  __shared__ float red[4][4][64][4];
  __shared__ float stats[64][2];
  if ((int)blockIdx.x < a.sk_blocks) {      // block-uniform
    lcl_skinny_block(a, blockIdx.x % a.sk_blocks_x, blockIdx.x / a.sk_blocks_x, red, stats);
    return;
  }
  // ---- LayerNorm rows (same arithmetic as layernorm_bf16_kernel: wave per row, two-pass f32 statistics) ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)(blockIdx.x - a.sk_blocks) * 4 + wave;
  if (row >= a.M) return;
  const int C = a.C, nchunks = C >> 3;
  const uint16_t* xr = a.x + row * C;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
      lcl_load_chunk(xr, a.pre_bias, ch, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + 64 * i < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; ss = fmaf(d, d, ss); }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + a.eps);
  uint16_t* yr = a.y + row * C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunks) {
      const s16x8 gq = *reinterpret_cast<const s16x8*>(a.gamma + ch * 8);
      const s16x8 bq = *reinterpret_cast<const s16x8*>(a.beta + ch * 8);
      s16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = (short)f32_to_bf16_bits((v[i][e] - mean) * rstd * bf16_bits_to_f32((uint16_t)gq[e]) +
                                       bf16_bits_to_f32((uint16_t)bq[e]));
      *reinterpret_cast<s16x8*>(yr + ch * 8) = o;
    }
  }
}

}  // namespace vpr

using namespace vpr;

extern "C" int vpr_bias_layernorm_cls_linear_bf16(const uint16_t* x, const float* pre_bias, const uint16_t* gamma,
                                                  const uint16_t* beta, float eps, uint16_t* y, long long M, int C,
                                                  long long cls_row0, int n_cls, const float* row_stats,
                                                  const uint16_t* W_scaled, int ldw, const float* colsum,
                                                  const float* cprime, const float* bprime, int gelu,
                                                  uint16_t* out, int ldo, int N, void* stream) {
  if (!row_stats || !colsum || !cprime || !bprime) return VPR_ERR_INVALID_ARG;
  const uint16_t* W = W_scaled;
  if (!x || !gamma || !beta || !y || !W_scaled || !out || M <= 0 || C <= 0 || n_cls <= 0 || N <= 0 || cls_row0 < 0)
    return VPR_ERR_INVALID_ARG;
  if (cls_row0 + n_cls > M) return VPR_ERR_INVALID_ARG;
  if ((C % 32) || C > 2048 || (ldw % 8) || ldw < C || ldo < N || M > 0x1fffffffcLL) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(W) |
       reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(pre_bias)) & 15)
    return VPR_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(out) & 7) return VPR_ERR_UNSUPPORTED;
  LnClsArgs a{x, pre_bias, gamma, beta, eps, y, M, C, cls_row0, n_cls, row_stats, C / 16, W, ldw, colsum, cprime, bprime, gelu, out, ldo, N, 0, 0};
  a.sk_blocks_x = (N + 15) / 16;
  a.sk_blocks = a.sk_blocks_x * ((n_cls + 63) / 64);
  const long long blocks = a.sk_blocks + (M + 3) / 4;
  if (blocks > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch ((C / 8 + 63) / 64) {
    case 1: VPR_TRY_LAUNCH(launch_kernel(ln_cls_linear_kernel<1>, grid, dim3(256), 0, st, a)); break;
    case 2: VPR_TRY_LAUNCH(launch_kernel(ln_cls_linear_kernel<2>, grid, dim3(256), 0, st, a)); break;
    case 3: VPR_TRY_LAUNCH(launch_kernel(ln_cls_linear_kernel<3>, grid, dim3(256), 0, st, a)); break;
    case 4: VPR_TRY_LAUNCH(launch_kernel(ln_cls_linear_kernel<4>, grid, dim3(256), 0, st, a)); break;
    default: return VPR_ERR_UNSUPPORTED;
  }
  return VPR_OK;
}
