// pose_head.hip — regression / angle heads of the geopose path, f32 end to end.
//
// vpr_pose_head          Linear(D,hidden) -> ReLU -> Linear(hidden,n_out) [+ unit-normalise a pair]
//   replaces DINOv2RegressionModel.regressor   dinov2salad/dinov2salad_validation.py:43-47,52
//            Swin-Base MLP head (Dropout = identity in eval)  swin_transformer/val_and_test_swin_2.py:168-177
//            sin/cos MLP head   angle_prediction/swin/swin_angle_finetuning_gemini.py:101-106
//   hidden == 0: single Linear(D,n_out)       swin_transformer/swin_validation.py:41,46
//            + F.normalize(dim=1,p=2,eps=1e-6) angle_prediction/swin/swin_angle_finetuning_sin_cos.py:56-62
// vpr_ln_meanpool_head   HF SwinModel pooler (LayerNorm -> mean over tokens) + linear head
//   replaces `outputs.pooler_output` + `self.regressor`  swin_transformer/swin_validation.py:43-46
//
// The first layer is a weight-streaming skinny GEMM (HBM/L2-bound on W1): split-K over the
// grid, exact-f32 MFMA (v_mfma_f32_16x16x4_f32 == an fmaf chain, cdna guide §3), partial slabs
// summed in a fixed order by the epilogue kernel — bitwise reproducible run to run.
#include <math.h>
#include <stdlib.h>
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int PH_BT = 64;   // batch rows per workgroup (4 waves x 16)
constexpr int PH_HT = 32;   // hidden units per workgroup (2 MFMA column blocks)
constexpr int PH_CH = 12;   // k-steps (16 of D each) loaded ahead of their MFMAs

__global__ __launch_bounds__(256) void pose_l1_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ W1, float* __restrict__ part,
    int B, int D, int hidden, int steps_per_slice) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h0 = blockIdx.x * PH_HT;
  const int ks = blockIdx.y;
  const int b0 = blockIdx.z * PH_BT + 16 * wave;
  const int nsteps = D >> 4;                       // 16 k per step (one float4 per lane)
  const int s_begin = ks * steps_per_slice;
  const int s_end = min(nsteps, s_begin + steps_per_slice);
  // operand maps of 16x16x4: A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15].
  // Each lane loads 4 consecutive k (one float4) and feeds element t to MFMA t: lanes of one
  // k-group use the same 4 k values for A and for B, so the k permutation cancels.
  const int r = lane & 15, kg = lane >> 4;
  const int brow = min(b0 + r, B - 1);
  const float4* xa = reinterpret_cast<const float4*>(x + (long long)brow * D) + kg;
  const float4* wa = reinterpret_cast<const float4*>(W1 + (long long)(h0 + r) * D) + kg;
  const float4* wb = reinterpret_cast<const float4*>(W1 + (long long)(h0 + 16 + r) * D) + kg;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // PH_CH steps of loads (3 x 16 B per lane each) are issued before their MFMAs, so a slice is
  // a few round trips to HBM/L2 instead of one per step.
  for (int s0 = s_begin; s0 < s_end; s0 += PH_CH) {
    float4 a[PH_CH], w0[PH_CH], w1[PH_CH];
#pragma unroll
    for (int i = 0; i < PH_CH; ++i) {
      const int s = min(s0 + i, s_end - 1);
      a[i] = xa[s * 4];
      w0[i] = wa[s * 4];
      w1[i] = wb[s * 4];
    }
#pragma unroll
    for (int i = 0; i < PH_CH; ++i) {
      if (s0 + i < s_end) {   // wave-uniform
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, w0[i].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, w1[i].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, w0[i].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, w1[i].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, w0[i].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, w1[i].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, w0[i].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, w1[i].w, acc1, 0, 0, 0);
      }
    }
  }
  // C/D: col (hidden) = lane&15, row (batch) = 4*(lane>>4) + e
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int b = b0 + 4 * kg + e;
    if (b < B) {
      float* p = part + ((long long)ks * B + b) * hidden + h0;
      p[r] = acc0[e];
      p[16 + r] = acc1[e];
    }
  }
}

__device__ __forceinline__ float block_sum_256p(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ void finish_outputs(float* outs, int n_out, int sincos_offset, float* dst) {
  // F.normalize(p=2, dim=1, eps=1e-6) of the pair [off, off+1]: v / max(||v||, eps)
  if (threadIdx.x == 0) {
    if (sincos_offset >= 0 && sincos_offset + 1 < n_out) {
      const float a = outs[sincos_offset], b = outs[sincos_offset + 1];
      const float den = fmaxf(sqrtf(a * a + b * b), 1e-6f);
      outs[sincos_offset] = a / den;
      outs[sincos_offset + 1] = b / den;
    }
    for (int o = 0; o < n_out; ++o) dst[o] = outs[o];
  }
}

// hidden > 0: one workgroup per batch row; sums the split-K slabs (fixed order), bias, ReLU,
// second layer, optional pair normalise.
__global__ __launch_bounds__(256) void pose_epilogue_kernel(
    const float* __restrict__ part, int nslice, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out,
    int B, int hidden, int n_out, int sincos_offset) {
  __shared__ float red[8][4];
  __shared__ float outs[8];
  const int b = blockIdx.x;
  float po[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) po[o] = 0.f;
  // a thread owns 4 consecutive hidden units: one float4 per slab, all slabs of a group of 16 requested
  // together (hidden = 1024, 16 slabs: a single round trip), summed in slice order
  for (int h = threadIdx.x * 4; h < hidden; h += 1024) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ks0 = 0; ks0 < nslice; ks0 += 16) {
      float4 t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        t[i] = *reinterpret_cast<const float4*>(part + ((long long)min(ks0 + i, nslice - 1) * B + b) * hidden + h);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (ks0 + i < nslice) { s.x += t[i].x; s.y += t[i].y; s.z += t[i].z; s.w += t[i].w; }
    }
    const float4 bb = *reinterpret_cast<const float4*>(b1 + h);
    s.x = fmaxf(s.x + bb.x, 0.f); s.y = fmaxf(s.y + bb.y, 0.f); s.z = fmaxf(s.z + bb.z, 0.f); s.w = fmaxf(s.w + bb.w, 0.f);
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < n_out) {
        const float4 w = *reinterpret_cast<const float4*>(W2 + (long long)o * hidden + h);
        po[o] = fmaf(s.x, w.x, fmaf(s.y, w.y, fmaf(s.z, w.z, fmaf(s.w, w.w, po[o]))));
      }
  }
  // all outputs reduced together: wave sums, one barrier, fixed order
#pragma unroll
  for (int o = 0; o < 8; ++o)
    if (o < n_out) {   // n_out is uniform
      const float v = wave_sum(po[o]);
      if ((threadIdx.x & 63) == 0) red[o][threadIdx.x >> 6] = v;
    }
  __syncthreads();
  if ((int)threadIdx.x < n_out) outs[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]) + b2[threadIdx.x];
  __syncthreads();
  finish_outputs(outs, n_out, sincos_offset, out + (long long)b * n_out);
}

// hidden == 0: out = W2 x + b2, one workgroup per batch row.
__global__ __launch_bounds__(256) void pose_linear_kernel(
    const float* __restrict__ x, const float* __restrict__ W2, const float* __restrict__ b2,
    float* __restrict__ out, int D, int n_out, int sincos_offset) {
  __shared__ float red[4];
  __shared__ float outs[8];
  const int b = blockIdx.x;
  float po[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) po[o] = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float xv = x[(long long)b * D + d];
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < n_out) po[o] = fmaf(xv, W2[(long long)o * D + d], po[o]);
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o < n_out) {
      const float tot = block_sum_256p(po[o], red);
      if (threadIdx.x == 0) outs[o] = tot + b2[o];
    }
  }
  __syncthreads();
  finish_outputs(outs, n_out, sincos_offset, out + (long long)b * n_out);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm -> mean over tokens -> optional linear head.  One workgroup per image, wave per
// token (4 tokens in flight), lane holds VPL = H/64 elements as VPL/4 4-element vectors
// (element = c*256 + lane*4 + e: 8-B bf16 / 16-B f32 loads, coalesced over the wave).
// Two-pass statistics in registers (mean, then centred sum of squares) like torch's LayerNorm.
// ---------------------------------------------------------------------------------------------
template <int VPL, bool BF16>
__global__ __launch_bounds__(1024) void ln_meanpool_head_kernel(
    const void* __restrict__ xin, int T, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ pooled_out,
    const float* __restrict__ Wh, const float* __restrict__ bh, int n_out, int sincos_offset,
    float* __restrict__ out) {
  constexpr int H = VPL * 64;
  constexpr int NV = VPL / 4;
  constexpr int NW = 16;           // 16 waves per image ...
  constexpr int TB = VPL <= 16 ? 4 : 2;   // ... each with TB token rows requested before the first reduction: T = 49 is ONE
                                   // memory round trip per wave, T = 144 three (round 1: 8 waves, one row ahead = 7 / 18
                                   // dependent trips: 12.8 / 21.3 us for 25.7 / 75.5 MB)
  __shared__ float pool[NW][H];
  __shared__ float outs[8];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  float gm[VPL], bt[VPL], accp[VPL];
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int h = c * 256 + lane * 4 + e;
      gm[c * 4 + e] = gamma[h];
      bt[c * 4 + e] = beta[h];
      accp[c * 4 + e] = 0.f;
    }

  auto load_row = [&](int t, float (&xv)[VPL]) {
    const int tc = t < T ? t : T - 1;                // clamped past the end: a harmless re-read of the last row
    if (BF16) {
      const uint16_t* row = reinterpret_cast<const uint16_t*>(xin) + ((long long)b * T + tc) * H;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(row + c * 256 + lane * 4);
        xv[c * 4 + 0] = bf16_bits_to_f32(q.x);
        xv[c * 4 + 1] = bf16_bits_to_f32(q.y);
        xv[c * 4 + 2] = bf16_bits_to_f32(q.z);
        xv[c * 4 + 3] = bf16_bits_to_f32(q.w);
      }
    } else {
      const float* row = reinterpret_cast<const float*>(xin) + ((long long)b * T + tc) * H;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const float4 q = *reinterpret_cast<const float4*>(row + c * 256 + lane * 4);
        xv[c * 4 + 0] = q.x; xv[c * 4 + 1] = q.y; xv[c * 4 + 2] = q.z; xv[c * 4 + 3] = q.w;
      }
    }
  };
  // token order per wave is fixed (wave, wave + NW, ...), and so is the order of its running sum: deterministic
  for (int t0 = wave; t0 < T; t0 += NW * TB) {
    float xv[TB][VPL];
#pragma unroll
    for (int j = 0; j < TB; ++j) load_row(t0 + NW * j, xv[j]);
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      if (t0 + NW * j < T) {                         // wave-uniform
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += xv[j][i];
        const float mean = wave_sum(s) / (float)H;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const float d = xv[j][i] - mean; ss = fmaf(d, d, ss); }
        const float var = wave_sum(ss) / (float)H;
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) accp[i] += (xv[j][i] - mean) * rstd * gm[i] + bt[i];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) pool[wave][c * 256 + lane * 4 + e] = accp[c * 4 + e];
  __syncthreads();
  // fixed-order cross-wave sum; thread h keeps pooled[h] for the head below
  for (int h = threadIdx.x; h < H; h += NW * 64) {
    float p = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 4) p += (pool[w][h] + pool[w + 1][h]) + (pool[w + 2][h] + pool[w + 3][h]);
    p /= (float)T;
    pool[0][h] = p;
    if (pooled_out) pooled_out[(long long)b * H + h] = p;
  }
  __syncthreads();
  if (Wh == nullptr || n_out <= 0) return;
  for (int o = wave; o < n_out; o += NW) {
    float s = 0.f;
    for (int h = lane; h < H; h += 64) s = fmaf(pool[0][h], Wh[(long long)o * H + h], s);
    s = wave_sum(s);
    if (lane == 0) outs[o] = s + bh[o];
  }
  __syncthreads();
  finish_outputs(outs, n_out, sincos_offset, out + (long long)b * n_out);
}

// ---- first layer on the bf16 matrix pipe at f32 accuracy -------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate: at B = 64, hidden = 1024 the 1.1 GFLOP of the first
// layer are 7 us of matrix-pipe time on their own, as much as the 34.6 MB weight stream.  Every f32 value
// is the sum of two bf16 values to 2^-17 (hi = bf16(v), lo = bf16(v - hi)), products of bf16 pairs are exact
// in f32, so  x w = (x_hi + x_lo)(w_hi + w_lo)  is four bf16 MFMAs with f32 accumulation: error per product
// <= 2^-16 relative (random sign), i.e. ~1e-7 absolute on outputs of magnitude 0.1 — the size of the f32
// summation error itself.  W1 is packed once into (hi, lo) bf16 planes: the same 4 bytes per weight.
__global__ __launch_bounds__(256) void pose_pack_split_kernel(const float* __restrict__ w, long long count,
                                                              uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float v = w[i];
  const uint16_t h = f32_to_bf16_bits(v);
  hi[i] = h;
  lo[i] = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
}

__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {       // hardware RNE conversions (v_cvt_pk_bf16_f32): 3 VALU ops per element
    hi[e] = (__bf16)v[e];
    lo[e] = (__bf16)(v[e] - (float)hi[e]);
  }
}

// grid (hidden / 64, nslice, ceil(B / 64)); workgroup = 64 hidden units x 64 batch rows x one K slice.  The four
// waves split the slice's K-steps (each wave: all 4 hidden blocks x all 4 batch blocks = 16 accumulators), so a
// wave's part is two or three memory round trips of 32 x 16-byte loads, x crosses the L2 once per 64 hidden
// units, and one workgroup per CU keeps 128 KB in flight.  Partial accumulators meet in LDS; wave w finishes
// hidden block w.
// NW = 8 (round 3 experiment, VPR_POSE_VARIANT=8): eight waves split the slice's K-steps, so a wave's part is ONE memory round
// trip (two K-steps = 32 x 16-byte loads) instead of two or three; waves 4-7 hand their accumulators to waves 0-3 through the
// same 64 KB of LDS before the four-way sum.  Measured slower (24.9 vs 22.4 us with the epilogue, scripts/pose_ab.py): the
// kernel is not bound by its dependent round trips either; NW = 4 stays the default.
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void pose_l1_split_kernel(
    const float* __restrict__ x, const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
    float* __restrict__ part, int B, int D, int hidden, int steps_per_slice) {
  extern __shared__ __attribute__((aligned(16))) float red[];        // [4 waves][4 cb][4 mb][64 lanes][4] = 64 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, ks = blockIdx.y, m0 = blockIdx.z * 64;
  const int ksteps = D >> 5;
  const int s_begin = ks * steps_per_slice, s_end = min(ksteps, s_begin + steps_per_slice);
  const int ns = max(s_end - s_begin, 0);
  const int kbeg = s_begin + ns * wave / NW, kend = s_begin + ns * (wave + 1) / NW;
  long long wrow[4];
  const float* xp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wrow[i] = (long long)min(n0 + i * 16 + r, hidden - 1) * D + 8 * g;
    xp[i] = x + (long long)min(m0 + i * 16 + r, B - 1) * D + 8 * g;
  }
  f32x4 acc[4][4];     // [cb][mb]
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[cb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const bf16x8 (&wh)[4], const bf16x8 (&wl)[4], const float4 (&xa)[4], const float4 (&xb)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      bf16x8 xh, xl;
      split8(xa[mb], xb[mb], xh, xl);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        f32x4 c = acc[cb][mb];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cb], xl, c, 0, 0, 0);     // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cb], xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cb], xl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cb], xh, c, 0, 0, 0);
        acc[cb][mb] = c;
      }
    }
  };
  int s = kbeg;
  for (; s + 2 <= kend; s += 2) {               // two K-steps per round trip: 32 x 16-byte loads in flight per lane
    bf16x8 wh[2][4], wl[2][4];
    float4 xa[2][4], xb[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh[u][i] = *reinterpret_cast<const bf16x8*>(Whi + wrow[i] + (s + u) * 32);
        wl[u][i] = *reinterpret_cast<const bf16x8*>(Wlo + wrow[i] + (s + u) * 32);
        xa[u][i] = *reinterpret_cast<const float4*>(xp[i] + (s + u) * 32);
        xb[u][i] = *reinterpret_cast<const float4*>(xp[i] + (s + u) * 32 + 4);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) compute(wh[u], wl[u], xa[u], xb[u]);
  }
  for (; s < kend; ++s) {
    bf16x8 wh[4], wl[4];
    float4 xa[4], xb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const bf16x8*>(Whi + wrow[i] + s * 32);
      wl[i] = *reinterpret_cast<const bf16x8*>(Wlo + wrow[i] + s * 32);
      xa[i] = *reinterpret_cast<const float4*>(xp[i] + s * 32);
      xb[i] = *reinterpret_cast<const float4*>(xp[i] + s * 32 + 4);
    }
    compute(wh, wl, xa, xb);
  }
  if constexpr (NW == 8) {                    // waves 4..7 -> LDS -> added by waves 0..3 (fixed order: own, then partner)
    if (wave >= 4) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          *reinterpret_cast<f32x4*>(red + (((((wave - 4) * 4 + cb) * 4 + mb) * 64 + lane) << 2)) = acc[cb][mb];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(red + ((((wave * 4 + cb) * 4 + mb) * 64 + lane) << 2));
          acc[cb][mb][0] += q[0]; acc[cb][mb][1] += q[1]; acc[cb][mb][2] += q[2]; acc[cb][mb][3] += q[3];
        }
    }
    __syncthreads();
  }
  if (wave < 4) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        *reinterpret_cast<f32x4*>(red + ((((wave * 4 + cb) * 4 + mb) * 64 + lane) << 2)) = acc[cb][mb];
  }
  __syncthreads();                            // (every wave of the workgroup takes part in every barrier: none leaves early)
  if (wave >= 4) return;
  // wave w finishes hidden block w; C/D: col = batch row (lane & 15) of block mb, rows 4g+e = 4 consecutive hidden units
  const int n = n0 + wave * 16 + 4 * g;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    f32x4 t = *reinterpret_cast<const f32x4*>(red + ((((0 * 4 + wave) * 4 + mb) * 64 + lane) << 2));
#pragma unroll
    for (int p = 1; p < 4; ++p) {               // fixed order: bitwise reproducible
      const f32x4 q = *reinterpret_cast<const f32x4*>(red + ((((p * 4 + wave) * 4 + mb) * 64 + lane) << 2));
      t[0] += q[0]; t[1] += q[1]; t[2] += q[2]; t[3] += q[3];
    }
    const int m = m0 + mb * 16 + r;
    if (m < B && n < hidden) *reinterpret_cast<f32x4*>(part + ((long long)ks * B + m) * hidden + n) = t;
  }
}

// ---- single-launch form (round 3) -------------------------------------------------------------------------------------
// Same arithmetic as pose_l1_split_kernel + pose_epilogue_kernel, one launch:
//  * W1's (hi, lo) planes in MFMA FRAGMENT order (vpr_pose_head_pack_w1_frag): a wave's weight load is one contiguous
//    1 KB instead of 16 rows x 64 B (16 cache lines per instruction, each line fetched in two halves by different K-steps);
//  * split-K finished by ARRIVAL COUNTERS instead of a second kernel: every workgroup stores its slab, fences, and bumps the
//    counter of its (hidden tile, batch tile); the workgroup that arrives last adds the slabs of that tile IN SLICE ORDER
//    (whoever it is: the sum is the same bits), applies bias + ReLU, multiplies by its 64 columns of W2 and stores a
//    [64 rows][8] second-layer partial; a second counter per batch tile elects the workgroup that adds those partials in
//    TILE ORDER, adds b2, normalises the (sin, cos) pair and writes the rows.  No spinning: a workgroup either finishes
//    the job or exits.  The counters live at the head of the workspace, must be zero before the first call and are left
//    zero by every call.  Bitwise reproducible like the two-launch form (fixed summation orders at both levels).
// four dwords as agent-scope (sc1: write-through) stores: visible to every XCD once acknowledged, no L2 write-back needed
__device__ __forceinline__ void agent_store4(float* p, const f32x4& v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) __hip_atomic_store(p + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PoseFusedArgs {
  const float* x; const uint16_t* Whi; const uint16_t* Wlo; const float* b1; const float* W2; const float* b2;
  float* out; int* cnt; float* part; float* part2;
  int B, D, hidden, n_out, sincos_offset, steps_per_slice, nslice, ntiles;
  int finish;      // 1: arrival counters finish the head in this launch; 0: slabs only (pose_epilogue_tiles_kernel follows)
};

__global__ __launch_bounds__(256, 1) void pose_fused_kernel(PoseFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];        // [4 waves][4 cb][4 mb][64 lanes][4] = 64 KB
  __shared__ int s_flag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int nt = blockIdx.x, ks = blockIdx.y, mt = blockIdx.z;
  const int n0 = nt * 64, m0 = mt * 64;
  const int B = a.B, D = a.D, hidden = a.hidden;
  const int ksteps = D >> 5;
  const int s_begin = ks * a.steps_per_slice, s_end = min(ksteps, s_begin + a.steps_per_slice);
  const int ns = max(s_end - s_begin, 0);
  const int kbeg = s_begin + ns * wave / 4, kend = s_begin + ns * (wave + 1) / 4;
  const int nblocks = hidden >> 4;
  long long wfrag[4];          // fragment base of hidden block cb: ((nb * ksteps) * 64 + lane) * 8
  const float* xp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wfrag[i] = ((long long)min(nt * 4 + i, nblocks - 1) * ksteps * 64 + lane) * 8;
    xp[i] = a.x + (long long)min(m0 + i * 16 + r, B - 1) * D + 8 * g;
  }
  f32x4 acc[4][4];     // [cb][mb]
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[cb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const bf16x8 (&wh)[4], const bf16x8 (&wl)[4], const float4 (&xa)[4], const float4 (&xb)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      bf16x8 xh, xl;
      split8(xa[mb], xb[mb], xh, xl);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        f32x4 c = acc[cb][mb];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cb], xl, c, 0, 0, 0);     // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cb], xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cb], xl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cb], xh, c, 0, 0, 0);
        acc[cb][mb] = c;
      }
    }
  };
  int s = kbeg;
  for (; s + 2 <= kend; s += 2) {               // two K-steps per round trip: 32 x 16-byte loads in flight per lane
    bf16x8 wh[2][4], wl[2][4];
    float4 xa[2][4], xb[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh[u][i] = *reinterpret_cast<const bf16x8*>(a.Whi + wfrag[i] + (long long)(s + u) * 512);
        wl[u][i] = *reinterpret_cast<const bf16x8*>(a.Wlo + wfrag[i] + (long long)(s + u) * 512);
        xa[u][i] = *reinterpret_cast<const float4*>(xp[i] + (s + u) * 32);
        xb[u][i] = *reinterpret_cast<const float4*>(xp[i] + (s + u) * 32 + 4);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) compute(wh[u], wl[u], xa[u], xb[u]);
  }
  for (; s < kend; ++s) {
    bf16x8 wh[4], wl[4];
    float4 xa[4], xb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const bf16x8*>(a.Whi + wfrag[i] + (long long)s * 512);
      wl[i] = *reinterpret_cast<const bf16x8*>(a.Wlo + wfrag[i] + (long long)s * 512);
      xa[i] = *reinterpret_cast<const float4*>(xp[i] + s * 32);
      xb[i] = *reinterpret_cast<const float4*>(xp[i] + s * 32 + 4);
    }
    compute(wh, wl, xa, xb);
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
      *reinterpret_cast<f32x4*>(red + ((((wave * 4 + cb) * 4 + mb) * 64 + lane) << 2)) = acc[cb][mb];
  __syncthreads();
  // wave w finishes hidden block w; C/D: col = batch row (lane & 15) of block mb, rows 4g+e = 4 consecutive hidden units.
  // Slab layout: part[ks][mt][nt][64 rows][64 cols] (tile-contiguous: the finisher reads 16 KB runs)
  float* slab = a.part + (((long long)ks * gridDim.z + mt) * a.ntiles + nt) * 4096;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    f32x4 t = *reinterpret_cast<const f32x4*>(red + ((((0 * 4 + wave) * 4 + mb) * 64 + lane) << 2));
#pragma unroll
    for (int p = 1; p < 4; ++p) {               // fixed order: bitwise reproducible
      const f32x4 q = *reinterpret_cast<const f32x4*>(red + ((((p * 4 + wave) * 4 + mb) * 64 + lane) << 2));
      t[0] += q[0]; t[1] += q[1]; t[2] += q[2]; t[3] += q[3];
    }
    agent_store4(slab + (mb * 16 + r) * 64 + wave * 16 + 4 * g, t);
  }
  if (!a.finish) return;
  // ---- level 1: last of the nslice workgroups of this (hidden tile, batch tile) ----
  // Release without a cache write-back: the slab went out as agent-scope (write-through) stores, so "visible device-wide"
  // is "acknowledged" = vmcnt(0).  A __threadfence() here is buffer_wbl2 — a write-back of the XCD's whole L2, serialised
  // per XCD: measured 1 us per workgroup, 56 us for this kernel instead of 14 (scripts/pose_ab.py, first cut).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* cnt1 = a.cnt + mt * (a.ntiles + 1) + 1 + nt;
  int* cnt2 = a.cnt + mt * (a.ntiles + 1);
  if (threadIdx.x == 0)
    s_flag = (__hip_atomic_fetch_add(cnt1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.nslice - 1);
  __syncthreads();
  if (!s_flag) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // acquire: invalidate (cheap), then read the other workgroups' slabs
  const int row = threadIdx.x >> 2, hq = threadIdx.x & 3;          // 64 rows x 4 column quarters (16 hidden units each)
  float4 hs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) hs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* tile0 = a.part + ((long long)mt * a.ntiles + nt) * 4096 + row * 64 + hq * 16;
  const long long slice_stride = (long long)gridDim.z * a.ntiles * 4096;
  for (int k0 = 0; k0 < a.nslice; k0 += 8) {             // 8 slices (32 float4) requested per round trip, added in slice order
    float4 t[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* p = tile0 + (long long)min(k0 + i, a.nslice - 1) * slice_stride;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[i][j] = reinterpret_cast<const float4*>(p)[j];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (k0 + i < a.nslice) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { hs[j].x += t[i][j].x; hs[j].y += t[i][j].y; hs[j].z += t[i][j].z; hs[j].w += t[i][j].w; }
      }
  }
  float po[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) po[o] = 0.f;
  const int hbase = n0 + hq * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int h = hbase + 4 * j;
    if (h < hidden) {                                    // hidden % 16 == 0: whole float4s
      const float4 bb = *reinterpret_cast<const float4*>(a.b1 + h);
      const float4 v = make_float4(fmaxf(hs[j].x + bb.x, 0.f), fmaxf(hs[j].y + bb.y, 0.f), fmaxf(hs[j].z + bb.z, 0.f), fmaxf(hs[j].w + bb.w, 0.f));
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o < a.n_out) {
          const float4 w = *reinterpret_cast<const float4*>(a.W2 + (long long)o * hidden + h);
          po[o] = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, po[o]))));
        }
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {                          // the four quarters of a row sit in adjacent lanes: fixed-order butterfly
    po[o] += __shfl_xor(po[o], 1, 64);
    po[o] += __shfl_xor(po[o], 2, 64);
  }
  float* p2 = a.part2 + (((long long)mt * a.ntiles + nt) * 64 + row) * 8;
  if (hq == 0) {
#pragma unroll
    for (int o = 0; o < 8; ++o) __hip_atomic_store(p2 + o, po[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- level 2: last of the ntiles level-1 finishers of this batch tile ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(cnt1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next call (nobody touches it again in this one)
    s_flag = (__hip_atomic_fetch_add(cnt2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ntiles - 1);
  }
  __syncthreads();
  if (!s_flag) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (threadIdx.x == 0) __hip_atomic_store(cnt2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* outs = red;                                     // [64 rows][8]
  {
    const int oq = threadIdx.x & 3;                      // outputs 2 oq, 2 oq + 1 of row `row`
    float s0 = 0.f, s1 = 0.f;
    const float* q = a.part2 + ((long long)mt * a.ntiles * 64 + row) * 8 + 2 * oq;
    for (int t = 0; t < a.ntiles; ++t) {                 // tile order: fixed
      const float2 v = *reinterpret_cast<const float2*>(q + (long long)t * 512);
      s0 += v.x; s1 += v.y;
    }
    if (2 * oq < a.n_out) s0 += a.b2[2 * oq];
    if (2 * oq + 1 < a.n_out) s1 += a.b2[2 * oq + 1];
    outs[row * 8 + 2 * oq] = s0;
    outs[row * 8 + 2 * oq + 1] = s1;
  }
  __syncthreads();
  if (threadIdx.x < 64 && m0 + (int)threadIdx.x < B) {
    float* o8 = outs + threadIdx.x * 8;
    const int so = a.sincos_offset;
    if (so >= 0 && so + 1 < a.n_out) {                   // F.normalize(p=2, dim=1, eps=1e-6) of the pair
      const float u = o8[so], v = o8[so + 1];
      const float den = fmaxf(sqrtf(u * u + v * v), 1e-6f);
      o8[so] = u / den;
      o8[so + 1] = v / den;
    }
    float* dst = a.out + (long long)(m0 + threadIdx.x) * a.n_out;
    for (int o = 0; o < a.n_out; ++o) dst[o] = o8[o];
  }
}

// Two-launch companion of pose_fused_kernel(finish = 0): pose_epilogue_kernel on the tile-contiguous slab layout
// part[ks][mt][nt][64 rows][64 cols].  One workgroup per batch row.
__global__ __launch_bounds__(256) void pose_epilogue_tiles_kernel(
    const float* __restrict__ part, int nslice, int mtiles, int ntiles, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out,
    int B, int hidden, int n_out, int sincos_offset) {
  __shared__ float red[8][4];
  __shared__ float outs[8];
  const int b = blockIdx.x;
  const int mt = b >> 6, row = b & 63;
  float po[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) po[o] = 0.f;
  const long long slice_stride = (long long)mtiles * ntiles * 4096;
  for (int h = threadIdx.x * 4; h < hidden; h += 1024) {
    const float* p0 = part + ((long long)mt * ntiles + (h >> 6)) * 4096 + row * 64 + (h & 63);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ks0 = 0; ks0 < nslice; ks0 += 16) {
      float4 t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = *reinterpret_cast<const float4*>(p0 + (long long)min(ks0 + i, nslice - 1) * slice_stride);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (ks0 + i < nslice) { s.x += t[i].x; s.y += t[i].y; s.z += t[i].z; s.w += t[i].w; }
    }
    const float4 bb = *reinterpret_cast<const float4*>(b1 + h);
    s.x = fmaxf(s.x + bb.x, 0.f); s.y = fmaxf(s.y + bb.y, 0.f); s.z = fmaxf(s.z + bb.z, 0.f); s.w = fmaxf(s.w + bb.w, 0.f);
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < n_out) {
        const float4 w = *reinterpret_cast<const float4*>(W2 + (long long)o * hidden + h);
        po[o] = fmaf(s.x, w.x, fmaf(s.y, w.y, fmaf(s.z, w.z, fmaf(s.w, w.w, po[o]))));
      }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o)
    if (o < n_out) {
      const float v = wave_sum(po[o]);
      if ((threadIdx.x & 63) == 0) red[o][threadIdx.x >> 6] = v;
    }
  __syncthreads();
  if ((int)threadIdx.x < n_out) outs[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]) + b2[threadIdx.x];
  __syncthreads();
  finish_outputs(outs, n_out, sincos_offset, out + (long long)b * n_out);
}

// (hi, lo) bf16 planes of W1 [hidden, D] in fragment order: plane[((nb * (D/32) + s) * 64 + lane) * 8 + e] =
// W1[nb*16 + (lane & 15)][s*32 + 8*(lane >> 4) + e]
__global__ __launch_bounds__(256) void pose_pack_frag_kernel(const float* __restrict__ w, int hidden, int D,
                                                             uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  const long long total = (long long)hidden * D / 8;     // 16-byte pieces
  const int ksteps = D >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long q = i >> 6;
    const int s = (int)(q % ksteps), nb = (int)(q / ksteps);
    const float* src = w + (long long)(nb * 16 + (lane & 15)) * D + s * 32 + 8 * (lane >> 4);
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint16_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h[e] = f32_to_bf16_bits(v[e]);
      l[e] = f32_to_bf16_bits(v[e] - bf16_bits_to_f32(h[e]));
    }
    uint4 ho, lo4;
    ho.x = h[0] | ((uint32_t)h[1] << 16); ho.y = h[2] | ((uint32_t)h[3] << 16); ho.z = h[4] | ((uint32_t)h[5] << 16); ho.w = h[6] | ((uint32_t)h[7] << 16);
    lo4.x = l[0] | ((uint32_t)l[1] << 16); lo4.y = l[2] | ((uint32_t)l[3] << 16); lo4.z = l[4] | ((uint32_t)l[5] << 16); lo4.w = l[6] | ((uint32_t)l[7] << 16);
    *reinterpret_cast<uint4*>(hi + i * 8) = ho;
    *reinterpret_cast<uint4*>(lo + i * 8) = lo4;
  }
}

struct PoseFusedPlan { int ks, ntiles, mtiles, sps; size_t off_part, off_part2, total; };
static bool pose_fused_plan(int B, int D, int hidden, PoseFusedPlan* p);

static int pick_slices_split(int B, int D, int hidden) {
  // about one workgroup per CU; a slice at least 8 K-steps (two per wave)
  const int tiles = ((hidden + 63) / 64) * ((B + 63) / 64);
  int ks = (256 + tiles - 1) / tiles;
  const int max_ks = (D / 32) / 8 > 0 ? (D / 32) / 8 : 1;
  if (ks > max_ks) ks = max_ks;
  if (ks > 32) ks = 32;
  if (ks < 1) ks = 1;
  const int e = tune_or(TUNE_POSE_KS, 0);         // A/B switch (slices per tile), clamped to what the workspace query allows
  if (e >= 1 && e <= 64 && e <= (D / 32)) ks = e;
  return ks;
}

static int pick_slices(int B, int D, int hidden) {
  // aim at ~512 workgroups; each slice at least 8 MFMA k-steps (128 of D)
  const int tiles = (hidden / PH_HT) * ((B + PH_BT - 1) / PH_BT);
  int ks = (512 + tiles - 1) / tiles;
  const int nsteps = D / 16;
  const int max_ks = nsteps / 8 > 0 ? nsteps / 8 : 1;
  if (ks > max_ks) ks = max_ks;
  if (ks > 64) ks = 64;
  if (ks < 1) ks = 1;
  return ks;
}

static bool pose_fused_plan(int B, int D, int hidden, PoseFusedPlan* p) {
  if (B <= 0 || D <= 0 || hidden <= 0 || (D % 32) || (hidden % 16)) return false;
  p->ntiles = (hidden + 63) / 64;
  p->mtiles = (B + 63) / 64;
  p->ks = pick_slices_split(B, D, hidden);
  const int ksteps = D / 32;
  p->sps = (ksteps + p->ks - 1) / p->ks;
  // arrival counters (zero between calls): a FIXED 4 KB at the head of the workspace, whatever the shape — calls of different
  // shapes may share one workspace, and a counter area that grew with the shape would overlap a smaller shape's slabs
  if ((size_t)(p->ntiles + 1) * p->mtiles * sizeof(int) > 4096) return false;
  size_t off = 4096;
  p->off_part = off;  off += align_up((size_t)p->ks * p->mtiles * p->ntiles * 4096 * sizeof(float), 256);
  p->off_part2 = off; off += align_up((size_t)p->mtiles * p->ntiles * 64 * 8 * sizeof(float), 256);
  p->total = off;
  return true;
}

}  // namespace vpr

using namespace vpr;

extern "C" size_t vpr_pose_head_fused_workspace_bytes(int B, int D, int hidden) {
  PoseFusedPlan p;
  return pose_fused_plan(B, D, hidden, &p) ? p.total : 0;
}

extern "C" size_t vpr_pose_head_fused_counter_bytes(int B, int D, int hidden) {
  PoseFusedPlan p;
  return pose_fused_plan(B, D, hidden, &p) ? p.off_part : 0;
}

extern "C" int vpr_pose_head_pack_w1_frag(const float* W1, int hidden, int D, uint16_t* hi, uint16_t* lo, void* stream) {
  if (!W1 || !hi || !lo || hidden <= 0 || D <= 0) return VPR_ERR_INVALID_ARG;
  if ((hidden % 16) || (D % 32)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15) return VPR_ERR_UNSUPPORTED;
  const long long total = (long long)hidden * D / 8;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  VPR_TRY_LAUNCH(launch_kernel(pose_pack_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), W1,
                               hidden, D, hi, lo));
  return VPR_OK;
}

extern "C" int vpr_pose_head_fused(const float* x, const uint16_t* W1_hi_frag, const uint16_t* W1_lo_frag, const float* b1,
                                   const float* W2, const float* b2, float* out, int B, int D, int hidden,
                                   int n_out, int sincos_offset, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !W1_hi_frag || !W1_lo_frag || !b1 || !W2 || !b2 || !out || !workspace || B <= 0 || D <= 0 || hidden <= 0 || n_out < 1)
    return VPR_ERR_INVALID_ARG;
  if (n_out > 8) return VPR_ERR_UNSUPPORTED;
  PoseFusedPlan p;
  if (!pose_fused_plan(B, D, hidden, &p)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W1_hi_frag) | reinterpret_cast<uintptr_t>(W1_lo_frag) |
       reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(W2)) & 15)
    return VPR_ERR_UNSUPPORTED;
  if (workspace_bytes < p.total) return VPR_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  // VPR_POSE_VARIANT (A/B): 0 / unset = measured default below, 1 = one launch (arrival counters), 2 = two launches
  const int variant = tune_or(TUNE_POSE_VARIANT, 0);
  const int finish = variant == 1 ? 1 : 0;
  PoseFusedArgs a{x, W1_hi_frag, W1_lo_frag, b1, W2, b2, out, reinterpret_cast<int*>(ws), reinterpret_cast<float*>(ws + p.off_part),
                  reinterpret_cast<float*>(ws + p.off_part2), B, D, hidden, n_out, sincos_offset, p.sps, p.ks, p.ntiles, finish};
  constexpr size_t lds = 4 * 4 * 4 * 64 * 4 * sizeof(float);   // 64 KB
  static PerDeviceFlag attr = {};
  VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(pose_fused_kernel), lds, attr));
  VPR_TRY_LAUNCH(launch_kernel(pose_fused_kernel, dim3(p.ntiles, p.ks, p.mtiles), dim3(256), lds, stream, a));
  if (!finish)
    VPR_TRY_LAUNCH(launch_kernel(pose_epilogue_tiles_kernel, dim3(B), dim3(256), 0, stream, a.part, p.ks, p.mtiles, p.ntiles, b1, W2, b2,
                                 out, B, hidden, n_out, sincos_offset));
  return VPR_OK;
}

extern "C" size_t vpr_pose_head_workspace_bytes(int B, int D, int hidden, int n_out) {
  (void)n_out;
  if (B <= 0 || D <= 0 || hidden < 0) return 0;
  if (hidden == 0) return 256;
  return align_up((size_t)pick_slices(B, D, hidden) * B * hidden * sizeof(float), 256);
}

extern "C" int vpr_pose_head(const float* x, const float* W1, const float* b1, const float* W2,
                             const float* b2, float* out, int B, int D, int hidden, int n_out,
                             int sincos_offset, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !W2 || !b2 || !out || B <= 0 || D <= 0 || hidden < 0 || n_out < 1) return VPR_ERR_INVALID_ARG;
  if (n_out > 8) return VPR_ERR_UNSUPPORTED;
  if (hidden == 0) {
    VPR_TRY_LAUNCH(launch_kernel(pose_linear_kernel, dim3(B), dim3(256), 0, stream, x, W2, b2, out, D, n_out, sincos_offset));
    return VPR_OK;
  }
  if (!W1 || !b1 || !workspace) return VPR_ERR_INVALID_ARG;
  if ((D % 16) || (hidden % PH_HT)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W1)) & 15) return VPR_ERR_UNSUPPORTED;
  const int ks = pick_slices(B, D, hidden);
  if (workspace_bytes < (size_t)ks * B * hidden * sizeof(float)) return VPR_ERR_WORKSPACE;
  const int nsteps = D / 16;
  const int sps = (nsteps + ks - 1) / ks;
  float* part = static_cast<float*>(workspace);
  VPR_TRY_LAUNCH(launch_kernel(pose_l1_partial_kernel, dim3(hidden / PH_HT, ks, (B + PH_BT - 1) / PH_BT), dim3(256), 0,
                     stream, x, W1, part, B, D, hidden, sps));
  VPR_TRY_LAUNCH(launch_kernel(pose_epilogue_kernel, dim3(B), dim3(256), 0, stream, part, ks, b1, W2, b2, out, B,
                     hidden, n_out, sincos_offset));
  return VPR_OK;
}

template <int VPL>
static int launch_ln(const void* x, int bf16, int B, int T, const float* gamma, const float* beta, float eps,
                      float* pooled, const float* Wh, const float* bh, int n_out, int so, float* out,
                      hipStream_t stream) {
  if (bf16)
    VPR_TRY_LAUNCH(launch_kernel((ln_meanpool_head_kernel<VPL, true>), dim3(B), dim3(1024), 0, stream, x, T, gamma, beta,
                       eps, pooled, Wh, bh, n_out, so, out));
  else
    VPR_TRY_LAUNCH(launch_kernel((ln_meanpool_head_kernel<VPL, false>), dim3(B), dim3(1024), 0, stream, x, T, gamma, beta,
                       eps, pooled, Wh, bh, n_out, so, out));
  return VPR_OK;
}

extern "C" int vpr_ln_meanpool_head(const void* x, int x_is_bf16, int B, int T, int H, const float* gamma,
                                    const float* beta, float eps, float* pooled_out, const float* Wh,
                                    const float* bh, int n_out, int sincos_offset, float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !gamma || !beta || B <= 0 || T <= 0 || n_out < 0) return VPR_ERR_INVALID_ARG;
  if (n_out > 8) return VPR_ERR_UNSUPPORTED;
  if (n_out > 0 && Wh && (!bh || !out)) return VPR_ERR_INVALID_ARG;
  if (!pooled_out && !(Wh && n_out > 0)) return VPR_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(x) & 15) return VPR_ERR_UNSUPPORTED;
  switch (H) {
    case 512: return launch_ln<8>(x, x_is_bf16, B, T, gamma, beta, eps, pooled_out, Wh, bh, n_out, sincos_offset, out, stream);
    case 768: return launch_ln<12>(x, x_is_bf16, B, T, gamma, beta, eps, pooled_out, Wh, bh, n_out, sincos_offset, out, stream);
    case 1024: return launch_ln<16>(x, x_is_bf16, B, T, gamma, beta, eps, pooled_out, Wh, bh, n_out, sincos_offset, out, stream);
    case 1536: return launch_ln<24>(x, x_is_bf16, B, T, gamma, beta, eps, pooled_out, Wh, bh, n_out, sincos_offset, out, stream);
    default: return VPR_ERR_UNSUPPORTED;
  }
  return VPR_OK;
}

/* W1 f32 [count] -> (hi, lo) bf16 planes with hi + lo == W1 to 2^-17 relative. */
extern "C" int vpr_pose_head_pack_w1(const float* W1, long long count, uint16_t* hi, uint16_t* lo, void* stream) {
  if (!W1 || !hi || !lo || count < 0) return VPR_ERR_INVALID_ARG;
  if (count == 0) return VPR_OK;
  if ((count + 255) / 256 > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  VPR_TRY_LAUNCH(launch_kernel(pose_pack_split_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                               static_cast<hipStream_t>(stream), W1, count, hi, lo));
  return VPR_OK;
}

extern "C" size_t vpr_pose_head_split_workspace_bytes(int B, int D, int hidden) {
  if (B <= 0 || D <= 0 || hidden <= 0) return 0;
  return align_up((size_t)pick_slices_split(B, D, hidden) * B * hidden * sizeof(float), 256);
}

extern "C" int vpr_pose_head_split(const float* x, const uint16_t* W1_hi, const uint16_t* W1_lo, const float* b1,
                                   const float* W2, const float* b2, float* out, int B, int D, int hidden,
                                   int n_out, int sincos_offset, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x || !W1_hi || !W1_lo || !b1 || !W2 || !b2 || !out || !workspace || B <= 0 || D <= 0 || hidden <= 0 || n_out < 1)
    return VPR_ERR_INVALID_ARG;
  if (n_out > 8 || (D % 32) || (hidden % 16)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W1_hi) | reinterpret_cast<uintptr_t>(W1_lo) |
       reinterpret_cast<uintptr_t>(workspace)) & 15)
    return VPR_ERR_UNSUPPORTED;
  const int ks = pick_slices_split(B, D, hidden);
  if (workspace_bytes < (size_t)ks * B * hidden * sizeof(float)) return VPR_ERR_WORKSPACE;
  const int ksteps = D / 32;
  const int sps = (ksteps + ks - 1) / ks;
  float* part = static_cast<float*>(workspace);
  constexpr size_t l1_lds = 4 * 4 * 4 * 64 * 4 * sizeof(float);   // 64 KB
  if (tune_or(TUNE_POSE_VARIANT, 0) != 8) {        // default: four waves per workgroup (8 waves, one round trip each: 24.9 vs 22.4 us)
    static PerDeviceFlag attr = {};
    VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(pose_l1_split_kernel<4>), l1_lds, attr));
    VPR_TRY_LAUNCH(launch_kernel(pose_l1_split_kernel<4>, dim3((hidden + 63) / 64, ks, (B + 63) / 64), dim3(256), l1_lds, stream, x,
                                 W1_hi, W1_lo, part, B, D, hidden, sps));
  } else {
    static PerDeviceFlag attr = {};
    VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(pose_l1_split_kernel<8>), l1_lds, attr));
    VPR_TRY_LAUNCH(launch_kernel(pose_l1_split_kernel<8>, dim3((hidden + 63) / 64, ks, (B + 63) / 64), dim3(512), l1_lds, stream, x,
                                 W1_hi, W1_lo, part, B, D, hidden, sps));
  }
  VPR_TRY_LAUNCH(launch_kernel(pose_epilogue_kernel, dim3(B), dim3(256), 0, stream, part, ks, b1, W2, b2, out, B,
                               hidden, n_out, sincos_offset));
  return VPR_OK;
}
