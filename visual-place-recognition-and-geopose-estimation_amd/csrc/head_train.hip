// head_train.hip — one optimizer step of the two-layer regression head on cached descriptors.
//
// vpr_head_train_step   forward, MSE (or Huber) loss, backward and AdamW update of Linear(D,hidden) -> ReLU -> Linear(hidden,n_out)
//   replaces, per batch, dinov2salad/dinov2salad_finetuning.py:119-125 (preds = model(inputs); loss = MSELoss;
//   zero_grad; backward; optimizer.step) with optimizer = torch.optim.AdamW(lr=1e-5) (:95) — for a FROZEN extractor whose
//   descriptors were computed once (the reference re-runs the backbone every epoch; see finetune.py).
//
// Everything is f32, as in the reference, and every sum has a fixed order: the step is bitwise reproducible.
// Three launches, no host synchronisation, no atomics:
//   1. head_fwd_partial_kernel   split-K partial sums of X[idx] W1^T (exact-f32 MFMA, the weight matrix streamed once)
//   2. head_mid_kernel           slabs -> z -> h = relu(z + b1); partial outputs per 16 hidden units; snapshot of (W2, b2)
//   3. head_update_kernel        outputs, dO, dz; gradient tile dz^T X formed in registers and consumed by the AdamW update
//                                of W1 in the same pass (the 17 MB gradient never exists in memory); b1 / W2 / b2 / loss
//                                by designated workgroups
// HBM-bound: algorithmic bytes per step = hidden*D*4 (forward read of W1) + 6*hidden*D*4 (W1, m, v read and written)
// + 2*B*D*4 (the batch rows, forward and backward) — 121.7 MB at D = 8448, hidden = 512, B = 16.
#include <math.h>
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int HT_HT = 32;     // hidden units per forward workgroup (2 MFMA column blocks)
constexpr int HT_BT = 16;     // batch rows per forward workgroup (one MFMA row block)
constexpr int HT_CH = 6;      // k-steps a wave requests before their MFMAs (3 x 16 B per lane each)
constexpr int HT_H2 = 16;     // hidden units per workgroup of the middle kernel
constexpr int HT_JT = 1024;   // W1 columns per update workgroup (256 threads x float4)
constexpr int HT_MAXB = 64;   // largest batch (LDS tables of the update kernel)
constexpr int HT_MAXO = 8;    // largest n_out

// 1. part[ks][b][h] = sum over K-slice ks of x[idx[b]][k] * W1[h][k].  Workgroup = 32 hidden units x one slice x
//   WIDE = false (B <= 16): 16 batch rows; the four waves take the slice's k-steps round robin and their accumulators are
//                           added in wave order;
//   WIDE = true  (B > 16):  64 batch rows, wave w = rows 16w .. 16w+15 over the whole slice (the four waves request the same
//                           weight fragments: one trip to L2, three L1 hits) — W1 crosses the L2 once per 64 rows.
template <bool WIDE>
__global__ __launch_bounds__(256) void head_fwd_partial_kernel(
    const float* __restrict__ X, long long x_stride, const int* __restrict__ idx, const float* __restrict__ W1,
    float* __restrict__ part, int B, int D, int hidden, int steps_per_slice, int* counter) {
  __shared__ float red[WIDE ? 1 : 4][HT_BT * HT_HT];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *counter = 0;   // head_mid_kernel's arrival counter
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h0 = blockIdx.x * HT_HT, ks = blockIdx.y;
  const int b0 = WIDE ? blockIdx.z * 4 * HT_BT + wave * HT_BT : blockIdx.z * HT_BT;
  const int nsteps = D >> 4;
  const int s_begin = ks * steps_per_slice;
  const int s_end = min(nsteps, s_begin + steps_per_slice);
  // operand maps of v_mfma_f32_16x16x4_f32: A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15]; a lane loads 4
  // consecutive k (one float4) and feeds element t to MFMA t — A and B use the same k permutation, so it cancels.
  const int r = lane & 15, kg = lane >> 4;
  const int brow = min(b0 + r, B - 1);
  const long long row = idx ? (long long)idx[brow] : (long long)brow;
  const float4* xa = reinterpret_cast<const float4*>(X + row * x_stride) + kg;
  const float4* wa = reinterpret_cast<const float4*>(W1 + (long long)(h0 + r) * D) + kg;
  const float4* wb = reinterpret_cast<const float4*>(W1 + (long long)(h0 + 16 + r) * D) + kg;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int STRIDE = WIDE ? 1 : 4;                       // k-steps between two steps of one wave
  const int first = WIDE ? 0 : wave;
  for (int base = s_begin; base < s_end; base += STRIDE * HT_CH) {     // uniform per workgroup
    float4 a[HT_CH], w0[HT_CH], w1[HT_CH];
#pragma unroll
    for (int i = 0; i < HT_CH; ++i) {
      const int s = min(base + first + STRIDE * i, s_end - 1);
      a[i] = xa[s * 4];
      w0[i] = wa[s * 4];
      w1[i] = wb[s * 4];
    }
#pragma unroll
    for (int i = 0; i < HT_CH; ++i) {
      if (base + first + STRIDE * i < s_end) {   // wave-uniform
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
  if constexpr (WIDE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int b = b0 + 4 * kg + e;
      if (b < B) {
        float* p = part + ((long long)ks * B + b) * hidden + h0;
        p[r] = acc0[e];
        p[16 + r] = acc1[e];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][(4 * kg + e) * HT_HT + r] = acc0[e];
      red[wave][(4 * kg + e) * HT_HT + 16 + r] = acc1[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < HT_BT * HT_HT; t += 256) {
      const int b = b0 + (t >> 5);
      if (b < B) part[((long long)ks * B + b) * hidden + h0 + (t & 31)] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
  }
}

// 2. Workgroup = 16 hidden units x all batch rows.  z = slabs in slice order + b1, h = relu(z) -> H; the workgroup's share of
// every output, po[wg][b][o] = sum over its 16 units of h * W2[o][unit] (DPP row sum: fixed tree); snap = W2 as it is BEFORE
// this step's update (the update kernel reads the copy while designated workgroups rewrite the original).
// The workgroup that arrives LAST at the counter (no spinning: the others have exited) adds the partial outputs in workgroup
// order, + b2, and leaves diff[b][o] = output - target (Huber: half the clipped residual) and the batch loss: the update kernel starts from diff instead of
// re-deriving the outputs in every workgroup (that serial chain of nwg2 loads cost 7 us of a 43 us step).
// Hand-over: the partials go out as agent-scope (write-through) stores, vmcnt(0) = acknowledged, relaxed agent-scope
// ticket, acquire fence in the last workgroup (the protocol of pose_fused_kernel).  The counter is zeroed by the forward
// kernel of the same step (stream order) and left alone otherwise: the workspace needs no initialisation.
__global__ __launch_bounds__(256) void head_mid_kernel(
    const float* __restrict__ part, int nslice, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ Y, long long y_stride, const int* __restrict__ idx,
    float* __restrict__ H, float* po, float* __restrict__ snap, float* __restrict__ diff, float* __restrict__ loss_out,
    int* counter, int B, int hidden, int n_out, float huber_delta) {
  __shared__ float s_red[256];
  __shared__ int s_last;
  const int h0 = blockIdx.x * HT_H2, hl = threadIdx.x & 15, h = h0 + hl;
  const float bias = b1[h];
  float w2[HT_MAXO];
#pragma unroll
  for (int o = 0; o < HT_MAXO; ++o) w2[o] = o < n_out ? W2[(long long)o * hidden + h] : 0.f;
  const int bpad = (B + 15) & ~15;
  for (int b = threadIdx.x >> 4; b < bpad; b += 16) {      // every lane of a wave runs the same number of rounds
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;
    float z = 0.f;
    for (int s0 = 0; s0 < nslice; s0 += 32) {             // up to 32 slabs per round trip, added in slice order
      float t[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) t[i] = part[((long long)min(s0 + i, nslice - 1) * B + bb) * hidden + h];
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if (s0 + i < nslice) z += t[i];
    }
    const float hv = valid ? fmaxf(z + bias, 0.f) : 0.f;
    if (valid) H[(long long)b * hidden + h] = hv;
#pragma unroll
    for (int o = 0; o < HT_MAXO; ++o)
      if (o < n_out) {       // uniform
        const float v = row16_sum(hv * w2[o]);
        if (hl == 0 && valid)
          __hip_atomic_store(po + ((long long)blockIdx.x * B + b) * n_out + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
  for (int t = threadIdx.x; t < HT_H2 * n_out; t += 256) {
    const int o = t / HT_H2, hh = h0 + t % HT_H2;
    snap[(long long)o * hidden + hh] = W2[(long long)o * hidden + hh];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = (__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // outputs: P threads per output take the workgroups' partials w = part, part + P, ... (ascending), then the P sums are
  // added in order — P depends on the shape only, so the result is reproducible
  const int nwg = gridDim.x, BO = B * n_out;
  int P = 1;
  while (P * 2 * BO <= 256 && P * 2 <= nwg) P *= 2;
  float sq = 0.f;                                        // this thread's share of sum diff^2 (threads < BO, strided)
  for (int base = 0; base < BO; base += 256 / P) {       // uniform
    const int t = base + (int)threadIdx.x % (256 / P), prt = threadIdx.x / (256 / P);
    float acc = 0.f;
    if (t < BO && prt < P) {
      for (int w0 = prt; w0 < nwg; w0 += 8 * P) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = po[(long long)min(w0 + i * P, nwg - 1) * BO + t];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (w0 + i * P < nwg) acc += v[i];
      }
    }
    __syncthreads();
    s_red[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < 256 / P && t < BO) {
      float o = 0.f;
      for (int q = 0; q < P; ++q) o += s_red[q * (256 / P) + threadIdx.x];
      const int b = t / n_out, oo = t - b * n_out;
      const long long row = idx ? (long long)idx[b] : (long long)b;
      const float d = (o + b2[oo]) - Y[row * y_stride + oo];
      if (huber_delta > 0.f) {       // nn.HuberLoss: 0.5 d^2 inside [-delta, delta], delta (|d| - 0.5 delta) outside; gradient d or +-delta.
        const float ad = fabsf(d);   // `diff` carries HALF the gradient numerator: the update kernel scales by 2 / (B n_out) as for MSE
        diff[t] = 0.5f * (ad < huber_delta ? d : copysignf(huber_delta, d));
        sq += ad < huber_delta ? 0.5f * d * d : huber_delta * (ad - 0.5f * huber_delta);
      } else {
        diff[t] = d;
        sq = fmaf(d, d, sq);
      }
    }
  }
  __syncthreads();
  s_red[threadIdx.x] = sq;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (loss_out) {
      float s = 0.f;
      const int nq = min(256 / P, BO);
      for (int q = 0; q < nq; ++q) s += s_red[q];
      *loss_out = s / (float)BO;
    }
  }
}

// AdamW, torch.optim's single-tensor update in its order of operations (the scalars are formed in double on the host,
// as Python forms them, and rounded to f32 once).
struct AdamConsts { float decay, one_minus_b1, beta2, one_minus_b2, sqrt_bc2, eps, step_size; };
__device__ __forceinline__ void adamw(float& p, float& m, float& v, float g, const AdamConsts& c) {
  p = p * c.decay;                                   // param.mul_(1 - lr * weight_decay)
  m = m + (g - m) * c.one_minus_b1;                  // exp_avg.lerp_(grad, 1 - beta1)
  v = v * c.beta2 + c.one_minus_b2 * g * g;          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / c.sqrt_bc2 + c.eps; // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  p = p - c.step_size * (m / denom);                 // param.addcdiv_(exp_avg, denom, value=-step_size)
}

struct HeadTrainArgs {
  const float* X; long long x_stride; const int* idx; const float* Y; long long y_stride;
  float *W1, *b1, *W2, *b2;
  float *m, *v;                         // moments, [W1 | b1 | W2 | b2]
  const float *H, *diff, *snap;         // from head_mid_kernel: h, output - target, W2 before the update
  int B, D, hidden, n_out;
  AdamConsts c;
  int abl;                              // timing-only build (-DVPR_ABLATION): which part of the update kernel is left out
};
#ifdef VPR_ABLATION
#define HT_ABL(a, x) ((a).abl == (x))
#else
#define HT_ABL(a, x) false
#endif

// 3. Workgroup = IT rows x 1024 columns of W1.  Every workgroup re-derives the outputs (a few hundred FMAs), dO = 2 (o - y) /
// (B n_out) and dz = (dO W2) * (h > 0) for its IT hidden units; a thread owns 4 columns: g[i] = sum_b dz[b][i] * x[b][cols]
// (b ascending, BC batch rows in registers at a time), then the AdamW update of its IT x 4 weights, RG rows of (W1, m, v)
// requested together.  Column-tile 0 also updates b1 and W2 for its IT units, workgroup (0,0) b2 and the loss.
// The kernel is a stream of 6 * hidden * D * 4 bytes: small IT = more workgroups per CU taking turns at load / update / store.
template <bool NT>
__device__ __forceinline__ void store4(float* p, const float4& v) {
  if constexpr (NT) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

template <int IT, int BC, int RG, bool NT = false>
__global__ __launch_bounds__(256) void head_update_kernel(HeadTrainArgs a) {
  __shared__ float s_diff[HT_MAXB * HT_MAXO];
  __shared__ __attribute__((aligned(16))) float s_dz[HT_MAXB * IT];
  __shared__ float s_h[HT_MAXB * IT];
  const int B = a.B, n_out = a.n_out, hidden = a.hidden, D = a.D;
  const int i0 = blockIdx.y * IT, j = blockIdx.x * HT_JT + threadIdx.x * 4;
  const float gscale = 2.0f / (float)(B * n_out);
  // the stream first: the first RG rows of (W1, m, v) and the first BC batch rows are requested BEFORE the dependent
  // chain of the prologue below (partials -> outputs -> dz: three round trips), so they arrive while it runs
  const bool active = j < D;
  float4 p4[RG], m4[RG], v4[RG], x4[BC];
  if (active) {
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const long long off = (long long)(i0 + i) * D + j;
      p4[i] = *reinterpret_cast<const float4*>(a.W1 + off);
      m4[i] = *reinterpret_cast<const float4*>(a.m + off);
      v4[i] = *reinterpret_cast<const float4*>(a.v + off);
    }
#pragma unroll
    for (int u = 0; u < BC; ++u) {
      const int bb = min(u, B - 1);
      const long long row = a.idx ? (long long)a.idx[bb] : (long long)bb;
      x4[u] = *reinterpret_cast<const float4*>(a.X + row * a.x_stride + j);
    }
  }
  if (!HT_ABL(a, 13)) {
  for (int t = threadIdx.x; t < B * n_out; t += 256) s_diff[t] = a.diff[t];
  __syncthreads();
  for (int t = threadIdx.x; t < B * IT; t += 256) {
    const int b = t / IT, i = t - b * IT;
    const float hv = a.H[(long long)b * hidden + i0 + i];
    float g = 0.f;
    for (int o = 0; o < n_out; ++o) g += (s_diff[b * n_out + o] * gscale) * a.snap[(long long)o * hidden + i0 + i];
    s_dz[t] = hv > 0.f ? g : 0.f;
    s_h[t] = hv;
  }
  __syncthreads();
  }
  if (active) {
    float4 g[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int bc = 0; bc < (HT_ABL(a, 12) ? 0 : B); bc += BC) {
      if (bc > 0) {
#pragma unroll
        for (int u = 0; u < BC; ++u) {
          const int bb = min(bc + u, B - 1);
          const long long row = a.idx ? (long long)a.idx[bb] : (long long)bb;
          x4[u] = *reinterpret_cast<const float4*>(a.X + row * a.x_stride + j);
        }
      }
#pragma unroll
      for (int u = 0; u < BC; ++u) {
        if (bc + u < B) {      // uniform
          const float4* dzr = reinterpret_cast<const float4*>(s_dz + (bc + u) * IT);
#pragma unroll
          for (int q = 0; q < IT / 4; ++q) {
            const float4 d = dzr[q];
            const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float4& gg = g[4 * q + e];
              gg.x = fmaf(dd[e], x4[u].x, gg.x); gg.y = fmaf(dd[e], x4[u].y, gg.y);
              gg.z = fmaf(dd[e], x4[u].z, gg.z); gg.w = fmaf(dd[e], x4[u].w, gg.w);
            }
          }
        }
      }
    }
#pragma unroll
    for (int grp = 0; grp < IT / RG; ++grp) {
      if (grp > 0) {
#pragma unroll
        for (int i = 0; i < RG; ++i) {
          const long long off = (long long)(i0 + grp * RG + i) * D + j;
          p4[i] = *reinterpret_cast<const float4*>(a.W1 + off);
          m4[i] = *reinterpret_cast<const float4*>(a.m + off);
          v4[i] = *reinterpret_cast<const float4*>(a.v + off);
        }
      }
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const long long off = (long long)(i0 + grp * RG + i) * D + j;
        const float4 gg = g[grp * RG + i];
        if (!HT_ABL(a, 11)) {
          adamw(p4[i].x, m4[i].x, v4[i].x, gg.x, a.c);
          adamw(p4[i].y, m4[i].y, v4[i].y, gg.y, a.c);
          adamw(p4[i].z, m4[i].z, v4[i].z, gg.z, a.c);
          adamw(p4[i].w, m4[i].w, v4[i].w, gg.w, a.c);
        } else {
          p4[i].x += gg.x; p4[i].y += gg.y; p4[i].z += gg.z; p4[i].w += gg.w;
        }
        if (!HT_ABL(a, 10) || p4[i].x == 123.456f) {
          store4<NT>(a.W1 + off, p4[i]);
          store4<NT>(a.m + off, m4[i]);
          store4<NT>(a.v + off, v4[i]);
        }
      }
    }
  }
  if (blockIdx.x == 0) {
    const long long off_b1 = (long long)hidden * D, off_w2 = off_b1 + hidden, off_b2 = off_w2 + (long long)n_out * hidden;
    for (int t = threadIdx.x; t < IT * (1 + n_out); t += 256) {
      const int which = t / IT, i = t - which * IT;
      float g = 0.f;
      if (which == 0) {
        for (int b = 0; b < B; ++b) g += s_dz[b * IT + i];
        const long long s = off_b1 + i0 + i;
        adamw(a.b1[i0 + i], a.m[s], a.v[s], g, a.c);
      } else {
        const int o = which - 1;
        for (int b = 0; b < B; ++b) g = fmaf(s_diff[b * n_out + o] * gscale, s_h[b * IT + i], g);
        const long long e = (long long)o * hidden + i0 + i;
        adamw(a.W2[e], a.m[off_w2 + e], a.v[off_w2 + e], g, a.c);
      }
    }
    if (blockIdx.y == 0) {
      if ((int)threadIdx.x < n_out) {
        float g = 0.f;
        for (int b = 0; b < B; ++b) g += s_diff[b * n_out + threadIdx.x] * gscale;
        adamw(a.b2[threadIdx.x], a.m[off_b2 + threadIdx.x], a.v[off_b2 + threadIdx.x], g, a.c);
      }
    }
  }
}

template <int IT, int BC, int RG, bool NT = false>
static int launch_head_update(const HeadTrainArgs& a, hipStream_t stream) {
  return launch_kernel(head_update_kernel<IT, BC, RG, NT>, dim3((a.D + HT_JT - 1) / HT_JT, a.hidden / IT), dim3(256), 0, stream, a);
}

static int head_train_slices(int B, int D, int hidden) {
  // about two forward workgroups per CU; a slice holds at least 4 k-steps (one per wave)
  const int bt = B > HT_BT ? 4 * HT_BT : HT_BT;            // batch rows per forward workgroup
  const int tiles = (hidden / HT_HT) * ((B + bt - 1) / bt);
  int ks = (512 + tiles - 1) / tiles;
  const int nsteps = D / 16;
  const int max_ks = nsteps / 4 > 0 ? nsteps / 4 : 1;
  if (ks > max_ks) ks = max_ks;
  if (ks > 64) ks = 64;
  if (ks < 1) ks = 1;
  return ks;
}

struct HeadTrainPlan { int ks, nwg2; size_t off_h, off_po, off_snap, off_diff, off_cnt, total; };
static bool head_train_plan(int B, int D, int hidden, int n_out, HeadTrainPlan* p) {
  if (B < 1 || B > HT_MAXB || D < 16 || (D % 16) || hidden < HT_HT || (hidden % HT_HT) || n_out < 1 || n_out > HT_MAXO) return false;
  p->ks = head_train_slices(B, D, hidden);
  p->nwg2 = hidden / HT_H2;
  size_t off = align_up((size_t)p->ks * B * hidden * sizeof(float), 256);
  p->off_h = off;    off += align_up((size_t)B * hidden * sizeof(float), 256);
  p->off_po = off;   off += align_up((size_t)p->nwg2 * B * n_out * sizeof(float), 256);
  p->off_snap = off; off += align_up((size_t)n_out * hidden * sizeof(float), 256);
  p->off_diff = off; off += align_up((size_t)B * n_out * sizeof(float), 256);
  p->off_cnt = off;  off += 256;
  p->total = off;
  return true;
}

}  // namespace vpr

using namespace vpr;

extern "C" size_t vpr_head_train_workspace_bytes(int B, int D, int hidden, int n_out) {
  HeadTrainPlan p;
  return head_train_plan(B, D, hidden, n_out, &p) ? p.total : 0;
}

extern "C" long long vpr_head_train_state_floats(int D, int hidden, int n_out) {
  if (D < 1 || hidden < 1 || n_out < 1) return 0;
  return (long long)hidden * D + hidden + (long long)n_out * hidden + n_out;
}

static int head_train_step_impl(const float* X, long long x_stride, const int* idx, const float* Y, long long y_stride,
                                int B, int D, int hidden, int n_out, float* W1, float* b1, float* W2, float* b2,
                                float* m, float* v, int step, double lr, double beta1, double beta2, double eps,
                                double weight_decay, int loss_kind, double huber_delta, float* loss_out, void* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
  if (!X || !Y || !W1 || !b1 || !W2 || !b2 || !m || !v || !workspace || step < 1) return VPR_ERR_INVALID_ARG;
  if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0))
    return VPR_ERR_INVALID_ARG;
  if ((loss_kind != VPR_LOSS_MSE && loss_kind != VPR_LOSS_HUBER) || (loss_kind == VPR_LOSS_HUBER && !(huber_delta > 0.0)))
    return VPR_ERR_INVALID_ARG;
  HeadTrainPlan p;
  if (B < 1 || D < 1 || hidden < 1 || n_out < 1) return VPR_ERR_INVALID_ARG;
  if (!head_train_plan(B, D, hidden, n_out, &p)) return VPR_ERR_UNSUPPORTED;
  if (x_stride < D || (x_stride % 4) || y_stride < n_out) return VPR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(workspace)) & 15)
    return VPR_ERR_UNSUPPORTED;
  if (workspace_bytes < p.total) return VPR_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  float* part = reinterpret_cast<float*>(ws);
  float* H = reinterpret_cast<float*>(ws + p.off_h);
  float* po = reinterpret_cast<float*>(ws + p.off_po);
  float* snap = reinterpret_cast<float*>(ws + p.off_snap);
  float* diff = reinterpret_cast<float*>(ws + p.off_diff);
  int* counter = reinterpret_cast<int*>(ws + p.off_cnt);
  const int nsteps = D / 16;
  const int sps = (nsteps + p.ks - 1) / p.ks;
  if (B > HT_BT)
    VPR_TRY_LAUNCH(launch_kernel(head_fwd_partial_kernel<true>, dim3(hidden / HT_HT, p.ks, (B + 4 * HT_BT - 1) / (4 * HT_BT)), dim3(256),
                                 0, stream, X, x_stride, idx, (const float*)W1, part, B, D, hidden, sps, counter));
  else
    VPR_TRY_LAUNCH(launch_kernel(head_fwd_partial_kernel<false>, dim3(hidden / HT_HT, p.ks, 1), dim3(256), 0, stream,
                                 X, x_stride, idx, (const float*)W1, part, B, D, hidden, sps, counter));
  VPR_TRY_LAUNCH(launch_kernel(head_mid_kernel, dim3(p.nwg2), dim3(256), 0, stream, (const float*)part, p.ks, (const float*)b1,
                               (const float*)W2, (const float*)b2, Y, y_stride, idx, H, po, snap, diff, loss_out, counter,
                               B, hidden, n_out, loss_kind == VPR_LOSS_HUBER ? (float)huber_delta : 0.f));
  HeadTrainArgs a;
  a.X = X; a.x_stride = x_stride; a.idx = idx; a.Y = Y; a.y_stride = y_stride;
  a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.m = m; a.v = v;
  a.H = H; a.diff = diff; a.snap = snap; a.B = B; a.D = D; a.hidden = hidden; a.n_out = n_out;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.c.decay = (float)(1.0 - lr * weight_decay);
  a.c.one_minus_b1 = (float)(1.0 - beta1);
  a.c.beta2 = (float)beta2;
  a.c.one_minus_b2 = (float)(1.0 - beta2);
  a.c.sqrt_bc2 = (float)sqrt(bc2);
  a.c.eps = (float)eps;
  a.c.step_size = (float)(lr / bc1);
  a.abl = 0;
#ifdef VPR_ABLATION
  if (tune_or(TUNE_HEAD_TRAIN_VARIANT, 0) >= 10) a.abl = tune_or(TUNE_HEAD_TRAIN_VARIANT, 0);
#endif
  switch (tune_or(TUNE_HEAD_TRAIN_VARIANT, 0)) {        // A/B: rows per workgroup / batch rows per chunk / rows per load group
    case 1:  VPR_TRY_LAUNCH((launch_head_update<8, 8, 8>(a, stream))); break;
    case 2:  VPR_TRY_LAUNCH((launch_head_update<4, 8, 4>(a, stream))); break;
    case 3:  VPR_TRY_LAUNCH((launch_head_update<8, 8, 4, true>(a, stream))); break;
    case 4:  VPR_TRY_LAUNCH((launch_head_update<16, 8, 4>(a, stream))); break;
    default: VPR_TRY_LAUNCH((launch_head_update<8, 8, 4>(a, stream))); break;
  }
  return VPR_OK;
}

extern "C" int vpr_head_train_step(const float* X, long long x_stride, const int* idx, const float* Y, long long y_stride,
                                   int B, int D, int hidden, int n_out, float* W1, float* b1, float* W2, float* b2,
                                   float* m, float* v, int step, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int loss_kind, double huber_delta, float* loss_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return head_train_step_impl(X, x_stride, idx, Y, y_stride, B, D, hidden, n_out, W1, b1, W2, b2, m, v, step, lr, beta1, beta2,
                              eps, weight_decay, loss_kind, huber_delta, loss_out, workspace, workspace_bytes,
                              static_cast<hipStream_t>(stream));
}

extern "C" int vpr_head_train_epoch(const float* X, long long x_stride, const int* order, int n, int batch_size,
                                    const float* Y, long long y_stride, int D, int hidden, int n_out,
                                    float* W1, float* b1, float* W2, float* b2, float* m, float* v, int first_step,
                                    double lr, double beta1, double beta2, double eps, double weight_decay,
                                    int loss_kind, double huber_delta, float* losses, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (!order || n < 1 || batch_size < 1 || first_step < 1) return VPR_ERR_INVALID_ARG;
  const int nb = (n + batch_size - 1) / batch_size;
  if ((long long)first_step + nb - 1 > 2147483647LL) return VPR_ERR_INVALID_ARG;
  // every batch is validated before the first launch: a refused epoch leaves the parameters untouched
  HeadTrainPlan p;
  if (D < 1 || hidden < 1 || n_out < 1) return VPR_ERR_INVALID_ARG;
  if (!head_train_plan(batch_size < n ? batch_size : n, D, hidden, n_out, &p)) return VPR_ERR_UNSUPPORTED;
  if (workspace_bytes < p.total) return VPR_ERR_WORKSPACE;
  for (int i = 0; i < nb; ++i) {
    const int lo = i * batch_size;
    const int B = n - lo < batch_size ? n - lo : batch_size;
    VPR_TRY_LAUNCH(head_train_step_impl(X, x_stride, order + lo, Y, y_stride, B, D, hidden, n_out, W1, b1, W2, b2, m, v,
                                        first_step + i, lr, beta1, beta2, eps, weight_decay, loss_kind, huber_delta,
                                        losses ? losses + i : nullptr, workspace, workspace_bytes,
                                        static_cast<hipStream_t>(stream)));
  }
  return VPR_OK;
}
