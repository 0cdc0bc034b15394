// head_train.hip — one optimizer step of the two-layer regression head on cached descriptors.
//
// vpr_head_train_step   forward, MSE loss, backward and AdamW update of Linear(D,hidden) -> ReLU -> Linear(hidden,n_out)
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
constexpr int HT_IT = 16;     // W1 rows per update workgroup
constexpr int HT_JT = 1024;   // W1 columns per update workgroup (256 threads x float4)
constexpr int HT_BC = 16;     // batch rows held in registers at a time by the update kernel
constexpr int HT_MAXB = 64;   // largest batch (LDS tables of the update kernel)
constexpr int HT_MAXO = 8;    // largest n_out

// 1. part[ks][b][h] = sum over K-slice ks of x[idx[b]][k] * W1[h][k].  Workgroup = 32 hidden units x 16 batch rows x one
// slice; its four waves take the slice's k-steps round robin and their accumulators are added in wave order.
__global__ __launch_bounds__(256) void head_fwd_partial_kernel(
    const float* __restrict__ X, long long x_stride, const int* __restrict__ idx, const float* __restrict__ W1,
    float* __restrict__ part, int B, int D, int hidden, int steps_per_slice) {
  __shared__ float red[4][HT_BT * HT_HT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h0 = blockIdx.x * HT_HT, ks = blockIdx.y, b0 = blockIdx.z * HT_BT;
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
  for (int base = s_begin; base < s_end; base += 4 * HT_CH) {     // uniform per workgroup
    float4 a[HT_CH], w0[HT_CH], w1[HT_CH];
#pragma unroll
    for (int i = 0; i < HT_CH; ++i) {
      const int s = min(base + wave + 4 * i, s_end - 1);
      a[i] = xa[s * 4];
      w0[i] = wa[s * 4];
      w1[i] = wb[s * 4];
    }
#pragma unroll
    for (int i = 0; i < HT_CH; ++i) {
      if (base + wave + 4 * i < s_end) {   // wave-uniform
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
    red[wave][(4 * kg + e) * HT_HT + r] = acc0[e];
    red[wave][(4 * kg + e) * HT_HT + 16 + r] = acc1[e];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < HT_BT * HT_HT; t += 256) {
    const int b = b0 + (t >> 5);
    if (b < B) part[((long long)ks * B + b) * hidden + h0 + (t & 31)] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
}

// 2. Workgroup = 16 hidden units x all batch rows.  z = slabs in slice order + b1, h = relu(z) -> H; the workgroup's share of
// every output, po[wg][b][o] = sum over its 16 units of h * W2[o][unit] (DPP row sum: fixed tree); snap = (W2 | b2) as they
// are BEFORE this step's update (the update kernel reads the copy while designated workgroups rewrite the originals).
__global__ __launch_bounds__(256) void head_mid_kernel(
    const float* __restrict__ part, int nslice, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, float* __restrict__ H, float* __restrict__ po, float* __restrict__ snap,
    int B, int hidden, int n_out) {
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
    for (int s0 = 0; s0 < nslice; s0 += 8) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = part[((long long)min(s0 + i, nslice - 1) * B + bb) * hidden + h];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (s0 + i < nslice) z += t[i];
    }
    const float hv = valid ? fmaxf(z + bias, 0.f) : 0.f;
    if (valid) H[(long long)b * hidden + h] = hv;
#pragma unroll
    for (int o = 0; o < HT_MAXO; ++o)
      if (o < n_out) {       // uniform
        const float v = row16_sum(hv * w2[o]);
        if (hl == 0 && valid) po[((long long)blockIdx.x * B + b) * n_out + o] = v;
      }
  }
  for (int t = threadIdx.x; t < HT_H2 * n_out; t += 256) {
    const int o = t / HT_H2, hh = h0 + t % HT_H2;
    snap[(long long)o * hidden + hh] = W2[(long long)o * hidden + hh];
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_out) snap[(long long)n_out * hidden + threadIdx.x] = b2[threadIdx.x];
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
  const float *H, *po, *snap;           // from head_mid_kernel
  int nwg2, B, D, hidden, n_out;
  AdamConsts c;
  float* loss_out;
};

// 3. Workgroup = 16 rows x 1024 columns of W1.  Every workgroup re-derives the outputs (a few hundred FMAs), dO = 2 (o - y) /
// (B n_out) and dz = (dO W2) * (h > 0) for its 16 hidden units; a thread owns 4 columns: g[i] = sum_b dz[b][i] * x[b][cols]
// (b ascending), then the AdamW update of its 16 x 4 weights.  Column-tile 0 also updates b1 and W2 for its 16 units,
// workgroup (0,0) b2 and the loss.
__global__ __launch_bounds__(256) void head_update_kernel(HeadTrainArgs a) {
  __shared__ float s_diff[HT_MAXB * HT_MAXO];
  __shared__ __attribute__((aligned(16))) float s_dz[HT_MAXB * HT_IT];
  __shared__ float s_h[HT_MAXB * HT_IT];
  const int B = a.B, n_out = a.n_out, hidden = a.hidden, D = a.D;
  const int i0 = blockIdx.y * HT_IT, j = blockIdx.x * HT_JT + threadIdx.x * 4;
  const float gscale = 2.0f / (float)(B * n_out);
  const float* snapb2 = a.snap + (long long)n_out * hidden;
  for (int t = threadIdx.x; t < B * n_out; t += 256) {
    const int b = t / n_out, o = t - b * n_out;
    float acc = 0.f;
    for (int w = 0; w < a.nwg2; ++w) acc += a.po[((long long)w * B + b) * n_out + o];
    acc += snapb2[o];
    const long long row = a.idx ? (long long)a.idx[b] : (long long)b;
    s_diff[t] = acc - a.Y[row * a.y_stride + o];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < B * HT_IT; t += 256) {
    const int b = t / HT_IT, i = t - b * HT_IT;
    const float hv = a.H[(long long)b * hidden + i0 + i];
    float g = 0.f;
    for (int o = 0; o < n_out; ++o) g += (s_diff[b * n_out + o] * gscale) * a.snap[(long long)o * hidden + i0 + i];
    s_dz[t] = hv > 0.f ? g : 0.f;
    s_h[t] = hv;
  }
  __syncthreads();
  if (j < D) {
    float4 g[HT_IT];
#pragma unroll
    for (int i = 0; i < HT_IT; ++i) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int bc = 0; bc < B; bc += HT_BC) {
      float4 x4[HT_BC];
#pragma unroll
      for (int u = 0; u < HT_BC; ++u) {
        const int bb = min(bc + u, B - 1);
        const long long row = a.idx ? (long long)a.idx[bb] : (long long)bb;
        x4[u] = *reinterpret_cast<const float4*>(a.X + row * a.x_stride + j);
      }
#pragma unroll
      for (int u = 0; u < HT_BC; ++u) {
        if (bc + u < B) {      // uniform
          const float4* dzr = reinterpret_cast<const float4*>(s_dz + (bc + u) * HT_IT);
#pragma unroll
          for (int q = 0; q < HT_IT / 4; ++q) {
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
    // the 3 x 16 row segments of (W1, m, v) are requested in two halves of 8 rows (24 x 16 B in flight per thread)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 p4[HT_IT / 2], m4[HT_IT / 2], v4[HT_IT / 2];
#pragma unroll
      for (int i = 0; i < HT_IT / 2; ++i) {
        const long long off = (long long)(i0 + half * (HT_IT / 2) + i) * D + j;
        p4[i] = *reinterpret_cast<const float4*>(a.W1 + off);
        m4[i] = *reinterpret_cast<const float4*>(a.m + off);
        v4[i] = *reinterpret_cast<const float4*>(a.v + off);
      }
#pragma unroll
      for (int i = 0; i < HT_IT / 2; ++i) {
        const long long off = (long long)(i0 + half * (HT_IT / 2) + i) * D + j;
        const float4 gg = g[half * (HT_IT / 2) + i];
        adamw(p4[i].x, m4[i].x, v4[i].x, gg.x, a.c);
        adamw(p4[i].y, m4[i].y, v4[i].y, gg.y, a.c);
        adamw(p4[i].z, m4[i].z, v4[i].z, gg.z, a.c);
        adamw(p4[i].w, m4[i].w, v4[i].w, gg.w, a.c);
        *reinterpret_cast<float4*>(a.W1 + off) = p4[i];
        *reinterpret_cast<float4*>(a.m + off) = m4[i];
        *reinterpret_cast<float4*>(a.v + off) = v4[i];
      }
    }
  }
  if (blockIdx.x == 0) {
    const long long off_b1 = (long long)hidden * D, off_w2 = off_b1 + hidden, off_b2 = off_w2 + (long long)n_out * hidden;
    for (int t = threadIdx.x; t < HT_IT * (1 + n_out); t += 256) {
      const int which = t / HT_IT, i = t - which * HT_IT;
      float g = 0.f;
      if (which == 0) {
        for (int b = 0; b < B; ++b) g += s_dz[b * HT_IT + i];
        const long long s = off_b1 + i0 + i;
        adamw(a.b1[i0 + i], a.m[s], a.v[s], g, a.c);
      } else {
        const int o = which - 1;
        for (int b = 0; b < B; ++b) g = fmaf(s_diff[b * n_out + o] * gscale, s_h[b * HT_IT + i], g);
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
      if (threadIdx.x == 64 && a.loss_out) {
        float s = 0.f;
        for (int t = 0; t < B * n_out; ++t) s = fmaf(s_diff[t], s_diff[t], s);
        *a.loss_out = s / (float)(B * n_out);
      }
    }
  }
}

static int head_train_slices(int B, int D, int hidden) {
  // about two forward workgroups per CU; a slice holds at least 4 k-steps (one per wave)
  const int tiles = (hidden / HT_HT) * ((B + HT_BT - 1) / HT_BT);
  int ks = (512 + tiles - 1) / tiles;
  const int nsteps = D / 16;
  const int max_ks = nsteps / 4 > 0 ? nsteps / 4 : 1;
  if (ks > max_ks) ks = max_ks;
  if (ks > 64) ks = 64;
  if (ks < 1) ks = 1;
  return ks;
}

struct HeadTrainPlan { int ks, nwg2; size_t off_h, off_po, off_snap, total; };
static bool head_train_plan(int B, int D, int hidden, int n_out, HeadTrainPlan* p) {
  if (B < 1 || B > HT_MAXB || D < 16 || (D % 16) || hidden < HT_HT || (hidden % HT_HT) || n_out < 1 || n_out > HT_MAXO) return false;
  p->ks = head_train_slices(B, D, hidden);
  p->nwg2 = hidden / HT_H2;
  size_t off = align_up((size_t)p->ks * B * hidden * sizeof(float), 256);
  p->off_h = off;    off += align_up((size_t)B * hidden * sizeof(float), 256);
  p->off_po = off;   off += align_up((size_t)p->nwg2 * B * n_out * sizeof(float), 256);
  p->off_snap = off; off += align_up(((size_t)n_out * hidden + n_out) * sizeof(float), 256);
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

extern "C" int vpr_head_train_step(const float* X, long long x_stride, const int* idx, const float* Y, long long y_stride,
                                   int B, int D, int hidden, int n_out, float* W1, float* b1, float* W2, float* b2,
                                   float* m, float* v, int step, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, float* loss_out, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!X || !Y || !W1 || !b1 || !W2 || !b2 || !m || !v || !workspace || step < 1) return VPR_ERR_INVALID_ARG;
  if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0))
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
  const int nsteps = D / 16;
  const int sps = (nsteps + p.ks - 1) / p.ks;
  VPR_TRY_LAUNCH(launch_kernel(head_fwd_partial_kernel, dim3(hidden / HT_HT, p.ks, (B + HT_BT - 1) / HT_BT), dim3(256), 0, stream,
                               X, x_stride, idx, (const float*)W1, part, B, D, hidden, sps));
  VPR_TRY_LAUNCH(launch_kernel(head_mid_kernel, dim3(p.nwg2), dim3(256), 0, stream, (const float*)part, p.ks, (const float*)b1,
                               (const float*)W2, (const float*)b2, H, po, snap, B, hidden, n_out));
  HeadTrainArgs a;
  a.X = X; a.x_stride = x_stride; a.idx = idx; a.Y = Y; a.y_stride = y_stride;
  a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.m = m; a.v = v;
  a.H = H; a.po = po; a.snap = snap; a.nwg2 = p.nwg2; a.B = B; a.D = D; a.hidden = hidden; a.n_out = n_out;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.c.decay = (float)(1.0 - lr * weight_decay);
  a.c.one_minus_b1 = (float)(1.0 - beta1);
  a.c.beta2 = (float)beta2;
  a.c.one_minus_b2 = (float)(1.0 - beta2);
  a.c.sqrt_bc2 = (float)sqrt(bc2);
  a.c.eps = (float)eps;
  a.c.step_size = (float)(lr / bc1);
  a.loss_out = loss_out;
  VPR_TRY_LAUNCH(launch_kernel(head_update_kernel, dim3((D + HT_JT - 1) / HT_JT, hidden / HT_IT), dim3(256), 0, stream, a));
  return VPR_OK;
}
