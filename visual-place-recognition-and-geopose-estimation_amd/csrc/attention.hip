// attention.hip — multi-head self-attention for the ViT backbone's short sequences
// (DINOv2/14 at 224 px: T = 257 tokens, head dim 64, non-causal, no mask, no dropout).
// PyTorch's SDPA (AOTriton flash kernel) needs ~100 us per layer at B=64, H=16 (~170 TFLOP/s):
// its tiles are sized for long sequences.  Here K and V of one (image, head) — 2 x 33 KB — live in
// LDS for the whole workgroup, so there is no online-softmax rescaling at all:
//   S^T = K Q^T  (v_mfma_f32_16x16x32_bf16; keys on the register axis, queries on the lanes)
//   softmax over keys: in-lane over 18 key blocks x 4 registers, then two cross-lane steps
//   O^T = V^T P^T (P stays in registers: the S^T accumulator layout IS the B-operand layout once
//        two 16-key blocks are packed into one 32-deep k-step; V sits row-major in LDS exactly
//        like K and its transposed A fragments come from ds_read_b64_tr_b16, the gfx950
//        hardware-transpose read: two 8-byte reads per MFMA, no transposed staging pass)
// One workgroup (4 waves) per (image, head); each wave owns query tiles of 16 rows.
// Input is the fused projection output qkv [B, T, 3, H, 64] (no q/k/v copies), output [B, T, H*64].
#include "vpr_common.cuh"
#include "vpr_internal.h"

namespace vpr {

constexpr int AT_D = 64;          // head dim
constexpr int AT_KP = 288;        // padded keys (18 blocks of 16 = 9 MFMA k-steps of 32)

__global__ __launch_bounds__(256, 2) void attention_kernel(
    const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int T, int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                                                   // [AT_KP][64] bf16, swizzled 128-B rows
  char* Vs = smem + AT_KP * 128;                                     // [AT_KP][64] bf16, same layout
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long tok_stride = 3LL * H * AT_D;
  const uint16_t* qb = qkv + (long long)b * T * tok_stride + (long long)h * AT_D;
  const uint16_t* kb_ = qb + (long long)H * AT_D;
  const uint16_t* vb = qb + 2LL * H * AT_D;

  // ---- stage K and V (row-major, swizzled 128-B rows); rows >= T are zero ----
  // (all 18 loads of a thread are issued before the first LDS write: one trip to memory, not nine)
  constexpr int NLD = AT_KP * 8 / 256;                // 9 chunk pairs per thread
  s16x8 kreg[NLD], vreg[NLD];
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = tid + 256 * it;
    const int key = i >> 3, ch = i & 7;
    const int kc = key < T ? key : T - 1;               // clamp the address, zero the value below
    kreg[it] = *reinterpret_cast<const s16x8*>(kb_ + kc * tok_stride + ch * 8);
    vreg[it] = *reinterpret_cast<const s16x8*>(vb + kc * tok_stride + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = tid + 256 * it;
    const int key = i >> 3, ch = i & 7;
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const s16x8 kv = key < T ? kreg[it] : z, vv = key < T ? vreg[it] : z;
    *reinterpret_cast<s16x8*>(Ks + tile_off(key, ch)) = kv;
    *reinterpret_cast<s16x8*>(Vs + tile_off(key, ch)) = vv;
  }
  __syncthreads();

  const int qcol = lane & 15, g = lane >> 4;
  const int ntile = (T + 15) >> 4;
  float npad = 0.f;                       // padded keys among this lane's keys {16kb + 4g + e}
  for (int kb = T >> 4; kb < AT_KP / 16; ++kb)
    for (int e = 0; e < 4; ++e) npad += (16 * kb + 4 * g + e >= T) ? 1.f : 0.f;
  for (int qt = wave; qt < ntile; qt += 4) {
    const int q0 = qt * 16;
    const int qrow = min(q0 + qcol, T - 1);
    // Q^T fragments (B operand): lane (query = lane&15, group g) holds Q[query][32s + 8g .. +7]
    bf16x8 bq[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
      bq[s] = *reinterpret_cast<const bf16x8*>(qb + qrow * tok_stride + 32 * s + 8 * g);

    // ---- S^T = K Q^T: 18 key blocks; C/D: col = query (lane&15), row = key 4g+e of the block ----
    f32x4 sacc[AT_KP / 16];
#pragma unroll
    for (int kb = 0; kb < AT_KP / 16; ++kb) {
      sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 a = lds_frag(Ks, 16 * kb + qcol, g + 4 * s);
        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[s], sacc[kb], 0, 0, 0);
      }
      if ((kb % 3) == 2) __builtin_amdgcn_sched_barrier(0);   // keep at most 6 K fragments in flight
    }

    // ---- softmax over keys, base 2: p = exp2((s - max) * scale*log2e) = exp2(fma(s, c, -max*c)).
    // No masking: padded K rows are zero, so padded keys score exactly 0; they may only raise the
    // stabiliser (still an upper bound of the row), their V rows are zero (no PV contribution),
    // and their exp2(-max*c) terms are subtracted from the row sum (npad of them in this lane). ----
    float mx = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < AT_KP / 16; ++kb)
      mx = fmaxf(fmaxf(mx, fmaxf(sacc[kb][0], sacc[kb][1])), fmaxf(sacc[kb][2], sacc[kb][3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = -mx * scale_log2e;
    float sum = 0.f;
    bf16x8 pb[AT_KP / 32];      // P^T as B operands: k-step t packs key blocks 2t (j<4) and 2t+1 (j>=4)
#pragma unroll
    for (int t = 0; t < AT_KP / 32; ++t) {
      float p[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[e] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * t][e], scale_log2e, nm));
        p[4 + e] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * t + 1][e], scale_log2e, nm));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { sum += p[j]; pb[t][j] = (__bf16)p[j]; }
    }
    sum -= npad * __builtin_amdgcn_exp2f(nm);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O^T = V^T P^T: 4 blocks of 16 output dims; A = V^T (row = dim, k = keys in P's order).
    // ds_read_b64_tr_b16 per 16-lane group: lane 4q+p supplies the address of row (key) k0+q,
    // columns (dims) 4p..4p+3, and receives column (dim) `lane&15` of the 4 rows — i.e. keys
    // k0..k0+3 for its dim: elements j=0..3 (block 2t, k0 = 32t+4g) and j=4..7 (block 2t+1).
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int trq = (lane & 15) >> 2, trp = lane & 3;
    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < AT_KP / 32; ++t) {
        const int r0 = 32 * t + 4 * g + trq;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(Vs + tile_off(r0, 2 * db + (trp >> 1)) + 8 * (trp & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(Vs + tile_off(r0 + 16, 2 * db + (trp >> 1)) + 8 * (trp & 1)));
        const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), pb[t], oacc[db], 0, 0, 0);
        if ((t % 3) == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // C/D: col = query (lane&15), row = dim 16db + 4g + e  -> 8-byte stores of 4 consecutive dims
    if (q0 + qcol < T) {
      uint16_t* orow = out + ((long long)b * T + q0 + qcol) * H * AT_D + (long long)h * AT_D;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        ushort4 o;
        o.x = f32_to_bf16_bits(oacc[db][0] * inv);
        o.y = f32_to_bf16_bits(oacc[db][1] * inv);
        o.z = f32_to_bf16_bits(oacc[db][2] * inv);
        o.w = f32_to_bf16_bits(oacc[db][3] * inv);
        *reinterpret_cast<ushort4*>(orow + 16 * db + 4 * g) = o;
      }
    }
  }
}

constexpr size_t AT_LDS = (size_t)2 * AT_KP * 128;

}  // namespace vpr

using namespace vpr;

extern "C" int vpr_attention_qkv_bf16(const uint16_t* qkv, uint16_t* out, int B, int T, int H, int head_dim,
                                      float scale, void* stream) {
  if (!qkv || !out || B <= 0 || T <= 0 || H <= 0) return VPR_ERR_INVALID_ARG;
  if (head_dim != AT_D || T > AT_KP || (long long)B * H > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) return VPR_ERR_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)AT_LDS) != hipSuccess)
      return VPR_ERR_LAUNCH;
    attr = true;
  }
  VPR_TRY_LAUNCH(launch_kernel(attention_kernel, dim3((unsigned)(B * H)), dim3(256), AT_LDS,
                               static_cast<hipStream_t>(stream), qkv, out, T, H, scale * 1.4426950408889634f));
  return VPR_OK;
}
