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
// One workgroup (8 waves) per (image, head); each wave owns query tiles of 16 rows.
// Input is the fused projection output qkv [B, T, 3, H, 64] (no q/k/v copies), output [B, T, H*64].
#include <stdlib.h>
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int AT_D = 64;          // head dim
constexpr int AT_KP = 288;        // padded keys (18 blocks of 16 = 9 MFMA k-steps of 32)


// V is only ever read through ds_read_b64_tr_b16; chunk c of key row r sits at c ^ (r & 7), which
// makes every 32-lane half of those reads bank-conflict-free (the K swizzle would be 2-way).
__device__ __forceinline__ int vtile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct AttnCtx {
  const char* Ks; const char* Vs; const uint16_t* qb; long long tok_stride; int T, qcol, g, trq, trp;
  float scale_log2e; uint16_t* out; long long out_stride;
  long long row_base, tail_base; int Tp;   // token t of this image lives in row (t < Tp ? row_base + t : tail_base + t - Tp)
};
__device__ __forceinline__ long long attn_row(const AttnCtx& cx, int t) {
  return t < cx.Tp ? cx.row_base + t : cx.tail_base + (t - cx.Tp);
}
typedef __attribute__((ext_vector_type(4))) short s16x4;

// Q^T fragments (B operand) of query tile qt: lane (query = lane&15, group g) holds Q[query][32s + 8g .. +7]
__device__ __forceinline__ void attn_load_q(const AttnCtx& cx, int qt, bf16x8 (&bq)[2]) {
  const int qrow = min(qt * 16 + cx.qcol, cx.T - 1);
#pragma unroll
  for (int s = 0; s < 2; ++s)
    bq[s] = *reinterpret_cast<const bf16x8*>(cx.qb + attn_row(cx, qrow) * cx.tok_stride + 32 * s + 8 * cx.g);
}
// S^T block kb = K[16kb..16kb+15] Q^T; C/D: col = query (lane&15), row = key 4g+e of the block
__device__ __forceinline__ f32x4 attn_qk_block(const AttnCtx& cx, int kb, const bf16x8 (&bq)[2]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(cx.Ks, 16 * kb + cx.qcol, cx.g + 4 * s), bq[s], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void attn_qk_all(const AttnCtx& cx, const bf16x8 (&bq)[2], f32x4 (&sc)[AT_KP / 16]) {
#pragma unroll
  for (int kb = 0; kb < AT_KP / 16; ++kb) {
    sc[kb] = attn_qk_block(cx, kb, bq);
    if ((kb % 3) == 2) __builtin_amdgcn_sched_barrier(0);
  }
}
// Row max over keys.  No masking: padded K rows are zero, so padded keys score exactly 0; they may
// only raise the stabiliser (still an upper bound of the row), their V rows are zero, and their
// exp2 terms are subtracted from the row sum in phase 2 (AT_KP - T of them per row).
__device__ __forceinline__ float attn_rowmax(const f32x4 (&sc)[AT_KP / 16]) {
  float mx = sc[0][0];
#pragma unroll
  for (int kb = 0; kb < AT_KP / 16; ++kb) {     // two v_max3_f32 per block
    mx = fmaxf(fmaxf(mx, sc[kb][0]), sc[kb][1]);
    mx = fmaxf(fmaxf(mx, sc[kb][2]), sc[kb][3]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  return fmaxf(mx, __shfl_xor(mx, 32, 64));
}
// phase 1: p = exp2(fma(s, c, -max*c)) of the current tile -> bf16 B operands pb (k-step t packs key
// blocks 2t (j<4) and 2t+1 (j>=4)); meanwhile the QK^T blocks 2t, 2t+1 of the next tile,
// written IN PLACE over the score registers just consumed (one score buffer, 72 VGPRs).
__device__ __forceinline__ float attn_phase1(const AttnCtx& cx, f32x4 (&sc)[AT_KP / 16], float mx,
                                             bf16x8 (&pb)[AT_KP / 32], bool has_next, const bf16x8 (&bq)[2]) {
  const float nm = -mx * cx.scale_log2e;
#pragma unroll
  for (int t = 0; t < AT_KP / 32; ++t) {
    float p[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[e] = __builtin_amdgcn_exp2f(fmaf(sc[2 * t][e], cx.scale_log2e, nm));
      p[4 + e] = __builtin_amdgcn_exp2f(fmaf(sc[2 * t + 1][e], cx.scale_log2e, nm));
    }
    if (has_next) {   // wave-uniform; the score registers of blocks 2t, 2t+1 are free again: the
      sc[2 * t] = attn_qk_block(cx, 2 * t, bq);           // next tile's S^T lands in place
      sc[2 * t + 1] = attn_qk_block(cx, 2 * t + 1, bq);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) pb[t][j] = (__bf16)p[j];
  }
  // what one padded key contributes to the row sum (bf16-rounded like every p): removed in phase 2
  return (float)(__bf16)__builtin_amdgcn_exp2f(nm);
}
// phase 2: O^T = V^T P^T of the current tile (4 blocks of 16 dims; A = V^T by ds_read_b64_tr_b16:
// lane 4q+p of a 16-lane group supplies the address of key row k0+q, dims 4p..4p+3 and receives
// dim `lane&15` of those 4 keys: j=0..3 from block 2t (k0 = 32t+4g), j=4..7 from block 2t+1),
// normalise, store; meanwhile the row max of the next tile.
__device__ __forceinline__ float attn_phase2(const AttnCtx& cx, const bf16x8 (&pb)[AT_KP / 32], float ppad,
                                             int qt, bool has_next, const f32x4 (&sn)[AT_KP / 16]) {
  // The row sum rides on the matrix pipe: a fifth "dim block" whose A operand is all ones gives
  // sum_k P^T[k][query] in every row of its accumulator — 9 MFMAs instead of 72 VALU adds and two
  // cross-lane steps, and it sums exactly the bf16 p values the numerator uses.
  const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
  f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < AT_KP / 32; ++t) sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[t], sacc, 0, 0, 0);
  f32x4 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < AT_KP / 32; ++t) {
      const int r0 = 32 * t + 4 * cx.g + cx.trq;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(cx.Vs + vtile_off(r0, 2 * db + (cx.trp >> 1)) + 8 * (cx.trp & 1)));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(cx.Vs + vtile_off(r0 + 16, 2 * db + (cx.trp >> 1)) + 8 * (cx.trp & 1)));
      const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), pb[t], oacc[db], 0, 0, 0);
      if ((t % 3) == 2) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float inv = 1.0f / (sacc[0] - (float)(AT_KP - cx.T) * ppad);   // minus the AT_KP - T padded keys
  float mxn = 0.f;
  if (has_next) mxn = attn_rowmax(sn);
  // C/D: col = query (lane&15), row = dim 16db + 4g + e  -> 8-byte stores of 4 consecutive dims
  const int q = qt * 16 + cx.qcol;
  if (q < cx.T) {
    uint16_t* orow = cx.out + attn_row(cx, q) * cx.out_stride;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      ushort4 o;
      o.x = f32_to_bf16_bits(oacc[db][0] * inv);
      o.y = f32_to_bf16_bits(oacc[db][1] * inv);
      o.z = f32_to_bf16_bits(oacc[db][2] * inv);
      o.w = f32_to_bf16_bits(oacc[db][3] * inv);
      *reinterpret_cast<ushort4*>(orow + 16 * db + 4 * cx.g) = o;
    }
  }
  return mxn;
}

template <int NW, bool PIPE, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW / 2) void attention_kernel(
    const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int T, int Tp, long long tail_row0, int H,
    float scale_log2e, int dephase) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // A/B (VPR_ATTN_VARIANT = 20 + n): the workgroups that fill the SECOND slot of every CU in the first wave of the grid
  // (256 <= blockIdx < 512 on 256 CUs) start n * ~1 us late, so that the two workgroups of a CU stop running their
  // staging / compute phases in lockstep
  if (dephase > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < dephase; ++i) __builtin_amdgcn_s_sleep(33);       // 33 * 64 clocks ~ 1 us
  char* Ks = smem;                                                   // [AT_KP][64] bf16, swizzled 128-B rows
  char* Vs = smem + AT_KP * 128;                                     // [AT_KP][64] bf16, same layout
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long tok_stride = 3LL * H * AT_D;
  const uint16_t* qb = qkv + (long long)h * AT_D;
  const uint16_t* kb_ = qb + (long long)H * AT_D;
  const uint16_t* vb = qb + 2LL * H * AT_D;

  // Q fragments of all of this wave's query tiles (qt = wave + 4j) are requested first, so their
  // latency hides under the K/V staging instead of stalling every tile.
  constexpr int NT = NW * 64;
  constexpr int AT_MAXT = (AT_KP / 16 + NW - 1) / NW;   // query tiles per wave
  bf16x8 qf[AT_MAXT][2];
  {
    AttnCtx c0;
    c0.qb = qb; c0.tok_stride = tok_stride; c0.T = T; c0.qcol = lane & 15; c0.g = lane >> 4;
    c0.Tp = Tp; c0.row_base = (long long)b * Tp; c0.tail_base = tail_row0 + (long long)b * (T - Tp);
#pragma unroll
    for (int j = 0; j < AT_MAXT; ++j) attn_load_q(c0, min(wave + NW * j, ((T + 15) >> 4) - 1), qf[j]);
  }
  // ---- stage K and V (row-major, swizzled 128-B rows); rows >= T are zero ----
  // (all 18 loads of a thread are issued before the first LDS write: one trip to memory, not nine)
  constexpr int NLD = (AT_KP * 8 + NT - 1) / NT;      // chunk pairs per thread (9 at 256 threads)
  s16x8 kreg[NLD], vreg[NLD];
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = min(tid + NT * it, AT_KP * 8 - 1);
    const int key = i >> 3, ch = i & 7;
    const int kc = key < T ? key : T - 1;               // clamp the address, zero the value below
    const long long krow = kc < Tp ? (long long)b * Tp + kc : tail_row0 + (long long)b * (T - Tp) + (kc - Tp);
    kreg[it] = *reinterpret_cast<const s16x8*>(kb_ + krow * tok_stride + ch * 8);
    vreg[it] = *reinterpret_cast<const s16x8*>(vb + krow * tok_stride + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = min(tid + NT * it, AT_KP * 8 - 1);      // a clamped duplicate rewrites the same bytes
    const int key = i >> 3, ch = i & 7;
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const s16x8 kv = key < T ? kreg[it] : z, vv = key < T ? vreg[it] : z;
    *reinterpret_cast<s16x8*>(Ks + tile_off(key, ch)) = kv;
    *reinterpret_cast<s16x8*>(Vs + vtile_off(key, ch)) = vv;
  }
  __syncthreads();

  if (ABL == 3) return;            // ablation: staging only
  const int qcol = lane & 15, g = lane >> 4;
  const int ntile = (T + 15) >> 4;

  AttnCtx cx;
  cx.Ks = Ks; cx.Vs = Vs; cx.qb = qb; cx.tok_stride = tok_stride; cx.T = T; cx.qcol = qcol; cx.g = g;
  cx.trq = (lane & 15) >> 2; cx.trp = lane & 3; cx.scale_log2e = scale_log2e;
  cx.out = out + (long long)h * AT_D; cx.out_stride = (long long)H * AT_D;
  cx.Tp = Tp; cx.row_base = (long long)b * Tp; cx.tail_base = tail_row0 + (long long)b * (T - Tp);

  // Software pipeline over this wave's query tiles (qt = wave, wave+4, ...), one score buffer:
  //   phase 1: exp/convert of the CURRENT tile (VALU) interleaved with the QK^T MFMAs of the NEXT
  //   phase 2: PV MFMAs of the current tile interleaved with the row max of the next (VALU)
  f32x4 sc[AT_KP / 16];
  bf16x8 pb[AT_KP / 32];
  if (wave >= ntile) return;
  if constexpr (PIPE) {
    attn_qk_all(cx, qf[0], sc);
    float mx = attn_rowmax(sc);
#pragma unroll
    for (int j = 0; j < AT_MAXT; ++j) {
      const int qt = wave + NW * j;
      if (qt < ntile) {   // wave-uniform
        const bool has_next = qt + NW < ntile;
        const float ppad = attn_phase1(cx, sc, mx, pb, has_next, qf[j + 1 < AT_MAXT ? j + 1 : j]);
        mx = attn_phase2(cx, pb, ppad, qt, has_next, sc);
      }
    }
  } else {   // one tile at a time: fewer live registers, more waves per SIMD
#pragma unroll
    for (int j = 0; j < AT_MAXT; ++j) {
      const int qt = wave + NW * j;
      if (qt < ntile) {
        attn_qk_all(cx, qf[j], sc);
        if (ABL == 4) {                  // ablation: QK^T only (scores summed into one store so nothing is dead)
          float a = 0.f;
#pragma unroll
          for (int kb = 0; kb < AT_KP / 16; ++kb) a += sc[kb][0] + sc[kb][1] + sc[kb][2] + sc[kb][3];
          if (qt * 16 + cx.qcol < T) cx.out[attn_row(cx, qt * 16 + cx.qcol) * cx.out_stride + 4 * cx.g] = f32_to_bf16_bits(a);
          continue;
        }
        const float mx = attn_rowmax(sc);
        const float ppad = attn_phase1(cx, sc, mx, pb, false, qf[j]);
        if (ABL == 2) {                  // ablation: no PV (probabilities summed into one store)
          float a = ppad;
#pragma unroll
          for (int t = 0; t < AT_KP / 32; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) a += (float)pb[t][e];
          if (qt * 16 + cx.qcol < T) cx.out[attn_row(cx, qt * 16 + cx.qcol) * cx.out_stride + 4 * cx.g] = f32_to_bf16_bits(a);
          continue;
        }
        attn_phase2(cx, pb, ppad, qt, false, sc);
      }
    }
  }
}

constexpr size_t AT_LDS = (size_t)2 * AT_KP * 128;

}  // namespace vpr

using namespace vpr;

static int attention_launch(const uint16_t* qkv, uint16_t* out, int B, int T, int Tp, long long tail_row0, int H,
                            int head_dim, float scale, void* stream) {
  if (!qkv || !out || B <= 0 || T <= 0 || H <= 0 || Tp < 0 || Tp > T || tail_row0 < 0) return VPR_ERR_INVALID_ARG;
  if (head_dim != AT_D || T > AT_KP || (long long)B * H > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) return VPR_ERR_UNSUPPORTED;
  const int variant = tune_or(TUNE_ATTN_VARIANT, 0);      // A/B switch; 0 = default
  const float c = scale * 1.4426950408889634f;
  const int dephase = (variant >= 20 && variant <= 60) ? variant - 20 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define VPR_ATTN_LAUNCH(NW, PIPE)                                                                        \
  do {                                                                                                   \
    static PerDeviceFlag attr = {};                                                                      \
    VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(attention_kernel<NW, PIPE>), AT_LDS, attr)); \
    VPR_TRY_LAUNCH(launch_kernel(attention_kernel<NW, PIPE>, dim3((unsigned)(B * H)), dim3(NW * 64), AT_LDS, st, \
                                 qkv, out, T, Tp, tail_row0, H, c, dephase));                                           \
  } while (0)
#define VPR_ATTN_ABL(A)                                                                                   \
  do {                                                                                                   \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<8, false, A>),                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_LDS) != hipSuccess)      \
      return VPR_ERR_LAUNCH;                                                                             \
    VPR_TRY_LAUNCH(launch_kernel(attention_kernel<8, false, A>, dim3((unsigned)(B * H)), dim3(512), AT_LDS, st, \
                                 qkv, out, T, Tp, tail_row0, H, c, 0));                                  \
  } while (0)
  // Ablations (variants 12-14, timing only): K/V + Q staging alone 11.4 us (101 MB of qkv at ~9 TB/s out of the
  // Infinity Cache), + QK^T 24.2, + softmax 32.9, full kernel 37.5 us: the phases add up, i.e. the two workgroups of a
  // CU run in lockstep and staging does not overlap compute.  A persistent variant would need both K/V sets in LDS
  // (144 KB: one workgroup, two waves per SIMD).  Tried: K and V staged by LDS-DMA with the first tile's QK^T + softmax
  // running before V has landed (padded keys masked in the scores instead of zero-filled rows): correct, but 45.7 us —
  // the scattered 128-byte DMA pieces, the mask and 6 spilled registers cost more than the overlap returns.
  // Measured at B=64, T=257, H=16 (PyTorch SDPA: 94-99 us): 8 waves, one tile at a time, 128 VGPRs,
  // 4 waves/SIMD: 37 us (default); 4 waves: 44 us; 4 waves software-pipelined across tiles
  // (208 VGPRs, 2 waves/SIMD): 45 us — occupancy beats intra-wave overlap here.
  switch (variant) {
    case 1: VPR_ATTN_LAUNCH(4, false); break;
    case 2: VPR_ATTN_LAUNCH(4, true); break;
    case 3: VPR_ATTN_LAUNCH(6, false); break;
    case 12: VPR_ATTN_ABL(2); break;   // ablations (timing only, wrong results): no PV
    case 13: VPR_ATTN_ABL(3); break;   // staging only
    case 14: VPR_ATTN_ABL(4); break;   // staging + QK^T
    default: VPR_ATTN_LAUNCH(8, false); break;
  }
#undef VPR_ATTN_LAUNCH
#undef VPR_ATTN_ABL
  return VPR_OK;
}

extern "C" int vpr_attention_qkv_bf16(const uint16_t* qkv, uint16_t* out, int B, int T, int H, int head_dim,
                                      float scale, void* stream) {
  return attention_launch(qkv, out, B, T, T, 0, H, head_dim, scale, stream);
}

extern "C" int vpr_attention_qkv_split_bf16(const uint16_t* qkv, uint16_t* out, int B, int T, int body_tokens,
                                            long long tail_row0, int H, int head_dim, float scale, void* stream) {
  return attention_launch(qkv, out, B, T, body_tokens, tail_row0, H, head_dim, scale, stream);
}
