// knn.hip — cosine top-k of bf16 descriptors against a gallery shard.
//
// The reference has no retrieval code; this is the north-star stage that sits between the
// SALAD descriptor (dinov2salad/dinov2salad_validation.py:51) and post-processing (:84).
//
// Pipeline (all on one stream, no host sync):
//   1. knn_scores_kernel   S[b,n] = <q_b, g_n>   — HBM-bound stream of the gallery, MFMA
//                          (v_mfma_f32_16x16x32_bf16) used only because a 64-query tile needs
//                          64 FLOP per gallery byte.  THE dominant kernel (roofline: HBM).
//   2. knn_select_kernel   per (query, 4096-score chunk): top-KP candidates by the MFMA score
//                          (KP = oversampled k), repeated until one chunk is left -> [B][KP].
//   3. knn_rescore_kernel  per (query, candidate): EXACT rescoring of the KP rows (bf16 products
//                          accumulated in f64, fixed order);  knn_order_kernel: final ordering
//                          by (f32(score) desc, index asc), write top-k.
// The exact rescoring makes the result independent of MFMA accumulation order: indices are
// bit-exact against oracle/knn.py as long as the true top-k lie inside the approximate top-KP
// (MFMA f32 error ~1e-6 vs. the k-th..KP-th score gap).
#include <math.h>
#include <stdlib.h>
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int KNN_QT = 64;         // queries per tile (4 waves x 16)
constexpr int KNN_CHUNK = 4096;    // candidates per workgroup of the register-held select levels
constexpr int KNN_STREAM_CHUNK = 8192;   // scores per workgroup of the streaming level-0 select

__host__ __device__ inline int knn_kp(int k) {
  int kp = 2 * k > k + 8 ? 2 * k : k + 8;
  return (kp + 7) / 8 * 8;
}

// ---------------------------------------------------------------------------------------------
// 1. score kernel.  Workgroup w owns gallery rows [N*w/nwg, N*(w+1)/nwg) — a balanced static
// partition over a grid that is fully resident (nwg <= CUs * WGPC), so every workgroup
// streams the same number of HBM bytes and they finish together; its rows are cut into equal
// tiles of <= KNN_TR rows.  Per tile and K-step (128 B of every row = 64 bf16 or 128 fp8): the
// gallery tile [th x 128 B] (HBM) and the query tile [64 x 128 B] (L2) land in LDS by LDS-DMA,
// 2-deep ring, one barrier per K-step.  Wave w multiplies queries 16w..16w+15 (MFMA A operand)
// with every 16-row gallery block (B operand): C[q][n], lane = gallery row n -> coalesced 64-B
// score stores.
// FP8 (OCP e4m3, per-row f32 scale): same LDS image and the same 16-B fragment reads; each 16-B
// chunk now holds 16 elements and feeds two v_mfma_f32_16x16x32_fp8_fp8 (low / high 8 bytes) —
// A and B use the same (permuted) k order, so the permutation cancels in the dot product — and
// the epilogue multiplies by the two row scales.
// ---------------------------------------------------------------------------------------------
template <bool FP8, int KNN_TR, int WGPC, int ABL>
__global__ __launch_bounds__(256, WGPC) void knn_scores_kernel(
    const void* __restrict__ Qv, const void* __restrict__ Gv, const float* __restrict__ q_scale,
    const float* __restrict__ g_scale, float* __restrict__ S, int B, int N, int row_bytes, int ldS) {
  constexpr int NB = KNN_TR / 16;
  constexpr int GG = KNN_TR / 8;            // gallery staging groups (8 rows each)
  constexpr int NGRP = GG + KNN_QT / 8;     // + query staging groups
  constexpr int GPW = (NGRP + 3) / 4;       // groups per wave (upper bound)
  constexpr int STAGE_BYTES = (KNN_TR + KNN_QT) * TILE_ROW_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char* Q = static_cast<const char*>(Qv);
  const char* G = static_cast<const char*>(Gv);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  const int qb = blockIdx.y;
  // balanced static partition in units of 4 rows (16-byte aligned score-row segments for the epilogue's float4 stores)
  const long long n4 = ((long long)N + 3) >> 2;
  const long long r_begin = 4 * (n4 * blockIdx.x / nwg);
  const long long r_end = min((long long)N, 4 * (n4 * (blockIdx.x + 1) / nwg));
  const int len = (int)(r_end - r_begin);
  if (len <= 0) return;
  const int ntile = (len + KNN_TR - 1) / KNN_TR;
  const int th_nom = ((len + ntile - 1) / ntile + 3) & ~3;      // <= KNN_TR (a multiple of 16)
  // K slice of this workgroup (small shards: gridDim.z > 1 slices, each into its own score slab, summed by the
  // level-0 select in slice order — see knn_ksplit)
  const int nk_all = row_bytes >> 7;
  const int k_lo = (int)((long long)nk_all * blockIdx.z / gridDim.z), k_hi = (int)((long long)nk_all * (blockIdx.z + 1) / gridDim.z);
  const int nk = k_hi - k_lo;
  Q += (long long)k_lo * 128;
  G += (long long)k_lo * 128;
  S += (long long)blockIdx.z * B * ldS;

  for (int t = 0; t < ntile; ++t) {
    const int row0 = (int)r_begin + t * th_nom;
    const int th = min(th_nom, (int)r_end - row0);   // valid rows of this tile
    if (th <= 0) break;                              // uniform; only for absurdly long row ranges

    // Source row pointers (bytes) of the staging groups this wave owns.
    const char* src[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int g = wave + 4 * i;
      const int tr = (g < GG ? g : g - GG) * 8 + (lane >> 3);   // row inside its tile
      const int sw = ((lane & 7) ^ ((tr >> 1) & 7)) << 4;
      if (g < GG) {
        const int r = row0 + min(tr, th - 1);
        src[i] = G + (long long)r * row_bytes + sw;
      } else {
        const int r = min(qb * KNN_QT + tr, B - 1);
        src[i] = Q + (long long)r * row_bytes + sw;
      }
    }
    const int gvalid = (th + 7) >> 3;   // gallery groups that hold at least one valid row

    auto stage = [&](int buf, int ks) {
      char* base = smem + buf * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < GPW; ++i) {
        const int g = wave + 4 * i;
        if (g < GG) {
          if (g < gvalid) glds16<(ABL & 4) ? 2 : 0>(src[i] + ks * 128, base + g * 8 * TILE_ROW_BYTES);
        } else if (g < NGRP) {
          if (!(ABL & 2) || ks == 0)
            glds16(src[i] + ks * 128, base + KNN_TR * TILE_ROW_BYTES + (g - GG) * 8 * TILE_ROW_BYTES);
        }
      }
    };

    f32x4 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();   // previous tile's last compute is done before its buffers are refilled
    stage(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (ks + 1 < nk) stage((ks + 1) & 1, ks + 1);
      const char* tg = smem + (ks & 1) * STAGE_BYTES;
      const char* tq = tg + KNN_TR * TILE_ROW_BYTES;
      const bf16x8 a0 = lds_frag(tq, 16 * wave + (lane & 15), (lane >> 4));
      const bf16x8 a1 = lds_frag(tq, 16 * wave + (lane & 15), 4 + (lane >> 4));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if ((ABL & 1) && ks > 0) break;
        if (nb * 16 < th) {
          const bf16x8 b0 = lds_frag(tg, 16 * nb + (lane & 15), (lane >> 4));
          const bf16x8 b1 = lds_frag(tg, 16 * nb + (lane & 15), 4 + (lane >> 4));
          if constexpr (FP8) {
            // one block-scaled MFMA (unit E8M0 scales) over the whole 128-byte K-step: a lane's 32 operand bytes
            // are its two 16-byte chunks (same lane -> K assignment for both operands); twice the rate of the
            // four v_mfma_f32_16x16x32_fp8_fp8 this replaces
            typedef __attribute__((ext_vector_type(8))) int i32x8;
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const i32x4 A0 = __builtin_bit_cast(i32x4, a0), A1 = __builtin_bit_cast(i32x4, a1);
            const i32x4 B0 = __builtin_bit_cast(i32x4, b0), B1 = __builtin_bit_cast(i32x4, b1);
            const i32x8 A = {A0[0], A0[1], A0[2], A0[3], A1[0], A1[1], A1[2], A1[3]};
            const i32x8 Bv = {B0[0], B0[1], B0[2], B0[3], B1[0], B1[1], B1[2], B1[3]};
            acc[nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, Bv, acc[nb], 0, 0, 0, 127, 0, 127);
          } else {
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[nb], 0, 0, 0);
          }
        }
      }
    }

    // C/D map of 16x16: col (gallery row) = lane&15, row (query) = 4*(lane>>4) + e.
    float qs[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (FP8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) qs[e] = q_scale[min(qb * KNN_QT + 16 * wave + 4 * (lane >> 4) + e, B - 1)];
    }
    if constexpr (ABL & 32) {
      // Score tile out through the (now idle) stage buffers as whole row segments: wave w owns query rows 16w..16w+15
      // of the tile on both sides, so only the first barrier (every wave is past its last operand read) is needed.
      // Pitch KNN_TR + 4 floats: the four query groups of a store land 16 banks apart (conflict-free), rows stay
      // 16-byte aligned.  One 1-KiB float4 store per query row instead of 16 dword stores of four 64-byte segments.
      constexpr int SP = KNN_TR + 4;
      static_assert(KNN_QT * SP * 4 <= 2 * STAGE_BYTES, "score tile must fit the stage buffers");
      float* st = reinterpret_cast<float*>(smem);
      __syncthreads();
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = 16 * nb + (lane & 15);
        if (n < th) {
          float gs = 1.f;
          if constexpr (FP8) gs = g_scale[row0 + n];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            st[(16 * wave + 4 * (lane >> 4) + e) * SP + n] = FP8 ? acc[nb][e] * qs[e] * gs : acc[nb][e];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int j = 0; j < 16; ++j) {
        const int q = qb * KNN_QT + 16 * wave + j;
        if (q >= B) break;
        float* dst = S + (long long)q * ldS + row0;
        const float* srow = st + (16 * wave + j) * SP;
        for (int c = lane * 4; c < th; c += 256) {
          if (c + 3 < th) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(srow + c);
            if constexpr (ABL & 16) __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(dst + c));
            else *reinterpret_cast<f32x4*>(dst + c) = v4;
          } else {
            for (int x = c; x < th; ++x) dst[x] = srow[x];
          }
        }
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = 16 * nb + (lane & 15);
        if (n < th) {
          float gs = 1.f;
          if constexpr (FP8) gs = g_scale[row0 + n];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = qb * KNN_QT + 16 * wave + 4 * (lane >> 4) + e;
            // ABL bit 3 (value 8, -DVPR_ABLATION builds only): no score stores (the impossible compare keeps the MFMAs alive)
            if ((ABL & 8) && acc[nb][e] != 12345.678f) continue;
            const float val = FP8 ? acc[nb][e] * qs[e] * gs : acc[nb][e];
            if (q < B) {
              if constexpr (ABL & 16) __builtin_nontemporal_store(val, &S[(long long)q * ldS + row0 + n]);
              else S[(long long)q * ldS + row0 + n] = val;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Block-wide iterative top-KP over <= 4096 (value, index) pairs held 16 per thread.
// Key order: value desc, index asc (packed into one u64 so a max-reduction does both).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long make_key(float v, int idx) {
  // idx >= 0 real entry; idx < 0 padding (sorts last among equal values, and carries -inf).
  return ((unsigned long long)f32_orderable(v) << 32) | (uint32_t)(0x7fffffff - idx);
}
__device__ __forceinline__ int key_idx(unsigned long long k) { return 0x7fffffff - (int)(uint32_t)k; }
__device__ __forceinline__ float key_val(unsigned long long k) { return f32_from_orderable((uint32_t)(k >> 32)); }

constexpr unsigned long long KEY_DEAD = 0ull;   // below every real key (orderable(-inf) > 0)
constexpr int SEL_CAP = 4096;                   // candidate list capacity (= elements per block)

struct SelectSmem {
  unsigned long long tmax[256];
  unsigned long long cand[SEL_CAP];
  unsigned long long outk[128];
  unsigned long long thr;
  int cnt;
};

// Block-wide top-kp of <= 4096 unique keys held 16 per thread, without a serial argmax loop:
//   a. every thread (or, for kp <= 64, every group of 4 threads) publishes the max of its keys;
//   b. the kp-th largest published max T (found by rank counting) is a lower bound of the kp-th
//      largest key overall, so every top-kp key is >= T;
//   c. keys >= T are appended to an LDS list (typically kp..2kp of them);
//   d. each listed key counts the listed keys above it = its rank; ranks < kp go to outk[rank].
// outk[0..kp) is complete (KEY_DEAD-padded) and visible to the whole block on return.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, true);
  return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}

__device__ __forceinline__ void block_select(const unsigned long long (&keys)[16], int kp, SelectSmem& sm) {
  const int tid = threadIdx.x;
  unsigned long long best = KEY_DEAD;
#pragma unroll
  for (int i = 0; i < 16; ++i) best = keys[i] > best ? keys[i] : best;
  if (tid == 0) { sm.thr = 1ull; sm.cnt = 0; }
  if (tid < kp) sm.outk[tid] = KEY_DEAD;
  if (kp <= 64) {
    // 64 group maxes (4 adjacent lanes each, DPP quad permutes) are enough to bound kp <= 64
    // keys, and ranking 64 values is 4x cheaper than ranking 256.
    unsigned long long o = dpp_u64<0xB1>(best);
    best = o > best ? o : best;
    o = dpp_u64<0x4E>(best);
    best = o > best ? o : best;
    if ((tid & 3) == 0) sm.tmax[tid >> 2] = best;
    __syncthreads();
    if (tid < 64) {
      const unsigned long long mine = sm.tmax[tid];
      int rank = 0;
      for (int s = 0; s < 64; s += 2) {
        const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(&sm.tmax[s]);
        rank += (t2.x > mine ? 1 : 0) + (t2.y > mine ? 1 : 0);
      }
      if (rank == kp - 1 && mine != KEY_DEAD) sm.thr = mine;   // keys are unique: one writer
    }
  } else {
    sm.tmax[tid] = best;
    __syncthreads();
    int rank = 0;
    for (int s = 0; s < 256; s += 2) {
      const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(&sm.tmax[s]);
      rank += (t2.x > best ? 1 : 0) + (t2.y > best ? 1 : 0);
    }
    if (rank == kp - 1 && best != KEY_DEAD) sm.thr = best;
  }
  __syncthreads();
  const unsigned long long T = sm.thr;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (keys[i] >= T) {
      const int pos = atomicAdd(&sm.cnt, 1);
      sm.cand[pos] = keys[i];
    }
  }
  __syncthreads();
  const int c = sm.cnt;
  for (int ci = tid; ci < c; ci += 256) {
    const unsigned long long mine = sm.cand[ci];
    int r = 0;
    for (int cj = 0; cj < c; ++cj) r += sm.cand[cj] > mine ? 1 : 0;
    if (r < kp) sm.outk[r] = mine;
  }
  __syncthreads();
}

// Branch-free load of up to 16 (value, index) pairs per thread: out-of-range positions are
// clamped for the load and turned into KEY_DEAD by a select (keeps the 16 keys in registers).
__device__ __forceinline__ void load_keys(unsigned long long (&keys)[16], const float* __restrict__ v,
                                          const int32_t* __restrict__ ix, int L, int base) {
  if (ix != nullptr) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = base + i * 256 + (int)threadIdx.x;
      const int pc = p < L ? p : L - 1;
      const int id = ix[pc];
      const float val = v[pc];
      keys[i] = (p < L && id >= 0) ? make_key(val, id) : KEY_DEAD;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = base + i * 256 + (int)threadIdx.x;
      const int pc = p < L ? p : L - 1;
      const float val = v[pc];
      keys[i] = p < L ? make_key(val, p) : KEY_DEAD;
    }
  }
}

// 2. select: grid (chunks, B).  in_idx == nullptr -> index = position; else explicit.
// Output [B][nchunk][kp] (value, index), KEY_DEAD -> (-inf, -1).
__global__ __launch_bounds__(256) void knn_select_kernel(
    const float* __restrict__ in_val, const int32_t* __restrict__ in_idx, int L, long long ld_in,
    float* __restrict__ out_val, int32_t* __restrict__ out_idx, int kp, int nchunk) {
  __shared__ SelectSmem sm;
  const int c = blockIdx.x, b = blockIdx.y;
  const float* v = in_val + (long long)b * ld_in;
  const int32_t* ix = in_idx ? in_idx + (long long)b * ld_in : nullptr;
  unsigned long long keys[16];
  load_keys(keys, v, ix, L, c * KNN_CHUNK);
  block_select(keys, kp, sm);
  if ((int)threadIdx.x < kp) {
    const unsigned long long k = sm.outk[threadIdx.x];
    const long long o = ((long long)b * nchunk + c) * kp + threadIdx.x;
    out_val[o] = k == KEY_DEAD ? -INFINITY : key_val(k);
    out_idx[o] = k == KEY_DEAD ? -1 : key_idx(k);
  }
}

// 2a. level-0 select straight from the score matrix: grid (chunks, B), chunk = `ch` scores (multiple of 1024,
// <= 8192).  The chunk is read ONCE: every thread requests its (up to) 8 float4 (32 scores, strided by 1024 so a wave
// reads 1 KiB lines) before the first compare and keeps them in registers for both phases:
//   max     the thread's best (value, then lower index) of its 32 scores, plain f32 compares in index-ascending
//           order (strict '>' keeps ties right);
//   rank    T = the kp-th largest of the 256 published thread maxes bounds the kp-th largest key of the chunk from
//           below; keys are unique, so at most 32 (kp-1) + 1 keys are >= T whatever the data — ties, sorted input,
//           all-equal scores: CAP = 1024 for kp <= 32 (k <= 16), 4096 otherwise;
//   filter  keys >= T (from the registers) appended to an LDS list (typically ~kp of them);
//   order   each listed key counts the listed keys above it = its rank; ranks < kp are written, rank-ordered.
// Round 1 read the chunk twice with dependent loads (8 round trips per pass) under 35 KB of LDS (4 workgroups per CU):
// 15.7 us at 100k rows, 117 us at 1M; this form needs 15 KB (kp <= 32) and one round trip.
template <int CAP>
__global__ __launch_bounds__(256) void knn_select_stream_kernel(
    const float* __restrict__ S, int N, long long ldS, int ch,
    float* __restrict__ out_val, int32_t* __restrict__ out_idx, int kp, int nchunk, int nslab, long long slab_stride) {
  __shared__ unsigned long long tmax[256];
  __shared__ unsigned long long wtop[4][128];
  __shared__ unsigned long long cand[CAP];
  __shared__ unsigned long long outk[128];
  __shared__ unsigned long long thr;
  __shared__ int cnt;
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* v = S + (long long)b * ldS;
  const int base = c * ch;
  const int len = min(ch, N - base);          // >= 1
  float4 q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid * 4 + i * 1024;
    // rows of S are padded to a multiple of 64 floats (ldS), so a float4 that starts inside the row stays inside it;
    // positions past `len` are masked below
    q[i] = p < len ? *reinterpret_cast<const float4*>(v + base + p) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // K-split shards (small N): the score is the sum of the slices' slabs, added in slice order (deterministic); the sum
  // goes back into slab 0 so that the workspace still holds THE score matrix afterwards (vpr_knn_scores_ptr)
  if (nslab > 1) {
    for (int z = 1; z < nslab; ++z) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = tid * 4 + i * 1024;
        if (p < len) {
          const float4 t = *reinterpret_cast<const float4*>(v + z * slab_stride + base + p);
          q[i].x += t.x; q[i].y += t.y; q[i].z += t.z; q[i].w += t.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid * 4 + i * 1024;
      if (p < len) *reinterpret_cast<float4*>(const_cast<float*>(v) + base + p) = q[i];
    }
  }
  float bv = -INFINITY;
  int bi = -1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid * 4 + i * 1024;
    const float qq[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (p + e < len && (qq[e] > bv || bi < 0)) { bv = qq[e]; bi = base + p + e; }
  }
  const unsigned long long best = bi >= 0 ? make_key(bv, bi) : KEY_DEAD;
  if (tid == 0) { thr = 1ull; cnt = 0; }
  if (tid < kp) outk[tid] = KEY_DEAD;
  for (int i = tid; i < 4 * 128; i += 256) wtop[i >> 7][i & 127] = KEY_DEAD;
  tmax[tid] = best;
  __syncthreads();
  // kp-th largest of the 256 thread maxes without the 256 x 256 all-pairs ranking (VALU-bound: 100 us at 1M rows):
  // rank inside the own wave (64 broadcast reads), the top kp of every wave go to a sorted list, and the global rank
  // of a listed key = own rank + binary searches in the other three lists.
  const int wv = tid >> 6;
  int wr = 0;
  for (int s = 0; s < 64; s += 2) {
    const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(&tmax[wv * 64 + s]);
    wr += (t2.x > best ? 1 : 0) + (t2.y > best ? 1 : 0);
  }
  const bool listed = best != KEY_DEAD && wr < kp;
  if (listed) wtop[wv][wr] = best;
  __syncthreads();
  if (listed) {
    int gr = wr;
    for (int w2 = 0; w2 < 4; ++w2) {
      if (w2 == wv) continue;
      int lo = 0, hi = kp;                      // descending list, KEY_DEAD (= 0) padding at the end
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (wtop[w2][mid] > best) lo = mid + 1; else hi = mid;
      }
      gr += lo;
    }
    if (gr == kp - 1) thr = best;               // unique keys: one writer
  }
  __syncthreads();
  const unsigned long long T = thr;
  // key >= T  <=>  value > Tv, or value == Tv and index <= Ti  (T == 1: no threshold, take all)
  const bool all = T == 1ull;
  const float Tv = all ? -INFINITY : key_val(T);
  const int Ti = all ? 0x7fffffff : key_idx(T);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid * 4 + i * 1024;
    const float qq[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = qq[e];
      if (p + e < len && (x > Tv || (x == Tv && base + p + e <= Ti) || all))
        cand[atomicAdd(&cnt, 1)] = make_key(x, base + p + e);
    }
  }
  __syncthreads();
  const int cn = cnt;
  for (int ci = tid; ci < cn; ci += 256) {
    const unsigned long long mine = cand[ci];
    int r = 0;
    for (int cj = 0; cj < cn; ++cj) r += cand[cj] > mine ? 1 : 0;
    if (r < kp) outk[r] = mine;
  }
  __syncthreads();
  if (tid < kp) {
    const unsigned long long k = outk[tid];
    const long long o = ((long long)b * nchunk + c) * kp + tid;
    out_val[o] = k == KEY_DEAD ? -INFINITY : key_val(k);
    out_idx[o] = k == KEY_DEAD ? -1 : key_idx(k);
  }
}

// 3a. exact rescoring: grid (kp, B), one workgroup per (candidate, query).  cand_idx [B][kp] are
// the approximate top-kp gallery rows (-1 = padding).  exact score = sum of bf16*bf16 products
// in f64 (each product is exact), in a fixed order (thread-strided 16-B chunks in sequence, wave
// butterfly, then the 4 wave sums) -> independent of how the candidate was found.  All of a
// thread's loads are issued before the first FMA: one trip to HBM per row.
constexpr int RS_U = 6;   // 16-B chunks per thread per trip
__device__ __forceinline__ double dot16_bf16(const s16x8& qa, const s16x8& ga, double acc) {
#pragma unroll
  for (int j = 0; j < 8; ++j)
    acc = fma((double)bf16_bits_to_f32((uint16_t)qa[j]), (double)bf16_bits_to_f32((uint16_t)ga[j]), acc);
  return acc;
}
__device__ __forceinline__ double dot16_fp8(const s16x8& qa, const s16x8& ga, double acc) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const i32x4 qi = __builtin_bit_cast(i32x4, qa), gi = __builtin_bit_cast(i32x4, ga);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    acc = fma((double)__builtin_amdgcn_cvt_f32_fp8(qi[w], 0), (double)__builtin_amdgcn_cvt_f32_fp8(gi[w], 0), acc);
    acc = fma((double)__builtin_amdgcn_cvt_f32_fp8(qi[w], 1), (double)__builtin_amdgcn_cvt_f32_fp8(gi[w], 1), acc);
    acc = fma((double)__builtin_amdgcn_cvt_f32_fp8(qi[w], 2), (double)__builtin_amdgcn_cvt_f32_fp8(gi[w], 2), acc);
    acc = fma((double)__builtin_amdgcn_cvt_f32_fp8(qi[w], 3), (double)__builtin_amdgcn_cvt_f32_fp8(gi[w], 3), acc);
  }
  return acc;
}

// Sum of squares of one 16-B operand chunk (f32; feeds the error bound of the certification step only).
__device__ __forceinline__ float sumsq16_bf16(const s16x8& a) {
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float x = bf16_bits_to_f32((uint16_t)a[j]); t = fmaf(x, x, t); }
  return t;
}
__device__ __forceinline__ float sumsq16_fp8(const s16x8& a) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const i32x4 w = __builtin_bit_cast(i32x4, a);
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float x = __builtin_amdgcn_cvt_f32_fp8(w[i], 0); t = fmaf(x, x, t);
    x = __builtin_amdgcn_cvt_f32_fp8(w[i], 1); t = fmaf(x, x, t);
    x = __builtin_amdgcn_cvt_f32_fp8(w[i], 2); t = fmaf(x, x, t);
    x = __builtin_amdgcn_cvt_f32_fp8(w[i], 3); t = fmaf(x, x, t);
  }
  return t;
}

template <bool FP8>
__global__ __launch_bounds__(256) void knn_rescore_kernel(
    const int32_t* __restrict__ cand_idx, const void* __restrict__ Qv, const void* __restrict__ Gv,
    const float* __restrict__ q_scale, const float* __restrict__ g_scale,
    int row_bytes, int kp, float* __restrict__ exact, float* __restrict__ qnorm) {
  __shared__ double red[4];
  __shared__ float redq[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const int id = cand_idx[(long long)b * kp + c];
  if (id < 0 && c != 0) {   // uniform (candidate 0 always runs: it also leaves |q| for the certification step)
    if (threadIdx.x == 0) exact[(long long)b * kp + c] = -INFINITY;
    return;
  }
  const char* qrow = static_cast<const char*>(Qv) + (long long)b * row_bytes;
  const char* grow = static_cast<const char*>(Gv) + (long long)(id < 0 ? 0 : id) * row_bytes;
  const int nchunks = row_bytes >> 4;
  double acc = 0.0;
  float qq = 0.f;
  for (int ch0 = threadIdx.x; ch0 < nchunks; ch0 += 256 * RS_U) {
    s16x8 qa[RS_U], ga[RS_U];
#pragma unroll
    for (int u = 0; u < RS_U; ++u) {
      const int ch = ch0 + 256 * u;
      const int chc = ch < nchunks ? ch : ch0;          // clamp: always a valid address
      qa[u] = *reinterpret_cast<const s16x8*>(qrow + chc * 16);
      ga[u] = *reinterpret_cast<const s16x8*>(grow + chc * 16);
    }
#pragma unroll
    for (int u = 0; u < RS_U; ++u) {
      if (ch0 + 256 * u < nchunks) {
        acc = FP8 ? dot16_fp8(qa[u], ga[u], acc) : dot16_bf16(qa[u], ga[u], acc);
        if (c == 0) qq += FP8 ? sumsq16_fp8(qa[u]) : sumsq16_bf16(qa[u]);
      }
    }
  }
  acc = wave_sum_f64(acc);
  qq = wave_sum(qq);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; redq[threadIdx.x >> 6] = qq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = (red[0] + red[1]) + (red[2] + red[3]);
    if (FP8 && id >= 0) tot = (tot * (double)q_scale[b]) * (double)g_scale[id];   // same order as oracle/knn.py
    exact[(long long)b * kp + c] = id < 0 ? -INFINITY : (float)tot;
    if (c == 0) {
      const float n2 = (redq[0] + redq[1]) + (redq[2] + redq[3]);
      qnorm[b] = sqrtf(n2) * (FP8 ? fabsf(q_scale[b]) : 1.0f);
    }
  }
}

// 3c. fused final stage: grid (B), 1024 threads.  Takes the level-0 candidates (<= 4096 per
// query), selects the top-kp by MFMA score, rescoring them exactly (16 waves, every load of a
// row issued before its first FMA, the query row staged once in LDS) and writes the ordered
// top-k — one launch instead of select + rescore + order.  Same arithmetic as the split kernels
// (same per-lane chunk order, same butterfly), so the results are bit-identical to them.
constexpr int FF_NT = 512;        // 8 waves
constexpr int FF_CPW = 3;         // candidates a wave rescoring together (all their loads in flight)
constexpr int FF_MAXCH = 9;       // 16-B chunks per lane per row in flight per trip
constexpr int FF_MAXC = 256;      // rescored candidates per query: kp + what the certification step adds
struct FinalSmem {
  unsigned long long tmax[256];
  unsigned long long cand[SEL_CAP];
  unsigned long long outk[FF_MAXC];
  unsigned long long thr;
  int cnt;
  float exact[FF_MAXC];
  float red[8];
  float ek;         // k-th best exact score of the rescored set
  int nvalid;       // real (non-padding) entries of the rescored set
  int extra;        // candidates the certification step adds
  int flag;         // 0 certified, 1 widen, 2 cannot be certified from the lists in hand
};

// Certification (the exactness contract, made checkable).  The result is the exact top-k iff no row OUTSIDE the
// rescored set has an exact score >= the k-th best exact score e_k of the set.  Every such row lost an
// approximate-score comparison: its MFMA score is <= a_min, the smallest approximate score kept (level-0 lists and
// the top-kp cut both keep the largest keys), and |MFMA score - exact score| <= eps = err_rel * |q| for gallery
// rows inside the norm bound the caller folded into err_rel (any summation order of D f32 additions of exact
// products: gamma_D * sum|q_i g_i| <= D 2^-24 |q| |g|).  So  e_k - eps > a_min  certifies the answer.
// Otherwise the set is widened to every level-0 candidate with MFMA score >= e_k - eps (they are rescored exactly
// too, status 1); if a level-0 chunk list is itself cut above that bar (>= kp rows of one chunk inside the band),
// or the widened set does not fit, the query is flagged (status 2) and the host re-runs it on exact scores
// (vpr_knn_exact_scores): ops.knn_topk(..., exact_fallback=True).
template <bool FP8>
__global__ __launch_bounds__(FF_NT) void knn_final_fused_kernel(
    const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx, int L,
    const void* __restrict__ Qv, const void* __restrict__ Gv, const float* __restrict__ q_scale,
    const float* __restrict__ g_scale, int row_bytes, int k, int kp, int index_base,
    float* __restrict__ out_val, int32_t* __restrict__ out_idx,
    float err_rel, int level0_lists, int32_t* __restrict__ status, int32_t* __restrict__ uncertified) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];      // [row_bytes] query row
  __shared__ FinalSmem sm;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // stage the query row (16-B chunks); |q|^2 on the way
  const char* qrow = static_cast<const char*>(Qv) + (long long)b * row_bytes;
  const int nchunks = row_bytes >> 4;
  float qq = 0.f;
  for (int ch = tid; ch < nchunks; ch += FF_NT) {
    const s16x8 c16 = *reinterpret_cast<const s16x8*>(qrow + ch * 16);
    *reinterpret_cast<s16x8*>(dyn + ch * 16) = c16;
    qq += FP8 ? sumsq16_fp8(c16) : sumsq16_bf16(c16);
  }
  qq = wave_sum(qq);
  if (lane == 0) sm.red[wave] = qq;

  // ---- top-kp of the candidates (8 keys per thread) ----
  unsigned long long keys[8];
  const float* v = cand_val + (long long)b * L;
  const int32_t* ix = cand_idx + (long long)b * L;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = i * FF_NT + tid;
    const int pc = p < L ? p : L - 1;
    const int id = ix[pc];
    const float val = v[pc];
    keys[i] = (p < L && id >= 0) ? make_key(val, id) : KEY_DEAD;
  }
  unsigned long long best = keys[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) best = keys[i] > best ? keys[i] : best;
  if (tid == 0) { sm.thr = 1ull; sm.cnt = 0; sm.extra = 0; sm.flag = 0; sm.nvalid = 0; sm.ek = -INFINITY; }
  if (tid < FF_MAXC) sm.outk[tid] = KEY_DEAD;
  {  // 256 group maxes: pairs of threads (quad-perm xor 1)
    const unsigned long long o = dpp_u64<0xB1>(best);
    best = o > best ? o : best;
    if ((tid & 1) == 0) sm.tmax[tid >> 1] = best;
  }
  __syncthreads();
  if (tid < 256) {
    const unsigned long long mine = sm.tmax[tid];
    int rank = 0;
    for (int s2 = 0; s2 < 256; s2 += 2) {
      const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(&sm.tmax[s2]);
      rank += (t2.x > mine ? 1 : 0) + (t2.y > mine ? 1 : 0);
    }
    if (rank == kp - 1 && mine != KEY_DEAD) sm.thr = mine;
  }
  __syncthreads();
  const unsigned long long T = sm.thr;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (keys[i] >= T) sm.cand[atomicAdd(&sm.cnt, 1)] = keys[i];
  __syncthreads();
  const int cn = sm.cnt;
  for (int ci = tid; ci < cn; ci += FF_NT) {
    const unsigned long long mine = sm.cand[ci];
    int r = 0;
    for (int cj = 0; cj < cn; ++cj) r += sm.cand[cj] > mine ? 1 : 0;
    if (r < kp) sm.outk[r] = mine;
  }
  __syncthreads();

  // ---- exact rescoring of outk[c_lo, c_hi): wave w takes candidates c_lo + w + 8*j; FF_CPW of them per trip, with
  // every 16-B chunk of all their rows requested before the first FMA (one HBM round trip per trip) ----
  auto rescore = [&](int c_lo, int c_hi) {
    for (int c0 = c_lo + wave; c0 < c_hi; c0 += 8 * FF_CPW) {
      int id[FF_CPW];
      const char* grow[FF_CPW];
      double acc[FF_CPW];
#pragma unroll
      for (int j = 0; j < FF_CPW; ++j) {
        const int c = c0 + 8 * j;
        const unsigned long long key = c < c_hi ? sm.outk[c] : KEY_DEAD;
        id[j] = key == KEY_DEAD ? -1 : key_idx(key);
        grow[j] = static_cast<const char*>(Gv) + (long long)(id[j] < 0 ? 0 : id[j]) * row_bytes;
        acc[j] = 0.0;
      }
      for (int base = 0; base < nchunks; base += 64 * FF_MAXCH) {   // 2 trips at D = 8448
        s16x8 ga[FF_CPW][FF_MAXCH];
#pragma unroll
        for (int j = 0; j < FF_CPW; ++j)
#pragma unroll
          for (int u = 0; u < FF_MAXCH; ++u) {
            const int ch = base + lane + 64 * u;
            ga[j][u] = *reinterpret_cast<const s16x8*>(grow[j] + (ch < nchunks ? ch : 0) * 16);
          }
#pragma unroll
        for (int u = 0; u < FF_MAXCH; ++u) {
          const int ch = base + lane + 64 * u;
          if (ch < nchunks) {
            const s16x8 qa = *reinterpret_cast<const s16x8*>(dyn + ch * 16);
#pragma unroll
            for (int j = 0; j < FF_CPW; ++j)
              acc[j] = FP8 ? dot16_fp8(qa, ga[j][u], acc[j]) : dot16_bf16(qa, ga[j][u], acc[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < FF_CPW; ++j) {
        const int c = c0 + 8 * j;
        double tot = wave_sum_f64(acc[j]);
        if (FP8 && id[j] >= 0) tot = (tot * (double)q_scale[b]) * (double)g_scale[id[j]];
        if (lane == 0 && c < c_hi) sm.exact[c] = id[j] < 0 ? -INFINITY : (float)tot;
      }
    }
  };
  // ---- final order of outk[0, tot) by (f32(exact) desc, index asc); leaves e_k and the entry count ----
  auto order = [&](int tot) {
    if (tid < tot) {
      const unsigned long long ki = sm.outk[tid];
      int rank;
      if (ki == KEY_DEAD) {
        rank = FF_MAXC + tid;                      // padding: behind every real entry
      } else {
        const unsigned long long mine = make_key(sm.exact[tid], key_idx(ki));
        rank = 0;
        for (int j = 0; j < tot; ++j) {
          const unsigned long long kj = sm.outk[j];
          rank += (kj != KEY_DEAD && make_key(sm.exact[j], key_idx(kj)) > mine) ? 1 : 0;
        }
        atomicAdd(&sm.nvalid, 1);
        if (rank == k - 1) sm.ek = sm.exact[tid];
      }
      if (rank < k) {
        out_val[(long long)b * k + rank] = sm.exact[tid];
        out_idx[(long long)b * k + rank] = key_idx(ki) + index_base;
      }
    }
  };

  rescore(0, kp);
  __syncthreads();
  order(kp);
  __syncthreads();
  const int nvalid = sm.nvalid;
  if (tid < k && tid >= nvalid) {                  // fewer real rows than k: (-inf, -1) tail
    out_val[(long long)b * k + tid] = -INFINITY;
    out_idx[(long long)b * k + tid] = -1;
  }
  // ---- certification ----
  const float qnorm = sqrtf(((sm.red[0] + sm.red[1]) + (sm.red[2] + sm.red[3])) + ((sm.red[4] + sm.red[5]) + (sm.red[6] + sm.red[7]))) *
                      (FP8 ? fabsf(q_scale[b]) : 1.0f);
  const float eps = err_rel * qnorm;
  const unsigned long long kmin = sm.outk[kp - 1];               // smallest approximate key kept (KEY_DEAD: list not full)
  const float bar = sm.ek - eps;
  __syncthreads();                                               // everybody has read nvalid / ek before they are reused
  if (kmin == KEY_DEAD || bar > key_val(kmin)) {                 // nothing was cut, or the margin covers the MFMA error
    if (tid == 0 && status) status[b] = 0;
    return;
  }
  // widen: every level-0 candidate outside the top-kp whose approximate score reaches the bar
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (keys[i] != KEY_DEAD && keys[i] < kmin && key_val(keys[i]) >= bar) {
      const int pos = atomicAdd(&sm.extra, 1);
      if (kp + pos < FF_MAXC) sm.outk[kp + pos] = keys[i];
    }
  }
  if (level0_lists) {                                            // a chunk list cut above the bar hides rows of the band
    const int nch = L / kp;
    for (int c = tid; c < nch; c += FF_NT) {
      const int last = c * kp + kp - 1;
      if (ix[last] >= 0 && v[last] >= bar) sm.flag = 2;
    }
  } else if (tid == 0) {
    sm.flag = 2;                                                 // lists of a later select level: per-chunk cuts unknown here
  }
  if (tid == 0) sm.nvalid = 0;
  __syncthreads();
  int extra = sm.extra;
  const bool overflow = kp + extra > FF_MAXC;
  if (overflow) extra = FF_MAXC - kp;
  rescore(kp, kp + extra);
  __syncthreads();
  order(kp + extra);
  if (tid == 0) {
    const int st = (sm.flag == 2 || overflow) ? 2 : 1;
    if (status) status[b] = st;
    if (st == 2 && uncertified) atomicAdd(uncertified, 1);
  }
}

// 3b. final order: grid (B), 128 threads.  Rank the kp rescored candidates by
// (f32(exact) desc, index asc) by counting; write the top-k with global indices.  Certification as in the
// fused kernel, without the widening step (this path serves rows wider than its LDS budget and galleries of
// more than ~1.4M rows per shard): status 0 or 2.
__global__ __launch_bounds__(128) void knn_order_kernel(
    const int32_t* __restrict__ cand_idx, const float* __restrict__ exact, int k, int kp, int index_base,
    float* __restrict__ out_val, int32_t* __restrict__ out_idx,
    const float* __restrict__ cand_approx, const float* __restrict__ qnorm, float err_rel,
    int32_t* __restrict__ status, int32_t* __restrict__ uncertified) {
  __shared__ unsigned long long key[128];
  __shared__ float ek;
  const int b = blockIdx.x, i = threadIdx.x;
  int id = -1;
  unsigned long long mine = KEY_DEAD;
  if (i < kp) {
    id = cand_idx[(long long)b * kp + i];
    if (id >= 0) mine = make_key(exact[(long long)b * kp + i], id);
  }
  key[i] = mine;
  if (i == 0) ek = -INFINITY;
  __syncthreads();
  if (i < kp) {
    int rank = 0;
    if (id < 0) {
      rank = i;     // padding already sits behind every real entry (select output is ordered)
    } else {
      for (int j = 0; j < kp; ++j) rank += key[j] > mine ? 1 : 0;
      if (rank == k - 1) ek = key_val(mine);
    }
    if (rank < k) {
      out_val[(long long)b * k + rank] = id < 0 ? -INFINITY : key_val(mine);
      out_idx[(long long)b * k + rank] = id < 0 ? -1 : id + index_base;
    }
  }
  __syncthreads();
  if (i == 0 && (status || uncertified)) {
    const bool full = cand_idx[(long long)b * kp + kp - 1] >= 0;          // the list is rank-ordered: last entry = a_min
    const bool ok = !full || ek - err_rel * qnorm[b] > cand_approx[(long long)b * kp + kp - 1];
    if (status) status[b] = ok ? 0 : 2;
    if (!ok && uncertified) atomicAdd(uncertified, 1);
  }
}

// Exact score matrix (the fallback of a query the certification step flags): S[b, n] = f32 of the exact (f64) dot
// product, as the rescoring kernels compute it.  One wave per gallery row (its 16-B chunks held in registers, <= 17
// per lane), looped over the (few) queries; grid = ceil(N / 4) workgroups of 4 waves.  HBM-bound on the gallery
// for a handful of queries; a rare path, not tuned further.
constexpr int EX_MAXCH = 17;      // 16-B chunks per lane: rows up to 17408 bytes
template <bool FP8>
__global__ __launch_bounds__(256) void knn_exact_scores_kernel(
    const void* __restrict__ Qv, const void* __restrict__ Gv, const float* __restrict__ q_scale,
    const float* __restrict__ g_scale, float* __restrict__ S, int B, int N, int row_bytes, int ldS) {
  const int lane = threadIdx.x & 63;
  const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int nchunks = row_bytes >> 4;
  const char* grow = static_cast<const char*>(Gv) + n * row_bytes;
  s16x8 ga[EX_MAXCH];
#pragma unroll
  for (int u = 0; u < EX_MAXCH; ++u) {
    const int ch = lane + 64 * u;
    ga[u] = *reinterpret_cast<const s16x8*>(grow + (ch < nchunks ? ch : 0) * 16);
  }
  const float gs = FP8 ? g_scale[n] : 1.f;
  for (int b = 0; b < B; ++b) {
    const char* qrow = static_cast<const char*>(Qv) + (long long)b * row_bytes;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < EX_MAXCH; ++u) {
      const int ch = lane + 64 * u;
      if (ch < nchunks) {
        const s16x8 qa = *reinterpret_cast<const s16x8*>(qrow + ch * 16);
        acc = FP8 ? dot16_fp8(qa, ga[u], acc) : dot16_bf16(qa, ga[u], acc);
      }
    }
    double tot = wave_sum_f64(acc);
    if (FP8) tot = (tot * (double)q_scale[b]) * (double)gs;
    if (lane == 0) S[(long long)b * ldS + n] = (float)tot;
  }
}

// Merge of per-shard lists: vals/idxs [shards, B, k] -> [B, k].  shards*k <= 4096.
__global__ __launch_bounds__(256) void topk_merge_kernel(
    const float* __restrict__ vals, const int32_t* __restrict__ idxs, int shards, int B, int k,
    float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  __shared__ SelectSmem sm;
  const int b = blockIdx.x;
  const int L = shards * k;
  unsigned long long keys[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = i * 256 + (int)threadIdx.x;
    const int pc = p < L ? p : L - 1;
    const int s = pc / k, j = pc % k;
    const long long o = ((long long)s * B + b) * k + j;
    const int id = idxs[o];
    const float val = vals[o];
    keys[i] = (p < L && id >= 0) ? make_key(val, id) : KEY_DEAD;
  }
  block_select(keys, k, sm);
  if ((int)threadIdx.x < k) {
    const unsigned long long key = sm.outk[threadIdx.x];
    out_val[(long long)b * k + threadIdx.x] = key == KEY_DEAD ? -INFINITY : key_val(key);
    out_idx[(long long)b * k + threadIdx.x] = key == KEY_DEAD ? -1 : key_idx(key);
  }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct KnnPlan {
  int Bpad, ldS, kp, ch0;
  int ks_max;            // score slabs the workspace holds (K-split of shards below 106k rows)
  int nrt;               // row tiles of the K-split form
  int nlevel;            // number of select levels before the final kernel
  int L[4], nchunk[4];   // input length / chunk count per level
  size_t off_S, off_cv[2], off_ci[2], off_qn, total;
};

static bool knn_plan(int B, int N, int D, int k, KnnPlan* p) {
  if (B <= 0 || N <= 0 || D <= 0 || k <= 0 || k > 64 || (D % 64) != 0) return false;
  p->Bpad = (B + KNN_QT - 1) / KNN_QT * KNN_QT;
  p->ldS = (N + 63) / 64 * 64;
  p->kp = knn_kp(k);
  // select levels: level 0 streams the score matrix in chunks of ch0 (<= 8192) scores, later
  // levels hold <= 4096 candidates per workgroup in registers; each level turns L entries per
  // query into nchunk*kp, and the last level has one chunk -> the [B][kp] list to rescore.
  p->ch0 = N > KNN_STREAM_CHUNK ? KNN_STREAM_CHUNK : (N + 1023) / 1024 * 1024;
  p->nlevel = 0;
  int L = N;
  for (;;) {
    if (p->nlevel >= 4) return false;
    const int chunk = p->nlevel == 0 ? p->ch0 : KNN_CHUNK;
    p->L[p->nlevel] = L;
    p->nchunk[p->nlevel] = (L + chunk - 1) / chunk;
    L = p->nchunk[p->nlevel] * p->kp;
    ++p->nlevel;
    if (p->nchunk[p->nlevel - 1] == 1) break;
  }
  // Small and medium shards (N < 106k rows: fewer 208-row tiles than the 512 resident workgroup slots).  Cutting the rows
  // finer to fill the chip multiplies the query traffic (every workgroup re-stages all 64 query rows: at 6k rows 16-row
  // tiles read 431 MB of queries for 108 MB of gallery) and leaves each tile a 132-step latency chain.  Instead: tall row
  // tiles (as few as fill the slots together with the K split, at least min(32, N/16)), and the K range of a tile split
  // over up to 16 workgroups, each writing its own score slab; the level-0 select adds the slabs.
  {
    const int by208 = (N + 207) / 208, by16 = (N + 15) / 16;
    const int fill = by16 < 32 ? by16 : 32;
    p->nrt = by208 > fill ? by208 : fill;
    int ks = 512 / p->nrt;
    p->ks_max = ks < 1 ? 1 : (ks > 16 ? 16 : ks);
    // gathered batches (> 64 queries) on the 256 x 256-tile GEMM: K slices until its one-per-CU workgroups fill 256 CUs
    if (B > KNN_QT) {
      const long long tiles = (long long)((B + 255) / 256) * ((N + 255) / 256);
      ks = tiles >= 256 ? 1 : (int)(256 / tiles);
      p->ks_max = ks > 4 ? 4 : ks;
    }
  }
  size_t off = 0;
  p->off_S = off;
  off += align_up((size_t)B * p->ldS * sizeof(float) * p->ks_max, 256);
  const size_t cand = (size_t)B * p->nchunk[0] * p->kp;
  for (int i = 0; i < 2; ++i) {
    p->off_cv[i] = off; off += align_up(cand * sizeof(float), 256);
    p->off_ci[i] = off; off += align_up(cand * sizeof(int32_t), 256);
  }
  p->off_qn = off; off += align_up((size_t)B * sizeof(float), 256);      // |q| per query (general rescore path)
  p->total = off;
  return true;
}

static int num_cus() { return device_cu_count(); }      // per device (vpr_internal.h)

struct KnnOperands {          // bf16: scales are null; fp8: per-row f32 scales
  const void* q; const void* g; const float* q_scale; const float* g_scale; bool fp8;
};

static int knn_check(const KnnOperands& o, int D) {
  if (!o.q || !o.g) return VPR_ERR_INVALID_ARG;
  if (o.fp8 && (!o.q_scale || !o.g_scale)) return VPR_ERR_INVALID_ARG;
  if (o.fp8 && (D % 128) != 0) return VPR_ERR_UNSUPPORTED;           // 128-B K-steps
  if ((reinterpret_cast<uintptr_t>(o.q) | reinterpret_cast<uintptr_t>(o.g)) & 15) return VPR_ERR_UNSUPPORTED;
  return VPR_OK;
}

// rows per workgroup of the fully resident grid (2 per CU) above one 208-row tile
static bool knn_tall_tiles(int N) {
  const int slots = num_cus() * 2;
  return (N + slots - 1) / slots > 208;
}

// more than one 256-row tile per workgroup of the resident grid (N > 131k rows)
static bool knn_multi_tile(int N) {
  const int slots = num_cus() * 2;
  return (N + slots - 1) / slots > 256;
}

// 256-row query tiles pay when at least 3/4 of their rows are real queries (256 gathered queries = 4 GPUs, 512 = 8 GPUs);
// VPR_KNN_FP8_GEMM256=0 forces the 128 x 128 kernel (A/B).
static bool knn_query_tile256_pays(int B) {
  if (tune_or(TUNE_KNN_FP8_GEMM256, 1) == 0) return false;
  const int tiles = (B + 255) / 256;
  return 4 * B >= 3 * tiles * 256;
}

static int knn_ksplit(const KnnPlan& p, int row_bytes) {
  int ks = p.ks_max;
  const int nk = row_bytes >> 7;
  if (ks > nk / 8) ks = nk / 8;                 // a slice keeps at least 8 K-steps
  return ks < 1 ? 1 : ks;
}

// Which kernel scores a [B] x [N] problem, and into how many K-slice slabs (the level-0 select adds them).  One rule,
// used by the score stage, the select stage and vpr_knn_scores_kernel_name.
enum KnnRoute { ROUTE_STREAM, ROUTE_GEMM128, ROUTE_GEMM256 };
static KnnRoute knn_route(const KnnPlan& p, bool fp8, int B, int N, int D, bool ksplit_ok, int* nslab) {
  const int gemm_min_b = tune_or(TUNE_KNN_GEMM_MIN_B, 65);     // A/B switch for the stream / GEMM crossover
  const int rb = fp8 ? D : D * 2;
  *nslab = 1;
  if (B < gemm_min_b) {                                 // streaming kernel; shards below 106k rows split K over tall tiles
    if (ksplit_ok) *nslab = knn_ksplit(p, rb);
    return ROUTE_STREAM;
  }
  if (!knn_query_tile256_pays(B) || (fp8 && D < 256)) return ROUTE_GEMM128;
  // 256 x 256 tiles, one workgroup per CU: with the K split a 512 x 12.5k problem (98 tiles) runs as 196 workgroups.
  // Without it (stand-alone score entry point, or VPR_KNN_GEMM_KSPLIT=0) bf16 needs >= 256 tiles to beat the 128 x 128
  // kernel's 4x as many workgroups.
  int ks = (ksplit_ok && B > KNN_QT && tune_or(TUNE_KNN_GEMM_KSPLIT, 1) != 0) ? knn_ksplit(p, rb) : 1;
  const long long tiles = (long long)((B + 255) / 256) * ((N + 255) / 256);
  if (!fp8 && tiles * ks < 192) return ROUTE_GEMM128;
  *nslab = ks;
  return ROUTE_GEMM256;
}

// ksplit_ok: the caller's select stage will add the slabs (vpr_knn_topk*); the stand-alone score entry point writes
// the one score matrix its contract promises.
int knn_scores(const KnnOperands& o, int B, int N, int D, void* ws, size_t ws_bytes, int k_for_plan,
               hipStream_t stream, bool ksplit_ok = false) {
  KnnPlan p;
  if (!ws) return VPR_ERR_INVALID_ARG;
  if (!knn_plan(B, N, D, k_for_plan, &p)) return VPR_ERR_UNSUPPORTED;
  const int st = knn_check(o, D);
  if (st != VPR_OK) return st;
  if (ws_bytes < p.total) return VPR_ERR_WORKSPACE;
  float* S = reinterpret_cast<float*>(static_cast<char*>(ws) + p.off_S);
  // More than one 64-query tile against a shard (the all-gathered batch of a multi-GPU job) moves
  // towards a compute-bound GEMM: the streaming kernel makes one gallery pass per 64 queries, the
  // 128x128-tile MFMA GEMM one per 128.  Measured, stream vs GEMM: 128 x 50k 367 / 222 us,
  // 192 x 33k 406 / 300, 256 x 25k 477 / 185, 512 x 12.5k 702 / 151 (scripts/knn_b_sweep.py).
  // >= 192 of every 256 gathered queries real (256 at 4 GPUs, 512 at 8): the 256 x 256-tile kernel with its LDS-DMA
  // stream kept in flight across barriers (gemm256.hip); a 128-query batch would leave half of such a tile row empty.
  int nslab = 1;
  const KnnRoute route = knn_route(p, o.fp8, B, N, D, ksplit_ok, &nslab);
  const long long slab_stride = (long long)B * p.ldS;
  if (route == ROUTE_GEMM256 && o.fp8)
    return launch_gemm256_fp8(static_cast<const uint8_t*>(o.q), D, o.q_scale, static_cast<const uint8_t*>(o.g), D,
                              o.g_scale, S, p.ldS, B, N, D, stream, nslab, slab_stride);
  if (route == ROUTE_GEMM128 && o.fp8)      // block-scaled fp8 MFMA GEMM (twice the bf16 rate), scales in its epilogue
    return launch_gemm_nt_fp8(static_cast<const uint8_t*>(o.q), D, o.q_scale, static_cast<const uint8_t*>(o.g), D,
                              o.g_scale, S, p.ldS, B, N, D, stream);
  if (route == ROUTE_GEMM256) {
    GemmProblem g{static_cast<const uint16_t*>(o.q), D, 0, 0, static_cast<const uint16_t*>(o.g), D, nullptr, 0,
                  S, p.ldS, 0, B, N, D, 0, 0};
    g.ksplit = nslab; g.slab_stride = slab_stride;
    return launch_gemm256(g, stream);
  }
  if (route == ROUTE_GEMM128)
    return launch_gemm_nt(static_cast<const uint16_t*>(o.q), D, 0, 0, static_cast<const uint16_t*>(o.g), D, nullptr,
                          0, S, p.ldS, 0, B, N, D, stream);
  // Tile height / residency / cache-policy variants (same arithmetic, same results); 0 is the default, the
  // others exist for A/B tuning in one process (VPR_KNN_VARIANT).  The ablation variants of round 1 (11-13: they
  // skip work and return wrong scores) are compiled only with -DVPR_ABLATION.
  const int variant = tune_or(TUNE_KNN_VARIANT, 0);
  int tr = 208, wgpc = 2;                         // default: 13 row blocks, 2 workgroups per CU
  if (variant == 2) { tr = 144; wgpc = 3; }
  // Shards whose workgroups own more than one 208-row tile (N > 106k) take 256-row tiles: 16 row blocks, 2 x 80 KB = the
  // whole 160 KB of LDS, fewer tiles and 20 % less query re-staging: +2.3 % at 500k bf16 rows, +1.4 % on the 1M-row e4m3
  // call; a 125k-row shard becomes one 244-row tile per workgroup instead of two of 122 (scripts/knn_ab.py, knn_ab_fp8.py).
  if (variant == 3 || ((variant == 0 || (variant >= 5 && variant <= 7) || variant == 14) && knn_tall_tiles(N))) { tr = 256; wgpc = 2; }
  // Score-store path.  Shards whose workgroups own several tiles (N > 131k) write each tile's scores as whole row
  // segments staged through the idle stage buffers (one 1-KiB float4 store per query row, nt policy) instead of dword
  // stores of four 64-byte segments: the stores of a large shard cost far more than their 1.5-3 % share of the bytes
  // (no-store ablation, score stage: 1M e4m3 rows 1518 -> 1245 us, 500k bf16 1391 -> 1252, 125k e4m3 170 -> 159,
  // 100k bf16 ~0).  Staged + nt: 1M e4m3 1497 -> 1342 us, 500k bf16 1374 -> 1300, 250k bf16 668 -> 642; single-tile
  // shards: no difference, they keep the direct stores.  Variants 5 / 6 / 7 force direct / staged / staged + nt.
  const bool staged = variant == 6 || variant == 7 || (variant == 0 && knn_multi_tile(N));
  const bool staged_nt = variant != 6;
  const int slots = num_cus() * wgpc;
  // Fully resident, balanced grid; never more workgroups than 16-row blocks of gallery.
  int nwg = slots;
  const int max_useful = (N + 15) / 16;
  if (nwg > max_useful) nwg = max_useful;
  const size_t lds = (size_t)2 * (tr + KNN_QT) * TILE_ROW_BYTES;
  const int rb = o.fp8 ? D : D * 2;
  const int ksplit = nslab;
  if (ksplit > 1) nwg = p.nrt;                    // tall tiles: the K split supplies the parallelism
  const dim3 grid(nwg, p.Bpad / KNN_QT, ksplit);
#define VPR_KNN_LAUNCH(F8, TR, W, A)                                                                   \
  do {                                                                                                 \
    static PerDeviceFlag attr = {};                                                                    \
    if (lds > 65536)                                                                                   \
      VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(knn_scores_kernel<F8, TR, W, A>), lds, attr)); \
    VPR_TRY_LAUNCH(launch_kernel(knn_scores_kernel<F8, TR, W, A>, grid, dim3(256), lds, stream, o.q, o.g, \
                                 o.q_scale, o.g_scale, S, B, N, rb, p.ldS));                           \
  } while (0)
  // ABL bit 2 (value 4) = nt cache policy on the gallery stream (read once; keeps the query tile and the score
  // matrix in L2 / Infinity Cache): +4.6 % on the bare stream, +5.2 % on this kernel at 100k rows.
  if (staged) {                                   // 4 | 32 (| 16)
#define VPR_KNN_STAGED(F8)                                                                           \
    do {                                                                                               \
      if (tr == 256) { if (staged_nt) VPR_KNN_LAUNCH(F8, 256, 2, 52); else VPR_KNN_LAUNCH(F8, 256, 2, 36); } \
      else { if (staged_nt) VPR_KNN_LAUNCH(F8, 208, 2, 52); else VPR_KNN_LAUNCH(F8, 208, 2, 36); }          \
    } while (0)
    if (o.fp8) VPR_KNN_STAGED(true); else VPR_KNN_STAGED(false);
#undef VPR_KNN_STAGED
    return VPR_OK;
  }
#ifdef VPR_ABLATION
  if (variant == 14) {                            // timing only: no score stores
    if (o.fp8) { if (tr == 256) VPR_KNN_LAUNCH(true, 256, 2, 12); else VPR_KNN_LAUNCH(true, 208, 2, 12); }
    else { if (tr == 256) VPR_KNN_LAUNCH(false, 256, 2, 12); else VPR_KNN_LAUNCH(false, 208, 2, 12); }
    return VPR_OK;
  }
#endif
  if (o.fp8) {
    if (variant == 2) VPR_KNN_LAUNCH(true, 144, 3, 4);
    else if (tr == 256) VPR_KNN_LAUNCH(true, 256, 2, 4);
    else if (variant == 1) VPR_KNN_LAUNCH(true, 208, 2, 0);
    else VPR_KNN_LAUNCH(true, 208, 2, 4);
  } else {
    switch (variant) {
      case 1: VPR_KNN_LAUNCH(false, 208, 2, 0); break;    // default cache policy (round 1's kernel)
      case 2: VPR_KNN_LAUNCH(false, 144, 3, 4); break;    // the round-1 first cut: 9 blocks, 3 per CU
      case 3: VPR_KNN_LAUNCH(false, 256, 2, 4); break;    // 16 blocks, all of the LDS
      case 4: VPR_KNN_LAUNCH(false, 208, 2, 4); break;    // 13 blocks whatever the shard size (A/B against the tall tiles)
#ifdef VPR_ABLATION     // timing-only builds (WRONG scores): never in the shipped library
      case 11: VPR_KNN_LAUNCH(false, 208, 2, 5); break;   // ablation: no MFMA after the first K-step
      case 12: VPR_KNN_LAUNCH(false, 208, 2, 6); break;   // ablation: no query staging after the first K-step
      case 13: VPR_KNN_LAUNCH(false, 208, 2, 7); break;   // ablation: both (pure gallery stream + barriers)
#endif
      default:
        if (tr == 256) VPR_KNN_LAUNCH(false, 256, 2, 4); else VPR_KNN_LAUNCH(false, 208, 2, 4);
        break;
    }
  }
#undef VPR_KNN_LAUNCH
  return VPR_OK;
}

// Default norm bounds of the unchecked entry points: L2-normalised descriptors.  A bf16-rounded unit vector has norm
// within 2^-9 of 1; an e4m3 row (3 mantissa bits, scale = max/448) within 2^-4.
constexpr float NORM_BOUND_BF16 = 1.002f;
constexpr float NORM_BOUND_FP8 = 1.0625f;

// Relative bound of |MFMA score - exact score| / (|q| |g|): D f32 additions of exact products in any order
// (gamma_D = D 2^-24 to first order; 1.1 covers the higher-order terms, the final f32 rounding of the exact score
// and the two scale multiplications of the fp8 path), times the caller's bound on the gallery row norms.
static float knn_err_rel(int D, float gallery_norm_bound) {
  return 1.1f * (float)D * 5.9604645e-8f * gallery_norm_bound;
}

int knn_select(const KnnOperands& o, int B, int N, int D, int k, int index_base, float* out_val,
               int32_t* out_idx, void* ws, size_t ws_bytes, hipStream_t stream,
               float gallery_norm_bound = NORM_BOUND_BF16, int32_t* status = nullptr, int32_t* uncertified = nullptr,
               bool ksplit_scores = false) {
  if (!(gallery_norm_bound > 0.f)) return VPR_ERR_INVALID_ARG;
  const float err_rel = knn_err_rel(D, gallery_norm_bound);
  KnnPlan p;
  if (!ws || !out_val || !out_idx) return VPR_ERR_INVALID_ARG;
  if (!knn_plan(B, N, D, k, &p)) return VPR_ERR_UNSUPPORTED;
  const int st = knn_check(o, D);
  if (st != VPR_OK) return st;
  if (ws_bytes < p.total) return VPR_ERR_WORKSPACE;
  char* w = static_cast<char*>(ws);
  const float* cur_v = reinterpret_cast<float*>(w + p.off_S);
  const int32_t* cur_i = nullptr;
  long long ld = p.ldS;
  int L = N;
  const int rb = o.fp8 ? D : D * 2;
  // slabs to add: whatever the score stage of this problem wrote
  int nslab = 1;
  knn_route(p, o.fp8, B, N, D, ksplit_scores, &nslab);
  const long long slab_stride = (long long)B * p.ldS;
  int lev = 0;
  for (; lev < p.nlevel; ++lev) {
    // the fused final kernel takes over as soon as the candidates of a query fit one workgroup
    // (its LDS = 36 KB of select state + the query row, kept under the 64 KB default limit)
    if (lev > 0 && L <= SEL_CAP && (size_t)rb <= 24 * 1024) break;
    float* ov = reinterpret_cast<float*>(w + p.off_cv[lev & 1]);
    int32_t* oi = reinterpret_cast<int32_t*>(w + p.off_ci[lev & 1]);
    if (lev == 0 && p.kp <= 32)
      VPR_TRY_LAUNCH(launch_kernel(knn_select_stream_kernel<1024>, dim3(p.nchunk[0], B), dim3(256), 0, stream,
                                   cur_v, N, ld, p.ch0, ov, oi, p.kp, p.nchunk[0], nslab, slab_stride));
    else if (lev == 0)
      VPR_TRY_LAUNCH(launch_kernel(knn_select_stream_kernel<4096>, dim3(p.nchunk[0], B), dim3(256), 0, stream,
                                   cur_v, N, ld, p.ch0, ov, oi, p.kp, p.nchunk[0], nslab, slab_stride));
    else
      VPR_TRY_LAUNCH(launch_kernel(knn_select_kernel, dim3(p.nchunk[lev], B), dim3(256), 0, stream,
                                   cur_v, cur_i, L, ld, ov, oi, p.kp, p.nchunk[lev]));
    cur_v = ov; cur_i = oi;
    L = p.nchunk[lev] * p.kp;
    ld = L;
  }
  if (lev > 0 && L <= SEL_CAP && (size_t)rb <= 24 * 1024) {
    const int level0 = lev == 1 ? 1 : 0;       // the lists in hand are level 0's [nchunk][kp] rank-ordered lists
    if (o.fp8)
      VPR_TRY_LAUNCH(launch_kernel(knn_final_fused_kernel<true>, dim3(B), dim3(FF_NT), (size_t)rb, stream, cur_v,
                                   cur_i, L, o.q, o.g, o.q_scale, o.g_scale, rb, k, p.kp, index_base, out_val, out_idx,
                                   err_rel, level0, status, uncertified));
    else
      VPR_TRY_LAUNCH(launch_kernel(knn_final_fused_kernel<false>, dim3(B), dim3(FF_NT), (size_t)rb, stream, cur_v,
                                   cur_i, L, o.q, o.g, o.q_scale, o.g_scale, rb, k, p.kp, index_base, out_val, out_idx,
                                   err_rel, level0, status, uncertified));
    return VPR_OK;
  }
  // general path: the last select level left the [B][kp] list; rescore and order it
  float* exact = reinterpret_cast<float*>(w + p.off_cv[p.nlevel & 1]);
  float* qn = reinterpret_cast<float*>(w + p.off_qn);
  if (o.fp8)
    VPR_TRY_LAUNCH(launch_kernel(knn_rescore_kernel<true>, dim3(p.kp, B), dim3(256), 0, stream, cur_i, o.q, o.g,
                                 o.q_scale, o.g_scale, D, p.kp, exact, qn));
  else
    VPR_TRY_LAUNCH(launch_kernel(knn_rescore_kernel<false>, dim3(p.kp, B), dim3(256), 0, stream, cur_i, o.q, o.g,
                                 o.q_scale, o.g_scale, D * 2, p.kp, exact, qn));
  VPR_TRY_LAUNCH(launch_kernel(knn_order_kernel, dim3(B), dim3(128), 0, stream, cur_i, exact, k, p.kp,
                               index_base, out_val, out_idx, cur_v, qn, err_rel, status, uncertified));
  return VPR_OK;
}

// Per-row symmetric quantisation to OCP e4m3: scale = max|x| / 448 (1 for an all-zero row),
// q = fp8_rne(x / scale).  One workgroup per row; v_cvt_pk_fp8_f32 does the rounding.
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const float* __restrict__ x, int D,
                                                                uint8_t* __restrict__ q, float* __restrict__ scale) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* xr = x + row * D;
  float m = 0.f;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(xr + d);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = m > 0.f ? m / 448.0f : 1.0f;
  if (threadIdx.x == 0) scale[row] = sc;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(xr + d);
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v.x / sc, v.y / sc, 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v.z / sc, v.w / sc, pk, true);
    *reinterpret_cast<int*>(q + row * D + d) = pk;
  }
}

}  // namespace vpr

using namespace vpr;

extern "C" size_t vpr_knn_workspace_bytes(int B, int N, int D, int k) {
  KnnPlan p;
  return knn_plan(B, N, D, k, &p) ? p.total : 0;
}

extern "C" const char* vpr_knn_scores_kernel_name(int is_fp8, int B, int N) {
  // the dispatch of vpr_knn_topk* (knn_route) at the descriptor width of the path, D = 8448: what a kernel trace
  // (rocprofv3) will show for this call
  KnnPlan p;
  if (!knn_plan(B, N, 8448, 10, &p)) return "";
  int nslab = 1;
  const KnnRoute route = knn_route(p, is_fp8 != 0, B, N, 8448, true, &nslab);
  if (route == ROUTE_GEMM256) return is_fp8 ? "vpr::gemm256_kernel<true, 10>" : "vpr::gemm256_kernel<false, 10>";     // (names as rocprofv3 prints them)
  if (route == ROUTE_GEMM128) return is_fp8 ? "vpr::gemm_nt_fp8_kernel" : "vpr::gemm_nt_kernel<128, 2, 2, 2>";
  const int variant = tune_or(TUNE_KNN_VARIANT, 0);
  const bool tall = variant == 3 || (variant == 0 && knn_tall_tiles(N));
  if (variant == 0 && knn_multi_tile(N))
    return is_fp8 ? "vpr::knn_scores_kernel<true, 256, 2, 52>" : "vpr::knn_scores_kernel<false, 256, 2, 52>";
  if (is_fp8) {
    return variant == 2 ? "vpr::knn_scores_kernel<true, 144, 3, 4>" : tall ? "vpr::knn_scores_kernel<true, 256, 2, 4>"
         : variant == 1 ? "vpr::knn_scores_kernel<true, 208, 2, 0>" : "vpr::knn_scores_kernel<true, 208, 2, 4>";
  }
  return variant == 2 ? "vpr::knn_scores_kernel<false, 144, 3, 4>" : tall ? "vpr::knn_scores_kernel<false, 256, 2, 4>"
       : variant == 1 ? "vpr::knn_scores_kernel<false, 208, 2, 0>" : "vpr::knn_scores_kernel<false, 208, 2, 4>";
}

extern "C" float* vpr_knn_scores_ptr(void* workspace, int B, int N, int D, int k, int* ld_out) {
  KnnPlan p;
  if (!workspace || !knn_plan(B, N, D, k, &p)) return nullptr;
  if (ld_out) *ld_out = p.ldS;
  return reinterpret_cast<float*>(static_cast<char*>(workspace) + p.off_S);
}

extern "C" int vpr_knn_scores(const uint16_t* q, const uint16_t* gallery, int B, int N, int D,
                              void* workspace, size_t workspace_bytes, void* stream) {
  // The plan's score-matrix offset does not depend on k; size checks use the caller's bytes.
  const KnnOperands o{q, gallery, nullptr, nullptr, false};
  return knn_scores(o, B, N, D, workspace, workspace_bytes, 1, static_cast<hipStream_t>(stream));
}

extern "C" int vpr_knn_select(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                              int index_base, float* out_val, int32_t* out_idx, void* workspace,
                              size_t workspace_bytes, void* stream) {
  const KnnOperands o{q, gallery, nullptr, nullptr, false};
  return knn_select(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream));
}

extern "C" int vpr_knn_select_checked(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                                      int index_base, float* out_val, int32_t* out_idx, void* workspace,
                                      size_t workspace_bytes, float gallery_norm_bound, int32_t* status,
                                      int32_t* uncertified, void* stream) {
  const KnnOperands o{q, gallery, nullptr, nullptr, false};
  return knn_select(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream), gallery_norm_bound, status, uncertified);
}

static int knn_topk_any(const KnnOperands& o, int B, int N, int D, int k, int index_base, float* out_val,
                        int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream,
                        float gallery_norm_bound, int32_t* status, int32_t* uncertified) {
  KnnPlan p;
  if (B <= 0 || N <= 0 || D <= 0 || k <= 0) return VPR_ERR_INVALID_ARG;
  if (!knn_plan(B, N, D, k, &p)) return VPR_ERR_UNSUPPORTED;
  if (workspace_bytes < p.total) return VPR_ERR_WORKSPACE;
  const int st = knn_scores(o, B, N, D, workspace, workspace_bytes, k, static_cast<hipStream_t>(stream), true);
  if (st != VPR_OK) return st;
  return knn_select(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream), gallery_norm_bound, status, uncertified, true);
}

// The two halves of vpr_knn_topk[_fp8]_checked as separate calls (same kernels, same workspace contents in between:
// the score stage may leave K-slice slabs that only this select stage knows to add), for callers that put events or
// other stream work between the HBM-bound score stage and the latency-bound tail.
extern "C" int vpr_knn_topk_scores_stage(const void* q, const float* q_scale, const void* gallery,
                                         const float* gallery_scale, int is_fp8, int B, int N, int D, int k,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0 || N <= 0 || D <= 0 || k <= 0) return VPR_ERR_INVALID_ARG;
  const KnnOperands o{q, gallery, q_scale, gallery_scale, is_fp8 != 0};
  return knn_scores(o, B, N, D, workspace, workspace_bytes, k, static_cast<hipStream_t>(stream), true);
}

extern "C" int vpr_knn_topk_select_stage(const void* q, const float* q_scale, const void* gallery,
                                         const float* gallery_scale, int is_fp8, int B, int N, int D, int k,
                                         int index_base, float* out_val, int32_t* out_idx, void* workspace,
                                         size_t workspace_bytes, float gallery_norm_bound, int32_t* status,
                                         int32_t* uncertified, void* stream) {
  if (B <= 0 || N <= 0 || D <= 0 || k <= 0) return VPR_ERR_INVALID_ARG;
  const KnnOperands o{q, gallery, q_scale, gallery_scale, is_fp8 != 0};
  return knn_select(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream), gallery_norm_bound, status, uncertified, true);
}

extern "C" int vpr_knn_topk(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                            int index_base, float* out_val, int32_t* out_idx, void* workspace,
                            size_t workspace_bytes, void* stream) {
  const KnnOperands o{q, gallery, nullptr, nullptr, false};
  return knn_topk_any(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes, stream,
                      NORM_BOUND_BF16, nullptr, nullptr);
}

extern "C" int vpr_knn_topk_checked(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                                    int index_base, float* out_val, int32_t* out_idx, void* workspace,
                                    size_t workspace_bytes, float gallery_norm_bound, int32_t* status,
                                    int32_t* uncertified, void* stream) {
  const KnnOperands o{q, gallery, nullptr, nullptr, false};
  return knn_topk_any(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes, stream,
                      gallery_norm_bound, status, uncertified);
}

extern "C" int vpr_knn_topk_fp8(const uint8_t* q, const float* q_scale, const uint8_t* gallery,
                                const float* gallery_scale, int B, int N, int D, int k, int index_base,
                                float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                                void* stream) {
  const KnnOperands o{q, gallery, q_scale, gallery_scale, true};
  return knn_topk_any(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes, stream,
                      NORM_BOUND_FP8, nullptr, nullptr);
}

extern "C" int vpr_knn_topk_fp8_checked(const uint8_t* q, const float* q_scale, const uint8_t* gallery,
                                        const float* gallery_scale, int B, int N, int D, int k, int index_base,
                                        float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                                        float gallery_norm_bound, int32_t* status, int32_t* uncertified,
                                        void* stream) {
  const KnnOperands o{q, gallery, q_scale, gallery_scale, true};
  return knn_topk_any(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes, stream,
                      gallery_norm_bound, status, uncertified);
}

extern "C" int vpr_knn_topk_exhaustive(const void* q, const float* q_scale, const void* gallery,
                                       const float* gallery_scale, int is_fp8, int B, int N, int D, int k,
                                       int index_base, float* out_val, int32_t* out_idx, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  KnnPlan p;
  if (B <= 0 || N <= 0 || D <= 0 || k <= 0 || !workspace) return VPR_ERR_INVALID_ARG;
  if (!knn_plan(B, N, D, k, &p)) return VPR_ERR_UNSUPPORTED;
  const KnnOperands o{q, gallery, q_scale, gallery_scale, is_fp8 != 0};
  const int st = knn_check(o, D);
  if (st != VPR_OK) return st;
  if (workspace_bytes < p.total) return VPR_ERR_WORKSPACE;
  const int rb = o.fp8 ? D : D * 2;
  if (rb > EX_MAXCH * 64 * 16) return VPR_ERR_UNSUPPORTED;
  float* S = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.off_S);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((N + 3) / 4));
  if (o.fp8)
    VPR_TRY_LAUNCH(launch_kernel(knn_exact_scores_kernel<true>, grid, dim3(256), 0, hs, o.q, o.g, o.q_scale, o.g_scale,
                                 S, B, N, rb, p.ldS));
  else
    VPR_TRY_LAUNCH(launch_kernel(knn_exact_scores_kernel<false>, grid, dim3(256), 0, hs, o.q, o.g, o.q_scale, o.g_scale,
                                 S, B, N, rb, p.ldS));
  // the scores are exact already: every cut keeps the true largest keys, whatever the bound says
  return knn_select(o, B, N, D, k, index_base, out_val, out_idx, workspace, workspace_bytes, hs, 1.0f, nullptr, nullptr);
}

extern "C" int vpr_quantize_fp8_rows(const float* x, long long rows, int D, uint8_t* q, float* scale, void* stream) {
  if (!x || !q || !scale || rows < 0 || D <= 0) return VPR_ERR_INVALID_ARG;
  if ((D % 4) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(q) & 3)) return VPR_ERR_UNSUPPORTED;
  if (rows == 0) return VPR_OK;
  if (rows > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  VPR_TRY_LAUNCH(launch_kernel(quantize_fp8_rows_kernel, dim3((unsigned)rows), dim3(256), 0,
                               static_cast<hipStream_t>(stream), x, D, q, scale));
  return VPR_OK;
}

extern "C" int vpr_topk_merge(const float* vals, const int32_t* idxs, int shards, int B, int k,
                              float* out_val, int32_t* out_idx, void* stream) {
  if (!vals || !idxs || !out_val || !out_idx || shards <= 0 || B <= 0 || k <= 0) return VPR_ERR_INVALID_ARG;
  if (k > 128 || (long long)shards * k > 4096) return VPR_ERR_UNSUPPORTED;
  VPR_TRY_LAUNCH(launch_kernel(topk_merge_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     vals, idxs, shards, B, k, out_val, out_idx));
  return VPR_OK;
}
