// gemm_nt.hip — C[M,N] = act(A[M,K] * W[N,K]^T + bias), bf16 operands, f32 accumulate (MFMA).
//
// Used for the SALAD token MLPs (score / cluster_features / token_features
// 1x1-conv + Linear layers of the aggregator called at
// dinov2salad/dinov2salad_validation.py:49-51).  MFMA-bound stage of the hot path.
//
// Structure (cdna guide §5, "minimum 2-phase"): 128 x BN output tile per 256-thread workgroup,
// BK = 64, both operand tiles brought in by LDS-DMA (global_load_lds_dwordx4) into a 2-deep LDS
// ring with the XOR swizzle of vpr_common.h on the source address, fragments read with
// ds_read_b128, v_mfma_f32_32x32x16_bf16, one barrier per K-step.  Workgroup ids are remapped
// so that the workgroups of one XCD (blockIdx % 8) own a contiguous run of tiles and re-use the
// same A panel out of that XCD's L2.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

struct GemmGroup { GemmProblem p[GEMM_MAX_GROUP]; int count; };

template <int BN, int WM, int WN, int STAGES = 2>
__device__ __forceinline__ void gemm_nt_tile(const GemmProblem& pr, int orig, char* smem) {
  constexpr int BM = 128;
  constexpr int TM = BM / WM / 32;  // 32x32 tiles per wave along M
  constexpr int TN = BN / WN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * TILE_ROW_BYTES;
  const uint16_t* __restrict__ A = pr.A;
  const uint16_t* __restrict__ W = pr.W;
  const float* __restrict__ bias = pr.bias;
  void* __restrict__ Cout = pr.C;
  const int lda = pr.lda, a_group_rows = pr.a_group_rows, ldw = pr.ldw, relu = pr.relu, ldc = pr.ldc;
  const long long a_group_stride = pr.a_group_stride;
  const int out_is_bf16 = pr.out_is_bf16, M = pr.M, N = pr.N, K = pr.K, tiles_m = pr.tiles_m, tiles_n = pr.tiles_n;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective remap (cdna guide §5 "XCD swizzle must be bijective").
  const int nwg = tiles_m * tiles_n;
  int tile;
  {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // Per-lane source row pointers for the staging groups this wave owns (group g = 8 tile rows;
  // wave w stages groups w, w+4, ...).
  constexpr int AG = BM / 8 / 4;  // A groups per wave
  constexpr int BG = BN / 8 / 4;  // W groups per wave
  const uint16_t* a_src[AG];
  const uint16_t* w_src[BG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    int r = m0 + (wave + 4 * i) * 8 + (lane >> 3);
    r = r < M ? r : M - 1;
    const uint16_t* p = a_group_rows > 0
        ? A + (long long)(r / a_group_rows) * a_group_stride + (long long)(r % a_group_rows) * lda
        : A + (long long)r * lda;
    const int tr = (wave + 4 * i) * 8 + (lane >> 3);
    a_src[i] = p + (((lane & 7) ^ ((tr >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int i = 0; i < BG; ++i) {
    int r = n0 + (wave + 4 * i) * 8 + (lane >> 3);
    r = r < N ? r : N - 1;
    const int tr = (wave + 4 * i) * 8 + (lane >> 3);
    w_src[i] = W + (long long)r * ldw + (((lane & 7) ^ ((tr >> 1) & 7)) << 3);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K >> 6;
  auto stage = [&](int buf, int ks) {
    char* ta = smem + buf * STAGE_BYTES;
    char* tw = ta + BM * TILE_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < AG; ++i) glds16(a_src[i] + ks * 64, ta + (wave + 4 * i) * 8 * TILE_ROW_BYTES);
#pragma unroll
    for (int i = 0; i < BG; ++i) glds16(w_src[i] + ks * 64, tw + (wave + 4 * i) * 8 * TILE_ROW_BYTES);
  };

  // STAGES-deep ring, STAGES-1 K-tiles in flight.  Before the barrier of step ks, tile ks must have landed:
  // the (STAGES-2) tiles issued after it may stay outstanding (counted vmcnt; every thread issues AG + BG
  // LDS-DMA instructions per tile).  In the tail fewer tiles follow, so the wait degrades to vmcnt(0).
  constexpr int LOADS_PER_STAGE = AG + BG;
#pragma unroll
  for (int p = 0; p < STAGES - 1; ++p)
    if (p < nk) stage(p, p);
  for (int ks = 0; ks < nk; ++ks) {
    if (STAGES > 2 && ks + STAGES - 2 < nk) {
      if constexpr (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS_PER_STAGE) : "memory");
      else if constexpr (STAGES == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LOADS_PER_STAGE) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ... then the barrier: tile ks has landed for every wave and every wave is done reading the buffer
    // that the next issue overwrites (the one read at step ks-1).  A raw s_barrier: __syncthreads() would put
    // `s_waitcnt vmcnt(0)` in front of it (checked in the ISA) and drain the tiles the counted wait left in flight.
    if constexpr (STAGES > 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      __syncthreads();
    }
    if (ks + STAGES - 1 < nk) stage((ks + STAGES - 1) % STAGES, ks + STAGES - 1);
    const char* ta = smem + (ks % STAGES) * STAGE_BYTES;
    const char* tw = ta + BM * TILE_ROW_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[TM], bfr[TN];
      const int chunk = (lane >> 5) + 2 * s;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = lds_frag(ta, (wm * TM + i) * 32 + (lane & 31), chunk);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = lds_frag(tw, (wn * TN + j) * 32 + (lane & 31), chunk);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // Epilogue.  C/D map of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
      const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][e] + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (m < M && n < N) {
          if (out_is_bf16)
            reinterpret_cast<uint16_t*>(Cout)[(long long)m * ldc + n] = f32_to_bf16_bits(v);
          else
            reinterpret_cast<float*>(Cout)[(long long)m * ldc + n] = v;
        }
      }
    }
}

template <int BN, int WM, int WN, int STAGES = 2>
__global__ __launch_bounds__(256, STAGES > 2 ? 1 : 2) void gemm_nt_kernel(GemmProblem pr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_tile<BN, WM, WN, STAGES>(pr, blockIdx.x, smem);
}

// Several independent small GEMMs in one launch (the SALAD layer-2 / token-MLP GEMMs are a few
// tiles each: as separate launches each costs a full launch + K-loop latency with the chip idle).
// Workgroup ids are dealt to the problems in order; every problem keeps its own XCD remap.
template <int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256, 2) void gemm_nt_group_kernel(GemmGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int id = blockIdx.x;
#pragma unroll
  for (int i = 0; i < GEMM_MAX_GROUP; ++i) {
    if (i < grp.count) {
      const int n = grp.p[i].tiles_m * grp.p[i].tiles_n;
      if (id >= 0 && id < n) { gemm_nt_tile<BN, WM, WN, STAGES>(grp.p[i], id, smem); id = -1; }
      else if (id >= 0) id -= n;
    }
  }
}

// ---- fp8 (OCP e4m3, per-row f32 scales) variant: C[m][n] = a_scale[m] * w_scale[n] * sum_k A[m][k] W[n][k] ----
// The kNN score tile for more than 64 gathered queries against an e4m3 shard (BASELINE config 5 on several GPUs).
// Same 128 x 128 tile, LDS image and 2-deep LDS-DMA ring as gemm_nt_tile — a 128-byte tile row is now 128 K
// values — but the products run on the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 with
// unit (E8M0 = 127) scales: twice the bf16 rate, where the plain fp8 MFMA runs at the bf16 rate.  A lane's
// operand is 32 contiguous K bytes (two swizzled 16-byte chunks); A and B use the same lane -> K assignment,
// which is all a dot product needs.
struct GemmFp8Problem {
  const uint8_t* A; int lda; const uint8_t* W; int ldw; const float* a_scale; const float* w_scale;
  float* C; int ldc; int M, N, K, tiles_m, tiles_n;
};
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ __launch_bounds__(256, 2) void gemm_nt_fp8_kernel(GemmFp8Problem pr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 128, BN = 128, WN = 2, TM = 2, TN = 2;      // 2 x 2 waves, 64 x 64 per wave
  constexpr int STAGE_BYTES = (BM + BN) * TILE_ROW_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nwg = pr.tiles_m * pr.tiles_n;
  int tile;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // tile_m fastest: the (few) query tiles that share a gallery panel run back to back on one XCD
  const int tm = tile % pr.tiles_m, tn = tile / pr.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  constexpr int AG = BM / 8 / 4, BG = BN / 8 / 4;
  const uint8_t* a_src[AG];
  const uint8_t* w_src[BG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int tr = (wave + 4 * i) * 8 + (lane >> 3);
    const int r = min(m0 + tr, pr.M - 1);
    a_src[i] = pr.A + (long long)r * pr.lda + (((lane & 7) ^ ((tr >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < BG; ++i) {
    const int tr = (wave + 4 * i) * 8 + (lane >> 3);
    const int r = min(n0 + tr, pr.N - 1);
    w_src[i] = pr.W + (long long)r * pr.ldw + (((lane & 7) ^ ((tr >> 1) & 7)) << 4);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = pr.K >> 7;                       // 128 K bytes per step
  auto stage = [&](int buf, int ks) {
    char* ta = smem + buf * STAGE_BYTES;
    char* tw = ta + BM * TILE_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < AG; ++i) glds16(a_src[i] + ks * 128, ta + (wave + 4 * i) * 8 * TILE_ROW_BYTES);
#pragma unroll
    for (int i = 0; i < BG; ++i) glds16(w_src[i] + ks * 128, tw + (wave + 4 * i) * 8 * TILE_ROW_BYTES);
  };
  stage(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ks + 1 < nk) stage((ks + 1) & 1, ks + 1);
    const char* ta = smem + (ks & 1) * STAGE_BYTES;
    const char* tw = ta + BM * TILE_ROW_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {              // two 64-deep MFMA steps per 128-byte row
      const int ch = 4 * s2 + 2 * (lane >> 5);
      i32x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (lane & 31);
        const i32x4 lo = *reinterpret_cast<const i32x4*>(ta + tile_off(row, ch));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(ta + tile_off(row, ch + 1));
        af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + (lane & 31);
        const i32x4 lo = *reinterpret_cast<const i32x4*>(tw + tile_off(row, ch));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(tw + tile_off(row, ch + 1));
        bfr[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bfr[j], acc[i][j], 0, 0, 0, 127, 0, 127);
    }
  }
  // C/D map of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
      const float ws = n < pr.N ? pr.w_scale[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < pr.M && n < pr.N) pr.C[(long long)m * pr.ldc + n] = acc[i][j][e] * pr.a_scale[m] * ws;
      }
    }
}

int launch_gemm_nt_fp8(const uint8_t* A, int lda, const float* a_scale, const uint8_t* W, int ldw, const float* w_scale,
                       float* C, int ldc, int M, int N, int K, hipStream_t stream) {
  if (!A || !W || !C || !a_scale || !w_scale || M <= 0 || N <= 0 || K <= 0) return VPR_ERR_INVALID_ARG;
  if ((K % 128) || lda < K || ldw < K || ldc < N || (lda % 16) || (ldw % 16)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) return VPR_ERR_UNSUPPORTED;
  GemmFp8Problem g{A, lda, W, ldw, a_scale, w_scale, C, ldc, M, N, K, (M + 127) / 128, (N + 127) / 128};
  constexpr size_t lds = 2 * (128 + 128) * TILE_ROW_BYTES;
  VPR_TRY_LAUNCH(launch_kernel(gemm_nt_fp8_kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g));
  return VPR_OK;
}

static int gemm_check(const GemmProblem& g) {
  if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return VPR_ERR_INVALID_ARG;
  if (g.K % 64 != 0 || g.lda < g.K || g.ldw < g.K || g.ldc < g.N) return VPR_ERR_UNSUPPORTED;
  if ((g.lda % 8) || (g.ldw % 8) || (g.a_group_rows > 0 && (g.a_group_stride % 8))) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) return VPR_ERR_UNSUPPORTED;
  return VPR_OK;
}

int launch_gemm_nt(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                   const uint16_t* W, int ldw, const float* bias, int relu, void* C, int ldc,
                   int out_is_bf16, int M, int N, int K, hipStream_t stream) {
  GemmProblem g{A, lda, a_group_rows, a_group_stride, W, ldw, bias, relu, C, ldc, out_is_bf16, M, N, K, 0, 0};
  const int st = gemm_check(g);
  if (st != VPR_OK) return st;
  g.tiles_m = (M + 127) / 128;
  if (N > 64) {
    g.tiles_n = (N + 127) / 128;
    // A/B switch: 3 = three-deep ring (96 KB: one workgroup per CU).  Measured on the kNN score tile, 2 vs 3
    // stages: 128 x 50k 217 / 291 us, 512 x 12.5k 141 / 205 us — two workgroups per CU beat the deeper ring.
    if (tune_or(TUNE_GEMM_NT_STAGES, 2) == 3) {
      constexpr size_t lds3 = 3 * (128 + 128) * TILE_ROW_BYTES;
      static PerDeviceFlag attr = {};
      VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm_nt_kernel<128, 2, 2, 3>), lds3, attr));
      VPR_TRY_LAUNCH(launch_kernel((gemm_nt_kernel<128, 2, 2, 3>), dim3(g.tiles_m * g.tiles_n), dim3(256), lds3, stream, g));
      return VPR_OK;
    }
    constexpr size_t lds = 2 * (128 + 128) * TILE_ROW_BYTES;
    VPR_TRY_LAUNCH(launch_kernel((gemm_nt_kernel<128, 2, 2>), dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g));
  } else {
    g.tiles_n = 1;
    constexpr size_t lds = 2 * (128 + 64) * TILE_ROW_BYTES;
    VPR_TRY_LAUNCH(launch_kernel((gemm_nt_kernel<64, 4, 1>), dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g));
  }
  return VPR_OK;
}

// Grouped launch with 128x128 tiles for every member (a member with N <= 64 wastes MFMA work on
// its half-empty tile, which is cheaper than a launch of its own).
int launch_gemm_nt_group(const GemmProblem* probs, int count, hipStream_t stream) {
  if (!probs || count < 1 || count > GEMM_MAX_GROUP) return VPR_ERR_INVALID_ARG;
  GemmGroup grp;
  grp.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    grp.p[i] = probs[i];
    const int st = gemm_check(grp.p[i]);
    if (st != VPR_OK) return st;
    grp.p[i].tiles_m = (grp.p[i].M + 127) / 128;
    grp.p[i].tiles_n = (grp.p[i].N + 127) / 128;
    total += grp.p[i].tiles_m * grp.p[i].tiles_n;
  }
  for (int i = count; i < GEMM_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
  // The grouped problems are short (K = 512: 8 K-tiles) and about one workgroup per CU: with the 2-deep ring each
  // K-step waits out one full memory round trip.  Variant 1 keeps two tiles in flight (3-deep ring) on 128 x 64
  // tiles (72 KB of LDS: still two workgroups per CU, so the ~390 workgroups stay one resident round; a 3-deep
  // ring on 128 x 128 tiles is 96 KB = one workgroup per CU and 258 workgroups then need a second round:
  // measured 101.6 vs 88.9 us for the whole SALAD stage).
  const int variant = tune_or(TUNE_GEMM_GROUP_VARIANT, 1);   // A/B switch: 0 = 128 x 128 tiles, 2-deep ring (89.9 us); 1 = default (84.4 us)
  if (variant == 1) {
    total = 0;
    for (int i = 0; i < count; ++i) {
      grp.p[i].tiles_n = (grp.p[i].N + 63) / 64;
      total += grp.p[i].tiles_m * grp.p[i].tiles_n;
    }
    for (int i = count; i < GEMM_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
    constexpr size_t lds3 = 3 * (128 + 64) * TILE_ROW_BYTES;
    static PerDeviceFlag attr = {};
    VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm_nt_group_kernel<64, 4, 1, 3>), lds3, attr));
    VPR_TRY_LAUNCH(launch_kernel((gemm_nt_group_kernel<64, 4, 1, 3>), dim3(total), dim3(256), lds3, stream, grp));
    return VPR_OK;
  }
  constexpr size_t lds = 2 * (128 + 128) * TILE_ROW_BYTES;
  VPR_TRY_LAUNCH(launch_kernel((gemm_nt_group_kernel<128, 2, 2, 2>), dim3(total), dim3(256), lds, stream, grp));
  return VPR_OK;
}

}  // namespace vpr
