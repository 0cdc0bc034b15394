// gemm256.hip — C[M,N] = act(A[M,K] * W[N,K]^T + bias) on a 256 x 256 output tile per workgroup.
//
// The 128 x 128 kernel of gemm_nt.hip stages 32 KB per 64-deep K-step for 2 MFLOP... 64 FLOP per
// staged byte: it cannot get near the MFMA peak.  A 256 x 256 tile doubles that ratio, and this
// kernel keeps the LDS-DMA traffic in flight ACROSS barriers (cdna guide §5, "Pipelining across
// barriers" / the 8-phase idea) instead of draining it every K-step:
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns a 128 x 64 output block = 4 quadrants of
//     64 x 32, one MFMA cluster (16 x v_mfma_f32_16x16x32_bf16 over BK = 64) per phase;
//   * a K-tile (A: 256 rows x 128 B, W: 256 rows x 128 B) is split into four half-tiles by WHEN a
//     wave needs them: A-early / W-early (quadrant row/col 0, read in phase 1), W-late (phase 2),
//     A-late (phase 3).  Two K-tile buffers (128 KB LDS); a half-tile slot is refilled for K-tile
//     kt+2 one phase after its last read, so five half-tiles (80 KB) are always in flight;
//   * every thread issues exactly 2 LDS-DMA instructions per phase in a fixed order, so
//     `s_waitcnt vmcnt(10)` (never 0 inside the loop) + raw s_barriers retire exactly the
//     half-tile the next phase is about to read; past the last K-tile the same addresses are
//     re-requested (identical bytes), which keeps the count constant;
//   * the two wave groups (wr = 0 / 1) run half a phase apart, so the MFMA cluster of one overlaps
//     the LDS reads of the other on every SIMD.
// Operand roles are swapped in the MFMA (A operand = W rows, B operand = A rows) so that a lane
// ends up with 4 consecutive output columns: 8-byte bf16 / 16-byte f32 stores.
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int G2_BM = 256, G2_BN = 256;
constexpr int G2_BUF_BYTES = (G2_BM + G2_BN) * TILE_ROW_BYTES;   // 64 KB per K-tile

// first tile row of 8-row staging group `ge` (0..15) of a half-tile
__device__ __forceinline__ int g2_a_row(int ge, bool late) { return (ge < 8 ? ge * 8 : 128 + (ge - 8) * 8) + (late ? 64 : 0); }
__device__ __forceinline__ int g2_w_row(int ge, bool late) { return (ge >> 2) * 64 + (ge & 3) * 8 + (late ? 32 : 0); }

// One operand fragment of a K-tile: the two 16-byte chunks (fch, fch + 4) of a tile row.  bf16: the operands of the two
// 32-deep MFMAs of a 64-deep K-tile; fp8 (FP8 = true: e4m3 bytes, a 128-byte tile row = 128 K values): together the
// 32-byte operand of ONE block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales) — same LDS image, same reads,
// the same 256 matrix-pipe cycles per phase for twice the K: the fp8 form of this kernel runs at twice the FLOP rate
// on the same byte stream.
struct G2Frag { bf16x8 lo, hi; };
__device__ __forceinline__ G2Frag g2_frag(const char* tile, int row, int fch) {
  return G2Frag{lds_frag(tile, row, fch), lds_frag(tile, row, fch + 4)};
}
template <bool FP8>
__device__ __forceinline__ f32x4 g2_mma(const G2Frag& w, const G2Frag& a, f32x4 c) {
  if constexpr (FP8) {
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const i32x4 w0 = __builtin_bit_cast(i32x4, w.lo), w1 = __builtin_bit_cast(i32x4, w.hi);
    const i32x4 a0 = __builtin_bit_cast(i32x4, a.lo), a1 = __builtin_bit_cast(i32x4, a.hi);
    const i32x8 wv = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
    const i32x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, c, 0, 0, 0, 127, 0, 127);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo, a.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, a.hi, c, 0, 0, 0);
  }
}

// FP8: pr.A / pr.W point at e4m3 bytes (lda / ldw / K in bytes = elements), pr.a_scale / pr.w_scale are the per-row
// f32 scales, the output is f32 = acc * a_scale[m] * w_scale[n] (no bias / relu): the kNN score tile of a 512-query
// gathered batch against an e4m3 shard (BASELINE config 5 on 8 GPUs).
typedef f32x4 G2Acc[2][2][4][2];   // [qi][qj][rb][cb]: rows n = wc*64 + qj*32 + cb*16 + 4g+e, cols m = wr*128 + qi*64 + rb*16 + lane&15

// Tile mapping, staging set-up and the K loop of one 256 x 256 output tile; returns with every wave past the last MFMA
// (the two wave groups re-aligned), the accumulators in `acc`, the LDS-DMA tail possibly still in flight.
// VM = the counted wait of the pipeline: 10 LDS-DMA instructions (five 16 KB half-tiles) stay in flight behind every wait.
// 6 / 2 (timing-only build, VPR_GEMM256_DEPTH): the same requests, waited for two / four half-tiles earlier than needed —
// an ablation of the prefetch distance (gathered-batch score GEMM).
template <int VM> __device__ __forceinline__ void g2_wait_vm() {
  if constexpr (VM == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (VM == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
}
template <int VM> __device__ __forceinline__ void g2_wait_vm_lgkm() {
  if constexpr (VM == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
  else if constexpr (VM == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
}

template <bool FP8, int VM = 10>
__device__ __forceinline__ void g2_mainloop(const GemmProblem& pr, char* smem, G2Acc& acc, int& m0_out, int& n0_out, int& tn_out) {
  constexpr int ES = FP8 ? 1 : 2;                       // operand element size in bytes
  const char* __restrict__ A = reinterpret_cast<const char*>(pr.A);
  const char* __restrict__ W = reinterpret_cast<const char*>(pr.W);
  const int M = pr.M, N = pr.N, K = pr.K;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware bijective remap; consecutive tiles of one XCD share the A panel (tile_n fastest)
  const int nwg = pr.tiles_m * pr.tiles_n;
  int tile;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // grouped raster inside the XCD's contiguous range: the ~32 tiles an XCD runs concurrently form
  // a 4 (M) x 8 (N) patch, i.e. 12 operand panels through its L2 instead of 33 for a 1 x 32 strip
  int tm, tn;
  {
    constexpr int GM = 4;
    const int per_group = GM * pr.tiles_n;
    const int grp = tile / per_group, in_grp = tile - grp * per_group;
    const int first_m = grp * GM;
    const int gsz = (pr.tiles_m - first_m) < GM ? (pr.tiles_m - first_m) : GM;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
  }
  const int m0 = tm * G2_BM, n0 = tn * G2_BN;

  // ---- staging: this wave owns groups 2*wave, 2*wave+1 of each of the four half-tiles ----
  // index h: 0 = A-early, 1 = W-early, 2 = W-late, 3 = A-late
  const char* src[4][2];
  int dst[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ge = 2 * wave + j;
      const bool isA = (h == 0 || h == 3), late = (h >= 2);
      const int row0 = isA ? g2_a_row(ge, late) : g2_w_row(ge, late);
      const int tr = row0 + (lane >> 3);
      const int sw = ((lane & 7) ^ ((tr >> 1) & 7)) << 4;      // bytes
      if (isA) {
        int r = m0 + tr;
        r = r < M ? r : M - 1;
        const char* p = pr.a_group_rows > 0
            ? A + ((long long)(r / pr.a_group_rows) * pr.a_group_stride + (long long)(r % pr.a_group_rows) * pr.lda) * ES
            : A + (long long)r * pr.lda * ES;
        src[h][j] = p + sw;
        dst[h][j] = row0 * TILE_ROW_BYTES;
      } else {
        int r = n0 + tr;
        r = r < N ? r : N - 1;
        src[h][j] = W + (long long)r * pr.ldw * ES + sw;
        dst[h][j] = G2_BM * TILE_ROW_BYTES + row0 * TILE_ROW_BYTES;
      }
    }
  // K-tiles of this workgroup: all of them, or — split-K (gridDim.y slices: kNN score tiles of shards too small to give
  // 256 tiles) — slice blockIdx.y, written to its own output slab (pr.slab_stride elements apart; the consumer adds them)
  const int nk_all = (K * ES) >> 7;                      // 128 bytes of every row per K-tile
  const int k_lo = (int)((long long)nk_all * blockIdx.y / gridDim.y);
  const int nk = (int)((long long)nk_all * (blockIdx.y + 1) / gridDim.y) - k_lo;
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) src[h][j] += (long long)k_lo * 128;
  // K walk: step kt of this tile reads K-tile (kt + koff) mod nk.  With koff = 0 every tile that shares an operand panel asks
  // the L2 for the same lines in the same microsecond; a per-tile rotation spreads those requests over time (Tensile calls
  // the same idea StaggerU).  A sum is a sum: only the f32 accumulation order of a tile changes, and it is fixed per tile.
  int koff = 0;
  switch (pr.kstagger) {
    case 1: koff = tn; break;                            // tiles sharing an A panel one K-tile apart
    case 2: koff = tm + tn; break;                       // neighbours in both directions apart
    case 3: koff = (tm & 3) * 4 + (tn & 3); break;
    case 4: koff = tm & 3; break;                        // tiles sharing a W panel apart
    case 5: koff = 2 * tn + (tm & 1); break;
    default: break;
  }
  koff %= nk;
  auto issue = [&](int h, int kt) {   // half-tile h of step kt -> buffer kt & 1 (kt clamped: dummy tail)
    const int kc = kt < nk ? kt : nk - 1;
    char* base = smem + (kc & 1) * G2_BUF_BYTES;
    const int kk = kc + koff < nk ? kc + koff : kc + koff - nk;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(src[h][j] + kk * 128, base + dst[h][j]);
  };

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: Ae0 We0 Wl0 Al0 Ae1 We1 Wl1 (the loop continues with Al(kt+1) at kt.P1)
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  issue(0, 1); issue(1, 1); issue(2, 1);

  const int frow = lane & 15, fch = lane >> 4;
  // Two wave groups (wr = 0 / 1; each SIMD hosts one wave of each) run the same segment stream half
  // a phase apart: every phase is  L {ds_read fragments, issue 2 LDS-DMA, retire next phase's data}
  // | barrier | C {16 MFMAs} | barrier, and group 1 starts one barrier late, so one group's MFMA
  // cluster overlaps the other's LDS reads on the same SIMD.  Hazards with the stagger:
  //   RAW  the data a phase reads was retired (counted vmcnt) by EVERY wave at the end of its
  //        previous L segment, which precedes a barrier both groups pass before either reads it;
  //   WAR  a slot is refilled in the L segment after the one that read it, and every L segment
  //        ends with lgkmcnt(0), so both groups' reads are complete two barriers earlier.
  g2_wait_vm<VM>();                                    // A-early / W-early of K-tile 0 (before ANY barrier)
  if (wr == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const char* ta = smem + (kt & 1) * G2_BUF_BYTES;
    const char* tw = ta + G2_BM * TILE_ROW_BYTES;
    G2Frag a[4], b0[2], b1[2];

    // ---- phase 1: A rows of quadrant-row 0, W rows of quadrant-col 0 ----
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) b0[cb] = g2_frag(tw, wc * 64 + cb * 16 + frow, fch);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) a[rb] = g2_frag(ta, wr * 128 + rb * 16 + frow, fch);
    issue(3, kt + 1);
    g2_wait_vm_lgkm<VM>();                                          // retires W-late(kt) for phase 2
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[0][0][rb][cb] = g2_mma<FP8>(b0[cb], a[rb], acc[0][0][rb][cb]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();

    // ---- phase 2: W rows of quadrant-col 1 ----
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) b1[cb] = g2_frag(tw, wc * 64 + 32 + cb * 16 + frow, fch);
    issue(0, kt + 2);          // A-early slot: both groups' phase-1 reads completed two barriers ago
    g2_wait_vm_lgkm<VM>();                                          // retires A-late(kt) for phase 3
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[0][1][rb][cb] = g2_mma<FP8>(b1[cb], a[rb], acc[0][1][rb][cb]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();

    // ---- phase 3: A rows of quadrant-row 1 ----
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) a[rb] = g2_frag(ta, wr * 128 + 64 + rb * 16 + frow, fch);
    issue(1, kt + 2);          // W-early slot (read in phase 1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[1][1][rb][cb] = g2_mma<FP8>(b1[cb], a[rb], acc[1][1][rb][cb]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();

    // ---- phase 4: no new operands (W quadrant-col 0 is still in registers) ----
    issue(2, kt + 2);          // W-late slot (read in phase 2)
    g2_wait_vm<VM>();                                               // retires A-early / W-early of K-tile kt+1
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[1][0][rb][cb] = g2_mma<FP8>(b0[cb], a[rb], acc[1][0][rb][cb]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // group 0 makes up the barrier group 1 spent at the start
  m0_out = m0; n0_out = n0; tn_out = tn;
}

template <bool FP8, int VM = 10>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmProblem pr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  G2Acc acc;
  int m0, n0, tn_unused;
  g2_mainloop<FP8, VM>(pr, smem, acc, m0, n0, tn_unused);
  const int M = pr.M, N = pr.N;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // ---- epilogue ----
  // C/D of (W-operand-as-A): a lane holds n = 4g+e (4 consecutive output columns) at row m = lane&15.
  // Written straight to global that is 32-byte pieces scattered over 16 rows per store; instead the
  // tile goes through the (now free) LDS in two 128-row passes and leaves as whole 512 B / 1 KB rows.
  const int g = lane >> 4;
  float4 bias4[2][2];
#pragma unroll
  for (int qj = 0; qj < 2; ++qj)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = n0 + wc * 64 + qj * 32 + cb * 16 + 4 * g;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* colv = FP8 ? pr.w_scale : pr.bias;        // bf16: additive bias; fp8: the W rows' scales (multiplicative)
      if (colv != nullptr) {
        if (n + 3 < N) b = *reinterpret_cast<const float4*>(colv + n);
        else {
          if (n < N) b.x = colv[n];
          if (n + 1 < N) b.y = colv[n + 1];
          if (n + 2 < N) b.z = colv[n + 2];
        }
      }
      bias4[qj][cb] = b;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // bias + the dummy tail of the LDS-DMA stream
  __builtin_amdgcn_s_barrier();                      // nobody stages before every wave's DMA has landed
  const int es = pr.out_is_bf16 ? 2 : 4;
  const int pitch = G2_BN * es + 16;                 // +16 B: 16 rows of one column no longer share a bank
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      char* rowp = smem + (wr * 64 + rb * 16 + (lane & 15)) * pitch;
      float as = 1.f;
      if constexpr (FP8) as = pr.a_scale[min(m0 + wr * 128 + qi * 64 + rb * 16 + (lane & 15), M - 1)];
#pragma unroll
      for (int qj = 0; qj < 2; ++qj)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int nl = wc * 64 + qj * 32 + cb * 16 + 4 * g;
          const float4 b = bias4[qj][cb];
          float v0, v1, v2, v3;
          if constexpr (FP8) {                                 // (acc * a_scale) * w_scale: the order of the other fp8 score paths
            v0 = acc[qi][qj][rb][cb][0] * as * b.x; v1 = acc[qi][qj][rb][cb][1] * as * b.y;
            v2 = acc[qi][qj][rb][cb][2] * as * b.z; v3 = acc[qi][qj][rb][cb][3] * as * b.w;
          } else {
            v0 = acc[qi][qj][rb][cb][0] + b.x; v1 = acc[qi][qj][rb][cb][1] + b.y;
            v2 = acc[qi][qj][rb][cb][2] + b.z; v3 = acc[qi][qj][rb][cb][3] + b.w;
          }
          if (pr.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          if (pr.out_is_bf16) {
            uint2 o;
            o.x = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
            o.y = (uint32_t)f32_to_bf16_bits(v2) | ((uint32_t)f32_to_bf16_bits(v3) << 16);
            *reinterpret_cast<uint2*>(rowp + nl * 2) = o;
          } else {
            *reinterpret_cast<float4*>(rowp + nl * 4) = make_float4(v0, v1, v2, v3);
          }
        }
    }
    __syncthreads();
    // read-out: wave w owns staged rows 16w .. 16w+15; a row is 32 (bf16) or 64 (f32) 16-byte chunks
    const int lanes_per_row = pr.out_is_bf16 ? 32 : 64;
    const int rows_per_inst = 64 / lanes_per_row;
    const int chunk = lane & (lanes_per_row - 1);
    const int epc = 16 / es;                           // elements per chunk
    const bool wide = ((pr.ldc * es) & 15) == 0;       // 16-byte row stores need a 16-byte row pitch
    const int n = n0 + chunk * epc;
    for (int i = 0; i < 16 / rows_per_inst; ++i) {
      const int lr = wave * 16 + i * rows_per_inst + (pr.out_is_bf16 ? (lane >> 5) : 0);
      const int m = m0 + (lr >> 6) * 128 + qi * 64 + (lr & 63);
      const uint4 v = *reinterpret_cast<const uint4*>(smem + lr * pitch + chunk * 16);
      if (m >= M || n >= N) continue;
      char* dstp = reinterpret_cast<char*>(pr.C) + ((long long)blockIdx.y * pr.slab_stride + (long long)m * pr.ldc + n) * es;
      if (n + epc <= N && wide) {
        *reinterpret_cast<uint4*>(dstp) = v;
      } else {                                         // ragged last chunk of the row
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < epc && n + e < N; ++e) {
          if (pr.out_is_bf16) reinterpret_cast<uint16_t*>(dstp)[e] = (uint16_t)(w4[e >> 1] >> (16 * (e & 1)));
          else reinterpret_cast<uint32_t*>(dstp)[e] = w4[e];
        }
      }
    }
    if (qi == 0) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// SALAD layer 1 + layer 2 in ONE kernel.  Layer 1 is  H = relu(X W1^T + b1)  with W1 = [score.0 |
// cluster_features.0] ([2*hidden, C]); the second layers are  S = H[:, :hidden] W2s^T + b2s  ([.., 64]) and
// F = H[:, hidden:] W2c^T + b2c  ([.., 128]).  A 256 x 256 tile of H is one image's 256 tokens x one quarter of the
// 2*hidden columns, i.e. a K slice of exactly one of the two second-layer products: the tile's epilogue rounds H to bf16
// (the rounding point the unfused path has), parks it in the now idle LDS in the operand layout of the main loop (four
// K-tiles of [256 rows][128 B], same XOR swizzle, so the fragment reads are the conflict-free ones of vpr_common.h),
// multiplies it with its [n_out, 256] slice of W2 (fragments straight from L2 into registers: 64-128 KB per matrix, shared
// by all tiles) and writes a PARTIAL sum  out[slab][row][n_out]  (slab = which 256-column slice of the head's hidden
// layer; slab 0 carries the bias).  The consumer (Sinkhorn kernel) adds the slabs in slab order: deterministic.
// Gone: the 33.5 MB write + read of H, and the grouped second-layer launch (18.6 us of an 82 us stage).
#ifdef VPR_ABLATION
__device__ long long* g_fuse2_clocks = nullptr;      // timing-only build: phase clocks of gemm256_fuse2_kernel, [tiles][8]
#define G2_CLOCK(k) do { if (g_fuse2_clocks && threadIdx.x == 0) g_fuse2_clocks[blockIdx.x * 8 + (k)] = vpr_clock_now(); } while (0)
#else
#define G2_CLOCK(k) do { } while (0)
#endif

struct Fuse2 {
  const uint16_t* W2[2];       // [n_out[h], hidden] bf16 row-major; h = 0 score head, 1 cluster head
  const uint16_t* W2frag[2];   // the same matrices in fragment order (vpr_salad_pack_w2_fragments) or null
  const float* b2[2];
  float* out[2];               // [slabs][M][n_out[h]] f32
  int n_out[2];                // 64 or 128 each
  int hidden;                  // per-head hidden width: a multiple of 256; tiles_n = 2 * hidden / 256
};

// Epilogue of gemm256_fuse2_kernel.  NOB = o-blocks (16 outputs) per wave = n_out / 64.
// Work split of the second GEMM ([256 rows] x [n_out] x [K = 256 hidden columns of this tile]): 2 row halves x 4 output
// quarters.  A wave keeps its W2 fragments in registers (NOB x 8 k-steps x 16 B per lane: 16 / 8 KB per wave, each
// fragment needed by exactly 2 waves) and reads the hidden tile from LDS (8 row blocks x 8 k-steps).  The first cut
// (4 row quarters x 2 output halves) pulled every W2 fragment through four waves — 256 KB per workgroup out of an L2 that
// all 256 workgroups hit at the same addresses at the same moment: 3.9 / 7.4 us (score / cluster tiles) waiting for it
// (phase clocks, scripts/sinkhorn_phases.py).  Here the fragments are requested BEFORE the hidden tile is parked in LDS
// (the main loop's LDS-DMA tail is older in the in-order vmcnt queue, so a counted wait still retires exactly the tail).
template <int NOB>
__device__ __forceinline__ void g2_fuse2_epilogue(const GemmProblem& pr, const Fuse2& f, G2Acc& acc, char* smem,
                                                  int head, int slab, int m0, int n0, int lane, int wave) {
  const int n_out = f.n_out[head];
  const int wr = wave >> 2, wc = wave & 3;
  const int frow = lane & 15, g = lane >> 4;
  const int mh = wave >> 2, oq = wave & 3;              // second GEMM: row half (128 rows), output quarter (16 * NOB outputs)
  // layer-1 bias first, then the W2 fragments: vmcnt retires in order, so the staging code below can wait for the bias
  // (needed at once) with the NOB * 8 younger fragment loads still in flight behind it
  float4 bias4[2][2];
#pragma unroll
  for (int qj = 0; qj < 2; ++qj)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      bias4[qj][cb] = *reinterpret_cast<const float4*>(pr.bias + n0 + wc * 64 + qj * 32 + cb * 16 + 4 * g);
  // W2 slice fragments (A operand: i = output o, k = hidden column): 8 k-steps of 32 per 256-column slab.
  // Fragment-order copy (vpr_salad_pack_w2_fragments): one contiguous 1 KB per wave load.  Row-major original: a wave
  // load touches 16 rows x 64 B = 16 cache lines for 1 KB, and the L1's tag rate then sets the pace — 4 / 7 us of
  // waiting per score / cluster tile (phase clocks) against 1 us with the packed copy.
  bf16x8 wf[NOB][8];
  if (f.W2frag[head] != nullptr) {
    const uint16_t* wp = f.W2frag[head] + ((long long)(slab * (n_out >> 4) + oq * NOB) * 8 * 64 + lane) * 8;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        wf[ob][ks] = *reinterpret_cast<const bf16x8*>(wp + (ob * 8 + ks) * 512);
  } else {
    const uint16_t* w2 = f.W2[head] + (long long)(oq * 16 * NOB + frow) * f.hidden + slab * 256 + 8 * g;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        wf[ob][ks] = *reinterpret_cast<const bf16x8*>(w2 + (long long)ob * 16 * f.hidden + ks * 32);
  }
  // the 4 + NOB * 8 loads above are the youngest entries of the vmcnt queue: everything older (the dummy tail of the
  // LDS-DMA stream) has landed once only they are outstanding
  if constexpr (NOB == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // nobody overwrites the K-tile buffers before every wave's DMA has landed
  G2_CLOCK(2);
  // relu(acc + b1) -> bf16 -> LDS, operand layout: wave column wc owns hidden columns wc*64 .. +63 of the tile = K-tile wc
  char* kt_base = smem + wc * (G2_BM * TILE_ROW_BYTES);
#pragma unroll
  for (int qi = 0; qi < 2; ++qi)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int row = wr * 128 + qi * 64 + rb * 16 + (lane & 15);
#pragma unroll
      for (int qj = 0; qj < 2; ++qj)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float4 b = bias4[qj][cb];
          const f32x4 a = acc[qi][qj][rb][cb];
          const float v0 = fmaxf(a[0] + b.x, 0.f), v1 = fmaxf(a[1] + b.y, 0.f), v2 = fmaxf(a[2] + b.z, 0.f), v3 = fmaxf(a[3] + b.w, 0.f);
          uint2 o;
          o.x = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
          o.y = (uint32_t)f32_to_bf16_bits(v2) | ((uint32_t)f32_to_bf16_bits(v3) << 16);
          const int chunk = qj * 4 + cb * 2 + (g >> 1);
          *reinterpret_cast<uint2*>(kt_base + tile_off(row, chunk) + (g & 1) * 8) = o;
        }
    }
  f32x4 acc2[8][NOB];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) acc2[mb][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                      // the whole hidden tile is in LDS (and the W2 fragments have arrived)
  G2_CLOCK(3);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const char* tile = smem + (ks >> 1) * (G2_BM * TILE_ROW_BYTES);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const bf16x8 hf = lds_frag(tile, mh * 128 + mb * 16 + frow, g + 4 * (ks & 1));
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
        acc2[mb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ob][ks], hf, acc2[mb][ob], 0, 0, 0);
    }
  }
  G2_CLOCK(4);
  // D[i = o][j = m]: the lane holds outputs o = 4g .. 4g+3 of row m = lane & 15: one 16-byte store
  float* outp = f.out[head] + ((long long)slab * pr.M + m0 + mh * 128 + frow) * n_out + oq * 16 * NOB + 4 * g;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slab == 0) b = *reinterpret_cast<const float4*>(f.b2[head] + oq * 16 * NOB + ob * 16 + 4 * g);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const f32x4 a = acc2[mb][ob];
      *reinterpret_cast<float4*>(outp + (long long)mb * 16 * n_out + ob * 16) = make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
    }
  }
}

__global__ __launch_bounds__(512, 2) void gemm256_fuse2_kernel(GemmProblem pr, Fuse2 f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  G2Acc acc;
  int m0, n0, tn;
  G2_CLOCK(0);
  g2_mainloop<false>(pr, smem, acc, m0, n0, tn);
  G2_CLOCK(1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_per_head = f.hidden >> 8;
  const int head = tn >= tiles_per_head ? 1 : 0;
  const int slab = tn - head * tiles_per_head;
  if (f.n_out[head] == 128) g2_fuse2_epilogue<2>(pr, f, acc, smem, head, slab, m0, n0, lane, wave);
  else g2_fuse2_epilogue<1>(pr, f, acc, smem, head, slab, m0, n0, lane, wave);
  G2_CLOCK(5);
}

constexpr size_t G2_LDS = 128 * (G2_BN * 4 + 16);   // >= the two K-tile buffers (128 KB); sized by the f32 epilogue staging

int launch_gemm256(const GemmProblem& in, hipStream_t stream) {
  GemmProblem g = in;
  if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return VPR_ERR_INVALID_ARG;
  if (g.K % 64 != 0 || g.K < 128 || g.lda < g.K || g.ldw < g.K || g.ldc < g.N) return VPR_ERR_UNSUPPORTED;
  if ((g.lda % 8) || (g.ldw % 8) || (g.ldc % 4) || (g.a_group_rows > 0 && (g.a_group_stride % 8))) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.C)) & 15)
    return VPR_ERR_UNSUPPORTED;
  g.tiles_m = (g.M + G2_BM - 1) / G2_BM;
  g.tiles_n = (g.N + G2_BN - 1) / G2_BN;
  g.a_scale = g.w_scale = nullptr;
  g.kstagger = tune_or(TUNE_GEMM256_STAGGER, 0);
  const int ksplit = g.ksplit > 1 ? g.ksplit : 1;
  if (ksplit > 1 && (g.bias != nullptr || g.relu || g.K / 64 < 2 * ksplit)) return VPR_ERR_UNSUPPORTED;   // slabs are linear partial sums
#ifdef VPR_ABLATION
  {   // timing-only build: shallower prefetch distance (same results)
    const int depth = tune_or(TUNE_GEMM256_DEPTH, 10);
    if (depth == 6 || depth == 2) {
      static PerDeviceFlag a6 = {}, a2 = {};
      if (depth == 6) {
        VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm256_kernel<false, 6>), G2_LDS, a6));
        VPR_TRY_LAUNCH(launch_kernel(gemm256_kernel<false, 6>, dim3(g.tiles_m * g.tiles_n, ksplit), dim3(512), G2_LDS, stream, g));
      } else {
        VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm256_kernel<false, 2>), G2_LDS, a2));
        VPR_TRY_LAUNCH(launch_kernel(gemm256_kernel<false, 2>, dim3(g.tiles_m * g.tiles_n, ksplit), dim3(512), G2_LDS, stream, g));
      }
      return VPR_OK;
    }
  }
#endif
  static PerDeviceFlag attr = {};
  VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm256_kernel<false>), G2_LDS, attr));
  VPR_TRY_LAUNCH(launch_kernel(gemm256_kernel<false>, dim3(g.tiles_m * g.tiles_n, ksplit), dim3(512), G2_LDS, stream, g));
  return VPR_OK;
}

__global__ __launch_bounds__(256) void pack_w2_fragments_kernel(const uint16_t* __restrict__ w2, int n_out, int hidden,
                                                                uint16_t* __restrict__ out) {
  // one thread per 16-byte fragment piece: index = ((s * (n_out/16) + ob) * 8 + ks) * 64 + lane
  const long long total = (long long)n_out * hidden / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), ks = (int)((i >> 6) & 7);
    const long long r = i >> 9;
    const int nob = n_out >> 4;
    const int ob = (int)(r % nob), sl = (int)(r / nob);
    const uint16_t* src = w2 + (long long)(ob * 16 + (lane & 15)) * hidden + sl * 256 + ks * 32 + 8 * (lane >> 4);
    *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(src);
  }
}

int launch_pack_w2_fragments(const uint16_t* w2, int n_out, int hidden, uint16_t* out, hipStream_t stream) {
  if (!w2 || !out || n_out <= 0 || hidden <= 0) return VPR_ERR_INVALID_ARG;
  if ((n_out % 16) || (hidden % 256)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(out)) & 15) return VPR_ERR_UNSUPPORTED;
  const long long total = (long long)n_out * hidden / 8;
  VPR_TRY_LAUNCH(launch_kernel(pack_w2_fragments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w2, n_out, hidden, out));
  return VPR_OK;
}

// SALAD score + cluster MLPs, both layers (see gemm256_fuse2_kernel).  X rows as in GemmProblem (row-group addressing),
// W1 [2*hidden, C] + b1, second layers W2s [m, hidden] + b2s -> S[slabs][M][m], W2c [l, hidden] + b2c -> F[slabs][M][l],
// slabs = hidden / 256.  Returns VPR_ERR_UNSUPPORTED for shapes outside the tile geometry (the caller falls back).
int launch_salad_mlps_fused(const uint16_t* X, int ldx, int group_rows, long long group_stride, const uint16_t* W1, const float* b1,
                            const uint16_t* W2s, const float* b2s, const uint16_t* W2c, const float* b2c,
                            float* S, float* F, int M, int C, int hidden, int m, int l, hipStream_t stream,
                            const uint16_t* W2s_frag, const uint16_t* W2c_frag) {
  if (!X || !W1 || !b1 || !W2s || !b2s || !W2c || !b2c || !S || !F || M <= 0) return VPR_ERR_INVALID_ARG;
  if ((M % G2_BM) || (hidden % 256) || hidden <= 0 || (C % 64) || C < 128) return VPR_ERR_UNSUPPORTED;
  if (!((m == 64 || m == 128) && (l == 64 || l == 128))) return VPR_ERR_UNSUPPORTED;      // 16 or 32 outputs per wave quarter
  if ((ldx % 8) || (group_rows > 0 && (group_stride % 8))) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2s) |
       reinterpret_cast<uintptr_t>(W2c) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2s) |
       reinterpret_cast<uintptr_t>(b2c) | reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(F)) & 15)
    return VPR_ERR_UNSUPPORTED;
  GemmProblem g{X, ldx, group_rows, group_stride, W1, C, b1, 1, nullptr, 0, 1, M, 2 * hidden, C, M / G2_BM, 2 * hidden / G2_BN,
                nullptr, nullptr, 1, 0, tune_or(TUNE_GEMM256_STAGGER, 0)};
  if ((reinterpret_cast<uintptr_t>(W2s_frag) | reinterpret_cast<uintptr_t>(W2c_frag)) & 15) return VPR_ERR_UNSUPPORTED;
  Fuse2 f{{W2s, W2c}, {W2s_frag, W2c_frag}, {b2s, b2c}, {S, F}, {m, l}, hidden};
  static PerDeviceFlag attr = {};
  VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm256_fuse2_kernel), G2_LDS, attr));
  VPR_TRY_LAUNCH(launch_kernel(gemm256_fuse2_kernel, dim3(g.tiles_m * g.tiles_n), dim3(512), G2_LDS, stream, g, f));
  return VPR_OK;
}

#ifdef VPR_ABLATION
extern "C" int vpr_gemm256_fuse2_set_clocks(long long* clocks) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fuse2_clocks), &clocks, sizeof(clocks)) == hipSuccess ? VPR_OK : VPR_ERR_LAUNCH;
}
#endif
// e4m3 operands with per-row scales, f32 out: C[m][n] = a_scale[m] * w_scale[n] * sum_k A[m][k] W[n][k]; K % 128 == 0, K >= 256.
int launch_gemm256_fp8(const uint8_t* A, int lda, const float* a_scale, const uint8_t* W, int ldw, const float* w_scale,
                       float* C, int ldc, int M, int N, int K, hipStream_t stream, int ksplit, long long slab_stride) {
  if (!A || !W || !C || !a_scale || !w_scale || M <= 0 || N <= 0 || K <= 0) return VPR_ERR_INVALID_ARG;
  if (ksplit < 1) ksplit = 1;
  if (K / 128 < 2 * ksplit) return VPR_ERR_UNSUPPORTED;
  if ((K % 128) || K < 256 || lda < K || ldw < K || ldc < N || (lda % 16) || (ldw % 16) || (ldc % 4)) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(C)) & 15)
    return VPR_ERR_UNSUPPORTED;
  GemmProblem g{reinterpret_cast<const uint16_t*>(A), lda, 0, 0, reinterpret_cast<const uint16_t*>(W), ldw, nullptr, 0,
                C, ldc, 0, M, N, K, (M + G2_BM - 1) / G2_BM, (N + G2_BN - 1) / G2_BN, a_scale, w_scale, ksplit, slab_stride,
                tune_or(TUNE_GEMM256_STAGGER, 0)};
  static PerDeviceFlag attr = {};
  VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(gemm256_kernel<true>), G2_LDS, attr));
  VPR_TRY_LAUNCH(launch_kernel(gemm256_kernel<true>, dim3(g.tiles_m * g.tiles_n, ksplit), dim3(512), G2_LDS, stream, g));
  return VPR_OK;
}

}  // namespace vpr

using namespace vpr;

extern "C" int vpr_salad_pack_w2_fragments(const uint16_t* w2, int n_out, int hidden, uint16_t* out, void* stream) {
  return launch_pack_w2_fragments(w2, n_out, hidden, out, static_cast<hipStream_t>(stream));
}

// Same contract as vpr_gemm_nt_bf16 (K >= 128, ldc % 4 == 0, C 16-byte aligned), 256 x 256 tiles.
extern "C" int vpr_gemm256_nt_bf16(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                                   const uint16_t* W, int ldw, const float* bias, int relu, void* C, int ldc,
                                   int out_is_bf16, int M, int N, int K, void* stream) {
  const GemmProblem g{A, lda, a_group_rows, a_group_stride, W, ldw, bias, relu, C, ldc, out_is_bf16, M, N, K, 0, 0};
  return launch_gemm256(g, static_cast<hipStream_t>(stream));
}
