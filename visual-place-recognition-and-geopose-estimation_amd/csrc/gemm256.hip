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
//     `s_waitcnt vmcnt(10)` (never 0 inside the loop) + one raw s_barrier retire exactly the
//     half-tile the phase is about to read; past the last K-tile the same addresses are
//     re-requested into slots nobody reads again, which keeps the count constant.
// Operand roles are swapped in the MFMA (A operand = W rows, B operand = A rows) so that a lane
// ends up with 4 consecutive output columns: 8-byte bf16 / 16-byte f32 stores.
#include "vpr_common.cuh"
#include "vpr_internal.h"

namespace vpr {

constexpr int G2_BM = 256, G2_BN = 256;
constexpr int G2_BUF_BYTES = (G2_BM + G2_BN) * TILE_ROW_BYTES;   // 64 KB per K-tile

// first tile row of 8-row staging group `ge` (0..15) of a half-tile
__device__ __forceinline__ int g2_a_row(int ge, bool late) { return (ge < 8 ? ge * 8 : 128 + (ge - 8) * 8) + (late ? 64 : 0); }
__device__ __forceinline__ int g2_w_row(int ge, bool late) { return (ge >> 2) * 64 + (ge & 3) * 8 + (late ? 32 : 0); }

__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmProblem pr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint16_t* __restrict__ A = pr.A;
  const uint16_t* __restrict__ W = pr.W;
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
  const int m0 = (tile / pr.tiles_n) * G2_BM, n0 = (tile % pr.tiles_n) * G2_BN;

  // ---- staging: this wave owns groups 2*wave, 2*wave+1 of each of the four half-tiles ----
  // index h: 0 = A-early, 1 = W-early, 2 = W-late, 3 = A-late
  const uint16_t* src[4][2];
  int dst[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ge = 2 * wave + j;
      const bool isA = (h == 0 || h == 3), late = (h >= 2);
      const int row0 = isA ? g2_a_row(ge, late) : g2_w_row(ge, late);
      const int tr = row0 + (lane >> 3);
      const int sw = ((lane & 7) ^ ((tr >> 1) & 7)) << 3;
      if (isA) {
        int r = m0 + tr;
        r = r < M ? r : M - 1;
        const uint16_t* p = pr.a_group_rows > 0
            ? A + (long long)(r / pr.a_group_rows) * pr.a_group_stride + (long long)(r % pr.a_group_rows) * pr.lda
            : A + (long long)r * pr.lda;
        src[h][j] = p + sw;
        dst[h][j] = row0 * TILE_ROW_BYTES;
      } else {
        int r = n0 + tr;
        r = r < N ? r : N - 1;
        src[h][j] = W + (long long)r * pr.ldw + sw;
        dst[h][j] = G2_BM * TILE_ROW_BYTES + row0 * TILE_ROW_BYTES;
      }
    }
  const int nk = K >> 6;
  auto issue = [&](int h, int kt) {   // half-tile h of K-tile kt -> buffer kt & 1 (kt clamped: dummy tail)
    const int kc = kt < nk ? kt : nk - 1;
    char* base = smem + (kc & 1) * G2_BUF_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(src[h][j] + kc * 64, base + dst[h][j]);
  };

  f32x4 acc[2][2][4][2];   // [qi][qj][rb][cb]: rows n = wc*64 + qj*32 + cb*16 + 4g+e, cols m = wr*128 + qi*64 + rb*16 + lane&15
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
  for (int kt = 0; kt < nk; ++kt) {
    const char* ta = smem + (kt & 1) * G2_BUF_BYTES;
    const char* tw = ta + G2_BM * TILE_ROW_BYTES;
    bf16x8 a[4][2], b0[2][2], b1[2][2];

    // ---- phase 1: A rows of quadrant-row 0, W rows of quadrant-col 0 ----
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int s = 0; s < 2; ++s) b0[cb][s] = lds_frag(tw, wc * 64 + cb * 16 + frow, fch + 4 * s);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int s = 0; s < 2; ++s) a[rb][s] = lds_frag(ta, wr * 128 + rb * 16 + frow, fch + 4 * s);
    issue(3, kt + 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[0][0][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[cb][s], a[rb][s], acc[0][0][rb][cb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);

    // ---- phase 2: W rows of quadrant-col 1 ----
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int s = 0; s < 2; ++s) b1[cb][s] = lds_frag(tw, wc * 64 + 32 + cb * 16 + frow, fch + 4 * s);
    issue(0, kt + 2);          // A-early slot: every wave finished its phase-1 reads before this barrier
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[0][1][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[cb][s], a[rb][s], acc[0][1][rb][cb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);

    // ---- phase 3: A rows of quadrant-row 1 ----
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int s = 0; s < 2; ++s) a[rb][s] = lds_frag(ta, wr * 128 + 64 + rb * 16 + frow, fch + 4 * s);
    issue(1, kt + 2);          // W-early slot (read in phase 1)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[1][1][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[cb][s], a[rb][s], acc[1][1][rb][cb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);

    // ---- phase 4: no new operands (W quadrant-col 0 is still in registers) ----
    issue(2, kt + 2);          // W-late slot: read in phase 2, and every wave has passed phase 3's barrier since
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[1][0][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[cb][s], a[rb][s], acc[1][0][rb][cb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy tail before the workgroup exits

  // ---- epilogue: C/D of (W-operand-as-A): row = n offset 4g+e, col = m (lane&15) ----
  const int g = lane >> 4;
#pragma unroll
  for (int qi = 0; qi < 2; ++qi)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int m = m0 + wr * 128 + qi * 64 + rb * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int qj = 0; qj < 2; ++qj)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int n = n0 + wc * 64 + qj * 32 + cb * 16 + 4 * g;
          if (n >= N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[qi][qj][rb][cb][e] + (pr.bias != nullptr && n + e < N ? pr.bias[n + e] : 0.f);
            if (pr.relu) v[e] = fmaxf(v[e], 0.f);
          }
          if (n + 3 < N) {
            if (pr.out_is_bf16) {
              ushort4 o;
              o.x = f32_to_bf16_bits(v[0]); o.y = f32_to_bf16_bits(v[1]); o.z = f32_to_bf16_bits(v[2]); o.w = f32_to_bf16_bits(v[3]);
              *reinterpret_cast<ushort4*>(reinterpret_cast<uint16_t*>(pr.C) + (long long)m * pr.ldc + n) = o;
            } else {
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(pr.C) + (long long)m * pr.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
          } else {
            for (int e = 0; e < 4 && n + e < N; ++e) {
              if (pr.out_is_bf16) reinterpret_cast<uint16_t*>(pr.C)[(long long)m * pr.ldc + n + e] = f32_to_bf16_bits(v[e]);
              else reinterpret_cast<float*>(pr.C)[(long long)m * pr.ldc + n + e] = v[e];
            }
          }
        }
    }
}

int launch_gemm256(const GemmProblem& in, hipStream_t stream) {
  GemmProblem g = in;
  if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return VPR_ERR_INVALID_ARG;
  if (g.K % 64 != 0 || g.K < 128 || g.lda < g.K || g.ldw < g.K || g.ldc < g.N) return VPR_ERR_UNSUPPORTED;
  if ((g.lda % 8) || (g.ldw % 8) || (g.ldc % 4) || (g.a_group_rows > 0 && (g.a_group_stride % 8))) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.C)) & 15)
    return VPR_ERR_UNSUPPORTED;
  g.tiles_m = (g.M + G2_BM - 1) / G2_BM;
  g.tiles_n = (g.N + G2_BN - 1) / G2_BN;
  constexpr size_t lds = 2 * (size_t)G2_BUF_BYTES;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return VPR_ERR_LAUNCH;
    attr = true;
  }
  VPR_TRY_LAUNCH(launch_kernel(gemm256_kernel, dim3(g.tiles_m * g.tiles_n), dim3(512), lds, stream, g));
  return VPR_OK;
}

}  // namespace vpr

using namespace vpr;

// Same contract as vpr_gemm_nt_bf16 (K >= 128, ldc % 4 == 0, C 16-byte aligned), 256 x 256 tiles.
extern "C" int vpr_gemm256_nt_bf16(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                                   const uint16_t* W, int ldw, const float* bias, int relu, void* C, int ldc,
                                   int out_is_bf16, int M, int N, int K, void* stream) {
  const GemmProblem g{A, lda, a_group_rows, a_group_stride, W, ldw, bias, relu, C, ldc, out_is_bf16, M, N, K, 0, 0};
  return launch_gemm256(g, static_cast<hipStream_t>(stream));
}
