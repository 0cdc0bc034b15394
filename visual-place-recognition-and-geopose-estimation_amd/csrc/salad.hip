// salad.hip — SALAD optimal-transport aggregation (arXiv:2311.15937).
//
// Replaces the aggregator inside `feature_extractor(x)` of
// dinov2salad/dinov2salad_validation.py:49-51 (hub model loaded at :65, output width 8448
// pinned at :44).  The arithmetic is not in the reference tree; oracle/salad.py restates the
// published algorithm and tests/test_salad*.py pin both against closed-form known answers.
//
// Stages (round 3; each has its own C entry point, the one-call forms run them in one stream):
//   T  vpr_salad_stage_token      skinny_linear_kernel x2: g = W2_t relu(W1_t cls + b1_t) + b2_t   [B, t] f32   (token MLP; in
//                                 the pipeline it runs on the backbone's cls-row stream)
//   M  vpr_salad_stage_mlps       gemm256_fuse2_kernel: H = relu(X W1_sc^T + b1) tile by tile, each tile multiplied from LDS
//                                 with its W2 slice -> S [2][B*n, m], F [2][B*n, l] f32 partial-sum slabs (no H in HBM)
//   A  vpr_salad_stage_aggregate  sinkhorn_aggregate_kernel (one workgroup per image): slab sum, dustbin row, Sinkhorn in the
//                                 exp domain (K = exp(M - rowmax) once, then alpha / beta updates: matrix-vector products),
//                                 P as two bf16 planes, V = F^T P^T as three exact-product bf16 MFMAs on the (hi, lo) terms
//                                 (f32 accuracy), the three L2 normalisations, 8448 outputs (f32 + bf16 copy for the kNN stage)
//   vpr_salad_aggregate_f32: the same at the reference's fp32 precision (three bf16 planes per operand, f32 hidden).
#include <math.h>
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

constexpr int SA_N = 256;   // patch tokens
constexpr int SA_M = 64;    // clusters
constexpr int SA_L = 128;   // cluster dim
constexpr int SA_T = 256;   // token dim
constexpr int SA_LD = SA_N + 1;   // LDS row stride (words) of the [m+1][n] score matrix
constexpr int SA_PLD = SA_N + 8;  // row stride (bf16) of the P planes: 528 B, 16-byte aligned rows, rows 4 banks apart
// LDS: [score matrix 65 x 257 f32  |  later: two P planes 64 x 264 bf16]  then the small arrays (alpha / beta / norms)
constexpr size_t SA_BIG_BYTES = ((SA_M + 1) * SA_LD * sizeof(float) > 2 * SA_M * SA_PLD * sizeof(uint16_t)
                                     ? (SA_M + 1) * SA_LD * sizeof(float) : 2 * SA_M * SA_PLD * sizeof(uint16_t));
constexpr size_t SA_SMALL_OFF = (SA_BIG_BYTES + 15) / 16 * 16;

// value of lane `i` (compile-time constant) of a wave-distributed register, as a scalar operand: v_readlane_b32
__device__ __forceinline__ float lane_bcast(float v, int i) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  // deterministic: wave butterflies, then a fixed-order sum of the 4 wave totals
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// NSLAB = 2: scores / feats arrive as two partial-sum slabs (the fused MLP kernel's K slices of the second layers,
// gemm256_fuse2_kernel), `slab_rows` * m resp. * l elements apart; they are added here, slab 0 first.
//
// Round 3 restructure (29 -> ~15 us at B = 64 with two slabs):
//  * every global load of the kernel — the score slabs AND all 128 KB (x NSLAB) of cluster features — is issued in the
//    first microsecond; the old kernel fetched the features chunk by chunk behind the iterations, eight dependent round
//    trips of ~1 us each;
//  * the iterations run in the exp domain.  With K_ij = exp(M_ij - r_i), r_i = max_j M_ij, alpha_i = exp(u_i + r_i),
//    beta_j = exp(v_j), the log-domain updates u = log a - LSE_j(M + v), v = log b - LSE_i(M + u) are exactly
//        alpha_i = a_i / sum_j K_ij beta_j,      beta_j = b_j / sum_i K_ij alpha_i        (beta = 1 at the start),
//    i.e. one exp per matrix entry ONCE instead of two per entry per iteration (the iterations were exp-issue-bound:
//    2.6 us each), and P = K alpha beta (n + m).  Range: every row holds a K = 1 (its maximum) and so does every column
//    (the dustbin row is constant, so K = 1 along it): all sums are >= one O(1) term and alpha, beta stay within a few
//    orders of magnitude of 1 for any finite scores; entries more than e^87 below their row maximum flush to 0, which is
//    where the log-domain form's exp(x - max) puts them too.
//  * the two bf16 planes of P re-use the LDS of the score matrix (dead once K sits in registers);
//  * barriers up to the aggregation order LDS traffic only (lds_barrier(): no vmcnt(0)), so the 128-256 KB of cluster
//    features keep streaming in behind the score-dependent critical path instead of being waited for at the first
//    __syncthreads() (a CU pulls ~70 GB/s: 384 KB of input are 5.6 us when waited for up front).
// Phase clocks (timing-only build: `make ablation`, scripts/sinkhorn_phases.py): with VPR_ABLATION defined, thread 0 of
// every workgroup stores s_memrealtime (100 MHz) at the phase boundaries into a 16-slot record behind the bf16 output pointer's twin
// (g_sinkhorn_clocks, set through vpr_salad_sinkhorn_set_clocks); compiled out of the shipped library.
#ifdef VPR_ABLATION
__device__ long long* g_sinkhorn_clocks = nullptr;
#define SA_CLOCK(k) do { if (g_sinkhorn_clocks && threadIdx.x == 0) g_sinkhorn_clocks[blockIdx.x * 16 + (k)] = vpr_clock_now(); } while (0)
#else
#define SA_CLOCK(k) do { } while (0)
#endif

//
// NQ = 4 (round 3 experiment, off by default — see launch_sinkhorn_aggregate): FOUR workgroups per image, so that a 64-image
// batch covers all 256 CUs instead of 64 — the front of this kernel is what ONE CU can pull (~70 GB/s).  Every workgroup of an image runs the whole Sinkhorn
// on the scores (redundant, cheap: 128 KB of loads, ~7 us of latency-bound arithmetic) but aggregates only its quarter of the
// cluster dims (32 of 128: a quarter of the feature bytes, of the MFMAs and of the output); inside the workgroup the four
// waves split the TOKENS (64 each) and their partial V meet in LDS.  The per-cluster norm runs over all 128 cluster dims:
// each workgroup stores its un-normalised quarter of V straight into its final place in out_f32 and its partial sums of
// squares into the image's token slots of out_f32 (written last anyway) — agent-scope write-through stores — and bumps the
// image's arrival counter; the workgroup that arrives last (no spinning: the others exit) adds the four partials in quarter
// order, normalises the whole row in place and writes the token part.  Same fixed summation orders whoever finishes:
// bitwise reproducible.  `counters`: one int per image, zero before the first call, left zero (workspace contract).
template <int NSLAB, int NQ>
__global__ __launch_bounds__(256, 1) void sinkhorn_aggregate_kernel(
    const float* __restrict__ scores,   // [NSLAB][B, n, m]
    const float* __restrict__ feats,    // [NSLAB][B, n, l]
    const float* __restrict__ tokfeat,  // [B, t]
    long long slab_rows, float dustbin, int iters,
    float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16, int* __restrict__ counters, int rows_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Mx = reinterpret_cast<float*>(smem);        // [65][257] raw scores; later the two P planes
  uint16_t* Phi = reinterpret_cast<uint16_t*>(smem);      // [64][SA_PLD] bf16: hi plane of P (16-byte aligned rows)
  uint16_t* Plo = Phi + SA_M * SA_PLD;                    // lo plane: P = hi + lo to 2^-17
  float* al = reinterpret_cast<float*>(smem + SA_SMALL_OFF);   // [65] (+3 pad): row maxima, then alpha
  float* be = al + 68;                               // [256] beta
  float* ssq = be + SA_N;                            // [4][64]
  float* red = ssq + 4 * SA_M;                       // [4]
  // rows_per_xcd > 0 (A/B, VPR_SALAD_VARIANT=4): workgroup i runs on XCD i & 7 (round-robin dispatch); take the image whose
  // score / feature slabs the MLP kernel's tiles wrote from THAT XCD (gemm256's tile map gives XCD x the row tiles
  // [x * rows_per_xcd, (x + 1) * rows_per_xcd)), in case its L2 still holds them
  const int b = NQ == 4 ? (int)(blockIdx.x >> 2)
                        : rows_per_xcd > 0 ? rows_per_xcd * (int)(blockIdx.x & 7) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int qd = NQ == 4 ? (int)(blockIdx.x & 3) : 0;          // which quarter of the cluster dims this workgroup aggregates
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int D_OUT = SA_T + SA_L * SA_M;

  SA_CLOCK(0);
  // ---- all global loads up front ----
  const float* sb = scores + (long long)b * SA_N * SA_M;
  const float4* sb4 = reinterpret_cast<const float4*>(sb);
  float4 q[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) q[it] = sb4[tid + 256 * it];
  if constexpr (NSLAB == 2) {
    const float4* sb4b = reinterpret_cast<const float4*>(sb + slab_rows * SA_M);
    float4 q2[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) q2[it] = sb4b[tid + 256 * it];
#pragma unroll
    for (int it = 0; it < 16; ++it) { q[it].x += q2[it].x; q[it].y += q2[it].y; q[it].z += q2[it].z; q[it].w += q2[it].w; }
  }
  // cluster features of this wave's 32 cluster-dim rows, all 256 tokens (32x32x16 A-operand map:
  // A[i = lane & 31][k = 8 * (lane >> 5) + e]); token = 32 * ch + 16 * (i >> 3) + 8 * kh + (i & 7).
  // (Tried: requesting them in chunks of 32 loads spread over the score-dependent phases, so that the wave never stalls
  // on its 64-entry load queue — the phases just shift: what bounds the front of this kernel is the ~70 GB/s one CU
  // pulls, 192-384 KB of input, not the order of the requests.)
  const float* fb = feats + (long long)b * SA_N * SA_L;
  const int l0 = NQ == 4 ? 32 * qd : 32 * wave;
  const int kh = lane >> 5, li = lane & 31;
  // NQ = 1: the wave's 32 cluster dims x all 256 tokens (8 chunks of 32 tokens); NQ = 4: the workgroup's 32 cluster dims x
  // the WAVE's 64 tokens (2 chunks): token = tok0 + 32 * ch + 16 * (i >> 3) + 8 * kh + (i & 7)
  constexpr int NCH = NQ == 4 ? 2 : SA_N / 32;
  const int tok0 = NQ == 4 ? 64 * wave : 0;
  float fa[NCH][16];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < 16; ++i) fa[ch][i] = fb[(tok0 + 32 * ch + 16 * (i >> 3) + 8 * kh + (i & 7)) * SA_L + l0 + li];
  float g = tokfeat[(long long)b * SA_T + tid];

  SA_CLOCK(1);      // loads issued
  // ---- scores transposed into LDS: Mx[i][j] = scores[b][j][i]; dustbin row i = m ----
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int e4 = tid + 256 * it;
    const int j = e4 >> 4, i = (e4 & 15) * 4;
    Mx[(i + 0) * SA_LD + j] = q[it].x;
    Mx[(i + 1) * SA_LD + j] = q[it].y;
    Mx[(i + 2) * SA_LD + j] = q[it].z;
    Mx[(i + 3) * SA_LD + j] = q[it].w;
  }
  Mx[SA_M * SA_LD + tid] = dustbin;
  be[tid] = 1.f;
  // second feature slab: requested now (the score registers are free again), added once K is computed
  float fa2[NSLAB == 2 ? NCH : 1][16];
  if constexpr (NSLAB == 2) {
    const float* fb2 = fb + slab_rows * SA_L;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int i = 0; i < 16; ++i) fa2[ch][i] = fb2[(tok0 + 32 * ch + 16 * (i >> 3) + 8 * kh + (i & 7)) * SA_L + l0 + li];
  }
  lds_barrier();

  SA_CLOCK(2);      // scores arrived and staged
  // every thread keeps its operands of both half-steps in registers: its column (65 values) and its 16 entries of each
  // of its 5 rows (one 16-lane group per matrix row, 16 rows in flight per workgroup)
  const int grp = tid >> 4, l16 = tid & 15;
  float mrow[5][16], mcol[SA_M + 1];
#pragma unroll
  for (int p5 = 0; p5 < 5; ++p5) {
    const int ic = min(16 * p5 + grp, SA_M);        // clamp (EXEC stays full for the DPP steps)
#pragma unroll
    for (int c = 0; c < 16; ++c) mrow[p5][c] = Mx[ic * SA_LD + l16 + 16 * c];
  }
#pragma unroll
  for (int i = 0; i <= SA_M; ++i) mcol[i] = Mx[i * SA_LD + tid];
  // row maxima r_i, then K = exp(M - r) in both register copies
#pragma unroll
  for (int p5 = 0; p5 < 5; ++p5) {
    float mx = mrow[p5][0];
#pragma unroll
    for (int c = 1; c < 16; ++c) mx = fmaxf(mx, mrow[p5][c]);
    mx = row16_max(mx);
    const int i = 16 * p5 + grp;
    if (l16 == 0 && i <= SA_M) al[i] = mx;
#pragma unroll
    for (int c = 0; c < 16; ++c) mrow[p5][c] = __expf(mrow[p5][c] - mx);
  }
  lds_barrier();                                       // (also: every thread is done reading Mx)
  // per-row values (row maxima here, alpha below) reach a thread's column loop as ONE LDS read per lane + v_readlane:
  // 65 broadcast ds_reads, which the compiler serialises behind s_waitcnt one by one under this register pressure,
  // cost 1.5 us per loop (phase clocks, scripts/sinkhorn_phases.py)
  {
    const float r_lane = al[lane], r_dust = al[SA_M];
#pragma unroll
    for (int i = 0; i < SA_M; ++i) mcol[i] = __expf(mcol[i] - lane_bcast(r_lane, i));
    mcol[SA_M] = __expf(mcol[SA_M] - r_dust);
  }
  lds_barrier();                                       // al[] is about to be rewritten as alpha
  if constexpr (NSLAB == 2) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int i = 0; i < 16; ++i) fa[ch][i] += fa2[ch][i];
  }

  SA_CLOCK(3);      // K in registers (and, NSLAB = 2, both feature slabs arrived)
  const float inv_nm = 1.f / (float)(SA_N + SA_M);
  const float a_reg = inv_nm, a_dust = (float)(SA_N - SA_M) * inv_nm, b_col = inv_nm;
  for (int it = 0; it < iters; ++it) {
    {                                                  // alpha_i = a_i / sum_j K_ij beta_j
      float bv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) bv[c] = be[l16 + 16 * c];
#pragma unroll
      for (int p5 = 0; p5 < 5; ++p5) {
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sx = fmaf(mrow[p5][c], bv[c], sx);
        sx = row16_sum(sx);
        const int i = 16 * p5 + grp;
        if (l16 == 0 && i <= SA_M) al[i] = (i == SA_M ? a_dust : a_reg) / sx;
      }
    }
    lds_barrier();  
    {                                                  // beta_j = b_j / sum_i K_ij alpha_i
      const float a_lane = al[lane];
      float s0 = mcol[SA_M] * al[SA_M], s1 = 0.f;      // two chains: half the dependent-FMA latency
#pragma unroll
      for (int i = 0; i < SA_M; i += 2) {
        s0 = fmaf(mcol[i], lane_bcast(a_lane, i), s0);
        s1 = fmaf(mcol[i + 1], lane_bcast(a_lane, i + 1), s1);
      }
      be[tid] = b_col / (s0 + s1);
    }
    lds_barrier();
  }

  SA_CLOCK(4);      // iterations done
  // ---- P = K alpha beta (n + m); dustbin row dropped.  Two bf16 planes (hi = bf16(P), lo = bf16(P - hi)) over the
  // LDS of the score matrix (nobody reads Mx after the barrier above the iterations). ----
  {
    const int j = tid;
    const float bj = be[j] * (float)(SA_N + SA_M);
    const float a_lane = al[lane];
#pragma unroll
    for (int i = 0; i < SA_M; ++i) {
      const float pv = mcol[i] * lane_bcast(a_lane, i) * bj;
      const __bf16 h = (__bf16)pv;
      const __bf16 l = (__bf16)(pv - (float)h);
      Phi[i * SA_PLD + j] = __builtin_bit_cast(uint16_t, h);
      Plo[i * SA_PLD + j] = __builtin_bit_cast(uint16_t, l);
    }
  }
  lds_barrier();

  SA_CLOCK(5);      // P planes in LDS
  // ---- V[l][m] = sum_j F[j][l] * P[m][j] on the bf16 matrix pipe at f32 accuracy ----
  // F and P each split into two bf16 terms (2^-17 relative): three v_mfma_f32_32x32x16_bf16 (hi*hi, hi*lo, lo*hi) with exact
  // products and f32 accumulation, error <= 2^-16 per product (~1e-8 absolute on descriptor entries of 1e-2; tolerance 1e-4).
  // 32x32x16 operand maps: A[i = lane&31][k = 8*(lane>>5) + e], B[k = 8*(lane>>5) + e][j = lane&31].
  // Wave w owns cluster-dim rows l0 = 32w .. 32w+31 and both 32-cluster column blocks.
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8 fh, fl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fh[e] = (__bf16)fa[ch][8 * st + e];
        fl[e] = (__bf16)(fa[ch][8 * st + e] - (float)fh[e]);
      }
      const int j0 = tok0 + 32 * ch + 16 * st + 8 * kh;
      const bf16x8 p0h = *reinterpret_cast<const bf16x8*>(Phi + li * SA_PLD + j0);
      const bf16x8 p0l = *reinterpret_cast<const bf16x8*>(Plo + li * SA_PLD + j0);
      const bf16x8 p1h = *reinterpret_cast<const bf16x8*>(Phi + (32 + li) * SA_PLD + j0);
      const bf16x8 p1l = *reinterpret_cast<const bf16x8*>(Plo + (32 + li) * SA_PLD + j0);
      // (the lo x lo term is 2^-34 of the product: below f32 resolution, dropped)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, p0h, acc0, 0, 0, 0);     // smallest terms first
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, p1h, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, p0l, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, p1l, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, p0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, p1h, acc1, 0, 0, 0);
    }
  }

  SA_CLOCK(6);      // aggregation MFMAs issued (features arrived)
  if constexpr (NQ == 4) {
    // ---- the four waves' partial V [32 l][64 m] (each over its 64 tokens) meet in LDS; thread t then owns cluster m = t & 63,
    //      cluster dims lq + 4 i (lq = t >> 6, i = 0..7) of this quarter ----
    float* vred = red + 4;                                  // [4 waves][32][64] f32 = 32 KB behind the small arrays
    __shared__ int s_last;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int l = (e & 3) + 8 * (e >> 2) + 4 * kh;        // C/D map of 32x32: col (cluster) = lane & 31, row (l) as here
      vred[(wave * 32 + l) * SA_M + li] = acc0[e];
      vred[(wave * 32 + l) * SA_M + 32 + li] = acc1[e];
    }
    __syncthreads();
    const int m = tid & 63, lq = tid >> 6;
    float v[8], s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int l = lq + 4 * i;
      v[i] = (vred[(0 * 32 + l) * SA_M + m] + vred[(1 * 32 + l) * SA_M + m]) + (vred[(2 * 32 + l) * SA_M + m] + vred[(3 * 32 + l) * SA_M + m]);
      s2 = fmaf(v[i], v[i], s2);
    }
    ssq[lq * SA_M + m] = s2;
    float* ob = out_f32 + (long long)b * D_OUT;
#pragma unroll
    for (int i = 0; i < 8; ++i)                             // un-normalised quarter, in its final place (agent-scope write-through)
      __hip_atomic_store(ob + SA_T + (32 * qd + lq + 4 * i) * SA_M + m, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid < SA_M)                                         // this quarter's sums of squares -> the image's token slots (scratch until the end)
      __hip_atomic_store(ob + qd * SA_M + tid, (ssq[tid] + ssq[SA_M + tid]) + (ssq[2 * SA_M + tid] + ssq[3 * SA_M + tid]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // write-through stores acknowledged = visible device-wide
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3);
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) __hip_atomic_store(counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SA_CLOCK(7);
    // ---- finisher: per-cluster norm over all 128 cluster dims (quarter order), the whole row normalised in place ----
    const float den = fmaxf(sqrtf((ob[m] + ob[SA_M + m]) + (ob[2 * SA_M + m] + ob[3 * SA_M + m])), 1e-12f);
    float w[32], part = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = ob[SA_T + (lq + 4 * i) * SA_M + m];
#pragma unroll
    for (int i = 0; i < 32; ++i) { w[i] = w[i] / den; part = fmaf(w[i], w[i], part); }
    const float gn = fmaxf(sqrtf(block_sum_256(g * g, red)), 1e-12f);      // (its barriers also fence the reads of the scratch slots)
    g = g / gn;
    const float tot = block_sum_256(part + g * g, red);
    const float gden = fmaxf(sqrtf(tot), 1e-12f);
    uint16_t* obh = out_bf16 ? out_bf16 + (long long)b * D_OUT : nullptr;
    {
      const float o = g / gden;
      ob[tid] = o;
      if (obh) obh[tid] = f32_to_bf16_bits(o);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float o = w[i] / gden;
      const int idx = SA_T + (lq + 4 * i) * SA_M + m;
      ob[idx] = o;
      if (obh) obh[idx] = f32_to_bf16_bits(o);
    }
    SA_CLOCK(8);
    return;
  }
  // ---- per-cluster L2 norm over l (F.normalize dim=1, eps 1e-12) ----
  {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0 += acc0[e] * acc0[e]; s1 += acc1[e] * acc1[e]; }
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (lane < 32) { ssq[wave * SA_M + lane] = s0; ssq[wave * SA_M + 32 + lane] = s1; }
  }
  __syncthreads();
  const float den0 = fmaxf(sqrtf((ssq[li] + ssq[SA_M + li]) + (ssq[2 * SA_M + li] + ssq[3 * SA_M + li])), 1e-12f);
  const float den1 = fmaxf(sqrtf((ssq[32 + li] + ssq[SA_M + 32 + li]) + (ssq[2 * SA_M + 32 + li] + ssq[3 * SA_M + 32 + li])), 1e-12f);
  float part = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc0[e] = acc0[e] / den0;
    acc1[e] = acc1[e] / den1;
    part += acc0[e] * acc0[e] + acc1[e] * acc1[e];
  }

  // ---- token vector: F.normalize(g) ----
  const float gn = fmaxf(sqrtf(block_sum_256(g * g, red)), 1e-12f);
  g = g / gn;

  // ---- global L2 over the concatenation [g | V.flatten] ----
  const float tot = block_sum_256(part + g * g, red);
  const float gden = fmaxf(sqrtf(tot), 1e-12f);

  SA_CLOCK(7);      // norms done
  float* ob = out_f32 + (long long)b * D_OUT;
  uint16_t* obh = out_bf16 ? out_bf16 + (long long)b * D_OUT : nullptr;
  {
    const float o = g / gden;
    ob[tid] = o;
    if (obh) obh[tid] = f32_to_bf16_bits(o);
  }
  // C/D map of 32x32: col (cluster) = lane&31, row (l) = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int l = l0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    const float o0 = acc0[e] / gden, o1 = acc1[e] / gden;
    const int base = SA_T + l * SA_M;
    ob[base + li] = o0;
    ob[base + 32 + li] = o1;
    if (obh) { obh[base + li] = f32_to_bf16_bits(o0); obh[base + 32 + li] = f32_to_bf16_bits(o1); }
  }
  SA_CLOCK(8);
}

constexpr size_t SINKHORN_LDS = SA_SMALL_OFF + (68 + SA_N + 4 * SA_M + 4) * sizeof(float);
constexpr size_t SINKHORN_LDS_Q = SINKHORN_LDS + 4 * 32 * SA_M * sizeof(float);        // + the cross-wave V buffer of the NQ = 4 form

int launch_sinkhorn_aggregate(const float* scores, const float* feats, const float* tokfeat,
                              int B, int n, int m, int l, int t, float dustbin, int iters,
                              float* out_f32, uint16_t* out_bf16, hipStream_t stream, int nslab, long long slab_rows,
                              int* counters) {
  if (!scores || !feats || !tokfeat || !out_f32 || B <= 0 || iters < 1) return VPR_ERR_INVALID_ARG;
  if (n != SA_N || m != SA_M || l != SA_L || t != SA_T || nslab < 1 || nslab > 2) return VPR_ERR_UNSUPPORTED;
  // Four workgroups per image (NQ = 4) — built in round 3 to put all 256 CUs on a 64-image batch, measured, and left OFF:
  // stage A 27.3 us against 20.8 us with one workgroup per image (scripts/salad_ab.py).  Every workgroup needs the whole
  // score matrix, so the batch reads 4 x 8.4 MB of scores + 16.8 MB of features = 50 MB at once and the front becomes
  // HBM-bound (5.9 us to issue the loads instead of 3.1), and the finisher's second pass over the row costs 4.2 us, which
  // eats what the quartered aggregation saves.  VPR_SALAD_VARIANT=3 selects it (A/B); it needs the zeroed counter area.
  const bool quarters = counters != nullptr && 4 * B <= device_cu_count() && tune_or(TUNE_SALAD_VARIANT, 0) == 3;
  const int rpx = (!quarters && nslab == 2 && B % 8 == 0 && tune_or(TUNE_SALAD_VARIANT, 0) == 4) ? B / 8 : 0;
#define VPR_SINKHORN_LAUNCH(NS, NQV, LDS, GRID)                                                                              \
  do {                                                                                                                       \
    static PerDeviceFlag attr = {};   /* > 64 KiB of dynamic LDS needs the opt-in once per device */                         \
    VPR_TRY_LAUNCH(optin_dynamic_lds(reinterpret_cast<const void*>(sinkhorn_aggregate_kernel<NS, NQV>), LDS, attr));         \
    VPR_TRY_LAUNCH(launch_kernel((sinkhorn_aggregate_kernel<NS, NQV>), dim3(GRID), dim3(256), LDS, stream, scores, feats, tokfeat, \
                                 nslab == 2 ? slab_rows : 0LL, dustbin, iters, out_f32, out_bf16, counters, rpx));            \
  } while (0)
  if (quarters) {
    if (nslab == 2) VPR_SINKHORN_LAUNCH(2, 4, SINKHORN_LDS_Q, 4 * B); else VPR_SINKHORN_LAUNCH(1, 4, SINKHORN_LDS_Q, 4 * B);
  } else {
    if (nslab == 2) VPR_SINKHORN_LAUNCH(2, 1, SINKHORN_LDS, B); else VPR_SINKHORN_LAUNCH(1, 1, SINKHORN_LDS, B);
  }
#undef VPR_SINKHORN_LAUNCH
  return VPR_OK;
}

struct SaladPlan {
  size_t off_cnt, off_H, off_S, off_F, off_Ht, off_g, total;
};
constexpr size_t SALAD_COUNTER_BYTES = 4096;     // arrival counters of the four-workgroups-per-image Sinkhorn kernel: a FIXED area at
                                                 // the head of the workspace (zero before first use, left zero), whatever the shape
static bool salad_plan(int B, int n, int C, int m, int l, int t, int hidden, SaladPlan* p) {
  if (B <= 0 || n <= 0 || C <= 0 || m <= 0 || l <= 0 || t <= 0 || hidden <= 0) return false;
  p->off_cnt = 0;
  size_t off = SALAD_COUNTER_BYTES;
  p->off_H = off;  off += align_up((size_t)B * n * 2 * hidden * sizeof(uint16_t), 256);     // unfused route only
  p->off_S = off;  off += align_up((size_t)2 * B * n * m * sizeof(float), 256);              // up to two partial-sum slabs
  p->off_F = off;  off += align_up((size_t)2 * B * n * l * sizeof(float), 256);
  p->off_Ht = off; off += align_up((size_t)B * hidden * sizeof(uint16_t), 256);
  p->off_g = off;  off += align_up((size_t)B * t * sizeof(float), 256);
  p->total = off;
  return true;
}

// Partial-sum slabs the MLP stage leaves in S / F: 2 = second layers fused into the layer-1 tile epilogue
// (gemm256_fuse2_kernel; hidden = 512 -> two 256-column K slices per head), 1 = separate launches (hidden activations
// through HBM).  One rule, used by the MLP stage and the aggregation stage.  VPR_SALAD_VARIANT=1 forces the unfused route (A/B).
static int salad_slabs(int n, int C, int m, int l, int hidden) {
  if (tune_or(TUNE_SALAD_VARIANT, 0) == 1) return 1;
  const bool fused = n == SA_N && hidden == 512 && (C % 64) == 0 && C >= 128 && m == SA_M && l == SA_L;
  return fused ? 2 : 1;
}

struct SaladArgs {
  int B, n, C, m, l, t, hidden;
  const vpr_salad_weights* w;
  char* ws;
  SaladPlan p;
};
static int salad_args(SaladArgs* a, int B, int n, int C, int m, int l, int t, int hidden, const vpr_salad_weights* w,
                      void* workspace, size_t workspace_bytes, bool need_w) {
  if (!workspace || B <= 0 || n < 1 || C <= 0) return VPR_ERR_INVALID_ARG;
  if (need_w) {
    if (!w) return VPR_ERR_INVALID_ARG;
    if (!w->w1_sc || !w->b1_sc || !w->w2_s || !w->b2_s || !w->w2_c || !w->b2_c || !w->w1_t || !w->b1_t || !w->w2_t || !w->b2_t)
      return VPR_ERR_INVALID_ARG;
  }
  if (n != SA_N || m != SA_M || l != SA_L || t != SA_T || (C % 64) || (hidden % 64)) return VPR_ERR_UNSUPPORTED;
  if (!salad_plan(B, n, C, m, l, t, hidden, &a->p)) return VPR_ERR_INVALID_ARG;
  if (workspace_bytes < a->p.total) return VPR_ERR_WORKSPACE;
  a->B = B; a->n = n; a->C = C; a->m = m; a->l = l; a->t = t; a->hidden = hidden; a->w = w;
  a->ws = static_cast<char*>(workspace);
  return VPR_OK;
}

// Stage T: token MLP on the B cls rows -> g [B, t] f32 in the workspace.  Independent of stage M (other inputs, other
// workspace regions): a caller with two streams runs it beside the big GEMM (ops.salad_aggregate_split does).
static int salad_stage_token(const SaladArgs& a, const uint16_t* cls, long long cls_stride, hipStream_t stream) {
  if (!cls) return VPR_ERR_INVALID_ARG;
  if ((cls_stride % 8) || cls_stride > 0x7fffffffLL) return VPR_ERR_UNSUPPORTED;
  uint16_t* Ht = reinterpret_cast<uint16_t*>(a.ws + a.p.off_Ht);
  float* g = reinterpret_cast<float*>(a.ws + a.p.off_g);
  const vpr_salad_weights* w = a.w;
  int st = launch_skinny_linear(cls, (int)cls_stride, w->w1_t, a.C, w->b1_t, 0, 3, Ht, a.hidden, a.B, a.hidden, a.C, nullptr, nullptr, stream);
  if (st == VPR_ERR_UNSUPPORTED) {
    const GemmProblem l1{cls, (int)cls_stride, 0, 0, w->w1_t, a.C, w->b1_t, 1, Ht, a.hidden, 1, a.B, a.hidden, a.C, 0, 0};
    st = launch_gemm_nt_group(&l1, 1, stream);
  }
  if (st != VPR_OK) return st;
  // layer 2: the same weight-streaming kernel with f32 output (mode 5); 4 us against 9 us on the 128-row GEMM tile
  st = launch_skinny_linear(Ht, a.hidden, w->w2_t, a.hidden, w->b2_t, 0, 5, reinterpret_cast<uint16_t*>(g), a.t, a.B, a.t, a.hidden,
                            nullptr, nullptr, stream);
  if (st != VPR_ERR_UNSUPPORTED) return st;
  const GemmProblem l2{Ht, a.hidden, 0, 0, w->w2_t, a.hidden, w->b2_t, 0, g, a.t, 0, a.B, a.t, a.hidden, 0, 0};
  return launch_gemm_nt_group(&l2, 1, stream);
}

// Stage M: score + cluster MLPs on the B*n patch rows -> S, F (salad_slabs() partial-sum slabs) in the workspace.
// patch row r of image b at patch + b*patch_img_stride + r*C (addressed in place).
static int salad_stage_mlps(const SaladArgs& a, const uint16_t* patch, long long patch_img_stride, hipStream_t stream) {
  if (!patch) return VPR_ERR_INVALID_ARG;
  if (patch_img_stride % 8) return VPR_ERR_UNSUPPORTED;
  uint16_t* H = reinterpret_cast<uint16_t*>(a.ws + a.p.off_H);
  float* S = reinterpret_cast<float*>(a.ws + a.p.off_S);
  float* F = reinterpret_cast<float*>(a.ws + a.p.off_F);
  const vpr_salad_weights* w = a.w;
  const int rows = a.B * a.n, hidden = a.hidden;
  if (salad_slabs(a.n, a.C, a.m, a.l, hidden) == 2)
    return launch_salad_mlps_fused(patch, a.C, a.n, patch_img_stride, w->w1_sc, w->b1_sc, w->w2_s, w->b2_s, w->w2_c, w->b2_c,
                                   S, F, rows, a.C, hidden, a.m, a.l, stream, w->w2_s_frag, w->w2_c_frag);
  // unfused: layer 1 on the 256 x 256-tile kernel (34 of SALAD's 38 GFLOP; B tiles x 4 = one full wave of workgroups at
  // B = 64), hidden activations through HBM as bf16, the two second layers as one grouped launch
  const GemmProblem l1_sc{patch, a.C, a.n, patch_img_stride, w->w1_sc, a.C, w->b1_sc, 1, H, 2 * hidden, 1, rows, 2 * hidden, a.C, 0, 0};
  int st = launch_gemm256(l1_sc, stream);
  if (st == VPR_ERR_UNSUPPORTED) st = launch_gemm_nt_group(&l1_sc, 1, stream);
  if (st != VPR_OK) return st;
  const GemmProblem l2[2] = {
      {H, 2 * hidden, 0, 0, w->w2_s, hidden, w->b2_s, 0, S, a.m, 0, rows, a.m, hidden, 0, 0},
      {H + hidden, 2 * hidden, 0, 0, w->w2_c, hidden, w->b2_c, 0, F, a.l, 0, rows, a.l, hidden, 0, 0}};
  return launch_gemm_nt_group(l2, 2, stream);
}

// Stage A: Sinkhorn + aggregation + normalisations on what stages T and M left in the workspace.
static int salad_stage_aggregate(const SaladArgs& a, float dustbin, int sinkhorn_iters, float* out_f32, uint16_t* out_bf16,
                                 hipStream_t stream) {
  if (!out_f32) return VPR_ERR_INVALID_ARG;
  const float* S = reinterpret_cast<const float*>(a.ws + a.p.off_S);
  const float* F = reinterpret_cast<const float*>(a.ws + a.p.off_F);
  const float* g = reinterpret_cast<const float*>(a.ws + a.p.off_g);
  int* counters = (size_t)a.B * sizeof(int) <= SALAD_COUNTER_BYTES ? reinterpret_cast<int*>(a.ws + a.p.off_cnt) : nullptr;
  return launch_sinkhorn_aggregate(S, F, g, a.B, a.n, a.m, a.l, a.t, dustbin, sinkhorn_iters, out_f32, out_bf16, stream,
                                   salad_slabs(a.n, a.C, a.m, a.l, a.hidden), (long long)a.B * a.n, counters);
}

}  // namespace vpr

using namespace vpr;

extern "C" size_t vpr_salad_workspace_bytes(int B, int n, int C, int m, int l, int t, int hidden) {
  SaladPlan p;
  return salad_plan(B, n, C, m, l, t, hidden, &p) ? p.total : 0;
}

extern "C" int vpr_salad_sinkhorn_aggregate(const float* scores, const float* feats, const float* tokfeat,
                                            int B, int n, int m, int l, int t, float dustbin,
                                            int sinkhorn_iters, float* out_f32, uint16_t* out_bf16,
                                            void* stream) {
  return launch_sinkhorn_aggregate(scores, feats, tokfeat, B, n, m, l, t, dustbin, sinkhorn_iters,
                                   out_f32, out_bf16, static_cast<hipStream_t>(stream));
}

#ifdef VPR_ABLATION
// timing-only build: where the phase clocks of sinkhorn_aggregate_kernel go ([B][16] int64 on the device; null = off)
extern "C" int vpr_salad_sinkhorn_set_clocks(long long* clocks) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sinkhorn_clocks), &clocks, sizeof(clocks)) == hipSuccess ? VPR_OK : VPR_ERR_LAUNCH;
}
#endif

extern "C" int vpr_salad_stage_token(const uint16_t* cls_tokens, long long cls_stride, int B, int n, int C,
                                     const vpr_salad_weights* w, int m, int l, int t, int hidden,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  SaladArgs a;
  VPR_TRY_LAUNCH(salad_args(&a, B, n, C, m, l, t, hidden, w, workspace, workspace_bytes, true));
  return salad_stage_token(a, cls_tokens, cls_stride, static_cast<hipStream_t>(stream));
}

extern "C" int vpr_salad_stage_mlps(const uint16_t* patch_tokens, long long patch_img_stride, int B, int n, int C,
                                    const vpr_salad_weights* w, int m, int l, int t, int hidden,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  SaladArgs a;
  VPR_TRY_LAUNCH(salad_args(&a, B, n, C, m, l, t, hidden, w, workspace, workspace_bytes, true));
  return salad_stage_mlps(a, patch_tokens, patch_img_stride, static_cast<hipStream_t>(stream));
}

extern "C" int vpr_salad_stage_aggregate(int B, int n, int C, float dustbin, int m, int l, int t, int hidden,
                                         int sinkhorn_iters, float* out_f32, uint16_t* out_bf16,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  SaladArgs a;
  VPR_TRY_LAUNCH(salad_args(&a, B, n, C, m, l, t, hidden, nullptr, workspace, workspace_bytes, false));
  return salad_stage_aggregate(a, dustbin, sinkhorn_iters, out_f32, out_bf16, static_cast<hipStream_t>(stream));
}

// patch row r of image b at patch + b*patch_img_stride + r*C; cls token of image b at cls + b*cls_stride
static int salad_run(const uint16_t* patch, long long patch_img_stride, const uint16_t* cls, long long cls_stride,
                     int B, int n, int C, const vpr_salad_weights* w, float dustbin,
                     int m, int l, int t, int hidden, int sinkhorn_iters,
                     float* out_f32, uint16_t* out_bf16,
                     void* workspace, size_t workspace_bytes, void* stream_) {
  if (!patch || !cls || !out_f32) return VPR_ERR_INVALID_ARG;
  SaladArgs a;
  VPR_TRY_LAUNCH(salad_args(&a, B, n, C, m, l, t, hidden, w, workspace, workspace_bytes, true));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  VPR_TRY_LAUNCH(salad_stage_token(a, cls, cls_stride, stream));
  VPR_TRY_LAUNCH(salad_stage_mlps(a, patch, patch_img_stride, stream));
  return salad_stage_aggregate(a, dustbin, sinkhorn_iters, out_f32, out_bf16, stream);
}

extern "C" int vpr_salad_aggregate(const uint16_t* tokens, int B, int tokens_per_image, int C,
                                   const vpr_salad_weights* w, float dustbin,
                                   int m, int l, int t, int hidden, int sinkhorn_iters,
                                   float* out_f32, uint16_t* out_bf16,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!tokens || tokens_per_image < 2 || C <= 0) return VPR_ERR_INVALID_ARG;
  const long long img_stride = (long long)tokens_per_image * C;
  return salad_run(tokens + C, img_stride, tokens, img_stride, B, tokens_per_image - 1, C, w, dustbin, m, l, t, hidden,
                   sinkhorn_iters, out_f32, out_bf16, workspace, workspace_bytes, stream);
}

extern "C" int vpr_salad_aggregate_split(const uint16_t* patch_tokens, const uint16_t* cls_tokens, int B,
                                         int patches_per_image, int C,
                                         const vpr_salad_weights* w, float dustbin,
                                         int m, int l, int t, int hidden, int sinkhorn_iters,
                                         float* out_f32, uint16_t* out_bf16,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  if (patches_per_image < 1 || C <= 0) return VPR_ERR_INVALID_ARG;
  return salad_run(patch_tokens, (long long)patches_per_image * C, cls_tokens, C, B, patches_per_image, C, w, dustbin,
                   m, l, t, hidden, sinkhorn_iters, out_f32, out_bf16, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// f32-accurate SALAD (the reference runs the extractor and the aggregator in fp32: dinov2salad_validation.py:65-66,80-81,
// `.cuda()` with no cast).  f32 tokens and f32 weights; every linear layer is computed to f32 accuracy on the bf16 matrix
// pipe by splitting BOTH operands into three bf16 planes, x = h + m + l exactly (8 + 8 + 8 mantissa bits), and keeping the
// six products whose magnitude is above 2^-24 of the leading one:  (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)  — the dropped
// ones (m·l, l·m, l·l) are below 2^-26 relative.  bf16 x bf16 products are exact in f32 and the MFMA accumulates in f32,
// so one ordinary bf16 GEMM over a K axis of 6 x K, smallest terms first, IS the f32 GEMM (error = f32 accumulation
// order, as in any f32 GEMM).  Hidden activations stay f32 and are split again for the second layers: no bf16 rounding
// point anywhere between the tokens and the descriptor.  ~6x the MFMA work of the bf16 stage: a precision mode for
// reference-parity runs (evaluate.*(dtype=torch.float32)), not the benchmark path.
namespace vpr {

// role 0 = activation operand: planes (l, h, m, m, h, h); role 1 = weight operand: planes (h, l, m, h, m, h)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, long long rows, int K, long long ld_src,
                                                     int group_rows, long long group_stride,
                                                     uint16_t* __restrict__ dst, long long ld_dst, int role) {
  const int k4n = K >> 2;
  const long long total = rows * k4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / k4n;
    const int k = (int)(i - r * k4n) << 2;
    const float* sp = group_rows > 0 ? src + (r / group_rows) * group_stride + (r % group_rows) * ld_src + k
                                     : src + r * ld_src + k;
    const float4 v = *reinterpret_cast<const float4*>(sp);
    const float x[4] = {v.x, v.y, v.z, v.w};
    ushort4 pl[3];
    uint16_t* ph = reinterpret_cast<uint16_t*>(&pl[0]);
    uint16_t* pm = reinterpret_cast<uint16_t*>(&pl[1]);
    uint16_t* pq = reinterpret_cast<uint16_t*>(&pl[2]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint16_t h = f32_to_bf16_bits(x[e]);
      const float r1 = x[e] - bf16_bits_to_f32(h);          // exact
      const uint16_t m = f32_to_bf16_bits(r1);
      const float r2 = r1 - bf16_bits_to_f32(m);            // exact
      ph[e] = h; pm[e] = m; pq[e] = f32_to_bf16_bits(r2);
    }
    uint16_t* dp = dst + r * ld_dst + k;
    const int order_a[6] = {2, 0, 1, 1, 0, 0}, order_w[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p)
      *reinterpret_cast<ushort4*>(dp + (long long)p * K) = pl[role ? order_w[p] : order_a[p]];
  }
}

static int launch_split3(const float* src, long long rows, int K, long long ld_src, int group_rows, long long group_stride,
                         uint16_t* dst, long long ld_dst, int role, hipStream_t stream) {
  if (rows <= 0) return VPR_OK;
  long long blocks = (rows * (K >> 2) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  VPR_TRY_LAUNCH(launch_kernel(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, rows, K, ld_src, group_rows,
                               group_stride, dst, ld_dst, role));
  return VPR_OK;
}

struct SaladF32Plan {
  size_t off_A1, off_W1, off_H, off_H2, off_W2s, off_W2c, off_cls, off_W1t, off_Ht, off_Ht2, off_W2t, off_S, off_F, off_g, total;
};
static bool salad_f32_plan(int B, int n, int C, int m, int l, int t, int hidden, SaladF32Plan* p) {
  if (B <= 0 || n <= 0 || C <= 0 || m <= 0 || l <= 0 || t <= 0 || hidden <= 0) return false;
  const size_t rows = (size_t)B * n;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  p->off_A1 = take(rows * 6 * C * 2);
  p->off_W1 = take((size_t)2 * hidden * 6 * C * 2);
  p->off_H = take(rows * 2 * hidden * 4);
  p->off_H2 = take(rows * 12 * hidden * 2);
  p->off_W2s = take((size_t)m * 6 * hidden * 2);
  p->off_W2c = take((size_t)l * 6 * hidden * 2);
  p->off_cls = take((size_t)B * 6 * C * 2);
  p->off_W1t = take((size_t)hidden * 6 * C * 2);
  p->off_Ht = take((size_t)B * hidden * 4);
  p->off_Ht2 = take((size_t)B * 6 * hidden * 2);
  p->off_W2t = take((size_t)t * 6 * hidden * 2);
  p->off_S = take(rows * m * 4);
  p->off_F = take(rows * l * 4);
  p->off_g = take((size_t)B * t * 4);
  p->total = off;
  return true;
}

}  // namespace vpr

extern "C" size_t vpr_salad_f32_workspace_bytes(int B, int n, int C, int m, int l, int t, int hidden) {
  SaladF32Plan p;
  return salad_f32_plan(B, n, C, m, l, t, hidden, &p) ? p.total : 0;
}

extern "C" int vpr_salad_aggregate_f32(const float* patch, long long patch_img_stride, const float* cls, long long cls_stride,
                                       int B, int n, int C, const vpr_salad_weights_f32* w, float dustbin,
                                       int m, int l, int t, int hidden, int sinkhorn_iters,
                                       float* out_f32, uint16_t* out_bf16,
                                       void* workspace, size_t workspace_bytes, void* stream_) {
  if (!patch || !cls || !w || !out_f32 || !workspace || B <= 0 || n < 1 || C <= 0) return VPR_ERR_INVALID_ARG;
  if (!w->w1_sc || !w->b1_sc || !w->w2_s || !w->b2_s || !w->w2_c || !w->b2_c || !w->w1_t || !w->b1_t || !w->w2_t || !w->b2_t)
    return VPR_ERR_INVALID_ARG;
  if (n != SA_N || m != SA_M || l != SA_L || t != SA_T || (C % 64) || (hidden % 64)) return VPR_ERR_UNSUPPORTED;
  if ((patch_img_stride % 4) || (cls_stride % 4) || patch_img_stride < (long long)n * C || cls_stride < C) return VPR_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(patch) | reinterpret_cast<uintptr_t>(cls) | reinterpret_cast<uintptr_t>(w->w1_sc) |
       reinterpret_cast<uintptr_t>(w->w2_s) | reinterpret_cast<uintptr_t>(w->w2_c) | reinterpret_cast<uintptr_t>(w->w1_t) |
       reinterpret_cast<uintptr_t>(w->w2_t) | reinterpret_cast<uintptr_t>(workspace)) & 15)
    return VPR_ERR_UNSUPPORTED;
  SaladF32Plan p;
  if (!salad_f32_plan(B, n, C, m, l, t, hidden, &p)) return VPR_ERR_INVALID_ARG;
  if (workspace_bytes < p.total) return VPR_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  auto u16 = [&](size_t off) { return reinterpret_cast<uint16_t*>(ws + off); };
  auto f32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  uint16_t *A1 = u16(p.off_A1), *W1 = u16(p.off_W1), *H2 = u16(p.off_H2), *W2s = u16(p.off_W2s), *W2c = u16(p.off_W2c);
  uint16_t *cls3 = u16(p.off_cls), *W1t = u16(p.off_W1t), *Ht2 = u16(p.off_Ht2), *W2t = u16(p.off_W2t);
  float *H = f32(p.off_H), *Ht = f32(p.off_Ht), *S = f32(p.off_S), *F = f32(p.off_F), *g = f32(p.off_g);
  const long long rows = (long long)B * n;
  const int K1 = 6 * C, K2 = 6 * hidden;
  // operand planes: tokens (row-group addressing resolved here, A1 is dense), cls rows, all five weight matrices
  VPR_TRY_LAUNCH(launch_split3(patch, rows, C, C, n, patch_img_stride, A1, K1, 0, stream));
  VPR_TRY_LAUNCH(launch_split3(cls, B, C, cls_stride, 0, 0, cls3, K1, 0, stream));
  VPR_TRY_LAUNCH(launch_split3(w->w1_sc, 2 * hidden, C, C, 0, 0, W1, K1, 1, stream));
  VPR_TRY_LAUNCH(launch_split3(w->w1_t, hidden, C, C, 0, 0, W1t, K1, 1, stream));
  VPR_TRY_LAUNCH(launch_split3(w->w2_s, m, hidden, hidden, 0, 0, W2s, K2, 1, stream));
  VPR_TRY_LAUNCH(launch_split3(w->w2_c, l, hidden, hidden, 0, 0, W2c, K2, 1, stream));
  VPR_TRY_LAUNCH(launch_split3(w->w2_t, t, hidden, hidden, 0, 0, W2t, K2, 1, stream));
  // layer 1 (f32 out, bias + ReLU in the epilogue)
  const GemmProblem l1_sc{A1, K1, 0, 0, W1, K1, w->b1_sc, 1, H, 2 * hidden, 0, (int)rows, 2 * hidden, K1, 0, 0};
  int st = launch_gemm256(l1_sc, stream);
  if (st == VPR_ERR_UNSUPPORTED) st = launch_gemm_nt_group(&l1_sc, 1, stream);
  if (st != VPR_OK) return st;
  const GemmProblem l1_t{cls3, K1, 0, 0, W1t, K1, w->b1_t, 1, Ht, hidden, 0, B, hidden, K1, 0, 0};
  VPR_TRY_LAUNCH(launch_gemm_nt_group(&l1_t, 1, stream));
  // hidden activations: f32 -> planes ([score 6h | cluster 6h] per row)
  VPR_TRY_LAUNCH(launch_split3(H, rows, hidden, 2 * hidden, 0, 0, H2, 2 * K2, 0, stream));
  VPR_TRY_LAUNCH(launch_split3(H + hidden, rows, hidden, 2 * hidden, 0, 0, H2 + K2, 2 * K2, 0, stream));
  VPR_TRY_LAUNCH(launch_split3(Ht, B, hidden, hidden, 0, 0, Ht2, K2, 0, stream));
  const GemmProblem l2[3] = {
      {H2, 2 * K2, 0, 0, W2s, K2, w->b2_s, 0, S, m, 0, (int)rows, m, K2, 0, 0},
      {H2 + K2, 2 * K2, 0, 0, W2c, K2, w->b2_c, 0, F, l, 0, (int)rows, l, K2, 0, 0},
      {Ht2, K2, 0, 0, W2t, K2, w->b2_t, 0, g, t, 0, B, t, K2, 0, 0}};
  VPR_TRY_LAUNCH(launch_gemm_nt_group(l2, 3, stream));
  return launch_sinkhorn_aggregate(S, F, g, B, n, m, l, t, dustbin, sinkhorn_iters, out_f32, out_bf16, stream);
}
