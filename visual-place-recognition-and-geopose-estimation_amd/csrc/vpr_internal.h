// vpr_internal.h — launch functions shared between the translation units of libvpr_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/vpr_amd.h"

namespace vpr {

int launch_gemm_nt(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                   const uint16_t* W, int ldw, const float* bias, int relu, void* C, int ldc,
                   int out_is_bf16, int M, int N, int K, hipStream_t stream);

// One member of a (grouped) GEMM launch: C = act(A W^T + bias); see gemm_nt.hip.  tiles_* are filled in
// by the launcher.
struct GemmProblem {
  const uint16_t* A; int lda; int a_group_rows; long long a_group_stride;
  const uint16_t* W; int ldw; const float* bias; int relu;
  void* C; int ldc; int out_is_bf16; int M, N, K; int tiles_m, tiles_n;
  const float* a_scale; const float* w_scale;       // gemm256 fp8 form only (per-row scales of A and W); null otherwise
  int ksplit; long long slab_stride;                // gemm256 split-K: K slices (0 / 1 = none); slice z writes C + z * slab_stride elements
};
constexpr int GEMM_MAX_GROUP = 3;
int launch_gemm_nt_group(const GemmProblem* probs, int count, hipStream_t stream);
// gemm_nt.hip: fp8 e4m3 operands with per-row scales, f32 out (kNN score tile for > 64 queries); K % 128 == 0
int launch_gemm_nt_fp8(const uint8_t* A, int lda, const float* a_scale, const uint8_t* W, int ldw, const float* w_scale,
                       float* C, int ldc, int M, int N, int K, hipStream_t stream);
int launch_gemm256(const GemmProblem& problem, hipStream_t stream);   // gemm256.hip: 256x256 tiles, K >= 128
// gemm256.hip, fp8 form: e4m3 operands with per-row scales, f32 out (kNN score tile of a >= 384-query gathered batch)
int launch_gemm256_fp8(const uint8_t* A, int lda, const float* a_scale, const uint8_t* W, int ldw, const float* w_scale,
                       float* C, int ldc, int M, int N, int K, hipStream_t stream, int ksplit = 1, long long slab_stride = 0);
// skinny.hip: a few rows x [N, K]^T; mode 0 bias, 1 bias+tanh-GELU, 2 accumulate into out, 3 bias+ReLU, 4 bias+erf-GELU
int launch_skinny_linear(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias, int bias_is_bf16,
                         int mode, uint16_t* out, int ldo, int M, int N, int K, const float* stats_bias,
                         float* row_stats, void* stream);

int launch_sinkhorn_aggregate(const float* scores, const float* feats, const float* tokfeat,
                              int B, int n, int m, int l, int t, float dustbin, int iters,
                              float* out_f32, uint16_t* out_bf16, hipStream_t stream);

// Launch through hipLaunchKernel(), whose return value is THIS launch's status.  (The
// hipGetLastError() idiom reads a per-thread sticky value that other libraries in the process —
// e.g. hipBLASLt's kernel lookups inside PyTorch — leave set, so it cannot be trusted here.)
template <typename... KArgs>
inline int launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds_bytes,
                         hipStream_t stream, typename std::common_type<KArgs>::type... args) {
  void* ptrs[sizeof...(KArgs)] = {const_cast<void*>(static_cast<const void*>(&args))...};
  const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds_bytes, stream);
  if (e != hipSuccess) {
    fprintf(stderr, "libvpr_amd: kernel launch failed: %s (grid %u,%u,%u block %u lds %zu)\n",
            hipGetErrorString(e), grid.x, grid.y, grid.z, block.x, lds_bytes);
    return VPR_ERR_LAUNCH;
  }
  return VPR_OK;
}

#define VPR_TRY_LAUNCH(expr) do { const int vpr_st_ = (expr); if (vpr_st_ != VPR_OK) return vpr_st_; } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace vpr
