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
  int kstagger;                                     // gemm256: order in which a tile walks its K-tiles (0 = 0, 1, 2, ...; see g2_mainloop)
};
constexpr int GEMM_MAX_GROUP = 3;
int launch_gemm_nt_group(const GemmProblem* probs, int count, hipStream_t stream);
// gemm_nt.hip: fp8 e4m3 operands with per-row scales, f32 out (kNN score tile for > 64 queries); K % 128 == 0
int launch_gemm_nt_fp8(const uint8_t* A, int lda, const float* a_scale, const uint8_t* W, int ldw, const float* w_scale,
                       float* C, int ldc, int M, int N, int K, hipStream_t stream);
int launch_gemm256(const GemmProblem& problem, hipStream_t stream);   // gemm256.hip: 256x256 tiles, K >= 128
// gemm256.hip: SALAD score + cluster MLPs with the second layers fused into the layer-1 tile epilogue; S / F receive
// hidden / 256 partial-sum slabs of [M][m] / [M][l] f32 (slab 0 carries the bias), to be added in slab order
int launch_salad_mlps_fused(const uint16_t* X, int ldx, int group_rows, long long group_stride, const uint16_t* W1, const float* b1,
                            const uint16_t* W2s, const float* b2s, const uint16_t* W2c, const float* b2c,
                            float* S, float* F, int M, int C, int hidden, int m, int l, hipStream_t stream,
                            const uint16_t* W2s_frag = nullptr, const uint16_t* W2c_frag = nullptr);
// gemm256.hip, fp8 form: e4m3 operands with per-row scales, f32 out (kNN score tile of a >= 384-query gathered batch)
int launch_gemm256_fp8(const uint8_t* A, int lda, const float* a_scale, const uint8_t* W, int ldw, const float* w_scale,
                       float* C, int ldc, int M, int N, int K, hipStream_t stream, int ksplit = 1, long long slab_stride = 0);
// skinny.hip: a few rows x [N, K]^T; mode 0 bias, 1 bias+tanh-GELU, 2 accumulate into out, 3 bias+ReLU, 4 bias+erf-GELU
int launch_skinny_linear(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias, int bias_is_bf16,
                         int mode, uint16_t* out, int ldo, int M, int N, int K, const float* stats_bias,
                         float* row_stats, void* stream);

int launch_sinkhorn_aggregate(const float* scores, const float* feats, const float* tokfeat,
                              int B, int n, int m, int l, int t, float dustbin, int iters,
                              float* out_f32, uint16_t* out_bf16, hipStream_t stream, int nslab = 1, long long slab_rows = 0,
                              int* counters = nullptr);

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

// ---- process-wide tuning switches (A/B experiments) --------------------------------------------------------------
// Read from the environment ONCE, when the library is loaded (capi.hip); a later setenv() has no effect on the
// library.  scripts/ and tests flip them through vpr_tuning_set().  TUNE_UNSET = the variable was absent.
enum TuneOpt {
  TUNE_KNN_VARIANT = 0, TUNE_KNN_GEMM_MIN_B, TUNE_KNN_GEMM_KSPLIT, TUNE_KNN_FP8_GEMM256, TUNE_GEMM_NT_STAGES,
  TUNE_GEMM_GROUP_VARIANT, TUNE_ATTN_VARIANT, TUNE_LN_ROWS, TUNE_POSE_KS, TUNE_SKINNY_NW, TUNE_SKINNY_MBW,
  TUNE_SALAD_VARIANT, TUNE_POSE_VARIANT, TUNE_LNHEAD_VARIANT, TUNE_GEMM256_DEPTH, TUNE_HEAD_TRAIN_VARIANT, TUNE_GEMM256_STAGGER, TUNE_COUNT
};
constexpr int TUNE_UNSET = -2147483647 - 1;
int tune(TuneOpt o);
inline int tune_or(TuneOpt o, int dflt) { const int v = tune(o); return v == TUNE_UNSET ? dflt : v; }

// ---- per-device launch state ---------------------------------------------------------------------------------------
// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) and the CU count belong to a DEVICE, not to the process: one
// flag / one cached value per device ordinal, looked up through hipGetDevice() at every launch (~50 ns).  A process
// that drives several GPUs (or switches device between calls) gets the opt-in on each of them.
constexpr int VPR_MAX_DEVICES = 64;
struct PerDeviceFlag { unsigned char set[VPR_MAX_DEVICES]; };
inline int optin_dynamic_lds(const void* kernel, size_t bytes, PerDeviceFlag& flag) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VPR_MAX_DEVICES) return VPR_ERR_LAUNCH;
  if (!flag.set[dev]) {      // two threads racing here both set the attribute: harmless
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
      return VPR_ERR_LAUNCH;
    flag.set[dev] = 1;
  }
  return VPR_OK;
}
int device_cu_count();      // capi.hip: multiProcessorCount of the current device, cached per device (256 on MI355X)

#define VPR_TRY_LAUNCH(expr) do { const int vpr_st_ = (expr); if (vpr_st_ != VPR_OK) return vpr_st_; } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace vpr
