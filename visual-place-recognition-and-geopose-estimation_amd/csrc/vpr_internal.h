// vpr_internal.h — launch functions shared between the translation units of libvpr_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vpr_amd.h"

namespace vpr {

int launch_gemm_nt(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                   const uint16_t* W, int ldw, const float* bias, int relu, void* C, int ldc,
                   int out_is_bf16, int M, int N, int K, hipStream_t stream);

int launch_sinkhorn_aggregate(const float* scores, const float* feats, const float* tokfeat,
                              int B, int n, int m, int l, int t, float dustbin, int iters,
                              float* out_f32, uint16_t* out_bf16, hipStream_t stream);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace vpr
