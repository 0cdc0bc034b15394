// capi.hip — C-ABI odds and ends of libvpr_amd.so (see include/vpr_amd.h for the contract).
#include <stdlib.h>
#include <string.h>
#include "vpr_common.h"
#include "vpr_internal.h"

namespace vpr {

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src,
                                                          uint16_t* __restrict__ dst, long long count) {
  // 4 elements per thread per trip (16-B loads, 8-B stores), grid-stride
  const long long nvec = count >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    ushort4 o;
    o.x = f32_to_bf16_bits(v.x); o.y = f32_to_bf16_bits(v.y);
    o.z = f32_to_bf16_bits(v.z); o.w = f32_to_bf16_bits(v.w);
    reinterpret_cast<ushort4*>(dst)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
    const long long i = (nvec << 2) + threadIdx.x;
    dst[i] = f32_to_bf16_bits(src[i]);
  }
}

// ---- tuning switches: environment read once at load --------------------------------------------------------------
static const char* const kTuneNames[TUNE_COUNT] = {
    "VPR_KNN_VARIANT", "VPR_KNN_GEMM_MIN_B", "VPR_KNN_GEMM_KSPLIT", "VPR_KNN_FP8_GEMM256", "VPR_GEMM_NT_STAGES",
    "VPR_GEMM_GROUP_VARIANT", "VPR_ATTN_VARIANT", "VPR_LN_ROWS", "VPR_POSE_KS", "VPR_SKINNY_NW", "VPR_SKINNY_MBW",
    "VPR_SALAD_VARIANT", "VPR_POSE_VARIANT", "VPR_LNHEAD_VARIANT", "VPR_GEMM256_DEPTH", "VPR_HEAD_TRAIN_VARIANT", "VPR_GEMM256_STAGGER"};
struct TuneTable {
  int v[TUNE_COUNT];
  TuneTable() {
    for (int i = 0; i < TUNE_COUNT; ++i) {
      const char* e = getenv(kTuneNames[i]);
      v[i] = (e && *e) ? atoi(e) : TUNE_UNSET;
    }
  }
};
static TuneTable g_tune;           // constructed by the dynamic loader, before any entry point can run
int tune(TuneOpt o) { return g_tune.v[o]; }

// ---- per-device CU count --------------------------------------------------------------------------------------------
int device_cu_count() {
  static int cached[VPR_MAX_DEVICES] = {};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VPR_MAX_DEVICES) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

}  // namespace vpr

using namespace vpr;

extern "C" int vpr_tuning_set(const char* name, int value, int unset) {
  if (!name) return VPR_ERR_INVALID_ARG;
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (strcmp(name, kTuneNames[i]) == 0) {
      g_tune.v[i] = unset ? TUNE_UNSET : value;
      return VPR_OK;
    }
  return VPR_ERR_INVALID_ARG;
}

extern "C" int vpr_tuning_get(const char* name, int* value) {
  if (!name || !value) return VPR_ERR_INVALID_ARG;
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (strcmp(name, kTuneNames[i]) == 0) {
      if (g_tune.v[i] == TUNE_UNSET) return 1;      // 1 = known switch, not set
      *value = g_tune.v[i];
      return VPR_OK;
    }
  return VPR_ERR_INVALID_ARG;
}

extern "C" const char* vpr_status_string(int status) {
  switch (status) {
    case VPR_OK: return "ok";
    case VPR_ERR_INVALID_ARG: return "invalid argument (null pointer or non-positive dimension)";
    case VPR_ERR_UNSUPPORTED: return "shape or alignment not supported by the gfx950 kernels";
    case VPR_ERR_WORKSPACE: return "workspace too small";
    case VPR_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown status";
  }
}

extern "C" int vpr_abi_version(void) { return VPR_AMD_ABI_VERSION; }

extern "C" int vpr_gemm_nt_bf16(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                                const uint16_t* W, int ldw, const float* bias, int relu, void* C, int ldc,
                                int out_is_bf16, int M, int N, int K, void* stream) {
  return launch_gemm_nt(A, lda, a_group_rows, a_group_stride, W, ldw, bias, relu, C, ldc, out_is_bf16, M, N, K,
                        static_cast<hipStream_t>(stream));
}

extern "C" int vpr_f32_to_bf16(const float* src, uint16_t* dst, long long count, void* stream) {
  if (!src || !dst || count < 0) return VPR_ERR_INVALID_ARG;
  if (count == 0) return VPR_OK;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7)) return VPR_ERR_UNSUPPORTED;
  long long blocks = ((count >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  VPR_TRY_LAUNCH(launch_kernel(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, dst, count));
  return VPR_OK;
}
