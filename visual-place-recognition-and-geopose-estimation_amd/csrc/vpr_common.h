// vpr_common.h — shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere; LDS tiles are [rows][64 bf16] (128-B rows) filled by
// LDS-DMA (global_load_lds_dwordx4) and read back as MFMA fragments with ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vpr {

typedef __attribute__((ext_vector_type(8))) short s16x8;     // 8 bf16 = one 16-B chunk
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // MFMA A/B fragment type
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int WAVE = 64;

// 100 MHz constant clock, read where the statement stands (volatile asm with a memory clobber: the compiler may not move
// it across the surrounding code) — phase clocks of the timing-only build.
__device__ __forceinline__ long long vpr_clock_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return (long long)t;
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// Round-to-nearest-even f32 -> bf16 bits.  The compiler emits v_cvt_pk_bf16_f32 for the cast
// (NaN stays NaN; see MI355X_MICROARCH "Correctness boundaries").
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() compiles to s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier,
// i.e. it also drains every global load the wave has in flight; a kernel that wants its global loads to keep flying
// across barriers (the compiler still waits for each loaded register before its first use) uses this one.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// All-reduce inside each 16-lane row with DPP only (no LDS traffic): xor 1, xor 2 as quad
// permutes, then row_half_mirror and row_mirror (after the first two steps a quad holds one
// value, so mirroring pairs the remaining groups).  Every lane ends with its row's result.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f32<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f32<0x141>(v));   // row_half_mirror
  v = fmaxf(v, dpp_f32<0x140>(v));   // row_mirror
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  return v;
}

// ---- LDS tile geometry -------------------------------------------------------------------
// A tile row is 64 bf16 = 128 B = 8 chunks of 16 B.  Chunk c of row r is stored at physical
// chunk c ^ ((r >> 1) & 7): with this XOR every ds_read_b128 lane group of both MFMA operand
// maps (16x16x32: row = lane&15, chunk = lane>>4 (+4); 32x32x16: row = lane&31,
// chunk = lane>>5 (+2s)) touches 16 distinct 16-B slots of the 256-B bank row (checked
// exhaustively on the host, tests/test_library_cpu.py::test_lds_swizzle_is_conflict_free) — conflict-free.
constexpr int TILE_ROW_BYTES = 128;
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return row * TILE_ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B land at lds_wave_base + lane*16 (the LDS side
// is linear; the swizzle is applied to the per-lane SOURCE address, cdna guide rule 21).
// AUX = cache-policy bits of the instruction: 0 default, 2 = nt (non-temporal: for bytes read once, e.g. the gallery
// stream of the kNN score kernel; MI355X_MICROARCH "nt-weights").
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// Fill 8 tile rows [row0, row0+8) of an LDS tile from a row-major bf16 matrix: lane i writes
// physical slot (row0 + i/8, i%8) and therefore fetches logical chunk (i%8) ^ swz(row).
// `src_row_ptr` is this lane's source row start (already offset to the k-step), i.e. the
// caller resolves row -> pointer (clamping / grouping) for row0 + (lane>>3).
__device__ __forceinline__ void stage8(const uint16_t* src_row_ptr, char* tile, int row0, int lane) {
  const int r = row0 + (lane >> 3);
  const int c = (lane & 7) ^ ((r >> 1) & 7);
  glds16(src_row_ptr + c * 8, tile + row0 * TILE_ROW_BYTES);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + tile_off(row, chunk));
}

}  // namespace vpr
