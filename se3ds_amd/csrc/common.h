// Shared helpers for the libse3ds_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/se3ds_hip.h"

namespace se3ds {

// Records the failing HIP status for se3ds_last_error(); defined in api.hip.
void set_last_error(hipError_t e, const char* where);

inline int check_launch(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(e, where);
    return SE3DS_E_LAUNCH;
  }
  return SE3DS_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8.
inline int grid_for(int64_t work_items, int block) {
  int64_t g = ceil_div(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

// bf16 <-> fp32 (round to nearest even), raw-bit helpers
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// 64-lane wave reductions (DPP/shuffle based)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

}  // namespace se3ds
