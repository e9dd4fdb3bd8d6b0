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

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

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
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round to nearest even, NaN -> quiet NaN); the
// integer formulation it replaces cost ~9 VALU instructions per element, which made the conv
// epilogues VALU-bound (3x3 128 -> 128 @512x1024: epilogue 27 % of the kernel, its global stores
// 2 % -- measured with the stores / the epilogue / the MFMAs compiled out).
typedef float se3ds_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 se3ds_bf16x2 __attribute__((ext_vector_type(2)));
// (a, b) -> a in the low half, b in the high half
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
  const se3ds_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, se3ds_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  return (uint16_t)(pack2_bf16(f, f) & 0xffffu);
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

// ---- typed 16-byte vector access (fp32 x4 / bf16 x8) and activation helpers
template <typename T> struct VT;
template <> struct VT<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct VT<uint16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const uint16_t* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = __uint_as_float(q[e] << 16);
      o[2 * e + 1] = __uint_as_float(q[e] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(uint16_t* p, const float (&o)[8]) {
    uint32_t q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      q[e] = pack2_bf16(o[2 * e], o[2 * e + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(q[0], q[1], q[2], q[3]);
  }
  static __device__ __forceinline__ float ld1(const uint16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st1(uint16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v > 0.f ? v : v * alpha;
  return v;
}
// derivative of the activation expressed through its OUTPUT y (alpha > 0 keeps the sign)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float alpha) {
  if (act == 1) return y > 0.f ? 1.f : 0.f;
  if (act == 2) return y > 0.f ? 1.f : alpha;
  return 1.f;
}
// the same from a stored "output > 0" bit
__device__ __forceinline__ float act_grad_from_bit(unsigned bit, int act, float alpha) {
  if (act == 1) return bit ? 1.f : 0.f;
  if (act == 2) return bit ? 1.f : alpha;
  return 1.f;
}

}  // namespace se3ds
