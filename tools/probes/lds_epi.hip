// Conflict counters of the LDS epilogue's two access shapes for candidate row strides RB:
//   kind 0: parking, ds_write_b64, lane (g, c): c * RB + g * 8          (16x16x32 layout)
//   kind 1: parking, ds_write_b64, lane (h, l): l * RB + h * 8          (32x32x16 layout)
//   kind 2: write-back, ds_read_b128, 8 lanes per pixel row: (lane / 8) * RB + (lane % 8) * 16
// one launch per (kind, RB); read back per dispatch (lds_epi.sh).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) epi_probe(int kind, int rb, int iters, uint32_t* out) {
  __shared__ __attribute__((aligned(256))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 16384 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x;
  int off;
  if (kind == 0) off = (lane & 15) * rb + (lane >> 4) * 8;
  else if (kind == 1) off = (lane & 31) * rb + (lane >> 5) * 8;
  else off = (lane >> 3) * rb + (lane & 7) * 16;
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(lds + off);
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (kind < 2) {
      asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(make_uint2(acc, it)) : "memory");
    } else {
      uint4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      acc ^= v.x ^ v.w;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x * 64 + lane] = acc;
}
int main() {
  uint32_t* out;
  (void)hipMalloc(&out, 256 * 64 * 4);
  const int rbs[8] = {128, 144, 160, 176, 192, 208, 224, 272};
  for (int kind = 0; kind < 3; ++kind)
    for (int i = 0; i < 8; ++i)
      hipLaunchKernelGGL(epi_probe, dim3(256), dim3(64), 0, 0, kind, rbs[i], 1024, out);
  (void)hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
