// Conflict counters of the 16-row x 4-chunk fragment read (v_mfma_f32_16x16x32 operands, 128-byte
// LDS rows) for candidate XOR swizzles f(row) and every start row 0..15: one launch per (f, r0),
// read back per dispatch (see lds_swz.sh).  Lane (g, i): row r0 + i, 16-byte chunk (g ^ f(row)) & 7.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ int fsw(int cand, int row) {
  switch (cand) {
    case 0: return (row >> 1) & 7;
    case 1: return row & 7;
    case 2: return ((row >> 1) & 3) | ((row & 1) << 2);
    case 3: return (row ^ (row >> 3)) & 7;
    case 4: return ((row >> 1) & 3) ^ ((row & 1) << 2) ^ (((row >> 3) & 1) << 2);
    case 5: return (row >> 2) & 7;
    case 6: return ((row >> 2) & 3) | ((row & 1) << 2);
    case 7: return (row & 3) | (((row >> 3) & 1) << 2);
    default: return 0;
  }
}
__global__ void __launch_bounds__(64) swz_probe(int cand, int r0, int kstep, int iters, uint32_t* out) {
  __shared__ __attribute__((aligned(256))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 16384 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i16 = lane & 15;
  const int row = r0 + i16;
  const int off = row * 128 + ((((kstep << 2) | g) ^ fsw(cand, row)) << 4);
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(lds + off);
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    acc ^= v.x ^ v.w;
  }
  if (acc == 0x12345678u) out[blockIdx.x * 64 + lane] = acc;
}
int main() {
  uint32_t* out;
  (void)hipMalloc(&out, 256 * 64 * 4);
  for (int cand = 0; cand < 8; ++cand)
    for (int r0 = 0; r0 < 16; ++r0)
      hipLaunchKernelGGL(swz_probe, dim3(256), dim3(64), 0, 0, cand, r0, r0 & 1, 1024, out);
  (void)hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
