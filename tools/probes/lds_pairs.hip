// Which lanes of a ds_read_b128 are checked against each other for bank conflicts?  Every lane reads
// its own 16 bytes (lane * 16: conflict-free) except lane b, which reads lane 0's banks at another
// address (+1024).  One launch per b = 1..63, same kernel: per-dispatch SQ_LDS_BANK_CONFLICT (in
// dispatch order) is non-zero exactly for the lanes that share a cycle with lane 0.  Second series:
// lane 16's banks (b' vs lane 16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) pair_probe(int a, int b, int iters, uint32_t* out) {
  __shared__ __attribute__((aligned(256))) unsigned char lds[8192];
  for (int i = threadIdx.x; i < 8192 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x;
  const int off = lane == b ? a * 16 + 1024 : lane * 16;
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
  for (int a = 0; a <= 16; a += 16)
    for (int b = 0; b < 64; ++b)
      hipLaunchKernelGGL(pair_probe, dim3(256), dim3(64), 0, 0, a, b == a ? 64 : b, 1024, out);
  (void)hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
