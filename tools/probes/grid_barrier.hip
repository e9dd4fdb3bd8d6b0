// What does ONE grid-wide barrier inside a persistent kernel cost on this 8-XCD part, and which memory
// operations make phase-1 data visible to phase 2 of the SAME launch?  (The sorted splat at 512 x 1024 is
// two launches of ~10 us of work each: launch / prologue / tail are half its time.  Rounds 3-4 measured
// `__threadfence()` behind a ticket at 51 / 613 us when thousands of workgroups each release; here one
// workgroup per CU releases once.)
//   hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier grid_barrier.hip ; ./build/grid_barrier
// Variants (phase 1: every workgroup writes `bytes` of its own slab; barrier; phase 2: it reads the slab
// of workgroup (b + shift) % G -- another XCD for shift = 1 -- and checks every word):
//   0  plain stores, __threadfence() before the arrive, __threadfence() after the wait, plain loads
//   1  agent-scope relaxed atomic stores (global_store ... sc1), no fence, agent-scope atomic loads (sc1)
//   2  sc1 stores, no fence, PLAIN loads (expected to read stale L2 lines on a second run: the control)
//   3  plain stores + release fence only; sc1 loads
// plus: the two phases as two launches (plain everything), an empty kernel, cooperative launch of the same.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void st_sc1(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_sc1(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void grid_barrier(uint32_t* ctr, uint32_t target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (bounded: a workgroup that cannot be resident must not hang the box)
    for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int VARIANT>
__global__ void __launch_bounds__(512) two_phase(uint64_t* slab, int words, uint32_t* ctr, uint32_t target,
                                                 uint64_t tag, int shift, uint32_t* bad) {
  const int G = gridDim.x, b = blockIdx.x;
  uint64_t* mine = slab + (size_t)b * words;
  for (int i = threadIdx.x; i < words; i += 512) {
    const uint64_t v = tag ^ ((uint64_t)b << 32) ^ (uint64_t)i;
    if (VARIANT == 1 || VARIANT == 2) st_sc1(mine + i, v); else mine[i] = v;
  }
  if (VARIANT == 0 || VARIANT == 3) __threadfence();
  grid_barrier(ctr, target);
  if (VARIANT == 0) __threadfence();
  const int o = (b + shift) % G;
  const uint64_t* other = slab + (size_t)o * words;
  uint32_t nbad = 0;
  for (int i = threadIdx.x; i < words; i += 512) {
    const uint64_t v = (VARIANT == 1 || VARIANT == 3) ? ld_sc1(other + i) : other[i];
    nbad += v != (tag ^ ((uint64_t)o << 32) ^ (uint64_t)i);
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ void __launch_bounds__(512) phase1(uint64_t* slab, int words, uint64_t tag) {
  uint64_t* mine = slab + (size_t)blockIdx.x * words;
  for (int i = threadIdx.x; i < words; i += 512) mine[i] = tag ^ ((uint64_t)blockIdx.x << 32) ^ (uint64_t)i;
}
__global__ void __launch_bounds__(512) phase2(const uint64_t* slab, int words, uint64_t tag, int shift, uint32_t* bad) {
  const int o = (blockIdx.x + shift) % gridDim.x;
  const uint64_t* other = slab + (size_t)o * words;
  uint32_t nbad = 0;
  for (int i = threadIdx.x; i < words; i += 512) nbad += other[i] != (tag ^ ((uint64_t)o << 32) ^ (uint64_t)i);
  if (nbad) atomicAdd(bad, nbad);
}
__global__ void empty_kernel(uint32_t* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  const int reps = 200;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  uint32_t *ctr, *bad;
  CK(hipMalloc(&ctr, 4 * (reps + 8) * 16));
  CK(hipMalloc(&bad, 4));
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    const int G = cus * wg_per_cu;
    for (int kb = 4; kb <= 256; kb *= 8) {   // 4 KB, 32 KB, 256 KB per workgroup
      const int words = kb * 1024 / 8;
      uint64_t* slab;
      CK(hipMalloc(&slab, (size_t)G * words * 8));
      for (int variant = 0; variant < 4; ++variant) {
        for (int shift = 1; shift <= 9; shift += 8) {
          CK(hipMemsetAsync(ctr, 0, 4 * (reps + 8) * 16, st));
          CK(hipMemsetAsync(bad, 0, 4, st));
          float best = 0;
          for (int pass = 0; pass < 2; ++pass) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) {
              // a fresh counter word (own 64-byte line) and a fresh tag per launch: stale data of the
              // previous launch is detectable
              uint32_t* c = ctr + 16 * r;
              const uint64_t tag = 0x9e3779b97f4a7c15ull * (uint64_t)(r + 1 + 1000 * pass + 77 * variant);
              switch (variant) {
                case 0: hipLaunchKernelGGL(two_phase<0>, dim3(G), dim3(512), 0, st, slab, words, c, (uint32_t)G, tag, shift, bad); break;
                case 1: hipLaunchKernelGGL(two_phase<1>, dim3(G), dim3(512), 0, st, slab, words, c, (uint32_t)G, tag, shift, bad); break;
                case 2: hipLaunchKernelGGL(two_phase<2>, dim3(G), dim3(512), 0, st, slab, words, c, (uint32_t)G, tag, shift, bad); break;
                default: hipLaunchKernelGGL(two_phase<3>, dim3(G), dim3(512), 0, st, slab, words, c, (uint32_t)G, tag, shift, bad); break;
              }
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&best, e0, e1));
            if (pass == 0) CK(hipMemsetAsync(ctr, 0, 4 * (reps + 8) * 16, st));
          }
          uint32_t hb = 0;
          CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
          printf("G=%d (%d/CU) %3d KB/wg variant %d shift %d: %.2f us/launch, bad words %u\n", G,
                 wg_per_cu, kb, variant, shift, 1e3 * best / reps, hb);
        }
      }
      // the same as two launches
      CK(hipMemsetAsync(bad, 0, 4, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) {
        const uint64_t tag = 0x9e3779b97f4a7c15ull * (uint64_t)(r + 5);
        hipLaunchKernelGGL(phase1, dim3(G), dim3(512), 0, st, slab, words, tag);
        hipLaunchKernelGGL(phase2, dim3(G), dim3(512), 0, st, slab, words, tag, 1, bad);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      uint32_t hb = 0;
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("G=%d %3d KB/wg as TWO launches: %.2f us per pair, bad %u\n", G, kb, 1e3 * ms / reps, hb);
      CK(hipFree(slab));
    }
  }
  // launch overheads: empty kernel, back to back; with a 4-byte memset in between; cooperative
  for (int G : {1, cus, 8 * cus}) {
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(512), 0, st, (uint32_t*)nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel grid %d: %.2f us per launch\n", G, 1e3 * ms / reps);
  }
  {
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(ctr, 0, 4, st));
      hipLaunchKernelGGL(empty_kernel, dim3(cus), dim3(512), 0, st, (uint32_t*)nullptr);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("4-byte memset + empty kernel grid %d: %.2f us per pair\n", cus, 1e3 * ms / reps);
  }
  {
    uint32_t* np = nullptr;
    void* args[] = {&np};
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
      CK(hipLaunchCooperativeKernel((const void*)empty_kernel, dim3(cus), dim3(512), args, 0, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("cooperative empty kernel grid %d: %.2f us per launch\n", cus, 1e3 * ms / reps);
  }
  return 0;
}
