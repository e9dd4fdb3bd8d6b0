#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __attribute__((aligned(128))) unsigned char zero_page[256];
__global__ void k(const uint4* in, uint4* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* src = (lane & 1) ? in + threadIdx.x : reinterpret_cast<const uint4*>(zero_page);
  unsigned char* dst = lds + wave * 1024;   // wave-uniform base
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<uint4*>(lds + threadIdx.x * 16);
}
int main(){ uint4 *a,*b; hipMalloc(&a,256*16); hipMalloc(&b,256*16);
 uint32_t h[1024]; for(int i=0;i<1024;++i)h[i]=i; hipMemcpy(a,h,4096,hipMemcpyHostToDevice);
 k<<<1,256>>>(a,b); uint32_t r[1024]; hipMemcpy(r,b,4096,hipMemcpyDeviceToHost);
 for(int t=0;t<8;++t) printf("t%d: %u %u %u %u\n", t, r[t*4],r[t*4+1],r[t*4+2],r[t*4+3]);
 printf("t65: %u  t130: %u t255: %u\n", r[65*4], r[130*4], r[255*4]); return 0; }
