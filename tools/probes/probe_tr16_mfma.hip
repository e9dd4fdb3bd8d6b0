#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(const uint16_t* in, uint16_t* out, float* o2) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64*64];
  for (int i = threadIdx.x; i < 64*64; i += 64) lds[i] = in[i];
  __syncthreads();
  int l = threadIdx.x;
  // linear: lane l points at elements 4l..4l+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4*l));
  for (int j = 0; j < 4; ++j) out[l*4+j] = (uint16_t)v[j];
  bf16x8 a, b; 
  for (int j=0;j<8;++j){ a[j]=(__bf16)1.0f; b[j]=(__bf16)2.0f; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  o2[l] = c[0];
  float fa = 1.0f, fb = 3.0f;
  f32x16 c2 = {0};
  c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
  o2[64+l] = c2[5];
}
int main(){
  uint16_t h[64*64]; for(int i=0;i<64*64;++i) h[i]=i;
  uint16_t *d,*o; float* o2; hipMalloc(&d,sizeof(h)); hipMalloc(&o,64*4*2); hipMalloc(&o2,128*4);
  hipMemcpy(d,h,sizeof(h),hipMemcpyHostToDevice);
  k<<<1,64>>>(d,o,o2);
  uint16_t r[256]; hipMemcpy(r,o,sizeof(r),hipMemcpyDeviceToHost);
  float r2[128]; hipMemcpy(r2,o2,sizeof(r2),hipMemcpyDeviceToHost);
  for(int l=0;l<64;++l){ printf("lane %2d: %4d %4d %4d %4d\n", l, r[l*4],r[l*4+1],r[l*4+2],r[l*4+3]); }
  printf("mfma bf16 c0=%f  f32 c5=%f\n", r2[0], r2[64]);
  return 0;
}
