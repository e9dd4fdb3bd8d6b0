// Which lane -> address patterns of ds_read_b128 / ds_read_b64_tr_b16 are bank-conflict-free on gfx950?
// One kernel name per pattern (template index), 256 workgroups x 1 wave x 4096 reads; run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- ./lds_conflict
// and compare the counters per kernel (tools/pmc_query.py <db> probe).
// hipcc --offload-arch=gfx950 -O3 -o build/lds_conflict lds_conflict.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_seg;

__device__ __forceinline__ int pattern(int v, int lane) {
  const int half = lane >> 5, l32 = lane & 31, g = lane >> 4, r16 = lane & 15;
  const int jrow = r16 >> 2, qcol = r16 & 3;
  switch (v) {
    // ---- ds_read_b128, 128-byte rows
    case 0: return l32 * 128 + ((half ^ ((l32 >> 1) & 7)) << 4);             // 32x32x16 fragment (old)
    case 1: return r16 * 128 + ((g ^ ((r16 >> 1) & 7)) << 4);                // 16x16x32 fragment, same swizzle
    case 2: return lane * 16;                                                // linear
    case 3: return r16 * 128 + ((g ^ (r16 & 7)) << 4);                       // swizzle by row & 7
    case 4: return r16 * 128 + (((g + 4 * (r16 & 1)) ^ ((r16 >> 1) & 7)) << 4);   // odd rows use the other half of the row
    case 5: return r16 * 128 + (g << 4);                                     // no swizzle (worst case)
    // ---- ds_read_b64_tr_b16, 256-byte rows (dy of the tap-fused weight gradient)
    case 6: {   // 32x32x16 mapping: lane groups = (pixel octet half, channel block g16)
      const int g16 = (lane >> 4) & 1, ycol = g16 * 16 + qcol * 4;
      return (half * 8 + jrow) * 256 + (((ycol >> 3) ^ ((jrow & 3) << 2)) << 4) + ((ycol & 4) << 1);
    }
    case 7: {   // 16x16x32 mapping: lane groups = pixel octets g, one channel block
      const int ycol = qcol * 4;
      return (g * 8 + jrow) * 256 + (((ycol >> 3) ^ ((jrow & 3) << 2)) << 4) + ((ycol & 4) << 1);
    }
    case 8: {   // ... with chunk bit 1 ^= bit 3 of the row (= g & 1)
      const int ycol = qcol * 4, row = g * 8 + jrow;
      return row * 256 + ((((ycol >> 3) ^ ((jrow & 3) << 2)) ^ (((row >> 3) & 1) << 1)) << 4) + ((ycol & 4) << 1);
    }
    case 9: {   // ... with chunk bits 1,0... only bit 1 ^= row bit 3, and byte bit 3 (8-byte slot) ^= row bit 4
      const int ycol = qcol * 4, row = g * 8 + jrow;
      return (row * 256 + ((((ycol >> 3) ^ ((jrow & 3) << 2)) ^ (((row >> 3) & 1) << 1)) << 4) + ((ycol & 4) << 1)) ^
             (((row >> 4) & 1) << 7);
    }
    // ---- ds_write_b64 of the epilogue's parking phase (rows of 144 bytes)
    case 10: return l32 * 144 + half * 8;                                    // 32x32x16 layout
    case 11: return r16 * 144 + g * 8;                                       // 16x16x32 layout
    case 12: return r16 * 144 + g * 32;                                      // 16x16x32, blocks i = g (other loop order)
    // ---- ds_read_b128 of the epilogue's write-back (8 lanes per pixel row)
    case 13: return (lane >> 3) * 144 + (lane & 7) * 16;
    // ---- fragment reads that start at an arbitrary patch row (tap offsets)
    case 14: { const int r = r16 + 1;  return r * 128 + ((g ^ ((r >> 1) & 7)) << 4); }
    case 15: { const int r = r16 + 35; return r * 128 + ((g ^ ((r >> 1) & 7)) << 4); }
    case 16: { const int r = l32 + 1;  return r * 128 + ((half ^ ((r >> 1) & 7)) << 4); }
    case 17: { const int r = r16 + 35; return r * 128 + (((4 + g) ^ ((r >> 1) & 7)) << 4); }   // k32 step 1
    default: return lane * 16;
  }
}

template <int V>
__global__ void __launch_bounds__(64) probe(int iters, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x;
  const unsigned char* ptr = lds + pattern(V, lane);
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (V >= 10 && V <= 12) {
      asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::
                   "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)ptr),
                   "v"(make_uint2(acc, it)) : "memory");
    } else if (V <= 5 || V >= 13) {
      uint4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v)
                   : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)ptr) : "memory");
      acc ^= v.x ^ v.w;
    } else {
      s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)ptr);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc ^= (uint32_t)v[0] ^ (uint32_t)v[3];
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x * 64 + lane] = acc;
}

template <int V> void run(uint32_t* out) {
  hipLaunchKernelGGL(probe<V>, dim3(256), dim3(64), 0, 0, 4096, out);
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 256 * 64 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out);
    run<6>(out); run<7>(out); run<8>(out); run<9>(out);
    run<10>(out); run<11>(out); run<12>(out); run<13>(out);
    run<14>(out); run<15>(out); run<16>(out); run<17>(out);
  }
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
