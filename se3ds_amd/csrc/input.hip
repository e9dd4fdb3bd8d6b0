// Device-side half of the training input pipeline: everything
// datasets/indoor_datasets.py does between the decoded frames and the batch dict the step
// consumes -- int -> float conversion (:185-228), random band masking of proj_mask (:281-304),
// bilinear / nearest resize (:312-314), horizontal roll + flip (:34-61,:319-324), random crop
// (:326-330) and the batch transform proj_image *= proj_mask, proj_depth *= proj_mask (:577-585)
// -- as ONE gather kernel: an output pixel pulls its sources straight from the raw uint8 / uint16
// frames, no intermediate (resized, rolled, concatenated) tensor is ever written.  HBM bound:
// reads <= 4 taps x 3 B (image) + 13 B of nearest samples, writes 40 B per output pixel.
// The random draws are made on the host (one small parameter row per sample).
// Compiled with -ffp-contract=off: every fp32 op rounds once, as in the reference's op chain.
#include "common.h"

namespace se3ds {
namespace {

constexpr int kB = 256;

struct InputXf {
  // raw frames, (N,H0,W0[,C]) row-major
  const uint8_t* image;        // (N,H0,W0,3)
  const uint8_t* proj_image;   // (N,H0,W0,3)
  const uint16_t* depth;       // (N,H0,W0)
  const uint16_t* proj_depth;  // (N,H0,W0)
  const uint8_t* proj_mask;    // (N,H0,W0)
  const uint8_t* blurred_mask; // (N,H0,W0)
  const uint8_t* segmentation; // (N,H0,W0)
  const int32_t* ip;           // [N][8]: rh, rw, roll, flip, crop_y, crop_x, hmode, vmode
  const float* fp;             // [N][4]: hstart, hend, vstart, vend
  int n, h0, w0, h, w;
  float* o_image;       // (N,h,w,3)
  float* o_proj_image;  // (N,h,w,3)
  float* o_proj_mask;   // (N,h,w,1)
  float* o_proj_depth;  // (N,h,w,1)
  float* o_depth;       // (N,h,w,1)
  float* o_blurred;     // (N,h,w,1)
  int32_t* o_seg;       // (N,h,w,1)
};

__device__ __forceinline__ int floormod(int a, int m) {
  int r = a % m;
  return r < 0 ? r + m : r;
}

__global__ void __launch_bounds__(kB)
input_transform_kernel(const InputXf p) {
  const int64_t total = (int64_t)p.n * p.h * p.w;
  const float s8 = 1.0f / 255.0f, s16 = 1.0f / 65535.0f;   // convert_image_dtype scales
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    const int b = (int)(i / ((int64_t)p.h * p.w));
    const int rem = (int)(i - (int64_t)b * p.h * p.w);
    const int y = rem / p.w, x = rem - y * p.w;
    const int32_t* ip = p.ip + b * 8;
    const float* fq = p.fp + b * 4;
    const int rh = ip[0], rw = ip[1], roll = ip[2], flip = ip[3];
    // crop -> flip -> roll, backwards: position in the resized frame
    const int yr = y + ip[4];
    int xs = x + ip[5];
    if (flip) xs = rw - 1 - xs;
    const int xr = floormod(xs - roll, rw);
    const int64_t base = (int64_t)b * p.h0 * p.w0;
    // ---- nearest sources (tf.image.resize 'nearest', half-pixel centres)
    const float sy = (float)p.h0 / (float)rh, sx = (float)p.w0 / (float)rw;
    int ny = (int)floorf(((float)yr + 0.5f) * sy);
    int nx = (int)floorf(((float)xr + 0.5f) * sx);
    ny = ny < p.h0 - 1 ? ny : p.h0 - 1;
    nx = nx < p.w0 - 1 ? nx : p.w0 - 1;
    const int64_t q = base + (int64_t)ny * p.w0 + nx;
    float pm = (float)(p.proj_mask[q] > 0 ? 1 : 0);
    // band masks live on the RAW grid (:281-304)
    const float fx = (float)nx, fy = (float)ny;
    if (ip[6] == 1) pm = pm * (((fx > fq[0]) && (fx < fq[1])) ? 1.0f : 0.0f);
    if (ip[6] == 2) pm = pm * (((fx > fq[0]) || (fx < fq[1])) ? 1.0f : 0.0f);
    if (ip[7] == 1) pm = pm * (((fy > fq[2]) && (fy < fq[3])) ? 1.0f : 0.0f);
    const float dpt = (float)p.depth[q] * s16;
    const float pdp = (float)p.proj_depth[q] * s16;
    p.o_proj_mask[i] = pm;
    p.o_depth[i] = dpt;
    p.o_proj_depth[i] = pdp * pm;
    p.o_blurred[i] = (float)(p.blurred_mask[q] > 0 ? 1 : 0);
    p.o_seg[i] = (int32_t)p.segmentation[q];
#pragma unroll
    for (int c = 0; c < 3; ++c) p.o_proj_image[i * 3 + c] = ((float)p.proj_image[q * 3 + c] * s8) * pm;
    // ---- bilinear image (tf.image.resize default), then clip to [0, 1]
    const float srcy = ((float)yr + 0.5f) * sy - 0.5f;
    const float srcx = ((float)xr + 0.5f) * sx - 0.5f;
    const float fly = floorf(srcy), flx = floorf(srcx);
    const float wy = srcy - fly, wx = srcx - flx;
    int y0 = (int)fly, x0 = (int)flx;
    int y1 = y0 + 1, x1 = x0 + 1;
    y0 = y0 < 0 ? 0 : (y0 > p.h0 - 1 ? p.h0 - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > p.h0 - 1 ? p.h0 - 1 : y1);
    x0 = x0 < 0 ? 0 : (x0 > p.w0 - 1 ? p.w0 - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 > p.w0 - 1 ? p.w0 - 1 : x1);
    const uint8_t* r0 = p.image + (base + (int64_t)y0 * p.w0) * 3;
    const uint8_t* r1 = p.image + (base + (int64_t)y1 * p.w0) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tl = (float)r0[x0 * 3 + c] * s8, tr = (float)r0[x1 * 3 + c] * s8;
      const float bl = (float)r1[x0 * 3 + c] * s8, br = (float)r1[x1 * 3 + c] * s8;
      float v;
      if (rh == p.h0 && rw == p.w0) {
        v = tl;   // identity size: tf.image.resize returns the input
      } else {
        const float top = tl + (tr - tl) * wx;
        const float bot = bl + (br - bl) * wx;
        v = top + (bot - top) * wy;
      }
      p.o_image[i * 3 + c] = fminf(fmaxf(v, 0.0f), 1.0f);
    }
  }
}

}  // namespace
}  // namespace se3ds

using namespace se3ds;

extern "C" int se3ds_input_transform(const uint8_t* image, const uint8_t* proj_image,
                                     const uint16_t* depth, const uint16_t* proj_depth,
                                     const uint8_t* proj_mask, const uint8_t* blurred_mask,
                                     const uint8_t* segmentation, const int32_t* iparams,
                                     const float* fparams, int n, int h0, int w0, int h, int w,
                                     float* o_image, float* o_proj_image, float* o_proj_mask,
                                     float* o_proj_depth, float* o_depth, float* o_blurred,
                                     int32_t* o_seg, void* stream) {
  if (n <= 0 || h0 <= 0 || w0 <= 0 || h <= 0 || w <= 0) return SE3DS_E_BADSHAPE;
  InputXf p;
  p.image = image; p.proj_image = proj_image; p.depth = depth; p.proj_depth = proj_depth;
  p.proj_mask = proj_mask; p.blurred_mask = blurred_mask; p.segmentation = segmentation;
  p.ip = iparams; p.fp = fparams; p.n = n; p.h0 = h0; p.w0 = w0; p.h = h; p.w = w;
  p.o_image = o_image; p.o_proj_image = o_proj_image; p.o_proj_mask = o_proj_mask;
  p.o_proj_depth = o_proj_depth; p.o_depth = o_depth; p.o_blurred = o_blurred; p.o_seg = o_seg;
  hipLaunchKernelGGL(input_transform_kernel, dim3(grid_for((int64_t)n * h * w, kB)), dim3(kB), 0,
                     as_stream(stream), p);
  return check_launch("input_transform");
}
