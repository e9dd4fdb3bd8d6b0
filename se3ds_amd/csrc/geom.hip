// Point-cloud warp kernels for gfx950: equirect unproject, fused equirect projection +
// z-buffer splat (zmin / resolve / finalize), bilinear gather, coordinate generators,
// mask_pano, stream compaction.  All HBM-bound integer/float scatter-gather work: one
// point (or pixel) per lane, coalesced row reads of the (N,4,M) coordinate planes, L2
// atomics for the z-buffer, wave-level pre-reduction for the reference's "sink" pixel.
//
// Reference semantics: utils/pano_utils.py, utils/point_cloud_utils.py (see the per-entry
// citations in include/se3ds_hip.h).  Index math lives in include/se3ds_geom_math.h and is
// shared bit-for-bit with the CPU oracle.  Built with -ffp-contract=off.
#include <atomic>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "common.h"
#include "../../include/se3ds_geom_math.h"

namespace se3ds {
namespace {

constexpr int kBlock = 256;

template <typename T> struct FeatIO;
template <> struct FeatIO<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ float cast_void(float v) { return v; }
};
template <> struct FeatIO<int32_t> {
  static __device__ __forceinline__ float load(const int32_t* p) { return (float)*p; }
  static __device__ __forceinline__ int32_t cast_void(float v) { return (int32_t)v; }
};
template <> struct FeatIO<uint8_t> {
  static __device__ __forceinline__ float load(const uint8_t* p) { return (float)*p; }
  static __device__ __forceinline__ uint8_t cast_void(float v) { return (uint8_t)v; }
};

// ------------------------------------------------------------------ unproject (equirect)
// One pixel per lane; grid.y = batch.  pano_utils.py:219-235 with precomputed tables.
template <typename T>
__global__ void __launch_bounds__(kBlock)
unproject_equirect_kernel(const T* __restrict__ feats, const float* __restrict__ depth,
                          const float* __restrict__ sin_el, const float* __restrict__ cos_el,
                          const float* __restrict__ sin_hd, const float* __restrict__ cos_hd,
                          const float* __restrict__ position, int height, int width, int channels,
                          float void_class, float depth_scale, float* __restrict__ xyz1,
                          T* __restrict__ feats_out, int64_t m_total, int64_t m_offset, int write_ones) {
  // outputs are windows [m_offset, m_offset + H*W) of a (N,4,m_total) / (N,m_total,C) memory
  const int b = blockIdx.y;
  const int64_t p = (int64_t)height * width;
  float pos_x = 0.f, pos_y = 0.f, pos_z = 0.f;
  if (position) {
    pos_x = position[b * 3 + 0];
    pos_y = position[b * 3 + 1];
    pos_z = position[b * 3 + 2];
  }
  const T vc = FeatIO<T>::cast_void(void_class);
  float* X = xyz1 + (int64_t)b * 4 * m_total + m_offset;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < p;
       i += (int64_t)gridDim.x * kBlock) {
    int r = (int)(i / width);
    int col = (int)(i - (int64_t)r * width);
    float d = depth[(int64_t)b * p + i];
    bool valid = (d > 0.0f) && (d < 1.0f);
    float mask = valid ? 1.0f : 0.0f;
    float rad = (d * depth_scale) * mask;
    float rs = rad * sin_el[r];
    float x = rs * cos_hd[col];
    float y = rs * sin_hd[col];
    float z = rad * cos_el[r];
    if (position) {
      x = x + pos_x;
      y = y + pos_y;
      z = z + pos_z;
    }
    X[i] = x;
    X[m_total + i] = y;
    X[2 * m_total + i] = z;
    if (write_ones) X[3 * m_total + i] = 1.0f;
    const T* fi = feats + ((int64_t)b * p + i) * channels;
    T* fo = feats_out + ((int64_t)b * m_total + m_offset + i) * channels;
    for (int k = 0; k < channels; ++k) fo[k] = valid ? fi[k] : vc;
  }
}

// Four pixels per thread (round 5): 4-byte features, C = 1 or 3, width % 4 == 0, 16-byte aligned
// bases and windows.  The scalar kernel above moves 44 B per pixel through eleven dword accesses
// (three of them stride-12 loads, three stride-12 stores) and a 64-bit division: 25 us for the
// 92 MB of a 1024 x 2048 view = 3.7 TB/s.  Here a thread owns four consecutive pixels of one row:
// depth, the heading tables and the four coordinate planes are one 16-byte access each, the C = 3
// features three 16-byte loads and stores; row / column come from one 32-bit division per thread.
// Same arithmetic per pixel, in the same order (-ffp-contract=off): bit-identical outputs.
template <typename T, int C>
__global__ void __launch_bounds__(kBlock)
unproject_equirect_vec4_kernel(const T* __restrict__ feats, const float* __restrict__ depth,
                               const float* __restrict__ sin_el, const float* __restrict__ cos_el,
                               const float* __restrict__ sin_hd, const float* __restrict__ cos_hd,
                               const float* __restrict__ position, int height, int width,
                               float void_class, float depth_scale, float* __restrict__ xyz1,
                               T* __restrict__ feats_out, int64_t m_total, int64_t m_offset,
                               int write_ones) {
  static_assert(sizeof(T) == 4 && (C == 1 || C == 3), "4-byte features, 1 or 3 channels");
  const int b = blockIdx.y;
  const uint32_t qw = (uint32_t)width >> 2;          // quads per row
  const uint32_t nq = (uint32_t)height * qw;         // quads per image (< 2^29: checked by the launcher)
  const int64_t p = (int64_t)height * width;
  float pos_x = 0.f, pos_y = 0.f, pos_z = 0.f;
  if (position) {
    pos_x = position[b * 3 + 0];
    pos_y = position[b * 3 + 1];
    pos_z = position[b * 3 + 2];
  }
  const T vc = FeatIO<T>::cast_void(void_class);
  float* X = xyz1 + (int64_t)b * 4 * m_total + m_offset;
  for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < nq; q += gridDim.x * kBlock) {
    const uint32_t r = q / qw;
    const uint32_t col = (q - r * qw) << 2;
    const int64_t i = (int64_t)q << 2;
    const float4 d4 = *reinterpret_cast<const float4*>(depth + (int64_t)b * p + i);
    const float4 ch4 = *reinterpret_cast<const float4*>(cos_hd + col);
    const float4 sh4 = *reinterpret_cast<const float4*>(sin_hd + col);
    const float se = sin_el[r], ce = cos_el[r];
    T f[4 * C];
    const T* fi = feats + ((int64_t)b * p + i) * C;
#pragma unroll
    for (int k = 0; k < C; ++k)
      *reinterpret_cast<uint4*>(&f[4 * k]) = *reinterpret_cast<const uint4*>(fi + 4 * k);
    const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
    const float cc[4] = {ch4.x, ch4.y, ch4.z, ch4.w};
    const float ss[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    float xo[4], yo[4], zo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = dd[e];
      const bool valid = (d > 0.0f) && (d < 1.0f);
      const float mask = valid ? 1.0f : 0.0f;
      const float rad = (d * depth_scale) * mask;
      const float rs = rad * se;
      float x = rs * cc[e];
      float y = rs * ss[e];
      float z = rad * ce;
      if (position) {
        x = x + pos_x;
        y = y + pos_y;
        z = z + pos_z;
      }
      xo[e] = x; yo[e] = y; zo[e] = z;
#pragma unroll
      for (int k = 0; k < C; ++k) f[e * C + k] = valid ? f[e * C + k] : vc;
    }
    *reinterpret_cast<float4*>(X + i) = make_float4(xo[0], xo[1], xo[2], xo[3]);
    *reinterpret_cast<float4*>(X + m_total + i) = make_float4(yo[0], yo[1], yo[2], yo[3]);
    *reinterpret_cast<float4*>(X + 2 * m_total + i) = make_float4(zo[0], zo[1], zo[2], zo[3]);
    if (write_ones)   // (SE3DS_XYZ1_ONES_PRESET: the memory's row 3 was filled when it was allocated)
      *reinterpret_cast<float4*>(X + 3 * m_total + i) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    T* fo = feats_out + ((int64_t)b * m_total + m_offset + i) * C;
#pragma unroll
    for (int k = 0; k < C; ++k)
      *reinterpret_cast<uint4*>(fo + 4 * k) = *reinterpret_cast<const uint4*>(&f[4 * k]);
  }
}

// ------------------------------------------------------------------ unproject (perspective)
__global__ void __launch_bounds__(kBlock)
unproject_perspective_kernel(const int32_t* __restrict__ feats, const float* __restrict__ depth,
                             const float* __restrict__ xs, const float* __restrict__ ys,
                             const float* __restrict__ kinv, int height, int width, int channels,
                             float depth_scale, float* __restrict__ xyz,
                             float* __restrict__ feats_out) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)height * width;
  float k[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) k[i] = kinv[i];
  float* X = xyz + (int64_t)b * 4 * p;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < p;
       i += (int64_t)gridDim.x * kBlock) {
    int r = (int)(i / width);
    int col = (int)(i - (int64_t)r * width);
    float d = depth[(int64_t)b * p + i] * depth_scale;
    bool valid = (d > 0.0f) && (d < depth_scale);
    float m = valid ? 1.0f : 0.0f;
    float v0 = (xs[col] * d) * m, v1 = (ys[r] * d) * m, v2 = d * m, v3 = 1.0f * m;
    // tf.matmul(inv(K), xyz): 4-term dot products, accumulated left to right in fp32.
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      float acc = k[row * 4 + 0] * v0;
      acc = acc + k[row * 4 + 1] * v1;
      acc = acc + k[row * 4 + 2] * v2;
      acc = acc + k[row * 4 + 3] * v3;
      X[row * p + i] = acc;
    }
    const int32_t* fi = feats + ((int64_t)b * p + i) * channels;
    float* fo = feats_out + ((int64_t)b * p + i) * channels;
    for (int c = 0; c < channels; ++c) fo[c] = (float)(fi[c] * (valid ? 1 : 0));
  }
}

// ------------------------------------------------------------------------------ splat
// Workspace layout (bytes): [0,256) header: u32 sink_z (ordered), u32 sink_feat[C<=60];
//   u32 zpart[kMaxSinkBlocks], u32 fpart[kMaxSinkBlocks][C] (per-block sink partials: the
//   reference's sink pixel collects EVERY invalid point, and one atomic per wave on a single
//   address serialises at ~12 ns each -- 200-600 us at 4 M points; per-block partials plus a
//   one-block reduce cost ~3 us);  then int32 idx[N*M], float z[N*M].
constexpr int kMaxSinkBlocks = 2048;
// Word 3 of the header (bytes 12..15, between sink_z and the sink features) is STICKY: the
// packed / sorted splats OR a 1 into it when an int32 feature breaks the caller's
// SE3DS_FEAT_BYTE_RANGE promise and nothing in the library ever clears it -- the owner of the
// workspace zeroes the header once and reads it whenever convenient (se3ds_splat_promise_sticky).
constexpr int kPromiseStickyWord = 3;
// The PER-CALL verdict (se3ds_splat_promise_broken) lives in the control words of the packed /
// sorted workspaces: a violating workgroup EXCHANGES this call's epoch into ctl[1], the first
// workgroup stores the epoch into ctl[kPromiseEpochWord], and the reader compares the two.  Round 4
// zeroed ctl[1] from block (0, 0) of the same launch whose other blocks OR into it: a block that
// finished before block (0, 0) started lost its bit.  Nothing is zeroed now.
// ctl[1] is never initialised (the workspace is caller memory and the control words move with n * m):
// an epoch carries a tag in its upper byte so that left-over data -- small counts, histogram words --
// cannot equal the current one (a 24-bit call counter under 0xA5).
constexpr int kPromiseEpochWord = 6;
constexpr uint32_t kEpochTag = 0xA5000000u, kEpochTagMask = 0xff000000u;
inline uint32_t next_splat_epoch() {
  static std::atomic<uint32_t> epoch{0};
  return kEpochTag | (++epoch & ~kEpochTagMask);
}
struct SplatWs {
  uint32_t* sink_z;
  uint32_t* sink_feat;
  uint32_t* zpart;
  uint32_t* fpart;
  int32_t* idx;
  float* z;
};
constexpr int kMaxSplatChannels = 60;
__host__ __device__ inline size_t splat_hdr_bytes() {
  return 256 + sizeof(uint32_t) * (size_t)kMaxSinkBlocks * (size_t)(1 + kMaxSplatChannels);
}
__host__ __device__ inline SplatWs carve_ws(void* ws, int n, int64_t m) {
  SplatWs w;
  char* p = (char*)ws;
  w.sink_z = (uint32_t*)p;
  w.sink_feat = (uint32_t*)(p + 16);
  w.zpart = (uint32_t*)(p + 256);
  w.fpart = w.zpart + kMaxSinkBlocks;
  char* q = p + splat_hdr_bytes();
  w.idx = (int32_t*)q;
  w.z = (float*)(q + sizeof(int32_t) * (size_t)n * (size_t)m);
  return w;
}
// grid for the per-point kernels: at most kMaxSinkBlocks blocks over (x, batch)
inline dim3 point_grid(int64_t m, int n, int block) {
  int64_t gx = ceil_div(m, block);
  int64_t cap = kMaxSinkBlocks / n;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)n);
}

// K0: zmin := depth_scale (raw fp32 bits live in the depth output buffer), feat := void.
template <bool ORDERED>
__global__ void __launch_bounds__(kBlock)
splat_init_kernel(float* __restrict__ zmin, float* __restrict__ feat, int64_t npx, int channels,
                  float depth_scale, float output_void, SplatWs ws) {
  int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * kBlock;
  if (tid == 0) *ws.sink_z = 0xffffffffu;
  if (tid < channels) ws.sink_feat[tid] = 0u;
  const float fv = ORDERED ? __uint_as_float(se3ds_f32_to_ordered(output_void)) : output_void;
  for (int64_t i = tid; i < npx; i += stride) zmin[i] = depth_scale;
  for (int64_t i = tid; i < npx * channels; i += stride) feat[i] = fv;
}

// K1: per point -> (idx, z); atomic-min the z-buffer.  EQUIRECT fuses pano_utils.py:139-156.
template <typename T, bool EQUIRECT>
__global__ void __launch_bounds__(kBlock)
splat_zmin_kernel(const float* __restrict__ coords, const float* __restrict__ offset,
                  const T* __restrict__ feats, int64_t m, int64_t ld, int channels, int height,
                  int width, float input_void, float* __restrict__ zmin, SplatWs ws) {
  // ld: capacity (in points) of the memory the first m points of every image live in
  const int b = blockIdx.y;
  const int64_t hw = (int64_t)height * width;
  const float* X = coords + (int64_t)b * 4 * ld;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (EQUIRECT && offset) {
    ox = offset[b * 3 + 0];
    oy = offset[b * 3 + 1];
    oz = offset[b * 3 + 2];
  }
  uint32_t sink = 0xffffffffu;
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < m; i0 += (int64_t)gridDim.x * kBlock) {
    int64_t i = i0 + threadIdx.x;
    if (i < m) {
      float x = X[i], y = X[ld + i], z = X[2 * ld + i];
      float px, py, pz;
      if (EQUIRECT) {
        if (offset) {
          x = x - ox;
          y = y - oy;
          z = z - oz;
        }
        se3ds_equirect_project(x, y, z, &px, &py, &pz);
      } else {
        px = x;
        py = y;
        pz = z;
      }
      const T* f = feats + ((int64_t)b * ld + i) * channels;
      int fv = 1;
      for (int k = 0; k < channels; ++k) fv &= (FeatIO<T>::load(f + k) != input_void);
      int32_t idx = se3ds_splat_index(px, py, pz, width, height, fv);
      ws.idx[(int64_t)b * m + i] = idx;
      ws.z[(int64_t)b * m + i] = pz;
      if (idx >= 0) {
        // valid => pz > 0: raw fp32 bits are order-preserving as unsigned integers.
        atomicMin(reinterpret_cast<unsigned int*>(zmin + (int64_t)b * hw + idx),
                  __float_as_uint(pz));
      } else if (pz == pz) {  // NaN never wins a min
        uint32_t o = se3ds_f32_to_ordered(pz);
        sink = o < sink ? o : sink;
      }
    }
  }
  // The reference redirects every invalid point to flat index 0 (point_cloud_utils.py:151-152):
  // reduce their z per block into a partial (no same-address atomics).
  __shared__ uint32_t s_sink[kBlock / 64];
  sink = wave_min_u32(sink);
  if ((threadIdx.x & 63) == 0) s_sink[threadIdx.x >> 6] = sink;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t v = s_sink[0];
    for (int i = 1; i < kBlock / 64; ++i) v = s_sink[i] < v ? s_sink[i] : v;
    ws.zpart[blockIdx.y * gridDim.x + blockIdx.x] = v;
  }
}

// one block: sink_z = min over the per-block partials
__global__ void __launch_bounds__(kBlock)
splat_sink_z_kernel(SplatWs ws, int nparts) {
  __shared__ uint32_t s_sink[kBlock / 64];
  uint32_t v = 0xffffffffu;
  for (int i = threadIdx.x; i < nparts; i += kBlock) v = ws.zpart[i] < v ? ws.zpart[i] : v;
  v = wave_min_u32(v);
  if ((threadIdx.x & 63) == 0) s_sink[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < kBlock / 64; ++i) v = s_sink[i] < v ? s_sink[i] : v;
    v = s_sink[0] < v ? s_sink[0] : v;
    *ws.sink_z = v;
  }
}

// one block: sink_feat[c] = max over the per-block partials
__global__ void __launch_bounds__(kBlock)
splat_sink_feat_kernel(SplatWs ws, int nparts, int channels) {
  __shared__ uint32_t s_red[kBlock / 64];
  for (int c = 0; c < channels; ++c) {
    uint32_t v = 0u;
    for (int i = threadIdx.x; i < nparts; i += kBlock) {
      uint32_t t = ws.fpart[(int64_t)i * channels + c];
      v = t > v ? t : v;
    }
    v = wave_max_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < kBlock / 64; ++i) v = s_red[i] > v ? s_red[i] : v;
      ws.sink_feat[c] = v;
    }
  }
}

// K2: survivors (z < zmin + 0.1) scatter-max their features; the rest go to the sink.
template <typename T, bool ORDERED>
__global__ void __launch_bounds__(kBlock)
splat_resolve_kernel(const T* __restrict__ feats, int64_t m, int64_t ld, int channels, int height,
                     int width, const float* __restrict__ zmin, float* __restrict__ feat,
                     SplatWs ws) {
  const int b = blockIdx.y;
  const int64_t hw = (int64_t)height * width;
  constexpr int kMaxC = 4;  // wave-reduced sink channels per pass
  const float sink_z = se3ds_ordered_to_f32(*ws.sink_z);
  const bool have_sink_z = (*ws.sink_z != 0xffffffffu);
  for (int c0 = 0; c0 < channels; c0 += kMaxC) {
    uint32_t smax[kMaxC];
#pragma unroll
    for (int k = 0; k < kMaxC; ++k) smax[k] = 0u;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < m; i0 += (int64_t)gridDim.x * kBlock) {
      int64_t i = i0 + threadIdx.x;
      if (i < m) {
        int32_t idx = ws.idx[(int64_t)b * m + i];
        float z = ws.z[(int64_t)b * m + i];
        int64_t fl = idx < 0 ? 0 : (int64_t)b * hw + idx;
        float zm = zmin[fl];
        if (fl == 0 && have_sink_z) zm = sink_z < zm ? sink_z : zm;
        bool keep = (idx >= 0) && (z < zm + 0.1f);
        const T* f = feats + ((int64_t)b * ld + i) * channels;
#pragma unroll
        for (int k = 0; k < kMaxC; ++k) {
          if (c0 + k < channels) {
            float v = FeatIO<T>::load(f + c0 + k);
            if (keep) {
              if (ORDERED) {
                atomicMax(reinterpret_cast<unsigned int*>(feat + fl * channels + c0 + k),
                          se3ds_f32_to_ordered(v));
              } else if (v > 0.0f) {  // buffer starts at output_void >= 0: only v > 0 can win
                atomicMax(reinterpret_cast<int*>(feat + fl * channels + c0 + k),
                          __float_as_int(v));
              }
            } else if (v == v) {
              uint32_t o = se3ds_f32_to_ordered(v);
              smax[k] = o > smax[k] ? o : smax[k];
            }
          }
        }
      }
    }
    __shared__ uint32_t s_f[kMaxC][kBlock / 64];
#pragma unroll
    for (int k = 0; k < kMaxC; ++k) {
      uint32_t s = wave_max_u32(smax[k]);
      if ((threadIdx.x & 63) == 0) s_f[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < kMaxC && c0 + (int)threadIdx.x < channels) {
      uint32_t v = 0u;
      for (int i = 0; i < kBlock / 64; ++i) v = s_f[threadIdx.x][i] > v ? s_f[threadIdx.x][i] : v;
      ws.fpart[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * channels + c0 + threadIdx.x] = v;
    }
    __syncthreads();
  }
}

// K3: per pixel: fold the sink into flat pixel 0, normalise depth, decode features, mask.
template <bool ORDERED>
__global__ void __launch_bounds__(kBlock)
splat_finalize_kernel(float* __restrict__ depth, float* __restrict__ feat,
                      float* __restrict__ mask, int64_t npx, int channels, float depth_scale,
                      float mask_void, SplatWs ws) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npx;
       i += (int64_t)gridDim.x * kBlock) {
    float z = depth[i];
    if (i == 0 && *ws.sink_z != 0xffffffffu) {
      float sz = se3ds_ordered_to_f32(*ws.sink_z);
      z = sz < z ? sz : z;
    }
    float d = z < 0.0f ? 0.0f : (z > depth_scale ? depth_scale : z);
    d = d / depth_scale;
    depth[i] = d;
    bool all_ok = true;
    for (int k = 0; k < channels; ++k) {
      float v = feat[i * channels + k];
      if (ORDERED) v = se3ds_ordered_to_f32(__float_as_uint(v));
      if (i == 0 && ws.sink_feat[k] != 0u) {
        float sv = se3ds_ordered_to_f32(ws.sink_feat[k]);
        v = sv > v ? sv : v;
      }
      if (ORDERED || i == 0) feat[i * channels + k] = v;
      all_ok = all_ok && (v != mask_void);
    }
    if (mask) mask[i] = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
  }
}

// ---- the fp32 index screen of TWO points at a time on the packed fp32 pipe (round 6).
// S1 is VALU-bound (profiles/r05_warp_valu_pmc.json: 3 719 VALU instructions per 1 024-point wave = 232 per
// point, VALU 0.73 busy).  gfx950 issues v_pk_mul / v_pk_add / v_pk_fma_f32 -- two lanes' worth of work
// per issue slot -- so the screen (se3ds_equirect_fxy_fast: sum of squares, two polynomial arc tangents,
// the scalings) runs on float2 pairs: the same operations in the same order, each individually rounded
// (no contraction in this file), so rad -- an OUTPUT -- is bit for bit the scalar chain's.  Selects,
// the IEEE square root and the IEEE quotient z / rad stay per point.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) {
  f32x2 r = {v, v};
  return r;
}
// se3ds_atan2_fast on two points.  POS_Y: both y are >= 0 (the elevation's sqrt term): no sign step.
template <bool POS_Y>
__device__ __forceinline__ f32x2 atan2_fast2(f32x2 y, f32x2 x) {
  const float ay0 = __builtin_fabsf(y.x), ax0 = __builtin_fabsf(x.x);
  const float ay1 = __builtin_fabsf(y.y), ax1 = __builtin_fabsf(x.y);
  const bool sw0 = ay0 > ax0, sw1 = ay1 > ax1;
  const f32x2 a = {sw0 ? ax0 : ay0, sw1 ? ax1 : ay1};
  const f32x2 b = {sw0 ? ay0 : ax0, sw1 ? ay1 : ax1};
  const f32x2 rb = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
  const f32x2 q = a * rb;   // b == 0 -> NaN / inf -> the margin test fails
  const f32x2 sq = q * q;
  f32x2 p = splat2(-0.004668773151934147f);
  p = __builtin_elementwise_fma(p, sq, splat2(0.02416618913412094f));
  p = __builtin_elementwise_fma(p, sq, splat2(-0.0593671016395092f));
  p = __builtin_elementwise_fma(p, sq, splat2(0.09906096756458282f));
  p = __builtin_elementwise_fma(p, sq, splat2(-0.14016585052013397f));
  p = __builtin_elementwise_fma(p, sq, splat2(0.19969235360622406f));
  p = __builtin_elementwise_fma(p, sq, splat2(-0.33331960439682007f));
  p = __builtin_elementwise_fma(p, sq, splat2(0.9999998807907104f));
  f32x2 r = q * p;
  const f32x2 r2 = splat2(1.57079632679489661923f) - r;
  r.x = sw0 ? r2.x : r.x;
  r.y = sw1 ? r2.y : r.y;
  const f32x2 r3 = splat2(3.14159265358979323846f) - r;
  r.x = x.x < 0.0f ? r3.x : r.x;
  r.y = x.y < 0.0f ? r3.y : r.y;
  if (!POS_Y) {
    r.x = y.x < 0.0f ? -r.x : r.x;
    r.y = y.y < 0.0f ? -r.y : r.y;
  }
  return r;
}
// The IEEE square root and quotient of the screen WITHOUT their range scaling (round 6: 27 of the screen's
// 74 instructions per point were these two).  sqrt_rn_mid is the compiler's own expansion of sqrtf --
// v_sqrt_f32 (1 ulp), then the two neighbours checked with exact fma residuals -- minus the rescaling of
// arguments below 2^-96, so it returns the correctly rounded root (bit for bit `__builtin_sqrtf`) for
// x >= 2^-96; the screen only ACCEPTS points with rad > 1e-14 (rad^2 > 2^-93), everything else takes the
// exact chain, which computes its own rad.  div_rn_mid likewise is the v_div_* sequence minus
// v_div_scale / v_div_fixup: identical to IEEE division unless an intermediate leaves the normal range,
// which for |z| <= rad in [1e-14, 1e30] only happens for quotients so small that the elevation does not
// see their last bits (the reference's rounded quotient matters next to the poles, |w| -> 1).
__device__ __forceinline__ float sqrt_rn_mid(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = rm <= 0.0f ? sm : s;
  s = rp > 0.0f ? sp : s;
  return s;
}
__device__ __forceinline__ float div_rn_mid(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  float q = n * r;
  q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
  return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
// se3ds_equirect_fxy_fast on two points (camera-relative x, y, z): fx, fy and rad.
__device__ __forceinline__ void equirect_fxy_fast2(f32x2 x, f32x2 y, f32x2 z, float fwidth, float fheight,
                                                   f32x2* fx, f32x2* fy, f32x2* rad_out) {
  const f32x2 s2 = (x * x + y * y) + z * z;
  const f32x2 rad = {sqrt_rn_mid(s2.x), sqrt_rn_mid(s2.y)};   // correctly rounded where accepted: rad IS an output (depth)
  *rad_out = rad;
  f32x2 heading = splat2(SE3DS_F32_ONE_HALF_PI) - atan2_fast2<false>(y, x);
  // (1.5 pi - atan2 lies in [0.5 pi, 2.5 pi]: the scalar chain's "+ 2 pi if <= 0" never fires here)
  const f32x2 hw = heading - splat2(SE3DS_F32_TWO_PI);
  heading.x = heading.x > SE3DS_F32_TWO_PI ? hw.x : heading.x;
  heading.y = heading.y > SE3DS_F32_TWO_PI ? hw.y : heading.y;
  // w must be the reference's individually rounded quotient (se3ds_equirect_fxy_fast)
  const f32x2 w = {div_rn_mid(z.x, rad.x), div_rn_mid(z.y, rad.y)};
  const f32x2 t = (splat2(1.0f) - w) * (splat2(1.0f) + w);
  const f32x2 st = {__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
  const f32x2 elevation = atan2_fast2<true>(st, w);
  *fx = (heading * splat2(0.159154943091895336f)) * splat2(fwidth);
  *fy = (elevation * splat2(0.318309886183790672f)) * splat2(fheight);
}
// the verdict of se3ds_equirect_uv_fast on one point of a pair
__device__ __forceinline__ bool screen_decided(float fx, float fy, float rad, float mx, float my) {
  const float dx = __builtin_fabsf(fx - __builtin_rintf(fx)), dy = __builtin_fabsf(fy - __builtin_rintf(fy));
  // (rad > 1e-14: the range in which sqrt_rn_mid / div_rn_mid are the IEEE results, see there)
  return (dx > mx) && (dy > my) && (rad > 1.0e-14f) && (rad < 1.0e30f);
}

// Parity tap of the DEVICE fast screen (se3ds_geom_math.h: v_rcp_f32 / v_sqrt_f32 inside): fx, fy
// and the screen's verdict per point (decided: idx >= -1, undecided: -2), so that the tests can
// bound its deviation from the exact chain on the real hardware.  It runs the PACKED pair screen the
// sorted splat ships (equirect_fxy_fast2 / screen_decided: points i and i + 1 as one pair).
__global__ void __launch_bounds__(kBlock)
debug_fast_fxy_kernel(const float* __restrict__ xyz, int64_t m, int width, int height,
                      float* __restrict__ fx, float* __restrict__ fy, int32_t* __restrict__ verdict) {
  const float fw = (float)width, fh = (float)height;
  const float mx = SE3DS_FAST_MARGIN * fw, my = SE3DS_FAST_MARGIN * fh;
  for (int64_t i = 2 * ((int64_t)blockIdx.x * kBlock + threadIdx.x); i < m;
       i += 2 * (int64_t)gridDim.x * kBlock) {
    const int64_t j = i + 1 < m ? i + 1 : i;
    const f32x2 x = {xyz[i], xyz[j]}, y = {xyz[m + i], xyz[m + j]}, z = {xyz[2 * m + i], xyz[2 * m + j]};
    f32x2 gx, gy, rad;
    equirect_fxy_fast2(x, y, z, fw, fh, &gx, &gy, &rad);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t k = e ? j : i;
      const float ex = e ? gx.y : gx.x, ey = e ? gy.y : gy.x, er = e ? rad.y : rad.x;
      fx[k] = ex;
      fy[k] = ey;
      int32_t v = -2;
      if (screen_decided(ex, ey, er, mx, my)) {
        const bool ok = (ex > -1.0f) && (ex < fw) && (ey > -1.0f) && (ey < fh);
        v = ok ? (int32_t)ey * width + (int32_t)ex : -1;
      }
      verdict[k] = v;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
splat_debug_copy_kernel(SplatWs ws, int64_t total, int32_t* idx_out, float* z_out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    idx_out[i] = ws.idx[i];
    z_out[i] = ws.z[i];
  }
}

// ------------------------------------------------------------------ binned splat (owner computes)
// The scatter version above pays 4 agent-scope atomics per point on memory that eight
// non-coherent L2s share (executed memory-side, ~28 G/s: 350 us at 4 M points).  Here the points
// are first binned by TARGET tile (16 x 128 pixels): one LDS histogram per chunk of points ->
// column scan over the chunks -> scatter of (pixel-in-tile, z, features) records; then one
// workgroup per tile resolves z-min and the per-channel max in LDS and writes depth, features
// and mask directly.  min / max do not depend on the record order, so the result is bit-identical
// to the scatter version.  Used for <= 7 channels and <= 4096 tiles per image.
constexpr int kTileY = 16, kTileX = 128, kTilePx = kTileY * kTileX;
constexpr int kMaxTiles = 4096;      // per image (LDS histogram)
constexpr int kMaxBinChannels = 7;
// The points of one image are cut into contiguous CHUNKS (one workgroup each, the same cut in the
// count and the scatter pass).  Each chunk owns a histogram row, so record positions come from a
// column scan of those rows: no global atomics anywhere (2 M cursor atomics cost ~70 us before).
constexpr int kChunkThreads = 512;
constexpr int kChunkPoints = 8192;   // minimum points per chunk (16 per thread)
constexpr int kMaxChunks = 1024;     // per image
constexpr int kScanWaves = 16;
constexpr int kExactQueue = 2048;   // per chunk, points waiting for the binary64 path

struct ChunkGeom {
  int chunks;     // per image
  int64_t per;    // points per chunk (multiple of kChunkThreads)
};
__host__ __device__ inline ChunkGeom chunk_geom(int64_t m, int n) {
  int64_t cap = kMaxSinkBlocks / (n > 0 ? n : 1);
  cap = cap < 1 ? 1 : (cap > kMaxChunks ? kMaxChunks : cap);
  int64_t chunks = (m + kChunkPoints - 1) / kChunkPoints;
  chunks = chunks < 1 ? 1 : (chunks > cap ? cap : chunks);
  int64_t per = (m + chunks - 1) / chunks;
  per = (per + kChunkThreads - 1) / kChunkThreads * kChunkThreads;
  if (per < kChunkThreads) per = kChunkThreads;
  chunks = (m + per - 1) / per;
  ChunkGeom g;
  g.chunks = (int)(chunks < 1 ? 1 : chunks);
  g.per = per;
  return g;
}

constexpr uint32_t kSliceRecords = 8192;   // records per band workgroup aimed at
__host__ __device__ inline int tile_bands_log2(uint32_t count, uint32_t slice_records) {
  int lg = 0;
  while (lg < 3 && ((uint64_t)slice_records << lg) < count) ++lg;
  return lg;
}
// upper bound of the resolve items: every tile has one band, and a tile with c > slice_records
// records has a power of two < 2 c / slice_records + 1 < 3 c / slice_records of them
__host__ __device__ inline int64_t resolve_items_max(int64_t nb, int64_t points, uint32_t slice_records) {
  return nb + 3 * ((points + slice_records - 1) / slice_records);
}
// (SE3DS_SPLAT_SLICE overrides the slice size: the parity tests use small slices to exercise
// banded tiles on small images)
inline uint32_t slice_records() {
  static const uint32_t v = [] {
    const char* e = getenv("SE3DS_SPLAT_SLICE");
    const long x = e ? atol(e) : 0;
    return (uint32_t)(x >= 16 ? x : (long)kSliceRecords);
  }();
  return v;
}

struct BinWs {
  uint32_t* tile_count;   // [nb]
  uint32_t* hist;         // [n][chunks][ntiles]: counts, then exclusive prefix over the chunks
  uint32_t* fpart2;       // [items][C] sink partials of the occluded points, per resolve item
  uint32_t* rec;          // [N*M][2 + C]: pixel inside the tile, z bits, feature bits
};
__host__ __device__ inline size_t align16(size_t v) { return (v + 15) / 16 * 16; }
inline size_t bin_ws_bytes(int n, int64_t m, int height, int width, int channels) {
  const size_t ntiles = (size_t)ceil_div(height, kTileY) * ceil_div(width, kTileX);
  const size_t nb = (size_t)n * ntiles;
  const size_t pts = (size_t)n * (size_t)(m > 0 ? m : 0);
  const size_t chunks = (size_t)chunk_geom(m, n).chunks;
  const size_t items = (size_t)resolve_items_max((int64_t)nb, (int64_t)pts, slice_records());
  return align16(4 * nb) + align16(4 * nb * chunks) +
         align16(4 * items * channels) + align16(4 * pts * (2 + channels));
}
inline BinWs carve_bin_ws(void* base, int n, int64_t m, int height, int width, int channels) {
  const size_t ntiles = (size_t)ceil_div(height, kTileY) * ceil_div(width, kTileX);
  const size_t nb = (size_t)n * ntiles;
  const size_t chunks = (size_t)chunk_geom(m, n).chunks;
  const size_t items = (size_t)resolve_items_max(
      (int64_t)nb, (int64_t)n * (m > 0 ? m : 0), slice_records());
  char* p = (char*)base;
  BinWs w;
  w.tile_count = (uint32_t*)p; p += align16(4 * nb);
  w.hist = (uint32_t*)p; p += align16(4 * nb * chunks);
  w.fpart2 = (uint32_t*)p; p += align16(4 * items * channels);
  w.rec = (uint32_t*)p;
  return w;
}

// packed target of a valid point: tile << 11 | pixel inside the tile (kTilePx = 2048).
// y = idx / width through a host-made reciprocal: wmagic = 2^40 / width + 1 is exact while
// idx * width < 2^40 (checked by the launcher).
__device__ __forceinline__ int32_t pack_tile(int32_t idx, int width, uint64_t wmagic, int tiles_x) {
  const int y = (int)(((uint64_t)(uint32_t)idx * wmagic) >> 40), x = idx - y * width;
  const int ty = y / kTileY, tx = x / kTileX;
  return ((ty * tiles_x + tx) << 11) | ((y - ty * kTileY) * kTileX + (x - tx * kTileX));
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan of `v` over the workgroup (NW waves); *total = sum.  s_w: NW words of LDS.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_u32(v);
  __syncthreads();   // s_w may still be read from a previous call
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  uint32_t run = incl - v, all = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t t = s_w[i];
    if (i < w) run += t;
    all += t;
  }
  *total = all;
  return run;
}

// A: per point -> (packed target, z) as splat_zmin_kernel, no z-buffer atomics; one histogram row
// per chunk.
template <typename T, bool EQUIRECT>
__global__ void __launch_bounds__(kChunkThreads)
splat_bin_count_kernel(const float* __restrict__ coords, const float* __restrict__ offset,
                       const T* __restrict__ feats, int64_t m, int64_t ld, int64_t per, int channels,
                       int height, int width, uint64_t wmagic, float input_void, int ntiles,
                       int tiles_x, SplatWs ws, BinWs bw) {
  extern __shared__ uint32_t s_hist[];   // [ntiles]
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < ntiles; t += kChunkThreads) s_hist[t] = 0u;
  __syncthreads();
  const float* X = coords + (int64_t)b * 4 * ld;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (EQUIRECT && offset) {
    ox = offset[b * 3 + 0];
    oy = offset[b * 3 + 1];
    oz = offset[b * 3 + 2];
  }
  uint32_t sink = 0xffffffffu;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
  // the tail every point ends with, whichever path produced (idx, pz)
  auto finish = [&](int64_t i, int32_t idx, float pz) {
    if (idx >= 0) {
      atomicAdd(&s_hist[pack_tile(idx, width, wmagic, tiles_x) >> 11], 1u);
    } else if (pz == pz) {
      uint32_t o = se3ds_f32_to_ordered(pz);
      sink = o < sink ? o : sink;
    }
    ws.idx[(int64_t)b * m + i] = idx;
    ws.z[(int64_t)b * m + i] = pz;
  };
  auto exact = [&](int64_t i, float x, float y, float z, int fv) {
    float px, py, pz;
    if (EQUIRECT) {
      se3ds_equirect_project(x, y, z, &px, &py, &pz);
    } else {
      px = x;
      py = y;
      pz = z;
    }
    finish(i, se3ds_splat_index(px, py, pz, width, height, fv), pz);
  };
  auto load = [&](int64_t i, float* x, float* y, float* z, int* fv) {
    *x = X[i];
    *y = X[ld + i];
    *z = X[2 * ld + i];
    if (EQUIRECT && offset) {
      *x = *x - ox;
      *y = *y - oy;
      *z = *z - oz;
    }
    const T* f = feats + ((int64_t)b * ld + i) * channels;
    int v = 1;
    for (int k = 0; k < channels; ++k) v &= (FeatIO<T>::load(f + k) != input_void);
    *fv = v;
  };
  // Equirect: the fp32 screen (se3ds_geom_math.h) decides ~97 % of the points; the rest are
  // queued in LDS and take the binary64 path densely after the loop (a divergent fallback would
  // make nearly every wave pay for it).
  __shared__ uint32_t s_queue[kExactQueue];
  __shared__ uint32_t s_qn;
  if (threadIdx.x == 0) s_qn = 0u;
  __syncthreads();
  // the next point's inputs are requested before the current one is worked on
  float nx = 0.f, ny = 0.f, nz = 0.f;
  int nfv = 0;
  if (lo + threadIdx.x < hi) load(lo + threadIdx.x, &nx, &ny, &nz, &nfv);
  for (int64_t i = lo + threadIdx.x; i < hi; i += kChunkThreads) {
    const float x = nx, y = ny, z = nz;
    const int fv = nfv;
    if (i + kChunkThreads < hi) load(i + kChunkThreads, &nx, &ny, &nz, &nfv);
    if (EQUIRECT) {
      int32_t idx = -1;
      float pz;
      if (se3ds_equirect_index_fast(x, y, z, width, height, fv, &idx, &pz)) {
        finish(i, idx, pz);
      } else {
        const uint32_t slot = atomicAdd(&s_qn, 1u);
        if (slot < (uint32_t)kExactQueue)
          s_queue[slot] = (uint32_t)(i - lo);
        else
          exact(i, x, y, z, fv);
      }
    } else {
      exact(i, x, y, z, fv);
    }
  }
  if (EQUIRECT) {
    __syncthreads();
    const uint32_t qn = s_qn < (uint32_t)kExactQueue ? s_qn : (uint32_t)kExactQueue;
    for (uint32_t q = threadIdx.x; q < qn; q += kChunkThreads) {
      const int64_t i = lo + s_queue[q];
      float x, y, z;
      int fv;
      load(i, &x, &y, &z, &fv);
      exact(i, x, y, z, fv);
    }
  }
  __shared__ uint32_t s_sink[kChunkThreads / 64];
  sink = wave_min_u32(sink);
  if ((threadIdx.x & 63) == 0) s_sink[threadIdx.x >> 6] = sink;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t v = s_sink[0];
    for (int i = 1; i < kChunkThreads / 64; ++i) v = s_sink[i] < v ? s_sink[i] : v;
    ws.zpart[blockIdx.y * gridDim.x + blockIdx.x] = v;
  }
  uint32_t* row = bw.hist + ((int64_t)b * gridDim.x + blockIdx.x) * ntiles;
  for (int t = threadIdx.x; t < ntiles; t += kChunkThreads) row[t] = s_hist[t];
}

// B1: per tile, exclusive prefix of the chunk rows (in place) and the tile total.  One workgroup
// per 64 tiles (lanes, coalesced rows); its 16 waves split the chunks.
__global__ void __launch_bounds__(64 * kScanWaves)
splat_bin_colscan_kernel(BinWs bw, int chunks, int ntiles, int nb) {
  __shared__ uint32_t s_sum[kScanWaves][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gt = blockIdx.x * 64 + lane;
  const bool ok = gt < nb;
  const int b = ok ? gt / ntiles : 0, t = ok ? gt - b * ntiles : 0;
  const int rows = ceil_div(chunks, kScanWaves);
  const int c0 = w * rows, c1 = c0 + rows < chunks ? c0 + rows : chunks;
  uint32_t* H = bw.hist + (int64_t)b * chunks * ntiles + t;
  uint32_t sum = 0;
  if (ok) {
#pragma unroll 8
    for (int c = c0; c < c1; ++c) sum += H[(int64_t)c * ntiles];
  }
  s_sum[w][lane] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int i = 0; i < kScanWaves; ++i) {
    const uint32_t v = s_sum[i][lane];
    if (i < w) run += v;
    total += v;
  }
  if (ok) {
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
      const uint32_t v = H[(int64_t)c * ntiles];
      H[(int64_t)c * ntiles] = run;
      run += v;
    }
    if (w == 0) bw.tile_count[gt] = total;
  }
}

// C: scatter the valid points of a chunk into its slice of every tile's record range; invalid
// points feed the sink.
template <typename T>
__global__ void __launch_bounds__(kChunkThreads)
splat_bin_scatter_kernel(const T* __restrict__ feats, int64_t m, int64_t ld, int64_t per, int channels,
                         int width, uint64_t wmagic, int ntiles, int tiles_x, SplatWs ws,
                         BinWs bw) {
  extern __shared__ uint32_t s_base[];   // [ntiles] next free record of this chunk, per tile
  const int b = blockIdx.y;
  {
    // first record of (image b, tile t) = exclusive scan of the tile totals; every workgroup
    // redoes it in LDS (a few K values) rather than waiting for a one-block scan kernel
    __shared__ uint32_t s_w[kChunkThreads / 64];
    uint32_t before = 0, base;
    for (int64_t i = threadIdx.x; i < (int64_t)b * ntiles; i += kChunkThreads)
      before += bw.tile_count[i];
    (void)block_excl_scan_u32<kChunkThreads / 64>(before, s_w, &base);
    const uint32_t* cnt = bw.tile_count + (int64_t)b * ntiles;
    const uint32_t* row = bw.hist + ((int64_t)b * gridDim.x + blockIdx.x) * ntiles;
    const int pt = ceil_div(ntiles, kChunkThreads);
    const int t0 = threadIdx.x * pt, t1 = t0 + pt < ntiles ? t0 + pt : ntiles;
    uint32_t sum = 0, all;
    for (int t = t0; t < t1; ++t) sum += cnt[t];
    uint32_t run = base + block_excl_scan_u32<kChunkThreads / 64>(sum, s_w, &all);
    for (int t = t0; t < t1; ++t) {
      s_base[t] = run + row[t];
      run += cnt[t];
    }
  }
  __syncthreads();
  constexpr int kMaxC = kMaxBinChannels;
  constexpr int kU = 4;
  uint32_t smax[kMaxC];
#pragma unroll
  for (int k = 0; k < kMaxC; ++k) smax[k] = 0u;
  const int stride = 2 + channels;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
  const int32_t* I = ws.idx + (int64_t)b * m;
  const float* Z = ws.z + (int64_t)b * m;
  const T* F = feats + (int64_t)b * ld * channels;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += kU * kChunkThreads) {
    int32_t idx[kU];
    float z[kU], f[kU][kMaxC];
    bool in[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + u * kChunkThreads;
      in[u] = i < hi;
      const int64_t j = in[u] ? i : lo;
      idx[u] = I[j];
      z[u] = Z[j];
#pragma unroll
      for (int k = 0; k < kMaxC; ++k)
        if (k < channels) f[u][k] = FeatIO<T>::load(F + j * channels + k);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (!in[u]) continue;
      if (idx[u] >= 0) {
        const int32_t pk = pack_tile(idx[u], width, wmagic, tiles_x);
        const uint32_t pos = atomicAdd(&s_base[pk >> 11], 1u);
        uint32_t* r = bw.rec + (int64_t)pos * stride;
        r[0] = (uint32_t)(pk & (kTilePx - 1));
        r[1] = __float_as_uint(z[u]);
#pragma unroll
        for (int k = 0; k < kMaxC; ++k)
          if (k < channels) r[2 + k] = __float_as_uint(f[u][k]);
      } else {
#pragma unroll
        for (int k = 0; k < kMaxC; ++k)
          if (k < channels) {
            const float v = f[u][k];
            if (v == v) {
              const uint32_t o = se3ds_f32_to_ordered(v);
              smax[k] = o > smax[k] ? o : smax[k];
            }
          }
      }
    }
  }
  __shared__ uint32_t s_f[kMaxC][kChunkThreads / 64];
#pragma unroll
  for (int k = 0; k < kMaxC; ++k) {
    uint32_t v = wave_max_u32(smax[k]);
    if ((threadIdx.x & 63) == 0) s_f[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < channels) {
    uint32_t v = 0u;
    for (int i = 0; i < kChunkThreads / 64; ++i)
      v = s_f[threadIdx.x][i] > v ? s_f[threadIdx.x][i] : v;
    ws.fpart[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * channels + threadIdx.x] = v;
  }
}

// D: one workgroup per target tile: z-min, tolerance test, per-channel max, finalize.  Records
// are read four at a time per thread (independent loads in flight; the loops are latency bound).
// With SC > 0 (channels <= SC) a thread's first kStash records stay in registers between the two
// passes, so they are read once.
constexpr int kResolveThreads = 512;
constexpr int kStash = 8;
constexpr int kStashChannels = 3;
template <bool ORDERED, int SC>
__global__ void __launch_bounds__(kResolveThreads)
splat_tile_resolve_kernel(int channels, int height, int width, int ntiles, int tiles_x, int nb,
                          uint32_t slice_records, float depth_scale, float output_void,
                          float mask_void, float* __restrict__ depth, float* __restrict__ feat,
                          float* __restrict__ mask, SplatWs ws, BinWs bw, uint32_t zpart_count) {
  extern __shared__ uint32_t s_tile[];   // z[kTilePx], feat[channels][kTilePx]
  uint32_t* s_z = s_tile;
  uint32_t* s_fe = s_tile + kTilePx;
  __shared__ uint32_t s_w[kResolveThreads / 64];
  __shared__ int s_item[3];      // tile, band, log2(bands of the tile)
  __shared__ uint32_t s_r0;
  // Work items: a tile with many records (a surface seen from farther away compresses into few
  // target pixels; 14x the mean was measured on a smooth depth map) is cut into 2 / 4 / 8 bands
  // of rows.  Every band workgroup reads all records of the tile and keeps the ones of its rows,
  // so bands never exchange anything.  item -> (tile, band) by a scan of the bands per tile.
  const uint32_t item = blockIdx.x;
  if (threadIdx.x == 0) s_item[0] = -1;
  {
    const int per = ceil_div(nb, kResolveThreads);
    const int t0 = threadIdx.x * per, t1 = t0 + per < nb ? t0 + per : nb;
    uint32_t sum_s = 0, sum_c = 0;
    for (int t = t0; t < t1; ++t) {
      const uint32_t c = bw.tile_count[t];
      sum_s += 1u << tile_bands_log2(c, slice_records);
      sum_c += c;
    }
    uint32_t tot;
    uint32_t run_s = block_excl_scan_u32<kResolveThreads / 64>(sum_s, s_w, &tot);
    uint32_t run_c = block_excl_scan_u32<kResolveThreads / 64>(sum_c, s_w, &tot);
    if (item >= run_s && item < run_s + sum_s) {
      for (int t = t0; t < t1; ++t) {
        const uint32_t c = bw.tile_count[t];
        const int lg = tile_bands_log2(c, slice_records);
        if (item < run_s + (1u << lg)) {
          s_item[0] = t;
          s_item[1] = (int)(item - run_s);
          s_item[2] = lg;
          s_r0 = run_c;
          break;
        }
        run_s += 1u << lg;
        run_c += c;
      }
    }
  }
  __syncthreads();
  const int bt = s_item[0];
  if (bt < 0) {   // beyond the last item: this row of the sink partials must still be defined
    if ((int)threadIdx.x < channels) bw.fpart2[(int64_t)item * channels + threadIdx.x] = 0u;
    return;
  }
  const int band = s_item[1], band_shift = 4 - s_item[2];   // rows per band = 16 >> log2(bands)
  const bool banded = s_item[2] != 0;
  const uint32_t tcount = bw.tile_count[bt];
  // records are packed tile after tile
  const uint32_t r0 = s_r0;
  const uint32_t r1 = r0 + tcount;
  const int b = bt / ntiles, t = bt - b * ntiles;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const uint32_t fvoid = se3ds_f32_to_ordered(output_void);
  auto mine = [&](uint32_t li) { return !banded || (int)((li >> 7) >> band_shift) == band; };
  const bool first = bt == 0 && band == 0;   // holds flat pixel 0, which also receives the sink
  uint32_t sink_o = 0xffffffffu;   // min z of the invalid points (all chunks, all images)
  if (first) {
    uint32_t v = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < zpart_count; i += kResolveThreads) {
      const uint32_t z = ws.zpart[i];
      v = z < v ? z : v;
    }
    v = wave_min_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    for (int i = 0; i < kResolveThreads / 64; ++i) sink_o = s_w[i] < sink_o ? s_w[i] : sink_o;
  }
  const bool have_sink_z = sink_o != 0xffffffffu;
  const float sink_z = se3ds_ordered_to_f32(sink_o);
  for (int p = threadIdx.x; p < kTilePx; p += kResolveThreads) s_z[p] = __float_as_uint(depth_scale);
  for (int p = threadIdx.x; p < kTilePx * channels; p += kResolveThreads) s_fe[p] = fvoid;
  __syncthreads();
  constexpr int kU = 4;
  const int stride = 2 + channels;
  constexpr int NS = SC > 0 ? kStash : 1, NC = SC > 0 ? SC : 1;
  uint32_t st_li[NS], st_z[NS], st_f[NS][NC];
  if (SC > 0) {
#pragma unroll
    for (int k0 = 0; k0 < NS; k0 += kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t q = r0 + threadIdx.x + (k0 + u) * kResolveThreads;
        const uint32_t* r = bw.rec + (int64_t)(q < r1 ? q : r0) * stride;
        const bool in = q < r1;
        st_li[k0 + u] = in ? r[0] : 0u;
        st_z[k0 + u] = in ? r[1] : 0u;
#pragma unroll
        for (int k = 0; k < NC; ++k) st_f[k0 + u][k] = (in && k < channels) ? r[2 + k] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (r0 + threadIdx.x + (k0 + u) * kResolveThreads < r1 && mine(st_li[k0 + u]))
          atomicMin(&s_z[st_li[k0 + u]], st_z[k0 + u]);
    }
  }
  const uint32_t rest = r0 + threadIdx.x + (SC > 0 ? kStash * kResolveThreads : 0);
  for (uint32_t q0 = rest; q0 < r1; q0 += kU * kResolveThreads) {
    uint32_t li[kU], zb[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint32_t q = q0 + u * kResolveThreads;
      const uint32_t* r = bw.rec + (int64_t)(q < r1 ? q : r0) * stride;
      li[u] = r[0];
      zb[u] = r[1];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (q0 + u * kResolveThreads < r1 && mine(li[u])) atomicMin(&s_z[li[u]], zb[u]);   // valid => z > 0
  }
  __syncthreads();
  constexpr int kMaxC = kMaxBinChannels;
  uint32_t smax[kMaxC];
#pragma unroll
  for (int k = 0; k < kMaxC; ++k) smax[k] = 0u;
  if (SC > 0) {
#pragma unroll
    for (int k0 = 0; k0 < NS; ++k0) {
      if (r0 + threadIdx.x + k0 * kResolveThreads >= r1 || !mine(st_li[k0])) continue;
      const uint32_t li = st_li[k0];
      const float z = __uint_as_float(st_z[k0]);
      float zm = __uint_as_float(s_z[li]);
      if (first && li == 0 && have_sink_z) zm = sink_z < zm ? sink_z : zm;
      const bool keep = z < zm + 0.1f;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < channels) {
          const float v = __uint_as_float(st_f[k0][k]);
          if (keep) {
            if (ORDERED || v > 0.0f) atomicMax(&s_fe[k * kTilePx + li], se3ds_f32_to_ordered(v));
          } else if (v == v) {
            const uint32_t o = se3ds_f32_to_ordered(v);
            smax[k] = o > smax[k] ? o : smax[k];
          }
        }
    }
  }
  for (uint32_t q0 = rest; q0 < r1; q0 += kU * kResolveThreads) {
    uint32_t li[kU], zb[kU], fb[kU][kMaxC];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint32_t q = q0 + u * kResolveThreads;
      const uint32_t* r = bw.rec + (int64_t)(q < r1 ? q : r0) * stride;
      li[u] = r[0];
      zb[u] = r[1];
#pragma unroll
      for (int k = 0; k < kMaxC; ++k)
        if (k < channels) fb[u][k] = r[2 + k];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (q0 + u * kResolveThreads >= r1 || !mine(li[u])) continue;
      const float z = __uint_as_float(zb[u]);
      float zm = __uint_as_float(s_z[li[u]]);
      if (first && li[u] == 0 && have_sink_z) zm = sink_z < zm ? sink_z : zm;
      const bool keep = z < zm + 0.1f;
#pragma unroll
      for (int k = 0; k < kMaxC; ++k)
        if (k < channels) {
          const float v = __uint_as_float(fb[u][k]);
          if (keep) {
            if (ORDERED || v > 0.0f)
              atomicMax(&s_fe[k * kTilePx + li[u]], se3ds_f32_to_ordered(v));
          } else if (v == v) {
            const uint32_t o = se3ds_f32_to_ordered(v);
            smax[k] = o > smax[k] ? o : smax[k];
          }
        }
    }
  }
  __shared__ uint32_t s_f[kMaxC][kResolveThreads / 64];
#pragma unroll
  for (int k = 0; k < kMaxC; ++k) {
    uint32_t v = wave_max_u32(smax[k]);
    if ((threadIdx.x & 63) == 0) s_f[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < channels) {
    uint32_t v = 0u;
    for (int i = 0; i < kResolveThreads / 64; ++i) v = s_f[threadIdx.x][i] > v ? s_f[threadIdx.x][i] : v;
    bw.fpart2[(int64_t)item * channels + threadIdx.x] = v;
  }
  // finalize (pixel 0's feature / mask fold of the sink happens in splat_sink_feat2_kernel)
  const int64_t hw = (int64_t)height * width;
  for (int p = threadIdx.x; p < kTilePx; p += kResolveThreads) {
    const int y = ty * kTileY + p / kTileX, x = tx * kTileX + (p % kTileX);
    if (y >= height || x >= width || !mine((uint32_t)p)) continue;
    const int64_t i = (int64_t)b * hw + (int64_t)y * width + x;
    float z = __uint_as_float(s_z[p]);
    if (i == 0 && have_sink_z) z = sink_z < z ? sink_z : z;
    float d = z < 0.0f ? 0.0f : (z > depth_scale ? depth_scale : z);
    d = d / depth_scale;
    depth[i] = d;
    bool all_ok = true;
    for (int k = 0; k < channels; ++k) {
      const float v = se3ds_ordered_to_f32(s_fe[k * kTilePx + p]);
      feat[i * channels + k] = v;
      all_ok = all_ok && (v != mask_void);
    }
    if (mask) mask[i] = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
  }
}

// sink_feat[c] = max over the per-block (invalid points) and per-tile (occluded points) partials
// ... and flat pixel 0 then folds the sink features in and redoes its mask
// (point_cloud_utils.py:151-152)
__global__ void __launch_bounds__(kBlock)
splat_sink_feat2_kernel(SplatWs ws, BinWs bw, int nparts, int nb, int channels, float* depth,
                        float* feat, float* mask, float mask_void) {
  __shared__ uint32_t s_red[kBlock / 64];
  for (int c = 0; c < channels; ++c) {
    uint32_t v = 0u;
    for (int i = threadIdx.x; i < nparts; i += kBlock) {
      uint32_t t = ws.fpart[(int64_t)i * channels + c];
      v = t > v ? t : v;
    }
    for (int i = threadIdx.x; i < nb; i += kBlock) {
      uint32_t t = bw.fpart2[(int64_t)i * channels + c];
      v = t > v ? t : v;
    }
    v = wave_max_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < kBlock / 64; ++i) v = s_red[i] > v ? s_red[i] : v;
      ws.sink_feat[c] = v;
    }
  }
  if (threadIdx.x == 0) {   // (same thread wrote sink_feat above)
    const float d = depth[0];
    bool all_ok = true;
    for (int k = 0; k < channels; ++k) {
      float v = feat[k];
      if (ws.sink_feat[k] != 0u) {
        const float sv = se3ds_ordered_to_f32(ws.sink_feat[k]);
        v = sv > v ? sv : v;
      }
      feat[k] = v;
      all_ok = all_ok && (v != mask_void);
    }
    if (mask) mask[0] = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
  }
}

template <typename T, bool EQUIRECT>
int launch_splat_binned(const float* coords, const float* offset, const T* feats, int n, int64_t m,
                        int64_t ld,
                        int channels, int height, int width, float depth_scale, float input_void,
                        float output_void, float* depth, float* feat, float* mask, float mask_void,
                        void* workspace, hipStream_t stream) {
  SplatWs ws = carve_ws(workspace, n, m);
  const size_t base = splat_hdr_bytes() + (sizeof(int32_t) + sizeof(float)) * (size_t)n * (size_t)m;
  BinWs bw = carve_bin_ws((char*)workspace + align16(base), n, m, height, width, channels);
  const int tiles_x = ceil_div(width, kTileX), tiles_y = ceil_div(height, kTileY);
  const int ntiles = tiles_x * tiles_y, nb = n * ntiles;
  const bool ordered = !(output_void >= 0.0f);
  const ChunkGeom cg = chunk_geom(m, n);
  const uint64_t wmagic = ((uint64_t)1 << 40) / (uint64_t)width + 1;
  const dim3 g_pt((unsigned)cg.chunks, (unsigned)n);
  const int nparts = cg.chunks * n;
  // three passes: count, column scan, scatter.  (Round 2's single-pass binning kernel --
  // splat_bin_fused_kernel, SE3DS_SPLAT_FUSED: count, reservation and scatter in one launch, equal on
  // a random depth map and slower on a smooth one, DESIGN.md section 3.3 -- left the library in round 5.)
  {
    hipLaunchKernelGGL((splat_bin_count_kernel<T, EQUIRECT>), g_pt, dim3(kChunkThreads), 4 * ntiles,
                       stream, coords, offset, feats, m, ld, cg.per, channels, height, width, wmagic,
                       input_void, ntiles, tiles_x, ws, bw);
    hipLaunchKernelGGL(splat_bin_colscan_kernel, dim3(ceil_div(nb, 64)), dim3(64 * kScanWaves), 0,
                       stream, bw, cg.chunks, ntiles, nb);
    hipLaunchKernelGGL((splat_bin_scatter_kernel<T>), g_pt, dim3(kChunkThreads), 4 * ntiles, stream,
                       feats, m, ld, cg.per, channels, width, wmagic, ntiles, tiles_x, ws, bw);
  }
  const size_t tile_lds = 4 * (size_t)kTilePx * (1 + channels);
  const uint32_t slice = slice_records();
  const int items = (int)resolve_items_max(nb, (int64_t)n * m, slice);
#define SE3DS_RESOLVE(ORD, SC)                                                                   \
  hipLaunchKernelGGL((splat_tile_resolve_kernel<ORD, SC>), dim3(items), dim3(kResolveThreads),   \
                     tile_lds, stream, channels, height, width, ntiles, tiles_x, nb, slice,      \
                     depth_scale, output_void, mask_void, depth, feat, mask, ws, bw,             \
                     (uint32_t)nparts)
  if (channels <= kStashChannels) {
    if (ordered) SE3DS_RESOLVE(true, kStashChannels); else SE3DS_RESOLVE(false, kStashChannels);
  } else {
    if (ordered) SE3DS_RESOLVE(true, 0); else SE3DS_RESOLVE(false, 0);
  }
#undef SE3DS_RESOLVE
  hipLaunchKernelGGL(splat_sink_feat2_kernel, dim3(1), dim3(kBlock), 0, stream, ws, bw, nparts,
                     items, channels, depth, feat, mask, mask_void);
  return check_launch("splat(binned)");
}

// ------------------------------------------------------------ packed splat (round 3, 8-byte records)
// The three-pass binned splat above moves 3x its algorithmic bytes: (idx, z) per point written and
// re-read (16 B), 20-byte 4-byte-aligned records written with five scattered dword stores and read
// back with dword loads, the features read twice.  When the features of every VALID point are
// integers in [0, 255] (uint8 features, or int32 features under the caller's SE3DS_FEAT_BYTE_RANGE
// promise: RGB memories hold [-1, 255] with void -1) and there are at most 3 channels, a record
// fits 64 bits -- 9-bit pixel inside a 16 x 32 tile, 31-bit z (valid => z > 0: the sign bit is
// free), 3 x 8-bit features -- and the passes become:
//   P1 pack + count : coordinates and features are read ONCE (16-byte loads, 4 consecutive points
//                     per thread), the fp32 screen / binary64 queue decide the pixel, and the point's
//                     final record (8 B) and tile number (2 B) are written in point order with
//                     16-byte stores; one histogram row per chunk; the invalid points' sink partials.
//   P2 column scan  : as before, and its last workgroup to finish also scans the tile totals
//                     (tile_start) and builds the resolve's item table, so no consumer rescans.
//   P3 permute      : a pure 8-byte permutation into tile order (no feature reads).
//   P4 resolve      : one 256-thread workgroup per 512-pixel tile item, 8 KB of LDS, records in
//                     registers between the z-min and the feature pass; the last workgroup to
//                     finish folds the sink into flat pixel 0 (no separate launch).
// Results are bit-identical to the other paths (min / max do not depend on the record order).
constexpr int kPTileY = 16, kPTileX = 32, kPTilePx = kPTileY * kPTileX;   // 512 px
constexpr int kPThreads = 512, kPPts = 4;
constexpr int kPGroup = kPThreads * kPPts;     // points per workgroup iteration
constexpr int kPResolveThreads = 256;
constexpr int kPStash = 4;                     // records per thread kept in registers
constexpr uint32_t kPSlice = 4096;             // records per band workgroup aimed at
constexpr int kPMaxChannels = 3;
constexpr int kSinkSlots = 64;
constexpr uint16_t kNoTile = 0xffffu;
constexpr uint16_t kPendingTile = 0xfffeu;   // undecided by the fp32 screen: the exact pass fills it in

struct PackWs {
  uint32_t* ctl;          // [8]: 0 colscan tickets, 1 promise violations, 2 items, 3 resolve tickets
  uint32_t* tile_count;   // [nb]
  uint32_t* tile_start;   // [nb + 1] exclusive prefix of tile_count (published by the permute pass)
  uint32_t* tile_lstart;  // [nb]  ... within the tile's group of 64
  uint32_t* tile_litem;   // [nb]  first resolve item of the tile within its group
  uint32_t* group_tot;    // [ngroups][2] records, items per group of 64 tiles
  uint32_t* items;        // [items_max]  tile | band << 20 | log2(bands) << 24
  uint32_t* hist;         // [n][chunks][ntiles]
  uint32_t* fpart2;       // [kSinkSlots][C] ordered max feature of the occluded points (slot = item % kSinkSlots)
  uint16_t* tile_pt;      // [n][mp]  tile of the point, kNoTile: no record
  uint64_t* rec_pt;       // [n][mp]  records in point order
  uint64_t* rec;          // [n * m]  records in tile order
  int64_t mp;             // padded points per image (chunks * per)
  uint32_t epoch;         // this call's promise epoch (kPromiseEpochWord)
};
__host__ __device__ inline ChunkGeom pack_geom(int64_t m, int n) {
  int64_t cap = kMaxSinkBlocks / (n > 0 ? n : 1);
  cap = cap < 1 ? 1 : (cap > kMaxChunks ? kMaxChunks : cap);
  int64_t chunks = (m + kChunkPoints - 1) / kChunkPoints;
  chunks = chunks < 1 ? 1 : (chunks > cap ? cap : chunks);
  int64_t per = (m + chunks - 1) / chunks;
  per = (per + kPGroup - 1) / kPGroup * kPGroup;
  if (per < kPGroup) per = kPGroup;
  chunks = (m + per - 1) / per;
  ChunkGeom g;
  g.chunks = (int)(chunks < 1 ? 1 : chunks);
  g.per = per;
  return g;
}
inline uint32_t pack_slice() {
  static const uint32_t v = [] {
    const char* e = getenv("SE3DS_SPLAT_SLICE");
    const long x = e ? atol(e) : 0;
    return (uint32_t)(x >= 16 ? x : (long)kPSlice);
  }();
  return v;
}
inline size_t pack_ws_bytes(int n, int64_t m, int height, int width) {
  const size_t ntiles = (size_t)ceil_div(height, kPTileY) * ceil_div(width, kPTileX);
  const size_t nb = (size_t)n * ntiles;
  const ChunkGeom g = pack_geom(m, n);
  const size_t mp = (size_t)g.chunks * (size_t)g.per;
  const size_t items = (size_t)resolve_items_max((int64_t)nb, (int64_t)n * (m > 0 ? m : 0), pack_slice());
  return 64 + align16(4 * nb) + align16(4 * (nb + 1)) + 2 * align16(4 * nb) +
         align16(8 * (size_t)ceil_div(nb, 64)) + align16(4 * items) +
         align16(4 * nb * (size_t)g.chunks) + align16(4 * kSinkSlots * kPMaxChannels) +
         align16(2 * (size_t)n * mp) + align16(8 * (size_t)n * mp) +
         align16(8 * (size_t)n * (size_t)(m > 0 ? m : 0));
}
inline PackWs carve_pack_ws(void* base, int n, int64_t m, int height, int width) {
  const size_t ntiles = (size_t)ceil_div(height, kPTileY) * ceil_div(width, kPTileX);
  const size_t nb = (size_t)n * ntiles;
  const ChunkGeom g = pack_geom(m, n);
  const size_t mp = (size_t)g.chunks * (size_t)g.per;
  const size_t items = (size_t)resolve_items_max((int64_t)nb, (int64_t)n * (m > 0 ? m : 0), pack_slice());
  char* p = (char*)base;
  PackWs w;
  w.ctl = (uint32_t*)p; p += 64;
  w.tile_count = (uint32_t*)p; p += align16(4 * nb);
  w.tile_start = (uint32_t*)p; p += align16(4 * (nb + 1));
  w.tile_lstart = (uint32_t*)p; p += align16(4 * nb);
  w.tile_litem = (uint32_t*)p; p += align16(4 * nb);
  w.group_tot = (uint32_t*)p; p += align16(8 * (size_t)ceil_div(nb, 64));
  w.items = (uint32_t*)p; p += align16(4 * items);
  w.hist = (uint32_t*)p; p += align16(4 * nb * (size_t)g.chunks);
  w.fpart2 = (uint32_t*)p; p += align16(4 * kSinkSlots * kPMaxChannels);
  w.tile_pt = (uint16_t*)p; p += align16(2 * (size_t)n * mp);
  w.rec_pt = (uint64_t*)p; p += align16(8 * (size_t)n * mp);
  w.rec = (uint64_t*)p;
  w.mp = (int64_t)mp;
  return w;
}

// record = [ z bits (31) | pixel bit 8 ] [ pixel bits 0-7 | f0 | f1 | f2 ]
__device__ __forceinline__ uint64_t pack_rec(uint32_t pix, float z, uint32_t f0, uint32_t f1, uint32_t f2) {
  const uint32_t lo = __float_as_uint(z) | ((pix >> 8) << 31);
  const uint32_t hi = ((pix & 255u) << 24) | (f0 << 16) | (f1 << 8) | f2;
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t rec_pix(uint64_t r) {
  return (uint32_t)(r >> 56) | (((uint32_t)r >> 31) << 8);
}
__device__ __forceinline__ uint32_t rec_zbits(uint64_t r) { return (uint32_t)r & 0x7fffffffu; }
__device__ __forceinline__ uint32_t rec_feat(uint64_t r, int k) {
  return ((uint32_t)(r >> 32) >> (16 - 8 * k)) & 255u;
}

// 4 consecutive elements of a feature array as floats (through the type's conversion) and raw
template <typename T> struct Feat4;
template <> struct Feat4<int32_t> {
  static __device__ __forceinline__ void load(const int32_t* p, bool vec, int32_t (&o)[4]) {
    if (vec) {
      const int4 v = *reinterpret_cast<const int4*>(p);
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
      for (int e = 0; e < 4; ++e) o[e] = p[e];
    }
  }
};
template <> struct Feat4<uint8_t> {
  static __device__ __forceinline__ void load(const uint8_t* p, bool vec, int32_t (&o)[4]) {
    if (vec) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
      o[0] = v & 255u; o[1] = (v >> 8) & 255u; o[2] = (v >> 16) & 255u; o[3] = v >> 24;
    } else {
      for (int e = 0; e < 4; ++e) o[e] = p[e];
    }
  }
};

// P1.  T in {int32_t, uint8_t}; C = channels (1..3).  vec: 16-byte loads are legal (ld % 4 == 0,
// aligned bases).  DEBUG also writes the (idx, z) parity tap.
template <typename T, bool EQUIRECT, int C, bool DEBUG>
__global__ void __launch_bounds__(kPThreads)
splat_pack_count_kernel(const float* __restrict__ coords, const float* __restrict__ offset,
                        const T* __restrict__ feats, int64_t m, int64_t ld, int64_t per, int height,
                        int width, uint64_t wmagic, float input_void, int ntiles, int tiles_x,
                        int vec, SplatWs ws, PackWs pw) {
  extern __shared__ uint32_t s_hist[];   // [ntiles]
  __shared__ uint32_t s_queue[kExactQueue];
  __shared__ uint32_t s_qn;
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < ntiles; t += kPThreads) s_hist[t] = 0u;
  if (threadIdx.x == 0) s_qn = 0u;
  if (blockIdx.x == 0 && b == 0) {   // state of the later passes (this kernel runs first)
    if (threadIdx.x == 0) pw.ctl[kPromiseEpochWord] = pw.epoch;
    if ((int)threadIdx.x < kSinkSlots * C) pw.fpart2[threadIdx.x] = 0u;
  }
  __syncthreads();
  const float* X = coords + (int64_t)b * 4 * ld;
  const T* F = feats + (int64_t)b * ld * C;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (EQUIRECT && offset) {
    ox = offset[b * 3 + 0];
    oy = offset[b * 3 + 1];
    oz = offset[b * 3 + 2];
  }
  uint32_t sink = 0xffffffffu;   // ordered min z of the invalid points
  uint32_t smax[C];              // ordered max feature of the invalid points, per channel
#pragma unroll
  for (int k = 0; k < C; ++k) smax[k] = 0u;
  uint32_t bad = 0u;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
  uint64_t* R = pw.rec_pt + (int64_t)b * pw.mp;
  uint16_t* TP = pw.tile_pt + (int64_t)b * pw.mp;

  // a point whose pixel (u, v) is known: histogram + record; invalid: sink partials
  auto emit = [&](bool valid, int u, int v, float pz, const int32_t (&f)[C], uint64_t* rec,
                  uint32_t* tile) {
    if (valid) {
      const int ty = v / kPTileY, tx = u / kPTileX;
      const uint32_t t = (uint32_t)(ty * tiles_x + tx);
      const uint32_t pix = (uint32_t)((v - ty * kPTileY) * kPTileX + (u - tx * kPTileX));
      atomicAdd(&s_hist[t], 1u);
      uint32_t fb[3] = {0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < C; ++k) {
        fb[k] = (uint32_t)f[k] & 255u;
        bad |= (uint32_t)f[k] >> 8;   // the byte-range promise (negative or > 255)
      }
      *rec = pack_rec(pix, pz, fb[0], fb[1], fb[2]);
      *tile = t;
    } else {
      if (pz == pz) {
        const uint32_t o = se3ds_f32_to_ordered(pz);
        sink = o < sink ? o : sink;
      }
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const uint32_t o = se3ds_f32_to_ordered((float)f[k]);
        smax[k] = o > smax[k] ? o : smax[k];
      }
      *rec = 0ull;
      *tile = kNoTile;
    }
  };
  auto feat_valid = [&](const int32_t (&f)[C]) {
    int v = 1;
#pragma unroll
    for (int k = 0; k < C; ++k) v &= ((float)f[k] != input_void);
    return v;
  };
  auto exact = [&](int64_t i, float x, float y, float z, const int32_t (&f)[C], uint64_t* rec,
                   uint32_t* tile) {
    float px, py, pz;
    if (EQUIRECT) {
      se3ds_equirect_project(x, y, z, &px, &py, &pz);
    } else {
      px = x;
      py = y;
      pz = z;
    }
    const int32_t idx = se3ds_splat_index(px, py, pz, width, height, feat_valid(f));
    int u = 0, v = 0;
    if (idx >= 0) {
      v = (int)(((uint64_t)(uint32_t)idx * wmagic) >> 40);
      u = idx - v * width;
    }
    emit(idx >= 0, u, v, pz, f, rec, tile);
    if (DEBUG) {
      ws.idx[(int64_t)b * m + i] = idx;
      ws.z[(int64_t)b * m + i] = pz;
    }
  };

  // one group of 4 consecutive points: 3 + C 16-byte loads
  auto load_group = [&](int64_t i0, float (&x)[kPPts], float (&y)[kPPts], float (&z)[kPPts],
                        int32_t (&f)[kPPts][C]) {
    const bool full = i0 + kPPts <= hi;
    if (full && vec) {
      const float4 vx = *reinterpret_cast<const float4*>(X + i0);
      const float4 vy = *reinterpret_cast<const float4*>(X + ld + i0);
      const float4 vz = *reinterpret_cast<const float4*>(X + 2 * ld + i0);
      x[0] = vx.x; x[1] = vx.y; x[2] = vx.z; x[3] = vx.w;
      y[0] = vy.x; y[1] = vy.y; y[2] = vy.z; y[3] = vy.w;
      z[0] = vz.x; z[1] = vz.y; z[2] = vz.z; z[3] = vz.w;
      int32_t flat[kPPts * C];
#pragma unroll
      for (int q = 0; q < C; ++q) {
        int32_t o[4];
        Feat4<T>::load(F + i0 * C + 4 * q, true, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) flat[4 * q + e] = o[e];
      }
#pragma unroll
      for (int p = 0; p < kPPts; ++p)
#pragma unroll
        for (int k = 0; k < C; ++k) f[p][k] = flat[p * C + k];
    } else {
#pragma unroll
      for (int p = 0; p < kPPts; ++p) {
        const int64_t i = i0 + p < hi ? i0 + p : hi - 1;   // (the tail repeats the last point)
        x[p] = X[i];
        y[p] = X[ld + i];
        z[p] = X[2 * ld + i];
#pragma unroll
        for (int k = 0; k < C; ++k) f[p][k] = (int32_t)F[i * C + k];
      }
    }
  };
  // (measured: issuing the next group's loads before working on the current one changes nothing,
  // 39.2 -> 40.3 us; two workgroups per CU already overlap their loads and their arithmetic)
  for (int64_t i0 = lo + (int64_t)threadIdx.x * kPPts; i0 < hi; i0 += kPGroup) {
    float x[kPPts], y[kPPts], z[kPPts];
    int32_t f[kPPts][C];
    load_group(i0, x, y, z, f);
    uint64_t rec[kPPts];
    uint32_t tile[kPPts];
#pragma unroll
    for (int p = 0; p < kPPts; ++p) {
      rec[p] = 0ull;
      tile[p] = kNoTile;
      if (i0 + p >= hi) continue;
      float px = x[p], py = y[p], pzz = z[p];
      if (EQUIRECT && offset) {
        px = px - ox;
        py = py - oy;
        pzz = pzz - oz;
      }
      if (EQUIRECT) {
        int u = 0, v = 0, ok = 0;
        float pz;
        if (se3ds_equirect_uv_fast(px, py, pzz, width, height, feat_valid(f[p]), &u, &v, &ok, &pz)) {
          emit(ok != 0, u, v, pz, f[p], &rec[p], &tile[p]);
          if (DEBUG) {
            ws.idx[(int64_t)b * m + i0 + p] = ok ? v * width + u : -1;
            ws.z[(int64_t)b * m + i0 + p] = pz;
          }
        } else {
          // undecided (~2.5 %): marked, and queued for the dense binary64 pass below (a divergent
          // fallback here would make nearly every wave pay the ~700-instruction chain, and inlining
          // it four times doubles the kernel's register count)
          tile[p] = kPendingTile;
          const uint32_t slot = atomicAdd(&s_qn, 1u);
          if (slot < (uint32_t)kExactQueue) s_queue[slot] = (uint32_t)(i0 + p - lo);
        }
      } else {
        exact(i0 + p, px, py, pzz, f[p], &rec[p], &tile[p]);
      }
    }
    // point-order records: 2 x 16 bytes + 8 bytes of tile numbers per thread (the arrays are
    // padded to whole groups, so the tail stores land in the padding)
    uint4* rp = reinterpret_cast<uint4*>(R + i0);
    rp[0] = make_uint4((uint32_t)rec[0], (uint32_t)(rec[0] >> 32), (uint32_t)rec[1], (uint32_t)(rec[1] >> 32));
    rp[1] = make_uint4((uint32_t)rec[2], (uint32_t)(rec[2] >> 32), (uint32_t)rec[3], (uint32_t)(rec[3] >> 32));
    *reinterpret_cast<uint2*>(TP + i0) = make_uint2(tile[0] | (tile[1] << 16), tile[2] | (tile[3] << 16));
  }
  if (EQUIRECT) {
    __syncthreads();   // (also orders the group stores above before the queue's single stores)
    // queue overflow (> 25 % undecided: lattice clouds): sweep the whole chunk for the marks
    const bool sweep = s_qn > (uint32_t)kExactQueue;
    const uint32_t qn = sweep ? (uint32_t)(hi - lo) : s_qn;
    for (uint32_t q = threadIdx.x; q < qn; q += kPThreads) {
      const int64_t i = lo + (sweep ? q : s_queue[q]);
      if (sweep && TP[i] != kPendingTile) continue;
      float x = X[i], y = X[ld + i], z = X[2 * ld + i];
      if (offset) {
        x = x - ox;
        y = y - oy;
        z = z - oz;
      }
      int32_t f[C];
#pragma unroll
      for (int k = 0; k < C; ++k) f[k] = (int32_t)F[i * C + k];
      uint64_t rec;
      uint32_t tile;
      exact(i, x, y, z, f, &rec, &tile);
      R[i] = rec;
      TP[i] = (uint16_t)tile;
    }
  }
  __shared__ uint32_t s_red[1 + C][kPThreads / 64];
  sink = wave_min_u32(sink);
  bad = wave_max_u32(bad);
  if ((threadIdx.x & 63) == 0) s_red[0][threadIdx.x >> 6] = sink;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const uint32_t v = wave_max_u32(smax[k]);
    if ((threadIdx.x & 63) == 0) s_red[1 + k][threadIdx.x >> 6] = v;
  }
  if ((threadIdx.x & 63) == 0 && bad != 0u) {
    atomicExch(&pw.ctl[1], pw.epoch);
    atomicOr(&ws.sink_z[kPromiseStickyWord], 1u);   // sticky: survives later calls (host-cleared)
  }
  __syncthreads();
  const int part = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) {
    uint32_t v = s_red[0][0];
    for (int i = 1; i < kPThreads / 64; ++i) v = s_red[0][i] < v ? s_red[0][i] : v;
    ws.zpart[part] = v;
  }
  if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + C) {
    const int k = threadIdx.x - 64;
    uint32_t v = 0u;
    for (int i = 0; i < kPThreads / 64; ++i) v = s_red[1 + k][i] > v ? s_red[1 + k][i] : v;
    ws.fpart[(int64_t)part * C + k] = v;
  }
  uint32_t* row = pw.hist + ((int64_t)b * gridDim.x + blockIdx.x) * ntiles;
  for (int t = threadIdx.x; t < ntiles; t += kPThreads) row[t] = s_hist[t];
}

// P2.  Column scan: per tile, exclusive prefix of the chunk rows (in place) and the tile total.
// One workgroup per GROUP of 64 tiles (lanes, coalesced rows); its 16 waves split the chunks.  It
// also leaves everything a consumer needs to place a tile without a scan over all tiles: the
// exclusive prefix of the totals (and of the resolve items: a tile with more than `slice` records
// is cut into 2 / 4 / 8 bands of rows) WITHIN the group, and the group's sums -- a consumer then
// scans <= a few hundred group sums in one wave.  (Measured and rejected: a one-workgroup scan
// kernel over all tiles, 6 us as its own launch; every permute workgroup redoing the full scan in
// LDS, +10 us; rows held in registers for a single read of the table, 10 -> 19 us; the last
// workgroup to finish publishing the scan behind a ticket + __threadfence -- on this multi-XCD
// part an agent-scope release is an L2 write-back per workgroup: 51 us here, 613 us in a resolve
// with the same pattern.)
__global__ void __launch_bounds__(64 * kScanWaves)
splat_pack_colscan_kernel(PackWs pw, int chunks, int ntiles, int nb, uint32_t slice) {
  __shared__ uint32_t s_sum[kScanWaves][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gt = blockIdx.x * 64 + lane;
  const bool ok = gt < nb;
  const int b = ok ? gt / ntiles : 0, t = ok ? gt - b * ntiles : 0;
  const int rows = ceil_div(chunks, kScanWaves);
  const int c0 = w * rows, c1 = c0 + rows < chunks ? c0 + rows : chunks;
  uint32_t* H = pw.hist + (int64_t)b * chunks * ntiles + t;
  uint32_t sum = 0;
  if (ok) {
#pragma unroll 8
    for (int c = c0; c < c1; ++c) sum += H[(int64_t)c * ntiles];
  }
  s_sum[w][lane] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int i = 0; i < kScanWaves; ++i) {
    const uint32_t v = s_sum[i][lane];
    if (i < w) run += v;
    total += v;
  }
  if (ok) {
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
      const uint32_t v = H[(int64_t)c * ntiles];
      H[(int64_t)c * ntiles] = run;
      run += v;
    }
  }
  if (w == 0) {
    const uint32_t cnt = ok ? total : 0u;
    const uint32_t bands = ok ? (1u << tile_bands_log2(cnt, slice)) : 0u;
    const uint32_t ic = wave_incl_scan_u32(cnt), is = wave_incl_scan_u32(bands);
    if (ok) {
      pw.tile_count[gt] = cnt;
      pw.tile_lstart[gt] = ic - cnt;
      pw.tile_litem[gt] = is - bands;
    }
    if (lane == 63) {
      pw.group_tot[2 * blockIdx.x] = ic;
      pw.group_tot[2 * blockIdx.x + 1] = is;
    }
  }
}

// exclusive prefix of the group sums (records, items) for group g, from one wave's scan over all
// groups; s_g: [2][ngroups + 1] words of LDS, filled by wave 0 and published by a barrier
__device__ __forceinline__ void scan_groups(const PackWs& pw, int ngroups, uint32_t* s_g) {
  if (threadIdx.x < 64) {
    uint32_t run_c = 0, run_s = 0;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
      const int g = g0 + threadIdx.x;
      const uint32_t c = g < ngroups ? pw.group_tot[2 * g] : 0u;
      const uint32_t sb = g < ngroups ? pw.group_tot[2 * g + 1] : 0u;
      const uint32_t ic = wave_incl_scan_u32(c), is = wave_incl_scan_u32(sb);
      if (g < ngroups) {
        s_g[g] = run_c + ic - c;
        s_g[ngroups + 1 + g] = run_s + is - sb;
      }
      run_c += (uint32_t)__shfl((int)ic, 63, 64);
      run_s += (uint32_t)__shfl((int)is, 63, 64);
    }
    if (threadIdx.x == 0) {
      s_g[ngroups] = run_c;
      s_g[2 * ngroups + 1] = run_s;
    }
  }
  __syncthreads();
}

// P3.  Pure permutation: point-order records -> tile order.  The first workgroup of every image
// also publishes tile_start and the resolve's item table.
__global__ void __launch_bounds__(kPThreads)
splat_pack_permute_kernel(int64_t m, int64_t per, int ntiles, int nimages, PackWs pw) {
  extern __shared__ uint32_t s_dyn[];   // [ntiles] next free record of this chunk, per tile; group bases
  uint32_t* s_base = s_dyn;
  const int nb = nimages * ntiles, ngroups = (int)ceil_div(nb, 64);
  uint32_t* s_g = s_dyn + ntiles;       // [2][ngroups + 1]
  const int b = blockIdx.y;
  scan_groups(pw, ngroups, s_g);
  {
    const uint32_t* row = pw.hist + ((int64_t)b * gridDim.x + blockIdx.x) * ntiles;
    const bool publish = blockIdx.x == 0;
    for (int t = threadIdx.x; t < ntiles; t += kPThreads) {
      const int gt = b * ntiles + t;
      const uint32_t start = s_g[gt >> 6] + pw.tile_lstart[gt];
      s_base[t] = start + row[t];
      if (publish) {
        pw.tile_start[gt] = start;
      }
    }
    if (publish) {
      // items of this image's tiles (bands per tile from the difference of consecutive first items)
      for (int t = threadIdx.x; t < ntiles; t += kPThreads) {
        const int gt = b * ntiles + t;
        const uint32_t first = s_g[ngroups + 1 + (gt >> 6)] + pw.tile_litem[gt];
        const uint32_t next = gt + 1 < nb ? s_g[ngroups + 1 + ((gt + 1) >> 6)] + pw.tile_litem[gt + 1]
                                          : s_g[2 * ngroups + 1];
        const uint32_t bands = next - first;
        uint32_t lg = 0;
        while ((1u << lg) < bands) ++lg;
        for (uint32_t band = 0; band < bands; ++band)
          pw.items[first + band] = (uint32_t)gt | (band << 20) | (lg << 24);
      }
      if (b == nimages - 1 && threadIdx.x == 0) {
        pw.tile_start[nb] = s_g[ngroups];
        pw.ctl[2] = s_g[2 * ngroups + 1];
      }
    }
  }
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
  const uint64_t* R = pw.rec_pt + (int64_t)b * pw.mp;
  const uint16_t* TP = pw.tile_pt + (int64_t)b * pw.mp;
  constexpr int kU = 2;   // two groups of 4 points in flight per thread
  for (int64_t i0 = lo + (int64_t)threadIdx.x * kPPts; i0 < hi; i0 += kU * kPGroup) {
    uint4 ra[kU], rb[kU];
    uint2 tl[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * kPGroup;
      const int64_t j = i < hi ? i : lo;   // (padded arrays: whole groups are readable)
      const uint4* rp = reinterpret_cast<const uint4*>(R + j);
      ra[u] = rp[0];
      rb[u] = rp[1];
      tl[u] = *reinterpret_cast<const uint2*>(TP + j);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (i0 + (int64_t)u * kPGroup >= hi) continue;
      const uint32_t t[4] = {tl[u].x & 0xffffu, tl[u].x >> 16, tl[u].y & 0xffffu, tl[u].y >> 16};
      const uint64_t r[4] = {((uint64_t)ra[u].y << 32) | ra[u].x, ((uint64_t)ra[u].w << 32) | ra[u].z,
                             ((uint64_t)rb[u].y << 32) | rb[u].x, ((uint64_t)rb[u].w << 32) | rb[u].z};
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (t[p] != kNoTile) pw.rec[atomicAdd(&s_base[t[p]], 1u)] = r[p];
    }
  }
}

// P4.  One workgroup per resolve item (tile, band of rows).
template <int C>
__global__ void __launch_bounds__(kPResolveThreads)
splat_pack_resolve_kernel(int height, int width, int ntiles, int tiles_x, float depth_scale,
                          float output_void, float mask_void, float* __restrict__ depth,
                          float* __restrict__ feat, float* __restrict__ mask, SplatWs ws, PackWs pw,
                          uint32_t zpart_count) {
  __shared__ uint32_t s_z[kPTilePx];
  __shared__ uint32_t s_fe[C][kPTilePx];
  __shared__ uint32_t s_w[kPResolveThreads / 64];
  __shared__ uint32_t s_c[C][kPResolveThreads / 64];
  const uint32_t item = blockIdx.x;
  const uint32_t nitems = pw.ctl[2];
  if (item < nitems) {
    const uint32_t it = pw.items[item];
    const int bt = (int)(it & 0xfffffu), band = (int)((it >> 20) & 15u), lg = (int)(it >> 24);
    const int band_shift = 4 - lg;   // rows per band = 16 >> log2(bands)
    const bool banded = lg != 0;
    const uint32_t r0 = pw.tile_start[bt], r1 = pw.tile_start[bt + 1];
    const int b = bt / ntiles, t = bt - b * ntiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    auto mine = [&](uint32_t li) { return !banded || (int)((li >> 5) >> band_shift) == band; };
    const bool first = bt == 0 && band == 0;   // holds flat pixel 0, which also receives the sink
    uint32_t sink_o = 0xffffffffu;
    if (first) {
      uint32_t v = 0xffffffffu;
      for (uint32_t i = threadIdx.x; i < zpart_count; i += kPResolveThreads) {
        const uint32_t z = ws.zpart[i];
        v = z < v ? z : v;
      }
      v = wave_min_u32(v);
      if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
      __syncthreads();
      for (int i = 0; i < kPResolveThreads / 64; ++i) sink_o = s_w[i] < sink_o ? s_w[i] : sink_o;
    }
    const bool have_sink_z = sink_o != 0xffffffffu;
    const float sink_z = se3ds_ordered_to_f32(sink_o);
    for (int p = threadIdx.x; p < kPTilePx; p += kPResolveThreads) {
      s_z[p] = __float_as_uint(depth_scale);
#pragma unroll
      for (int k = 0; k < C; ++k) s_fe[k][p] = 0u;
    }
    __syncthreads();
    // pass A: z-min; a thread's first kPStash records stay in registers for pass B
    uint64_t st[kPStash];
#pragma unroll
    for (int k = 0; k < kPStash; ++k) {
      const uint32_t q = r0 + threadIdx.x + k * kPResolveThreads;
      st[k] = q < r1 ? pw.rec[q] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < kPStash; ++k) {
      const uint32_t q = r0 + threadIdx.x + k * kPResolveThreads;
      const uint32_t li = rec_pix(st[k]);
      if (q < r1 && mine(li)) atomicMin(&s_z[li], rec_zbits(st[k]));   // valid => z > 0
    }
    const uint32_t rest = r0 + threadIdx.x + kPStash * kPResolveThreads;
    for (uint32_t q0 = rest; q0 < r1; q0 += 4 * kPResolveThreads) {
      uint64_t r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = q0 + u * kPResolveThreads;
        r[u] = q < r1 ? pw.rec[q] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t li = rec_pix(r[u]);
        if (q0 + u * kPResolveThreads < r1 && mine(li)) atomicMin(&s_z[li], rec_zbits(r[u]));
      }
    }
    __syncthreads();
    // pass B: survivors (z < zmin + 0.1) max their features; the rest feed the sink
    uint32_t smax[C];
#pragma unroll
    for (int k = 0; k < C; ++k) smax[k] = 0u;
    auto passb = [&](uint64_t r) {
      const uint32_t li = rec_pix(r);
      if (!mine(li)) return;
      const float z = __uint_as_float(rec_zbits(r));
      float zm = __uint_as_float(s_z[li]);
      if (first && li == 0 && have_sink_z) zm = sink_z < zm ? sink_z : zm;
      const bool keep = z < zm + 0.1f;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const uint32_t fv = rec_feat(r, k);
        if (keep) {
          if (fv != 0u) atomicMax(&s_fe[k][li], fv);
        } else {
          const uint32_t o = se3ds_f32_to_ordered((float)fv);
          smax[k] = o > smax[k] ? o : smax[k];
        }
      }
    };
#pragma unroll
    for (int k = 0; k < kPStash; ++k)
      if (r0 + threadIdx.x + k * kPResolveThreads < r1) passb(st[k]);
    for (uint32_t q0 = rest; q0 < r1; q0 += 4 * kPResolveThreads) {
      uint64_t r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = q0 + u * kPResolveThreads;
        r[u] = q < r1 ? pw.rec[q] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u * kPResolveThreads < r1) passb(r[u]);
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const uint32_t v = wave_max_u32(smax[k]);
      if ((threadIdx.x & 63) == 0) s_c[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
      uint32_t v = 0u;
      for (int i = 0; i < kPResolveThreads / 64; ++i) v = s_c[threadIdx.x][i] > v ? s_c[threadIdx.x][i] : v;
      // a few thousand atomics spread over kSinkSlots x C words (one word per channel would
      // serialise at ~12 ns each; a row per item costs the fold kernel 12 K loads)
      if (v != 0u) atomicMax(&pw.fpart2[(item % kSinkSlots) * C + threadIdx.x], v);
    }
    // finalize this item's pixels: a thread owns 4 consecutive pixels of a row (16-byte stores of
    // depth, mask and the 4 C feature floats) when the outputs allow it
    const int64_t hw = (int64_t)height * width;
    auto pixel = [&](int p, int64_t i, float* d_out, float (&f_out)[C], float* m_out) {
      float z = __uint_as_float(s_z[p]);
      if (i == 0 && have_sink_z) z = sink_z < z ? sink_z : z;
      float d = z < 0.0f ? 0.0f : (z > depth_scale ? depth_scale : z);
      d = d / depth_scale;
      bool all_ok = true;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        // scatter_max over fill(output_void >= 0): a feature of 0 and "no survivor" coincide
        const float fv = (float)s_fe[k][p];
        const float v = fv > output_void ? fv : output_void;
        f_out[k] = v;
        all_ok = all_ok && (v != mask_void);
      }
      *d_out = d;
      *m_out = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
    };
    // (16-byte output stores, 4 pixels per thread, measured 2 % slower than dword stores in round 3:
    // 100.3 vs 98.2 us per render on one box)
    for (int p = threadIdx.x; p < kPTilePx; p += kPResolveThreads) {
      const int y = ty * kPTileY + p / kPTileX, x = tx * kPTileX + (p % kPTileX);
      if (y >= height || x >= width || !mine((uint32_t)p)) continue;
      const int64_t i = (int64_t)b * hw + (int64_t)y * width + x;
      float d, mk, f[C];
      pixel(p, i, &d, f, &mk);
      depth[i] = d;
#pragma unroll
      for (int k = 0; k < C; ++k) feat[i * C + k] = f[k];
      if (mask) mask[i] = mk;
    }
  }
}

// P5 (one workgroup).  Folds the sink features into flat pixel 0 (point_cloud_utils.py:151-152:
// invalid and occluded points scatter into flat index 0) and redoes its mask.
template <int C>
__global__ void __launch_bounds__(1024)
splat_pack_sink_kernel(float* __restrict__ depth, float* __restrict__ feat, float* __restrict__ mask,
                       float mask_void, SplatWs ws, PackWs pw, uint32_t zpart_count) {
  __shared__ uint32_t s_c[C][16];
  uint32_t fm[C];
#pragma unroll
  for (int k = 0; k < C; ++k) fm[k] = 0u;
  for (uint32_t i = threadIdx.x; i < zpart_count * C; i += 1024) {
    const uint32_t v = ws.fpart[i];
    const int k = i % C;
#pragma unroll
    for (int kk = 0; kk < C; ++kk)
      if (kk == k) fm[kk] = v > fm[kk] ? v : fm[kk];
  }
  for (uint32_t i = threadIdx.x; i < (uint32_t)kSinkSlots * C; i += 1024) {
    const uint32_t v = pw.fpart2[i];
    const int k = i % C;
#pragma unroll
    for (int kk = 0; kk < C; ++kk)
      if (kk == k) fm[kk] = v > fm[kk] ? v : fm[kk];
  }
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const uint32_t v = wave_max_u32(fm[k]);
    if ((threadIdx.x & 63) == 0) s_c[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float d = depth[0];
    bool all_ok = true;
    for (int k = 0; k < C; ++k) {
      uint32_t sv = 0u;
      for (int i = 0; i < 16; ++i) sv = s_c[k][i] > sv ? s_c[k][i] : sv;
      float v = feat[k];
      if (sv != 0u) {
        const float sf = se3ds_ordered_to_f32(sv);
        v = sf > v ? sf : v;
      }
      feat[k] = v;
      all_ok = all_ok && (v != mask_void);
    }
    if (mask) mask[0] = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
feats_byte_range_kernel(const T* __restrict__ f, int64_t count, float void_class,
                        uint32_t* __restrict__ bad_out) {
  uint32_t bad = 0u;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * kBlock) {
    const int32_t v = (int32_t)f[i];
    bad += ((float)v != void_class && ((uint32_t)v >> 8) != 0u) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += (uint32_t)__shfl_xor((int)bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad != 0u) atomicAdd(bad_out, bad);
}

__global__ void splat_promise_kernel(PackWs pw, uint32_t* out) {
  const uint32_t e = pw.ctl[kPromiseEpochWord];
  out[0] = (pw.ctl[1] == e && (e & kEpochTagMask) == kEpochTag) ? 1u : 0u;
}
__global__ void splat_promise_sticky_kernel(uint32_t* hdr, uint32_t* out, int clear) {
  out[0] = hdr[kPromiseStickyWord] != 0u ? 1u : 0u;
  if (clear) hdr[kPromiseStickyWord] = 0u;
}

template <typename T, bool EQUIRECT>
int launch_splat_packed(const float* coords, const float* offset, const T* feats, int n, int64_t m,
                        int64_t ld, int channels, int height, int width, float depth_scale,
                        float input_void, float output_void, float* depth, float* feat,
                        float* mask, float mask_void, void* workspace, hipStream_t stream) {
  SplatWs ws = carve_ws(workspace, n, m);
  const size_t base = splat_hdr_bytes() + (sizeof(int32_t) + sizeof(float)) * (size_t)n * (size_t)m;
  PackWs pw = carve_pack_ws((char*)workspace + align16(base), n, m, height, width);
  pw.epoch = next_splat_epoch();
  const int tiles_x = ceil_div(width, kPTileX), tiles_y = ceil_div(height, kPTileY);
  const int ntiles = tiles_x * tiles_y, nb = n * ntiles;
  const ChunkGeom cg = pack_geom(m, n);
  const uint64_t wmagic = ((uint64_t)1 << 40) / (uint64_t)width + 1;
  const dim3 g_pt((unsigned)cg.chunks, (unsigned)n);
  const int nparts = cg.chunks * n;
  const uint32_t slice = pack_slice();
  const int items = (int)resolve_items_max(nb, (int64_t)n * m, slice);
  const int vec = (ld % 4 == 0) && ((uintptr_t)coords % 16 == 0) &&
                  ((uintptr_t)feats % (sizeof(T) == 1 ? 4 : 16) == 0);
  // (read per call: the parity tests switch the tap on for single calls)
  const char* e_dbg = getenv("SE3DS_SPLAT_DEBUG");
  const bool dbg = e_dbg && atoi(e_dbg) != 0;
#define SE3DS_P1(CC, DBG)                                                                         \
  hipLaunchKernelGGL((splat_pack_count_kernel<T, EQUIRECT, CC, DBG>), g_pt, dim3(kPThreads),      \
                     4 * ntiles, stream, coords, offset, feats, m, ld, cg.per, height, width,     \
                     wmagic, input_void, ntiles, tiles_x, vec, ws, pw)
  switch (channels) {
    case 1: if (dbg) SE3DS_P1(1, true); else SE3DS_P1(1, false); break;
    case 2: if (dbg) SE3DS_P1(2, true); else SE3DS_P1(2, false); break;
    default: if (dbg) SE3DS_P1(3, true); else SE3DS_P1(3, false); break;
  }
#undef SE3DS_P1
  hipLaunchKernelGGL(splat_pack_colscan_kernel, dim3(ceil_div(nb, 64)), dim3(64 * kScanWaves), 0,
                     stream, pw, cg.chunks, ntiles, nb, slice);
  hipLaunchKernelGGL(splat_pack_permute_kernel, g_pt, dim3(kPThreads),
                     4 * (ntiles + 2 * ((int)ceil_div(nb, 64) + 1)), stream, m, cg.per, ntiles, n, pw);
#define SE3DS_P4(CC)                                                                              \
  hipLaunchKernelGGL((splat_pack_resolve_kernel<CC>), dim3(items), dim3(kPResolveThreads), 0,     \
                     stream, height, width, ntiles, tiles_x, depth_scale, output_void, mask_void,  \
                     depth, feat, mask, ws, pw, (uint32_t)nparts);                        \
  hipLaunchKernelGGL((splat_pack_sink_kernel<CC>), dim3(1), dim3(1024), 0, stream, depth, feat,   \
                     mask, mask_void, ws, pw, (uint32_t)nparts)
  switch (channels) {
    case 1: SE3DS_P4(1); break;
    case 2: SE3DS_P4(2); break;
    default: SE3DS_P4(3); break;
  }
#undef SE3DS_P4
  return check_launch("splat(packed)");
}

// ------------------------------------------------------ sorted-chunk splat (round 4, two kernels)
// The packed path above still moves 2.4x its algorithmic bytes (PMC): the point-order round trip of
// the records (10 B per point out of P1, back into P3), an 8 MB chunk x tile histogram that P2 reads
// and rewrites, a scattered 8-byte permutation, a separate sink launch.  Here the permutation never
// touches HBM:
//   S1 sort    : as P1 (coordinates + features read once, fp32 screen / dense binary64 queue), but a
//                workgroup keeps its chunk's records in REGISTERS until the chunk's histogram over
//                the target SUPERTILES (128 x kSY pixels) is complete, places them in LDS in
//                supertile order and writes the chunk out with full-line stores: 8-byte records
//                (9-bit pixel, 31-bit z, 3 x 8-bit features, the format above) + one byte with the
//                pixel's upper bits (a supertile has 4096 / 8192 pixels), and one run descriptor
//                (offset, count) per (supertile, chunk).
//   S2 resolve : one workgroup per supertile GATHERS its runs from all chunks (a flat record index
//                is mapped to (chunk, offset) by a binary search over the run prefix in LDS, so
//                consecutive lanes read consecutive records whatever the run lengths are), z-min
//                and byte-wise feature max in LDS (one packed word per pixel, CAS), outputs
//                written directly.  A supertile with more than `slice` records is cut into 2 / 4 / 8
//                bands of rows (every band reads all runs and keeps its rows -- bit-identical).  The
//                sink (flat pixel 0 collects every invalid and occluded point) is folded by the LAST
//                workgroup to finish: every workgroup ends with one ticket atomic behind its sink
//                atomics; pixel 0 is published through atomics by its owner and written only by the
//                last workgroup (no fences, no spinning: returned device-scope atomics are performed
//                memory-side before the ticket is taken).
// Traffic per cfg5 render: 100 MB in + 38 MB of records out, 38 MB in + 42 MB out = 218 MB for 159 MB
// algorithmic (the packed path: ~310 nominal, 382 measured).  min / max do not depend on the record
// order: bit-identical to every other path.
constexpr int kSMaxPx = 4096;            // pixels per supertile, upper bound (LDS tile of the resolve: 2 words per pixel)
constexpr int kSDefaultPx = 2048;        // ... default: one full row of a 1024 x 2048 target
constexpr int kSThreads = 512;           // S1 workgroup
constexpr int kSQueue = 1024;            // S1: points waiting for the binary64 path (2 per thread)
constexpr int kSMaxSuper = 2048;         // supertiles per image (S1 LDS histogram, 11-bit field)
constexpr int kSMaxChunks = 4096;        // chunks per image (S2 LDS run prefix)
constexpr int kRThreads = 512;           // S2 workgroup
constexpr int kRStash = 12;              // S2: records per thread and batch (6 144 per workgroup: an average
                                         // supertile of cfg5 holds 4 096, its records stay in registers)
constexpr uint32_t kSMetaNone = 0xffffffffu, kSMetaPending = 0xfffffffeu;

struct SortWs {
  uint32_t* ctl;      // [16]: 0 ticket, 1 promise violations (same slot as PackWs), 4.. pixel 0 (z, C feature words)
  uint32_t* fpart2;   // [kSinkSlots][C] ordered max feature of the occluded points
  uint32_t* runs;     // [n * nsuper][chunks]  offset << 16 | count
  uint64_t* rec;      // [n * chunks][chunk_pts]
  uint8_t* hi;        // [n * chunks][chunk_pts]
  int chunks;         // per image
  int chunk_pts;
  int cstride;        // row length of `runs`: 8 * ceil(chunks / 8), see run_slot()
  uint32_t epoch;     // this call's promise epoch (kPromiseEpochWord)
};
// Column of chunk c in a supertile's row of the run table.  A chunk is one S1 workgroup and
// workgroup b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md; used for speed only): with
// the chunks of one XCD next to each other, the 4-byte descriptors that XCD's workgroups write into
// one supertile row fill whole lines of THAT XCD's L2 before they are written back.  Round 4 stored
// column c: every 128-byte line of the table was assembled from partial write-backs of all eight
// L2s (S1 wrote 51 MB for 39 MB of payload).
__host__ __device__ inline int run_slot(int c, int cstride) { return (c & 7) * (cstride >> 3) + (c >> 3); }
__host__ __device__ inline int run_chunk(int slot, int cstride) {
  const int per = cstride >> 3;
  return (slot % per) * 8 + slot / per;
}
// Supertile = `rows` FULL-WIDTH image rows (rows x width <= 2048 pixels; images wider than 2048 are
// cut into column strips).  The shape matters: the pole rows of every SOURCE view (thousands of
// pixels looking the same way) land on a near-vertical line of the target, and 32 x 128 supertiles
// put 30 000 (random depth) to 86 000 (smooth room) records into the supertiles on that line against
// a mean of 8 000 -- one workgroup then runs 4-10x longer than the rest.  A line crosses a row
// once: full-row supertiles hold at most 1.35x (random) / 2.3x (room) the mean.
struct SortGeom {
  int pts, chunk_pts, chunks, super_w, rlog, super_x, super_y, nsuper, px;
};
inline int sort_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// Geometry of one call.  Both kernels want TWO workgroups per CU (S1 is VALU- and S2 latency-bound: at one
// 8-wave workgroup per CU every barrier and every load round trip is exposed).  Round 5 sized both for
// the 4.2 M points of cfg5; at north_star's 512 x 1024 with two views that left 256 chunks and 256
// supertiles for 256 CUs and each kernel took 16.8 us for a quarter of the work that takes 34-39 us at
// 1024 x 2048 (profiles/r06_warp_512_*).  Now: points per thread = the largest of 16 / 8 / 4 that still
// gives kSTargetWgs chunks, pixels per supertile = the largest of 2048 / 1024 / 512 that gives as
// many supertiles (full image rows first, see above).
constexpr int kSTargetWgs = 512;
inline int sort_super_count(int height, int width, int super_px, int* super_w_out, int* rlog_out) {
  int super_w = width <= super_px ? width : super_px;
  if (width > super_w) {   // column strips of equal width (the last one may be ragged)
    const int strips = ceil_div(width, super_w);
    super_w = ceil_div(width, strips);
  }
  int rlog = 0;
  while ((2 << rlog) * super_w <= super_px && (1 << rlog) < height) ++rlog;
  if (super_w_out) *super_w_out = super_w;
  if (rlog_out) *rlog_out = rlog;
  return ceil_div(width, super_w) * ceil_div(height, 1 << rlog);
}
inline SortGeom sort_geom(int n, int64_t m, int height, int width, int pts_forced = 0, int px_forced = 0) {
  // SE3DS_SPLAT_PTS forces 4 / 8 / 16 points per thread, SE3DS_SPLAT_SUPERPX 256 .. 4096 pixels per
  // supertile (A/B runs and the parity tests' small images)
  static const int pts_forced_env = sort_env("SE3DS_SPLAT_PTS", 0);
  static const int px_forced_env = [] {
    const int v = sort_env("SE3DS_SPLAT_SUPERPX", 0);
    return v >= 256 && v <= kSMaxPx ? v : 0;
  }();
  const int64_t mm = m > 0 ? m : 1;
  const int nn = n > 0 ? n : 1;
  SortGeom g;
  g.pts = pts_forced ? pts_forced
                     : (pts_forced_env == 4 || pts_forced_env == 8 || pts_forced_env == 16) ? pts_forced_env : 0;
  if (!g.pts) {
    g.pts = 4;
    for (int pts = 16; pts > 4; pts >>= 1)
      if (ceil_div(mm, (int64_t)kSThreads * pts) * nn >= kSTargetWgs) {
        g.pts = pts;
        break;
      }
  }
  g.chunk_pts = kSThreads * g.pts;
  g.chunks = (int)ceil_div(mm, (int64_t)g.chunk_pts);
  int super_px = px_forced ? px_forced : px_forced_env;
  if (!super_px) {
    super_px = kSDefaultPx;
    while (super_px > 512 && (int64_t)nn * sort_super_count(height, width, super_px, nullptr, nullptr) < kSTargetWgs)
      super_px >>= 1;
  }
  g.nsuper = sort_super_count(height, width, super_px, &g.super_w, &g.rlog);
  g.super_x = ceil_div(width, g.super_w);
  g.super_y = ceil_div(height, 1 << g.rlog);
  g.px = g.super_w << g.rlog;
  return g;
}
inline size_t sort_ws_bytes(int n, int64_t m, int height, int width) {
  // (independent of the A/B environment switches: the worst case over every geometry a call may take)
  size_t worst = 0;
  for (int pts = 4; pts <= 16; pts <<= 1) {
    const SortGeom g = sort_geom(n, m, height, width, pts, 256);   // (256-pixel supertiles: the longest run table)
    const size_t b = 64 + align16(4 * kSinkSlots * kPMaxChannels) +
                     align16(4 * (size_t)n * kSMaxSuper * (size_t)(8 * ceil_div(g.chunks, 8))) +
                     align16(8 * (size_t)n * g.chunks * g.chunk_pts) +
                     align16((size_t)n * g.chunks * g.chunk_pts);
    worst = b > worst ? b : worst;
  }
  return worst;
}
inline SortWs carve_sort_ws(void* base, int n, const SortGeom& g) {
  char* p = (char*)base;
  SortWs w;
  w.ctl = (uint32_t*)p; p += 64;
  w.fpart2 = (uint32_t*)p; p += align16(4 * kSinkSlots * kPMaxChannels);
  w.cstride = 8 * ceil_div(g.chunks, 8);
  w.runs = (uint32_t*)p; p += align16(4 * (size_t)n * g.nsuper * w.cstride);
  w.rec = (uint64_t*)p; p += align16(8 * (size_t)n * g.chunks * g.chunk_pts);
  w.hi = (uint8_t*)p;
  w.chunks = g.chunks;
  w.chunk_pts = g.chunk_pts;
  w.epoch = 0u;
  return w;
}

#ifdef SE3DS_PROBE
// -DSE3DS_PROBE builds only (tools/warp_phases.py; never the shipped library): every workgroup of the two
// sorted-splat kernels stamps the 100 MHz wall clock at its phase boundaries.
constexpr int kProbeWgs = 4096, kProbeStamps = 12;
__device__ uint64_t g_probe[2][kProbeWgs][kProbeStamps];
#define SE3DS_STAMP(K, I)                                                                       \
  do {                                                                                          \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                   \
    if (threadIdx.x == 0 && wg_ < (unsigned)kProbeWgs) g_probe[K][wg_][I] = wall_clock64();     \
  } while (0)
#else
#define SE3DS_STAMP(K, I) do {} while (0)
#endif

// S1.  PTS points per thread (4, 8 or 16).  Supertile of pixel (u, v): (v >> rlog, u / super_w).
template <typename T, bool EQUIRECT, int C, bool DEBUG, int PTS>
__global__ void __launch_bounds__(kSThreads, 4)   // (<= 128 VGPRs: two workgroups per CU)
splat_sort_kernel(const float* __restrict__ coords, const float* __restrict__ offset,
                  const T* __restrict__ feats, int64_t m, int64_t ld, int height, int width,
                  uint64_t wmagic, float input_void, int nsuper, int super_x, int super_w, int rlog,
                  int vec, SplatWs ws, SortWs sw) {
  constexpr int kChunk = kSThreads * PTS;
  constexpr int kIter = PTS / kPPts;
  // queue of the undecided points: at most a quarter of the chunk (more: every thread takes its own),
  // coordinates / features / offsets parked in the record area, which is only written in phase 4
  constexpr int kQCap = kSQueue < kChunk / 4 ? kSQueue : kChunk / 4;
  constexpr int kQPer = (kQCap + kSThreads - 1) / kSThreads;
  static_assert((size_t)kQCap * (12 + 4 * C) <= (size_t)kChunk * 8, "the queue lives in the record area");
  extern __shared__ uint64_t s_dyn64[];          // [kChunk] records, [kChunk] hi bytes, counts, offsets
  uint64_t* s_rec = s_dyn64;
  uint8_t* s_hi = reinterpret_cast<uint8_t*>(s_dyn64 + kChunk);
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_hi + kChunk);   // [nsuper] counts, then offsets
  __shared__ uint16_t s_queue[kSQueue];   // (point offsets inside the chunk: < 8192)
  float* s_qxyz = reinterpret_cast<float*>(s_rec);                      // [3][kQCap] camera-relative x, y, z
  int32_t* s_qf = reinterpret_cast<int32_t*>(s_rec) + 3 * kQCap;        // [C][kQCap] features
  __shared__ uint32_t s_qn;
  __shared__ uint32_t s_w[kSThreads / 64];
  SE3DS_STAMP(0, 0);   // start
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < nsuper; t += kSThreads) s_cnt[t] = 0u;
  if (threadIdx.x == 0) s_qn = 0u;
  if (blockIdx.x == 0 && b == 0) {   // state of the resolve pass (this kernel runs first)
    // (not ctl[1]: other workgroups of THIS launch exchange the epoch into it, see kPromiseEpochWord)
    if (threadIdx.x < 16 && threadIdx.x != 1)
      sw.ctl[threadIdx.x] = (int)threadIdx.x == kPromiseEpochWord ? sw.epoch : 0u;
    if ((int)threadIdx.x < kSinkSlots * C) sw.fpart2[threadIdx.x] = 0u;
  }
  __syncthreads();
  const float* X = coords + (int64_t)b * 4 * ld;
  const T* F = feats + (int64_t)b * ld * C;
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (EQUIRECT && offset) {
    ox = offset[b * 3 + 0];
    oy = offset[b * 3 + 1];
    oz = offset[b * 3 + 2];
  }
  uint32_t sink = 0xffffffffu;
  uint32_t smax[C];
#pragma unroll
  for (int k = 0; k < C; ++k) smax[k] = 0u;
  uint32_t bad = 0u;
  const int64_t lo = (int64_t)blockIdx.x * kChunk, hi = lo + kChunk < m ? lo + kChunk : m;

  // a point whose pixel (u, v) is known: histogram rank + record; invalid: sink partials
  auto emit = [&](bool valid, int u, int v, float pz, const int32_t (&f)[C], uint64_t* rec,
                  uint32_t* meta) {
    if (valid) {
      const int sy_ = v >> rlog, sx_ = super_x == 1 ? 0 : u / super_w;
      const uint32_t st = (uint32_t)(sy_ * super_x + sx_);
      const uint32_t pix = (uint32_t)((v - (sy_ << rlog)) * super_w + (u - sx_ * super_w));
      const uint32_t rank = atomicAdd(&s_cnt[st], 1u);
      uint32_t fb[3] = {0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < C; ++k) {
        fb[k] = (uint32_t)f[k] & 255u;
        bad |= (uint32_t)f[k] >> 8;   // the byte-range promise (negative or > 255)
      }
      *rec = pack_rec(pix & 511u, pz, fb[0], fb[1], fb[2]);
      *meta = st | (rank << 11) | ((pix >> 9) << 25);   // 11-bit supertile, 14-bit rank, pixel bits 9..
    } else {
      if (pz == pz) {
        const uint32_t o = se3ds_f32_to_ordered(pz);
        sink = o < sink ? o : sink;
      }
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const uint32_t o = se3ds_f32_to_ordered((float)f[k]);
        smax[k] = o > smax[k] ? o : smax[k];
      }
      *rec = 0ull;
      *meta = kSMetaNone;
    }
  };
  auto feat_valid = [&](const int32_t (&f)[C]) {
    int v = 1;
#pragma unroll
    for (int k = 0; k < C; ++k) v &= ((float)f[k] != input_void);
    return v;
  };
  auto exact = [&](int64_t i, float x, float y, float z, const int32_t (&f)[C], uint64_t* rec,
                   uint32_t* meta) {
    float px, py, pz;
    if (EQUIRECT) {
      se3ds_equirect_project(x, y, z, &px, &py, &pz);
    } else {
      px = x;
      py = y;
      pz = z;
    }
    const int32_t idx = se3ds_splat_index(px, py, pz, width, height, feat_valid(f));
    int u = 0, v = 0;
    if (idx >= 0) {
      v = (int)(((uint64_t)(uint32_t)idx * wmagic) >> 40);
      u = idx - v * width;
    }
    emit(idx >= 0, u, v, pz, f, rec, meta);
    if (DEBUG) {
      ws.idx[(int64_t)b * m + i] = idx;
      ws.z[(int64_t)b * m + i] = pz;
    }
  };
  auto exact_at = [&](int64_t i, uint64_t* rec, uint32_t* meta) {   // reloads point i
    float x = X[i], y = X[ld + i], z = X[2 * ld + i];
    if (EQUIRECT && offset) {
      x = x - ox;
      y = y - oy;
      z = z - oz;
    }
    int32_t f[C];
#pragma unroll
    for (int k = 0; k < C; ++k) f[k] = (int32_t)F[i * C + k];
    exact(i, x, y, z, f, rec, meta);
  };
  auto load_group = [&](int64_t i0, float (&x)[kPPts], float (&y)[kPPts], float (&z)[kPPts],
                        int32_t (&f)[kPPts][C]) {
    const bool full = i0 + kPPts <= hi;
    if (full && vec) {
      const float4 vx = *reinterpret_cast<const float4*>(X + i0);
      const float4 vy = *reinterpret_cast<const float4*>(X + ld + i0);
      const float4 vz = *reinterpret_cast<const float4*>(X + 2 * ld + i0);
      x[0] = vx.x; x[1] = vx.y; x[2] = vx.z; x[3] = vx.w;
      y[0] = vy.x; y[1] = vy.y; y[2] = vy.z; y[3] = vy.w;
      z[0] = vz.x; z[1] = vz.y; z[2] = vz.z; z[3] = vz.w;
      int32_t flat[kPPts * C];
#pragma unroll
      for (int q = 0; q < C; ++q) {
        int32_t o[4];
        Feat4<T>::load(F + i0 * C + 4 * q, true, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) flat[4 * q + e] = o[e];
      }
#pragma unroll
      for (int p = 0; p < kPPts; ++p)
#pragma unroll
        for (int k = 0; k < C; ++k) f[p][k] = flat[p * C + k];
    } else {
#pragma unroll
      for (int p = 0; p < kPPts; ++p) {
        const int64_t i = i0 + p < hi ? i0 + p : hi - 1;   // (the tail repeats the last point)
        x[p] = X[i];
        y[p] = X[ld + i];
        z[p] = X[2 * ld + i];
#pragma unroll
        for (int k = 0; k < C; ++k) f[p][k] = (int32_t)F[i * C + k];
      }
    }
  };

  // ---- phase 1: every point's record and (supertile, rank) stay in registers
  uint64_t rec[PTS];
  uint32_t meta[PTS];
  // (round 6, the screened path: the two rare outcomes of a point are handled WITHOUT a divergent
  //  branch -- with 2-3 % of the points each, nearly every wave used to execute both side paths for
  //  every point.  An invalid point only feeds the sink: raw-bit / integer partials, converted once
  //  below.  An undecided point sets a bit; the bits are queued once per thread after the loop.)
  uint32_t pend = 0u;             // bit j: point j waits for the binary64 pass
  uint32_t inv_z = 0xffffffffu;   // min raw bits of rad over the screened-out points (rad > 0: bit order = value order)
  int32_t inv_f[C];               // ... max of their features, as integers ((float) and the ordered map are monotone)
#pragma unroll
  for (int k = 0; k < C; ++k) inv_f[k] = INT32_MIN;
  uint32_t fbits = 0u;            // OR of the valid points' feature words (the byte-range promise)
  const float fwidth = (float)width, fheight = (float)height;
  const float mgx = SE3DS_FAST_MARGIN * fwidth, mgy = SE3DS_FAST_MARGIN * fheight;
  const uint32_t rmask = (1u << rlog) - 1u;
  // a valid point of the screen: histogram rank + record
  auto emit_valid = [&](int u, int v, float pz, const int32_t (&f)[C], uint64_t* rec_o, uint32_t* meta_o) {
    const uint32_t sy_ = (uint32_t)v >> rlog, sx_ = super_x == 1 ? 0u : (uint32_t)u / (uint32_t)super_w;
    const uint32_t st = sy_ * (uint32_t)super_x + sx_;
    const uint32_t pix = ((uint32_t)v & rmask) * (uint32_t)super_w + ((uint32_t)u - sx_ * (uint32_t)super_w);
    const uint32_t rank = atomicAdd(&s_cnt[st], 1u);
    uint32_t fb[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < C; ++k) {
      fb[k] = (uint32_t)f[k] & 255u;
      fbits |= (uint32_t)f[k];
    }
    *rec_o = pack_rec(pix & 511u, pz, fb[0], fb[1], fb[2]);
    *meta_o = st | (rank << 11) | ((pix >> 9) << 25);   // 11-bit supertile, 14-bit rank, pixel bits 9..
  };
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int64_t i0 = lo + ((int64_t)it * kSThreads + threadIdx.x) * kPPts;
#pragma unroll
    for (int p = 0; p < kPPts; ++p) {
      rec[it * kPPts + p] = 0ull;
      meta[it * kPPts + p] = kSMetaNone;
    }
    float x[kPPts], y[kPPts], z[kPPts];
    int32_t f[kPPts][C];
    uint32_t pend_it = 0u;   // this group's undecided points (bit p)
    if (i0 < hi) {
      load_group(i0, x, y, z, f);
      if (EQUIRECT) {
        static_assert(kPPts % 2 == 0, "the screen works on pairs");
#pragma unroll
        for (int pp = 0; pp < kPPts; pp += 2) {
          // (x - 0 is exact: no separate path for a null offset)
          const f32x2 qx = {x[pp] - ox, x[pp + 1] - ox}, qy = {y[pp] - oy, y[pp + 1] - oy};
          const f32x2 qz = {z[pp] - oz, z[pp + 1] - oz};
          x[pp] = qx.x; x[pp + 1] = qx.y;   // (camera-relative from here on: what the queue parks)
          y[pp] = qy.x; y[pp + 1] = qy.y;
          z[pp] = qz.x; z[pp + 1] = qz.y;
          f32x2 fx2, fy2, rad2;
          equirect_fxy_fast2(qx, qy, qz, fwidth, fheight, &fx2, &fy2, &rad2);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int p = pp + e, j = it * kPPts + p;
            const float fx = e ? fx2.y : fx2.x, fy = e ? fy2.y : fy2.x, pz = e ? rad2.y : rad2.x;
            const bool in = i0 + p < hi;   // (the tail group repeats the last point)
            const bool decided = screen_decided(fx, fy, pz, mgx, mgy);
            const bool ok = (fx > -1.0f) && (fx < fwidth) && (fy > -1.0f) && (fy < fheight) &&
                            feat_valid(f[p]);
            const int u = (int)fx, v = (int)fy;
            if (in && decided && ok) emit_valid(u, v, pz, f[p], &rec[j], &meta[j]);
            const bool inv = in && decided && !ok;
            const uint32_t zb = __float_as_uint(pz);
            inv_z = inv && zb < inv_z ? zb : inv_z;
#pragma unroll
            for (int k = 0; k < C; ++k) inv_f[k] = inv && f[p][k] > inv_f[k] ? f[p][k] : inv_f[k];
            pend |= (in && !decided) ? (1u << j) : 0u;
            pend_it |= (in && !decided) ? (1u << p) : 0u;
            if (DEBUG && in && decided) {
              ws.idx[(int64_t)b * m + i0 + p] = ok ? v * width + u : -1;
              ws.z[(int64_t)b * m + i0 + p] = pz;
            }
          }
        }
      } else {
#pragma unroll
        for (int p = 0; p < kPPts; ++p) {
          if (i0 + p >= hi) continue;
          exact(i0 + p, x[p], y[p], z[p], f[p], &rec[it * kPPts + p], &meta[it * kPPts + p]);
        }
      }
    }
    if (EQUIRECT) {
      // the group's undecided points (~1.3 % of all) are queued for the dense binary64 pass WITH their
      // coordinates and features (round 6: the pass used to start with a global round trip to fetch
      // them again); one LDS atomic per wave, all lanes take part in the scan
      const uint32_t np = (uint32_t)__popc(pend_it);
      const uint32_t incl = wave_incl_scan_u32(np);
      const uint32_t wtot = (uint32_t)__shfl((int)incl, 63, 64);
      if (wtot) {
        uint32_t base = 0u;
        if ((threadIdx.x & 63) == 63) base = atomicAdd(&s_qn, wtot);
        base = (uint32_t)__shfl((int)base, 63, 64);
        uint32_t slot = base + incl - np;
#pragma unroll
        for (int p = 0; p < kPPts; ++p) {
          if (!((pend_it >> p) & 1u)) continue;
          if (slot < (uint32_t)kQCap) {
            s_qxyz[slot] = x[p];
            s_qxyz[kQCap + slot] = y[p];
            s_qxyz[2 * kQCap + slot] = z[p];
#pragma unroll
            for (int k = 0; k < C; ++k) s_qf[k * kQCap + slot] = f[p][k];
            s_queue[slot] = (uint16_t)((it * kSThreads + (int)threadIdx.x) * kPPts + p);
          }
          ++slot;
        }
      }
    }
  }
  SE3DS_STAMP(0, 1);   // loads + screen + histogram done (this wave)
  bad |= fbits >> 8;   // (negative or > 255 under the promise)
  if (inv_z != 0xffffffffu) {   // the screened-out points join the sink partials in their final form
    const uint32_t o = inv_z | 0x80000000u;   // se3ds_f32_to_ordered of a positive float
    sink = o < sink ? o : sink;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const uint32_t of = se3ds_f32_to_ordered((float)inv_f[k]);
      smax[k] = of > smax[k] ? of : smax[k];
    }
  }
  // point j of this thread, as an offset inside the chunk
  auto chunk_offset = [&](int j) {
    return (uint32_t)(((j / kPPts) * kSThreads + (int)threadIdx.x) * kPPts + (j % kPPts));
  };
  SE3DS_STAMP(0, 2);   // pending points queued
  // ---- phase 2: the undecided points, densely (their records stay with the thread that ran them)
  uint64_t qrec[kQPer];
  uint32_t qmeta[kQPer];
#pragma unroll
  for (int r = 0; r < kQPer; ++r) {
    qrec[r] = 0ull;
    qmeta[r] = kSMetaNone;
  }
  if (EQUIRECT) {
    __syncthreads();
    const uint32_t qn = s_qn;
    if (qn <= (uint32_t)kQCap) {
#pragma unroll
      for (int r = 0; r < kQPer; ++r) {
        const uint32_t q = threadIdx.x + r * kSThreads;
        if (q < qn) {
          int32_t qf[C];
#pragma unroll
          for (int k = 0; k < C; ++k) qf[k] = s_qf[k * kQCap + q];
          exact(lo + s_queue[q], s_qxyz[q], s_qxyz[kQCap + q], s_qxyz[2 * kQCap + q], qf, &qrec[r], &qmeta[r]);
        }
      }
    } else {
      // queue overflow (> 12 % undecided: lattice clouds): every thread takes its own marked points
#pragma unroll
      for (int j = 0; j < PTS; ++j) {
        if (!((pend >> j) & 1u)) continue;
        exact_at(lo + chunk_offset(j), &rec[j], &meta[j]);
      }
    }
  }
  __syncthreads();
  SE3DS_STAMP(0, 3);   // exact pass done, all waves through the barrier
  // ---- sink partials of the invalid points (as the packed path).  They are final here -- the exact
  // pass was the last to feed them -- and their reduction rides on the scan's barriers below (round 6:
  // as the kernel's tail it cost every workgroup a barrier and a serial loop behind its stores, ~0.7 us)
  __shared__ uint32_t s_red[1 + C][kSThreads / 64];
  sink = wave_min_u32(sink);
  bad = wave_max_u32(bad);
  if ((threadIdx.x & 63) == 0) s_red[0][threadIdx.x >> 6] = sink;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const uint32_t v = wave_max_u32(smax[k]);
    if ((threadIdx.x & 63) == 0) s_red[1 + k][threadIdx.x >> 6] = v;
  }
  if ((threadIdx.x & 63) == 0 && bad != 0u) {
    atomicExch(&sw.ctl[1], sw.epoch);
    atomicOr(&ws.sink_z[kPromiseStickyWord], 1u);
  }
  // ---- phase 3: exclusive scan of the supertile counts (<= 2048: four per thread), run descriptors
  uint32_t all;
  {
    const int t0 = 4 * threadIdx.x;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = t0 + q < nsuper ? s_cnt[t0 + q] : 0u;
      sum += c[q];
    }
    uint32_t ex = block_excl_scan_u32<kSThreads / 64>(sum, s_w, &all);
    {   // (s_red is visible behind the scan's barriers)
      const int part = blockIdx.y * gridDim.x + blockIdx.x;
      if (threadIdx.x == 0) {
        uint32_t v = s_red[0][0];
        for (int i = 1; i < kSThreads / 64; ++i) v = s_red[0][i] < v ? s_red[0][i] : v;
        ws.zpart[part] = v;
      }
      if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + C) {
        const int k = threadIdx.x - 64;
        uint32_t v = 0u;
        for (int i = 0; i < kSThreads / 64; ++i) v = s_red[1 + k][i] > v ? s_red[1 + k][i] : v;
        ws.fpart[(int64_t)part * C + k] = v;
      }
    }
    uint32_t* R = sw.runs + ((int64_t)b * nsuper) * sw.cstride + run_slot(blockIdx.x, sw.cstride);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (t0 + q < nsuper) {
        s_cnt[t0 + q] = ex;   // (in place: every count was read before the scan's barriers)
        R[(int64_t)(t0 + q) * sw.cstride] = (ex << 16) | c[q];
      }
      ex += c[q];
    }
  }
  {
    __syncthreads();
    SE3DS_STAMP(0, 4);   // scan + run descriptors
    // ---- phase 4: placement in LDS, then full-line stores of the chunk
    auto place = [&](uint64_t r, uint32_t mt) {
      if (mt >= kSMetaPending) return;
      const uint32_t pos = s_cnt[mt & 2047u] + ((mt >> 11) & 16383u);
      s_rec[pos] = r;
      s_hi[pos] = (uint8_t)(mt >> 25);
    };
#pragma unroll
    for (int j = 0; j < PTS; ++j) place(rec[j], meta[j]);
#pragma unroll
    for (int r = 0; r < kQPer; ++r) place(qrec[r], qmeta[r]);
    __syncthreads();
    SE3DS_STAMP(0, 5);   // placed in LDS
    uint64_t* G = sw.rec + ((int64_t)b * sw.chunks + blockIdx.x) * kChunk;
    uint8_t* H = sw.hi + ((int64_t)b * sw.chunks + blockIdx.x) * kChunk;
    for (uint32_t j = 2 * threadIdx.x; j < all; j += 2 * kSThreads)
      *reinterpret_cast<uint4*>(G + j) = *reinterpret_cast<const uint4*>(s_rec + j);
    for (uint32_t j = 16 * threadIdx.x; j < all; j += 16 * kSThreads)
      *reinterpret_cast<uint4*>(H + j) = *reinterpret_cast<const uint4*>(s_hi + j);
  }
  SE3DS_STAMP(0, 6);   // chunk stores issued
  SE3DS_STAMP(0, 7);   // end (the sink partials left with the scan)
}

// byte-wise max of two words of three feature bytes
__device__ __forceinline__ uint32_t bytemax3(uint32_t a, uint32_t b) {
  const uint32_t x0 = a & 0xffu, y0 = b & 0xffu, x1 = a & 0xff00u, y1 = b & 0xff00u;
  const uint32_t x2 = a & 0xff0000u, y2 = b & 0xff0000u;
  return (x0 > y0 ? x0 : y0) | (x1 > y1 ? x1 : y1) | (x2 > y2 ? x2 : y2);
}

// S2.  One workgroup per (image, supertile).
template <int C>
__global__ void __launch_bounds__(kRThreads, 6)   // (<= 80 VGPRs: three workgroups per CU.  Round 6 tried 64 VGPRs / kRStash 8 / four per CU so that all
                                                  //  1 024 supertiles of cfg5 are resident: lifetime 16.3 -> 22.6 us, span 30 -> 30 us, room 78 -> 85 us --
                                                  //  the kernel is bound by the LDS atomic rate, not by occupancy)
splat_sort_resolve_kernel(int height, int width, int nsuper, int super_x, int super_w, int rlog,
                          float depth_scale, float output_void, float mask_void,
                          float* __restrict__ depth, float* __restrict__ feat,
                          float* __restrict__ mask, SplatWs ws, SortWs sw, uint32_t zpart_count) {
  const int kPx = super_w << rlog;
  extern __shared__ uint32_t s_dyn[];   // [kPx] z, [kPx] feature words, [chunks + 1] run prefix, [chunks] run base
  uint32_t* s_z = s_dyn;
  uint32_t* s_f = s_dyn + kPx;
  uint32_t* s_cpre = s_f + kPx;             // NON-EMPTY runs only: first flat record index (+ sentinel)
  uint32_t* s_cbase = s_cpre + sw.chunks + 1;   // ... and the record number of the run's first record
  __shared__ uint32_t s_w[kRThreads / 64];
  __shared__ uint32_t s_c[C][kRThreads / 64];
  __shared__ uint32_t s_flag;
  __shared__ uint32_t s_q[kRStash * kRThreads / 64];   // first run of every 64-record block of a batch
  SE3DS_STAMP(1, 0);   // start
  const int chunks = sw.chunks;
  const int bs = blockIdx.x;
  const int b = bs / nsuper, st = bs - b * nsuper;
  const int sy_ = st / super_x, sx_ = st - sy_ * super_x;
  const bool first = bs == 0;   // holds flat pixel 0, which also receives the sink
  // this supertile's runs, compacted to the non-empty ones (a smooth scene leaves most chunks
  // without a record here), with their exclusive record prefix; total records
  uint32_t total = 0, nnz = 0;
  {
    // (the row holds the chunks XCD by XCD, run_slot(): the order of the runs is irrelevant -- min /
    // max do not depend on the record order -- only slot -> chunk must be undone for the addresses)
    const uint32_t* R = sw.runs + (int64_t)bs * sw.cstride;
    for (int c0 = 0; c0 < sw.cstride; c0 += kRThreads) {
      const int slot = c0 + threadIdx.x;
      const int c = slot < sw.cstride ? run_chunk(slot, sw.cstride) : chunks;
      const uint32_t d = c < chunks ? R[slot] : 0u;
      const uint32_t cnt = d & 0xffffu;
      uint32_t all, alln;
      const uint32_t ex = block_excl_scan_u32<kRThreads / 64>(cnt, s_w, &all);
      const uint32_t exn = block_excl_scan_u32<kRThreads / 64>(cnt ? 1u : 0u, s_w, &alln);
      if (cnt) {
        s_cpre[nnz + exn] = total + ex;
        s_cbase[nnz + exn] = (uint32_t)c * (uint32_t)sw.chunk_pts + (d >> 16);
      }
      total += all;
      nnz += alln;
    }
    if (threadIdx.x == 0) s_cpre[nnz] = total;   // sentinel: every walk below stops here
  }
  SE3DS_STAMP(1, 1);   // run row read, compacted, prefix
  {
    uint32_t sink_o = 0xffffffffu;
    if (first) {
      uint32_t v = 0xffffffffu;
      for (uint32_t i = threadIdx.x; i < zpart_count; i += kRThreads) {
        const uint32_t z = ws.zpart[i];
        v = z < v ? z : v;
      }
      v = wave_min_u32(v);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
      __syncthreads();
      for (int i = 0; i < kRThreads / 64; ++i) sink_o = s_w[i] < sink_o ? s_w[i] : sink_o;
    }
    const bool have_sink_z = sink_o != 0xffffffffu;
    const float sink_z = se3ds_ordered_to_f32(sink_o);
    for (int p = threadIdx.x; p < kPx; p += kRThreads) {
      s_z[p] = __float_as_uint(depth_scale);
      s_f[p] = 0u;
    }
    __syncthreads();
    SE3DS_STAMP(1, 2);   // LDS tile initialised
    // flat record index j -> record number inside this image: run k with cpre[k] <= j < cpre[k + 1],
    // record cbase[k] + (j - cpre[k]).  A binary search per RECORD made the LDS pipe the kernel's
    // bottleneck (11 conflicted reads per record against ~1.4 atomics): one search per aligned block
    // of 64 records (one wave iteration) gives the block's first run, and each lane walks forward
    // from there -- a few reads of nearly the same words, broadcast within the wave.
    int steps = 0;
    while ((1u << steps) < (nnz ? nnz : 1u)) ++steps;
    // (record numbers inside one image are < 2^25: 32-bit byte offsets from a uniform base keep the
    // address arithmetic out of the vector ALU)
    const char* RB = reinterpret_cast<const char*>(sw.rec + (int64_t)b * chunks * sw.chunk_pts);
    const uint8_t* HB = sw.hi + (int64_t)b * chunks * sw.chunk_pts;
    const uint32_t last = total ? total - 1u : 0u;
    auto block_table = [&](uint32_t base) {   // s_q[blk] for the blocks of the batch starting at `base`
      __syncthreads();                        // (the previous batch's walks are over)
      if (threadIdx.x < kRStash * kRThreads / 64) {
        uint32_t jb = base + 64u * threadIdx.x;
        jb = jb < total ? jb : last;
        int lo_ = 0, hi_ = (int)nnz - 1;
        for (int it = 0; it < steps; ++it) {
          const int mid = (lo_ + hi_ + 1) >> 1;
          const bool le = s_cpre[mid] <= jb;
          lo_ = le ? mid : lo_;
          hi_ = le ? hi_ : mid - 1;
        }
        s_q[threadIdx.x] = (uint32_t)lo_;
      }
      __syncthreads();
    };
    // (the walks of a thread advance in GROUPS of four: three predicated steps cover the usual case
    // with four independent LDS reads in flight per step, a loop takes the stragglers; a group's
    // loads are issued as soon as its addresses are known -- they fly under the next group's walk)
    constexpr int kWalk = 4;
    static_assert(kRStash % kWalk == 0, "walk groups");
    auto locate_group = [&](int q0, const uint32_t (&j)[kWalk], uint32_t (&a)[kWalk]) {
      uint32_t k[kWalk];
#pragma unroll
      for (int q = 0; q < kWalk; ++q) k[q] = s_q[(threadIdx.x >> 6) + (q0 + q) * (kRThreads / 64)];
#pragma unroll
      for (int stp = 0; stp < 3; ++stp) {
        uint32_t nx[kWalk];
#pragma unroll
        for (int q = 0; q < kWalk; ++q) nx[q] = s_cpre[k[q] + 1];
#pragma unroll
        for (int q = 0; q < kWalk; ++q) k[q] += nx[q] <= j[q] ? 1u : 0u;
      }
#pragma unroll
      for (int q = 0; q < kWalk; ++q)
        while (s_cpre[k[q] + 1] <= j[q]) ++k[q];
      uint32_t cb[kWalk], cp[kWalk];
#pragma unroll
      for (int q = 0; q < kWalk; ++q) {
        cb[q] = s_cbase[k[q]];
        cp[q] = s_cpre[k[q]];
      }
#pragma unroll
      for (int q = 0; q < kWalk; ++q) a[q] = cb[q] + (j[q] - cp[q]);
    };
    // Both passes walk the records in BATCHES of kRStash per thread (kRStash x 512 per workgroup):
    // all searches, then all 2 x kRStash loads, then the LDS work.  A supertile of average weight
    // is one batch and its records stay in registers between the passes; heavy ones (the source
    // views' pole rows pile up in a few supertiles of ANY target) stream their later batches a
    // second time (L2 hits).
    constexpr uint32_t kBatch = kRStash * kRThreads;
    uint64_t sr[kRStash];
    uint32_t sp[kRStash];
    auto load_batch = [&](uint32_t base) {
      block_table(base);
#pragma unroll
      for (int q0 = 0; q0 < kRStash; q0 += kWalk) {
        uint32_t jj[kWalk], sa[kWalk];
#pragma unroll
        for (int q = 0; q < kWalk; ++q) {
          const uint32_t j = base + threadIdx.x + (q0 + q) * kRThreads;
          jj[q] = j < total ? j : last;   // (out-of-range slots re-read the last record, unused)
        }
        locate_group(q0, jj, sa);
#pragma unroll
        for (int q = 0; q < kWalk; ++q) {
          sr[q0 + q] = *reinterpret_cast<const uint64_t*>(RB + (sa[q] << 3));
          sp[q0 + q] = (uint32_t)HB[sa[q]];
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // all 2 x kRStash loads are issued before the first use
#pragma unroll
      for (int k = 0; k < kRStash; ++k) sp[k] = rec_pix(sr[k]) | (sp[k] << 9);
    };
    // pass A: z-min
    for (uint32_t base = 0; base < total; base += kBatch) {
      load_batch(base);
#pragma unroll
      for (int k = 0; k < kRStash; ++k) {
        const uint32_t j = base + threadIdx.x + k * kRThreads;
        // valid => z > 0.  LDS atomics run at ~1 lane per clock and CU and are this kernel's floor: a
        // plain read first, the atomic only where the record can lower the minimum (a stale read costs
        // an unneeded atomic, never a missed one: the minimum only falls)
        if (j < total) {
          const uint32_t zb = rec_zbits(sr[k]);
          if (zb < s_z[sp[k]]) atomicMin(&s_z[sp[k]], zb);
        }
      }
    }
    SE3DS_STAMP(1, 3);   // pass A (searches, loads, z-min atomics) done by this wave
    __syncthreads();
    // pass B: survivors (z < zmin + 0.1) max their features; the rest feed the sink
    uint32_t sbytes = 0u;   // byte-wise max of the occluded records' feature words
    auto passb = [&](uint64_t r, uint32_t pix) {
      const float z = __uint_as_float(rec_zbits(r));
      float zm = __uint_as_float(s_z[pix]);
      if (first && pix == 0u && have_sink_z) zm = sink_z < zm ? sink_z : zm;
      const uint32_t fw = (uint32_t)(r >> 32) & 0xffffffu;
      if (z < zm + 0.1f) {
        uint32_t old = s_f[pix];
        while (true) {
          const uint32_t nw = bytemax3(old, fw);
          if (nw == old) break;
          const uint32_t prev = atomicCAS(&s_f[pix], old, nw);
          if (prev == old) break;
          old = prev;
        }
      } else {
        sbytes = bytemax3(sbytes, fw);   // (converted to the sink's ordered floats once, below)
      }
    };
    // (the registers still hold the LAST batch of pass A: walk the batches backwards)
    if (total) {
      const uint32_t nbat = (total + kBatch - 1) / kBatch;
      for (uint32_t bi = nbat; bi-- > 0;) {
        const uint32_t base = bi * kBatch;
        if (bi + 1 != nbat) load_batch(base);
#pragma unroll
        for (int k = 0; k < kRStash; ++k)
          if (base + threadIdx.x + k * kRThreads < total) passb(sr[k], sp[k]);
      }
    }
    SE3DS_STAMP(1, 4);   // pass B done by this wave
#pragma unroll
    for (int k = 0; k < C; ++k) {
      // (a feature of 0 never raises the sink's maximum over fill(0): ordered(0.0f) stays out)
      const uint32_t fb = (sbytes >> (16 - 8 * k)) & 255u;
      const uint32_t v = wave_max_u32(fb ? se3ds_f32_to_ordered((float)fb) : 0u);
      if ((threadIdx.x & 63) == 0) s_c[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    // (returned atomic: performed memory-side before this thread passes the ticket's barrier below.  Its
    //  round trip -- ~1.5 us, profiles/r06_warp_phases.txt -- flies under the output stores: the return
    //  value is only consumed behind them)
    uint32_t sink_ret = 0u;
    if ((int)threadIdx.x < C) {
      uint32_t v = 0u;
      for (int i = 0; i < kRThreads / 64; ++i) v = s_c[threadIdx.x][i] > v ? s_c[threadIdx.x][i] : v;
      if (v != 0u) sink_ret = atomicMax(&sw.fpart2[(blockIdx.x % kSinkSlots) * C + threadIdx.x], v);
    }
    SE3DS_STAMP(1, 5);   // sink slots
    // outputs of this item's pixels -- all but flat pixel 0, which only the last workgroup writes
    const int64_t hw = (int64_t)height * width;
    float* __restrict__ depth_b = depth + (int64_t)b * hw;
    float* __restrict__ feat_b = feat + (int64_t)b * hw * C;
    float* __restrict__ mask_b = mask ? mask + (int64_t)b * hw : nullptr;
    for (int p = threadIdx.x; p < kPx; p += kRThreads) {
      const int r = rlog ? p / super_w : 0;
      const int y = (sy_ << rlog) + r, x = sx_ * super_w + (p - r * super_w);
      if (y >= height || x >= width) continue;
      const uint32_t ip = (uint32_t)y * (uint32_t)width + (uint32_t)x;   // (< 2^31 / images: checked by the dispatcher)
      float z = __uint_as_float(s_z[p]);
      if (first && ip == 0u) {   // published for the fold: final z bits and the survivors' feature word
        if (have_sink_z) z = sink_z < z ? sink_z : z;
        s_w[0] = atomicExch(&sw.ctl[4], __float_as_uint(z));
        s_w[1] = atomicExch(&sw.ctl[5], s_f[p]);
        continue;
      }
      float d = z < 0.0f ? 0.0f : (z > depth_scale ? depth_scale : z);
      d = d / depth_scale;
      bool all_ok = true;
      const uint32_t fw = s_f[p];
#pragma unroll
      for (int k = 0; k < C; ++k) {
        // scatter_max over fill(output_void >= 0): a feature of 0 and "no survivor" coincide
        const float fv = (float)((fw >> (16 - 8 * k)) & 255u);
        const float v = fv > output_void ? fv : output_void;
        feat_b[ip * C + k] = v;
        all_ok = all_ok && (v != mask_void);
      }
      depth_b[ip] = d;
      if (mask_b) mask_b[ip] = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
    }
    if ((int)threadIdx.x < C) s_c[threadIdx.x][0] = sink_ret;   // (the wait for the sink atomic)
    SE3DS_STAMP(1, 6);   // outputs stored
    if (first) {
      // the invalid points' feature maxima (S1's per-chunk partials, a kernel ago) join the occluded
      // points' slots here, so that the last workgroup only reads the slots
      uint32_t fm[C];
#pragma unroll
      for (int k = 0; k < C; ++k) fm[k] = 0u;
      for (uint32_t i = threadIdx.x; i < zpart_count * C; i += kRThreads) {
        const uint32_t v = ws.fpart[i];
        const int k = i % C;
#pragma unroll
        for (int kk = 0; kk < C; ++kk)
          if (kk == k) fm[kk] = v > fm[kk] ? v : fm[kk];
      }
      __syncthreads();   // (s_c: this workgroup's sink reduction above has been consumed)
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const uint32_t v = wave_max_u32(fm[k]);
        if ((threadIdx.x & 63) == 0) s_c[k][threadIdx.x >> 6] = v;
      }
      __syncthreads();
      if ((int)threadIdx.x < C) {
        uint32_t v = 0u;
        for (int i = 0; i < kRThreads / 64; ++i) v = s_c[threadIdx.x][i] > v ? s_c[threadIdx.x][i] : v;
        if (v != 0u) s_w[2 + threadIdx.x] = atomicMax(&sw.fpart2[threadIdx.x], v);   // (returned)
      }
    }
  }
  // ---- ticket: the last workgroup to arrive folds the sink into flat pixel 0
  __syncthreads();   // (this workgroup's sink / pixel-0 atomics have returned)
  if (threadIdx.x == 0) s_flag = atomicAdd(&sw.ctl[0], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  SE3DS_STAMP(1, 7);   // ticket taken
  if (s_flag == 0u) return;
  uint32_t fm[C];
#pragma unroll
  for (int k = 0; k < C; ++k) fm[k] = 0u;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kSinkSlots * C; i += kRThreads) {   // occluded + invalid points (atomics)
    const uint32_t v = atomicMax(&sw.fpart2[i], 0u);
    const int k = i % C;
#pragma unroll
    for (int kk = 0; kk < C; ++kk)
      if (kk == k) fm[kk] = v > fm[kk] ? v : fm[kk];
  }
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const uint32_t v = wave_max_u32(fm[k]);
    if ((threadIdx.x & 63) == 0) s_c[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float z = __uint_as_float(atomicOr(&sw.ctl[4], 0u));
    const uint32_t fw = atomicOr(&sw.ctl[5], 0u);
    float d = z < 0.0f ? 0.0f : (z > depth_scale ? depth_scale : z);
    d = d / depth_scale;
    bool all_ok = true;
    for (int k = 0; k < C; ++k) {
      uint32_t sv = 0u;
      for (int i = 0; i < kRThreads / 64; ++i) sv = s_c[k][i] > sv ? s_c[k][i] : sv;
      const float fv = (float)((fw >> (16 - 8 * k)) & 255u);
      float v = fv > output_void ? fv : output_void;
      if (sv != 0u) {
        const float sf = se3ds_ordered_to_f32(sv);
        v = sf > v ? sf : v;
      }
      feat[k] = v;
      all_ok = all_ok && (v != mask_void);
    }
    depth[0] = d;
    if (mask) mask[0] = (d > 0.0f && d < 1.0f && all_ok) ? 1.0f : 0.0f;
  }
}

// (set once per kernel and size: hipFuncSetAttribute on every launch is host time for nothing)
inline bool sort_lds_attr(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> done;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(fn);
  if (it != done.end() && it->second >= bytes) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return false;
  done[fn] = bytes;
  return true;
}

template <typename T, bool EQUIRECT, int C, int PTS>
int launch_splat_sorted_c(const float* coords, const float* offset, const T* feats, int n, int64_t m,
                          int64_t ld, int height, int width, float depth_scale, float input_void,
                          float output_void, float* depth, float* feat, float* mask, float mask_void,
                          SplatWs ws, SortWs sw, const SortGeom& g, hipStream_t stream) {
  const uint64_t wmagic = ((uint64_t)1 << 40) / (uint64_t)width + 1;
  const dim3 g_pt((unsigned)g.chunks, (unsigned)n);
  const int vec = (ld % 4 == 0) && ((uintptr_t)coords % 16 == 0) &&
                  ((uintptr_t)feats % (sizeof(T) == 1 ? 4 : 16) == 0);
  const char* e_dbg = getenv("SE3DS_SPLAT_DEBUG");
  const bool dbg = e_dbg && atoi(e_dbg) != 0;
  const size_t lds1 = (size_t)g.chunk_pts * 9 + 4 * (size_t)g.nsuper;
  const size_t lds2 = 4 * (2 * (size_t)g.px + 2 * (size_t)g.chunks + 1);
  const void* k1 = dbg ? (const void*)splat_sort_kernel<T, EQUIRECT, C, true, PTS>
                       : (const void*)splat_sort_kernel<T, EQUIRECT, C, false, PTS>;
  const void* k2 = (const void*)splat_sort_resolve_kernel<C>;
  if (!sort_lds_attr(k1, lds1) || !sort_lds_attr(k2, lds2)) return SE3DS_E_LAUNCH;
  if (dbg)
    hipLaunchKernelGGL((splat_sort_kernel<T, EQUIRECT, C, true, PTS>), g_pt, dim3(kSThreads), lds1,
                       stream, coords, offset, feats, m, ld, height, width, wmagic, input_void,
                       g.nsuper, g.super_x, g.super_w, g.rlog, vec, ws, sw);
  else
    hipLaunchKernelGGL((splat_sort_kernel<T, EQUIRECT, C, false, PTS>), g_pt, dim3(kSThreads), lds1,
                       stream, coords, offset, feats, m, ld, height, width, wmagic, input_void,
                       g.nsuper, g.super_x, g.super_w, g.rlog, vec, ws, sw);
  hipLaunchKernelGGL((splat_sort_resolve_kernel<C>), dim3((unsigned)(n * g.nsuper)), dim3(kRThreads),
                     lds2, stream, height, width, g.nsuper, g.super_x, g.super_w, g.rlog, depth_scale,
                     output_void, mask_void, depth, feat, mask, ws, sw, (uint32_t)(g.chunks * n));
  return check_launch("splat(sorted)");
}

template <typename T, bool EQUIRECT>
int launch_splat_sorted(const float* coords, const float* offset, const T* feats, int n, int64_t m,
                        int64_t ld, int channels, int height, int width, float depth_scale,
                        float input_void, float output_void, float* depth, float* feat, float* mask,
                        float mask_void, void* workspace, const SortGeom& g, hipStream_t stream) {
  SplatWs ws = carve_ws(workspace, n, m);
  const size_t base = splat_hdr_bytes() + (sizeof(int32_t) + sizeof(float)) * (size_t)n * (size_t)m;
  SortWs sw = carve_sort_ws((char*)workspace + align16(base), n, g);
  sw.epoch = next_splat_epoch();
#define SE3DS_S(CC, PP)                                                                           \
  return launch_splat_sorted_c<T, EQUIRECT, CC, PP>(coords, offset, feats, n, m, ld, height, width, \
                                                    depth_scale, input_void, output_void, depth,  \
                                                    feat, mask, mask_void, ws, sw, g, stream)
#define SE3DS_SC(CC) if (g.pts == 4) SE3DS_S(CC, 4); else if (g.pts == 8) SE3DS_S(CC, 8); else SE3DS_S(CC, 16);
  switch (channels) {
    case 1: SE3DS_SC(1)
    case 2: SE3DS_SC(2)
    default: SE3DS_SC(3)
  }
#undef SE3DS_SC
#undef SE3DS_S
}

template <typename T, bool EQUIRECT>
int launch_splat(const float* coords, const float* offset, const T* feats, int n, int64_t m,
                 int64_t ld,
                 int channels, int height, int width, float depth_scale, float input_void,
                 float output_void, float* depth, float* feat, float* mask, float mask_void,
                 void* workspace, hipStream_t stream, bool byte_range = false) {
  {
    static const bool no_bin = getenv("SE3DS_SPLAT_SCATTER") != nullptr;
    // 8-byte packed records (round 3): features of valid points are bytes by type (uint8) or by
    // the caller's promise (int32 | SE3DS_FEAT_BYTE_RANGE), <= 3 channels, non-negative output
    // void.  SE3DS_SPLAT_PACKED=0 keeps the 20-byte-record path (A/B runs, parity tests).
    if constexpr (!std::is_same<T, float>::value) {
      static const bool no_pack = [] {
        const char* e = getenv("SE3DS_SPLAT_PACKED");
        return e && atoi(e) == 0;
      }();
      const int64_t ptiles = (int64_t)ceil_div(height, kPTileY) * ceil_div(width, kPTileX);
      // round 4: sorted chunks + gathering resolve (SE3DS_SPLAT_SORT: 0 off, 1 = when the image has
      // enough supertiles to fill the chip (default), 2 = always)
      static const int sort_mode = sort_env("SE3DS_SPLAT_SORT", 1);
      const bool packable = !no_bin && !no_pack && m > 0 && channels <= kPMaxChannels &&
          (std::is_same<T, uint8_t>::value || byte_range) && output_void >= 0.0f &&
          (int64_t)n * m < ((int64_t)1 << 31) && (int64_t)height * width * width < ((int64_t)1 << 40);
      if (packable && sort_mode != 0) {
        const SortGeom sg = sort_geom(n, m, height, width);
        if (sg.nsuper <= kSMaxSuper && sg.chunks <= kSMaxChunks &&
            (int64_t)sg.chunks * n <= kMaxSinkBlocks && (int64_t)n * sg.nsuper < ((int64_t)1 << 24) &&
            (sort_mode == 2 || (int64_t)n * sg.nsuper >= 128))
          return launch_splat_sorted<T, EQUIRECT>(coords, offset, feats, n, m, ld, channels, height,
                                                  width, depth_scale, input_void, output_void, depth,
                                                  feat, mask, mask_void, workspace, sg, stream);
      }
      if (!no_bin && !no_pack && m > 0 && channels <= kPMaxChannels && ptiles <= kMaxTiles &&
          (std::is_same<T, uint8_t>::value || byte_range) && output_void >= 0.0f &&
          (int64_t)n * ptiles < ((int64_t)1 << 20) && (int64_t)n * m < ((int64_t)1 << 31) &&
          (int64_t)height * width * width < ((int64_t)1 << 40))
        return launch_splat_packed<T, EQUIRECT>(coords, offset, feats, n, m, ld, channels, height,
                                                width, depth_scale, input_void, output_void, depth,
                                                feat, mask, mask_void, workspace, stream);
    }
    const int64_t ntiles = (int64_t)ceil_div(height, kTileY) * ceil_div(width, kTileX);
    if (!no_bin && m > 0 && channels <= kMaxBinChannels && ntiles <= kMaxTiles &&
        (int64_t)n * m < ((int64_t)1 << 31) &&
        (int64_t)height * width * width < ((int64_t)1 << 40))
      return launch_splat_binned<T, EQUIRECT>(coords, offset, feats, n, m, ld, channels, height, width,
                                              depth_scale, input_void, output_void, depth, feat,
                                              mask, mask_void, workspace, stream);
  }
  SplatWs ws = carve_ws(workspace, n, m);
  const int64_t npx = (int64_t)n * height * width;
  const bool ordered = !(output_void >= 0.0f);
  int g_px = grid_for(npx * (channels > 1 ? channels : 1), kBlock);
  if (ordered)
    hipLaunchKernelGGL(splat_init_kernel<true>, dim3(g_px), dim3(kBlock), 0, stream, depth, feat,
                       npx, channels, depth_scale, output_void, ws);
  else
    hipLaunchKernelGGL(splat_init_kernel<false>, dim3(g_px), dim3(kBlock), 0, stream, depth, feat,
                       npx, channels, depth_scale, output_void, ws);
  if (m > 0) {
    dim3 g_pt = point_grid(m, n, kBlock);
    const int nparts = (int)(g_pt.x * g_pt.y);
    hipLaunchKernelGGL((splat_zmin_kernel<T, EQUIRECT>), g_pt, dim3(kBlock), 0, stream, coords,
                       offset, feats, m, ld, channels, height, width, input_void, depth, ws);
    hipLaunchKernelGGL(splat_sink_z_kernel, dim3(1), dim3(kBlock), 0, stream, ws, nparts);
    if (ordered)
      hipLaunchKernelGGL((splat_resolve_kernel<T, true>), g_pt, dim3(kBlock), 0, stream, feats, m,
                         ld, channels, height, width, depth, feat, ws);
    else
      hipLaunchKernelGGL((splat_resolve_kernel<T, false>), g_pt, dim3(kBlock), 0, stream, feats, m,
                         ld, channels, height, width, depth, feat, ws);
    hipLaunchKernelGGL(splat_sink_feat_kernel, dim3(1), dim3(kBlock), 0, stream, ws, nparts,
                       channels);
  }
  int g_fin = grid_for(npx, kBlock);
  if (ordered)
    hipLaunchKernelGGL(splat_finalize_kernel<true>, dim3(g_fin), dim3(kBlock), 0, stream, depth,
                       feat, mask, npx, channels, depth_scale, mask_void, ws);
  else
    hipLaunchKernelGGL(splat_finalize_kernel<false>, dim3(g_fin), dim3(kBlock), 0, stream, depth,
                       feat, mask, npx, channels, depth_scale, mask_void, ws);
  return check_launch("splat");
}

template <bool EQUIRECT>
int dispatch_splat(const float* coords, const float* offset, const void* feats, int feat_dtype,
                   int n, int64_t m, int64_t ld, int channels, int height, int width, float depth_scale,
                   float input_void, float output_void, float* depth, float* feat, float* mask,
                   float mask_void, void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || m < 0 || ld < m || channels <= 0 || channels > 60 || height <= 0 || width <= 0)
    return SE3DS_E_BADSHAPE;
  if ((int64_t)n * height * width >= (int64_t)1 << 31) return SE3DS_E_BADSHAPE;
  const bool byte_range = (feat_dtype & SE3DS_FEAT_BYTE_RANGE) != 0;
  feat_dtype &= ~SE3DS_FEAT_BYTE_RANGE;
  if (workspace_bytes < se3ds_splat_workspace_bytes(n, m, height, width, channels))
    return SE3DS_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  switch (feat_dtype) {
    case SE3DS_F32:
      return launch_splat<float, EQUIRECT>(coords, offset, (const float*)feats, n, m, ld, channels,
                                           height, width, depth_scale, input_void, output_void,
                                           depth, feat, mask, mask_void, workspace, s);
    case SE3DS_I32:
      return launch_splat<int32_t, EQUIRECT>(coords, offset, (const int32_t*)feats, n, m, ld, channels,
                                             height, width, depth_scale, input_void, output_void,
                                             depth, feat, mask, mask_void, workspace, s, byte_range);
    case SE3DS_U8:
      return launch_splat<uint8_t, EQUIRECT>(coords, offset, (const uint8_t*)feats, n, m, ld, channels,
                                             height, width, depth_scale, input_void, output_void,
                                             depth, feat, mask, mask_void, workspace, s);
    default:
      return SE3DS_E_BADDTYPE;
  }
}

// ------------------------------------------------------------------------- bilinear gather
// tfa.image.interpolate_bilinear: one (query, channel) pair per lane, channel fastest so a
// wave reads runs of contiguous channels of the four taps.
__global__ void __launch_bounds__(kBlock)
interp_bilinear_kernel(const float* __restrict__ grid, const float* __restrict__ query,
                       int height, int width, int channels, int64_t q, int xy,
                       float* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t total = q * channels;
  const float* G = grid + (int64_t)b * height * width * channels;
  const float* Q = query + (int64_t)b * q * 2;
  float* O = out + (int64_t)b * q * channels;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    int64_t qi = t / channels;
    int c = (int)(t - qi * channels);
    float q0 = Q[qi * 2 + 0], q1 = Q[qi * 2 + 1];
    float qy = xy ? q1 : q0;
    float qx = xy ? q0 : q1;
    float fy = fminf(fmaxf(0.0f, floorf(qy)), (float)(height - 2));
    float fx = fminf(fmaxf(0.0f, floorf(qx)), (float)(width - 2));
    // NaN queries: fmaxf(0, NaN) = 0 keeps the gather in range; alpha stays NaN -> NaN out.
    int iy = (int)fy, ix = (int)fx;
    float ay = qy - fy, ax = qx - fx;
    ay = (ay != ay) ? ay : fminf(fmaxf(0.0f, ay), 1.0f);
    ax = (ax != ax) ? ax : fminf(fmaxf(0.0f, ax), 1.0f);
    float tl = G[((int64_t)iy * width + ix) * channels + c];
    float tr = G[((int64_t)iy * width + ix + 1) * channels + c];
    float bl = G[((int64_t)(iy + 1) * width + ix) * channels + c];
    float br = G[((int64_t)(iy + 1) * width + ix + 1) * channels + c];
    float top = ax * (tr - tl) + tl;
    float bot = ax * (br - bl) + bl;
    O[t] = ay * (bot - top) + top;
  }
}

// ---------------------------------------------------------------- coordinate generators
__global__ void __launch_bounds__(kBlock)
rotate_coords_kernel(const float* __restrict__ rays, const float* __restrict__ matrix, int64_t q,
                     int src_h, int src_w, float* __restrict__ out) {
  const int b = blockIdx.y;
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = matrix[b * 9 + i];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < q;
       i += (int64_t)gridDim.x * kBlock) {
    float r0 = rays[i], r1 = rays[q + i], r2 = rays[2 * q + i];
    float x = (m[0] * r0 + m[1] * r1) + m[2] * r2;
    float y = (m[3] * r0 + m[4] * r1) + m[5] * r2;
    float z = (m[6] * r0 + m[7] * r1) + m[8] * r2;
    float pitch = se3ds_acosf(-y);
    float heading = se3ds_atan2f(x, z);
    float hp = (heading / SE3DS_F32_TWO_PI + 0.5f) * (float)(src_w - 1);
    float pp = pitch / SE3DS_F32_PI * (float)(src_h - 1);
    out[((int64_t)b * q + i) * 2 + 0] = pp;
    out[((int64_t)b * q + i) * 2 + 1] = hp;
  }
}

__global__ void __launch_bounds__(kBlock)
perspective_coords_kernel(const float* __restrict__ rays, const float* __restrict__ w2i, int64_t q,
                          int round_nearest, float add, float* __restrict__ out) {
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = w2i[i];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < q;
       i += (int64_t)gridDim.x * kBlock) {
    float r0 = rays[i], r1 = rays[q + i], r2 = rays[2 * q + i];
    float x = (m[0] * r0 + m[1] * r1) + m[2] * r2;
    float y = (m[3] * r0 + m[4] * r1) + m[5] * r2;
    float z = (m[6] * r0 + m[7] * r1) + m[8] * r2;
    float cx = z > 0.0f ? x / z : -1.0f;
    float cy = z > 0.0f ? y / z : -1.0f;
    if (round_nearest) {
      cx = rintf(cx);
      cy = rintf(cy);
    }
    out[i * 2 + 0] = cx + add;
    out[i * 2 + 1] = cy + add;
  }
}

__global__ void __launch_bounds__(kBlock)
persp_from_equirect_coords_kernel(const float* __restrict__ kinv_t, const float* __restrict__ rot,
                                  int height, int width, int eq_h, int eq_w,
                                  float* __restrict__ out) {
  float k[9], r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    k[i] = kinv_t[i];
    r[i] = rot[i];
  }
  const int64_t total = (int64_t)height * width;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    float px = (float)(i % width), py = (float)(i / width), pz = 1.0f;
    // row vector times K^-T, then times R (pano_utils.py:464)
    float a0 = (px * k[0] + py * k[3]) + pz * k[6];
    float a1 = (px * k[1] + py * k[4]) + pz * k[7];
    float a2 = (px * k[2] + py * k[5]) + pz * k[8];
    float x = (a0 * r[0] + a1 * r[3]) + a2 * r[6];
    float y = (a0 * r[1] + a1 * r[4]) + a2 * r[7];
    float z = (a0 * r[2] + a1 * r[5]) + a2 * r[8];
    float norm = __builtin_sqrtf((x * x + y * y) + z * z);
    float xn = x / norm, yn = y / norm, zn = z / norm;
    float lon = se3ds_atan2f(xn, zn);
    float lat = se3ds_asinf(yn);
    out[i * 2 + 0] = (lon / SE3DS_F32_TWO_PI + 0.5f) * (float)(eq_w - 1);
    out[i * 2 + 1] = (lat / SE3DS_F32_PI + 0.5f) * (float)(eq_h - 1);
  }
}

// ---------------------------------------------------------------- fused perspective paths
// SURVEY 8f-4 (RE10K use case, notebooks/SE3DS_RE10K_Colab.ipynb cells 15 / 17).  The op-by-op
// chains write two 1024x2048 equirect intermediates (RGB 25 MB, depth 8 MB) just to read them back
// once; the fused kernels below evaluate the same arithmetic per output element (one rounding
// per reference op, same order) and never materialise them.

// tfa.image.interpolate_bilinear on one query (indexing 'xy': query = (x, y)) of a virtual
// (hp, wp) grid; returns the corner indices and the clamped weights exactly as
// interp_bilinear_kernel computes them.
__device__ __forceinline__ void bilinear_setup(float qx, float qy, int hp, int wp, int* iy, int* ix,
                                               float* ay, float* ax) {
  float fy = fminf(fmaxf(0.0f, floorf(qy)), (float)(hp - 2));
  float fx = fminf(fmaxf(0.0f, floorf(qx)), (float)(wp - 2));
  *iy = (int)fy;
  *ix = (int)fx;
  float a = qy - fy, b = qx - fx;
  *ay = (a != a) ? a : fminf(fmaxf(0.0f, a), 1.0f);
  *ax = (b != b) ? b : fminf(fmaxf(0.0f, b), 1.0f);
}
__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, float ax, float ay) {
  float top = ax * (tr - tl) + tl;
  float bot = ax * (br - bl) + bl;
  return ay * (bot - top) + top;
}

// Cell 15: project_perspective_image(rgb) and (depth) [pano_utils.py:344-417, constant padding],
// int32(rgb * 255), equirectangular_to_pointcloud [:164-242] (+ position) in one pass over the
// equirect grid.  image (h,w,C<=4) fp32, depth (h,w) fp32; rays (3,Q); w2i = K.R (3x3).
__global__ void __launch_bounds__(kBlock)
perspective_to_pointcloud_kernel(const float* __restrict__ image, const float* __restrict__ depth,
                                 int ih, int iw, int channels, const float* __restrict__ rays,
                                 const float* __restrict__ w2i, int round_nearest, float pad_value,
                                 const float* __restrict__ sin_el, const float* __restrict__ cos_el,
                                 const float* __restrict__ sin_hd, const float* __restrict__ cos_hd,
                                 const float* __restrict__ position, int height, int width,
                                 float void_class, float depth_scale, float* __restrict__ xyz1,
                                 int32_t* __restrict__ feats_out) {
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = w2i[i];
  const int64_t q = (int64_t)height * width;
  const int hp = ih + 2, wp = iw + 2;
  float pos_x = 0.f, pos_y = 0.f, pos_z = 0.f;
  if (position) { pos_x = position[0]; pos_y = position[1]; pos_z = position[2]; }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < q;
       i += (int64_t)gridDim.x * kBlock) {
    const float r0 = rays[i], r1 = rays[q + i], r2 = rays[2 * q + i];
    const float x = (m[0] * r0 + m[1] * r1) + m[2] * r2;
    const float y = (m[3] * r0 + m[4] * r1) + m[5] * r2;
    const float z = (m[6] * r0 + m[7] * r1) + m[8] * r2;
    float cx = z > 0.0f ? x / z : -1.0f;
    float cy = z > 0.0f ? y / z : -1.0f;
    if (round_nearest) { cx = rintf(cx); cy = rintf(cy); }
    cx = cx + 1.0f;   // account for the 1-pixel constant padding
    cy = cy + 1.0f;
    int iy, ix;
    float ay, ax;
    bilinear_setup(cx, cy, hp, wp, &iy, &ix, &ay, &ax);
    // taps of the virtual padded image
    auto inside = [&](int yy, int xx) { return yy >= 1 && yy <= ih && xx >= 1 && xx <= iw; };
    const bool in00 = inside(iy, ix), in01 = inside(iy, ix + 1);
    const bool in10 = inside(iy + 1, ix), in11 = inside(iy + 1, ix + 1);
    const int64_t o00 = (int64_t)(iy - 1) * iw + (ix - 1);
    const float d = bilerp(in00 ? depth[o00] : pad_value, in01 ? depth[o00 + 1] : pad_value,
                           in10 ? depth[o00 + iw] : pad_value, in11 ? depth[o00 + iw + 1] : pad_value,
                           ax, ay);
    const bool valid = (d > 0.0f) && (d < 1.0f);
    const float mask = valid ? 1.0f : 0.0f;
    const int r = (int)(i / width), col = (int)(i - (int64_t)r * width);
    const float rad = (d * depth_scale) * mask;
    const float rs = rad * sin_el[r];
    float px = rs * cos_hd[col], py = rs * sin_hd[col], pz = rad * cos_el[r];
    if (position) { px = px + pos_x; py = py + pos_y; pz = pz + pos_z; }
    xyz1[i] = px;
    xyz1[q + i] = py;
    xyz1[2 * q + i] = pz;
    xyz1[3 * q + i] = 1.0f;
    for (int c = 0; c < channels; ++c) {
      const float v = bilerp(in00 ? image[o00 * channels + c] : pad_value,
                             in01 ? image[(o00 + 1) * channels + c] : pad_value,
                             in10 ? image[(o00 + iw) * channels + c] : pad_value,
                             in11 ? image[(o00 + iw + 1) * channels + c] : pad_value, ax, ay);
      // tf.cast(rgb * 255, tf.int32): truncation
      feats_out[i * channels + c] = valid ? (int32_t)(v * 255.0f) : (int32_t)void_class;
    }
  }
}

// Cell 17: three get_perspective_from_equirectangular_image calls [pano_utils.py:443-476] -- RGB,
// depth and the validity mask (depth != 0, != 1, all(rgb != 0)) -- plus the guidance glue:
// rgb / 255 clipped to [0,1], mask == 1, products with the mask.  pred_rgb (H,W,3), pred_depth
// (H,W) are the splat outputs; outputs are the generator's proj_image / proj_depth / proj_mask.
__global__ void __launch_bounds__(kBlock)
perspective_guidance_kernel(const float* __restrict__ pred_rgb, const float* __restrict__ pred_depth,
                            int eq_h, int eq_w, const float* __restrict__ kinv_t,
                            const float* __restrict__ rot, int height, int width,
                            float* __restrict__ proj_image, float* __restrict__ proj_depth,
                            float* __restrict__ proj_mask) {
  float k[9], r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { k[i] = kinv_t[i]; r[i] = rot[i]; }
  const int64_t total = (int64_t)height * width;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const float px = (float)(i % width), py = (float)(i / width), pz = 1.0f;
    const float a0 = (px * k[0] + py * k[3]) + pz * k[6];
    const float a1 = (px * k[1] + py * k[4]) + pz * k[7];
    const float a2 = (px * k[2] + py * k[5]) + pz * k[8];
    const float x = (a0 * r[0] + a1 * r[3]) + a2 * r[6];
    const float y = (a0 * r[1] + a1 * r[4]) + a2 * r[7];
    const float z = (a0 * r[2] + a1 * r[5]) + a2 * r[8];
    const float norm = __builtin_sqrtf((x * x + y * y) + z * z);
    const float xn = x / norm, yn = y / norm, zn = z / norm;
    const float lon = se3ds_atan2f(xn, zn);
    const float lat = se3ds_asinf(yn);
    const float u = (lon / SE3DS_F32_TWO_PI + 0.5f) * (float)(eq_w - 1);
    const float v = (lat / SE3DS_F32_PI + 0.5f) * (float)(eq_h - 1);
    int iy, ix;
    float ay, ax;
    bilinear_setup(u, v, eq_h, eq_w, &iy, &ix, &ay, &ax);
    const int64_t o = (int64_t)iy * eq_w + ix;
    const int64_t taps[4] = {o, o + 1, o + eq_w, o + eq_w + 1};
    float dt[4], mk[4], rgb[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      dt[t] = pred_depth[taps[t]];
      bool ok = (dt[t] != 1.0f) && (dt[t] != 0.0f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        rgb[t][c] = pred_rgb[taps[t] * 3 + c];
        ok = ok && (rgb[t][c] != 0.0f);
      }
      mk[t] = ok ? 1.0f : 0.0f;
    }
    const float gm = bilerp(mk[0], mk[1], mk[2], mk[3], ax, ay);
    const float pm = (gm == 1.0f) ? 1.0f : 0.0f;
    const float gd = bilerp(dt[0], dt[1], dt[2], dt[3], ax, ay);
    proj_mask[i] = pm;
    proj_depth[i] = pm * gd;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = bilerp(rgb[0][c], rgb[1][c], rgb[2][c], rgb[3][c], ax, ay);
      proj_image[i * 3 + c] = pm * fminf(fmaxf(g / 255.0f, 0.0f), 1.0f);
    }
  }
}

// ------------------------------------------------------------------------------ mask_pano
template <typename T>
__global__ void __launch_bounds__(kBlock)
mask_pano_kernel(const T* __restrict__ pano, int height, int64_t row_elems, int masked_height,
                 T value, int64_t total, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    int r = (int)((i / row_elems) % height);
    bool keep = (r >= masked_height) && (r <= height - masked_height);
    out[i] = keep ? pano[i] : value;
  }
}

// ------------------------------------------------------------------------- compaction
// ------------------------------------------------------------------------------ tf.image.resize
// Half-pixel centres (TF 2.x): nearest src = floor((i + 0.5) * in / out), bilinear src =
// (i + 0.5) * in / out - 0.5 with the taps clamped to the image and the lerp written as
// a + (b - a) * t; every fp32 op rounded on its own (-ffp-contract=off), as oracle/warp_np.py does.
// Callers: equirectangular_to_pointcloud(size_mult != 1) (pano_utils.py:203-208) and
// crop_pano(resize_to_original=True) (:299-301; its antialias=True only changes DOWN-scaling).
template <typename T>
__global__ void __launch_bounds__(kBlock)
resize_nearest_kernel(const T* __restrict__ x, int n, int h, int w, int c, int oh, int ow,
                      T* __restrict__ y) {
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  const int64_t total = (int64_t)n * oh * ow * c;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int k = (int)(i % c);
    const int64_t p = i / c;
    const int ox = (int)(p % ow);
    const int64_t q = p / ow;
    const int oy = (int)(q % oh), b = (int)(q / oh);
    int iy = (int)floorf(((float)oy + 0.5f) * sy), ix = (int)floorf(((float)ox + 0.5f) * sx);
    iy = iy < h - 1 ? iy : h - 1;
    ix = ix < w - 1 ? ix : w - 1;
    y[i] = x[(((int64_t)b * h + iy) * w + ix) * c + k];
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
resize_bilinear_kernel(const T* __restrict__ x, int n, int h, int w, int c, int oh, int ow,
                       float* __restrict__ y) {
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  const int64_t total = (int64_t)n * oh * ow * c;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int k = (int)(i % c);
    const int64_t p = i / c;
    const int ox = (int)(p % ow);
    const int64_t q = p / ow;
    const int oy = (int)(q % oh), b = (int)(q / oh);
    const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    const float ly = floorf(fy), lx = floorf(fx);
    const float ty = fy - ly, tx = fx - lx;
    int y0 = (int)ly, x0 = (int)lx, y1 = y0 + 1, x1 = x0 + 1;
    y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
    x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
    const T* base = x + (int64_t)b * h * w * c + k;
    const float v00 = (float)base[((int64_t)y0 * w + x0) * c], v01 = (float)base[((int64_t)y0 * w + x1) * c];
    const float v10 = (float)base[((int64_t)y1 * w + x0) * c], v11 = (float)base[((int64_t)y1 * w + x1) * c];
    const float top = v00 + (v01 - v00) * tx, bot = v10 + (v11 - v10) * tx;
    y[i] = top + (bot - top) * ty;
  }
}

// mean of an fp32 array in binary64 (one workgroup; pad_mode='mean' of project_perspective_image,
// pano_utils.py:403-407: a few hundred thousand elements)
__global__ void __launch_bounds__(1024)
mean_f32_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ double s_p[1024];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += (double)x[i];
  s_p[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_p[threadIdx.x] += s_p[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(s_p[0] / (double)n);
}

// flags -> per-block counts -> exclusive scan of block counts (one block) -> scatter.
constexpr int kCBlock = 256;
constexpr int kCItems = 4;  // points per thread
constexpr int kCTile = kCBlock * kCItems;

template <typename T>
__device__ __forceinline__ bool point_valid(const T* feats, int n, int64_t m, int channels,
                                            int64_t j, T vc) {
  bool any = false;
  for (int b = 0; b < n; ++b)
    for (int k = 0; k < channels; ++k) any |= (feats[((int64_t)b * m + j) * channels + k] != vc);
  return any;
}

template <typename T>
__global__ void __launch_bounds__(kCBlock)
compact_count_kernel(const T* __restrict__ feats, int n, int64_t m, int channels, float void_class,
                     uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const T vc = FeatIO<T>::cast_void(void_class);
  uint32_t c = 0;
  int64_t base = (int64_t)blockIdx.x * kCTile;
  for (int it = 0; it < kCItems; ++it) {
    int64_t j = base + it * kCBlock + threadIdx.x;
    if (j < m && point_valid(feats, n, m, channels, j, vc)) ++c;
  }
  c = (uint32_t)wave_sum((float)c);  // <= 256 per wave: exact in fp32
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt;
}

__global__ void __launch_bounds__(1024)
compact_scan_kernel(uint32_t* __restrict__ block_counts, int64_t nblocks,
                    int64_t* __restrict__ count_out) {
  // single block: serial chunks + Hillis-Steele over 1024 partial sums
  __shared__ uint64_t s[1024];
  int64_t per = (nblocks + 1023) / 1024;
  int64_t lo = (int64_t)threadIdx.x * per;
  int64_t hi = lo + per < nblocks ? lo + per : nblocks;
  uint64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += block_counts[i];
  s[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    uint64_t v = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
    __syncthreads();
    s[threadIdx.x] += v;
    __syncthreads();
  }
  uint64_t excl = s[threadIdx.x] - sum;
  for (int64_t i = lo; i < hi; ++i) {
    uint32_t c = block_counts[i];
    block_counts[i] = (uint32_t)excl;
    excl += c;
  }
  if (threadIdx.x == 1023) *count_out = (int64_t)s[1023];
}

template <typename T>
__global__ void __launch_bounds__(kCBlock)
compact_scatter_kernel(const float* __restrict__ xyz1, const T* __restrict__ feats, int n,
                       int64_t m, int channels, float void_class,
                       const uint32_t* __restrict__ block_offsets, float* __restrict__ xyz1_out,
                       T* __restrict__ feats_out) {
  __shared__ uint32_t s_wave[kCItems][kCBlock / 64];
  const T vc = FeatIO<T>::cast_void(void_class);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t base = (int64_t)blockIdx.x * kCTile;
  bool valid[kCItems];
  uint32_t rank[kCItems];
  for (int it = 0; it < kCItems; ++it) {
    int64_t j = base + it * kCBlock + threadIdx.x;
    valid[it] = j < m && point_valid(feats, n, m, channels, j, vc);
    unsigned long long ball = __ballot(valid[it]);
    rank[it] = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[it][wave] = __popcll(ball);
  }
  __syncthreads();
  uint32_t run = block_offsets[blockIdx.x];
  for (int it = 0; it < kCItems; ++it) {
    uint32_t before = 0;
    for (int w = 0; w < kCBlock / 64; ++w) {
      uint32_t c = s_wave[it][w];
      if (w < wave) before += c;
    }
    uint32_t total = 0;
    for (int w = 0; w < kCBlock / 64; ++w) total += s_wave[it][w];
    if (valid[it]) {
      int64_t j = base + it * kCBlock + threadIdx.x;
      int64_t dst = (int64_t)run + before + rank[it];
      for (int b = 0; b < n; ++b) {
        for (int r = 0; r < 4; ++r)
          xyz1_out[((int64_t)b * 4 + r) * m + dst] = xyz1[((int64_t)b * 4 + r) * m + j];
        for (int k = 0; k < channels; ++k)
          feats_out[((int64_t)b * m + dst) * channels + k] =
              feats[((int64_t)b * m + j) * channels + k];
      }
    }
    run += total;
  }
}

template <typename T>
int launch_compact(const float* xyz1, const T* feats, int n, int64_t m, int channels,
                   float void_class, float* xyz1_out, T* feats_out, int64_t* count_out,
                   void* workspace, hipStream_t s) {
  uint32_t* counts = (uint32_t*)workspace;
  int64_t nblocks = ceil_div(m, kCTile);
  if (nblocks == 0) {
    (void)hipMemsetAsync(count_out, 0, sizeof(int64_t), s);
    return check_launch("compact");
  }
  hipLaunchKernelGGL(compact_count_kernel<T>, dim3((unsigned)nblocks), dim3(kCBlock), 0, s, feats,
                     n, m, channels, void_class, counts);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, counts, nblocks, count_out);
  hipLaunchKernelGGL(compact_scatter_kernel<T>, dim3((unsigned)nblocks), dim3(kCBlock), 0, s, xyz1,
                     feats, n, m, channels, void_class, counts, xyz1_out, feats_out);
  return check_launch("compact");
}

}  // namespace
}  // namespace se3ds

using namespace se3ds;

extern "C" {

int se3ds_unproject_equirect_into(const void* feats, int feat_dtype, const float* depth,
                                  const float* sin_el, const float* cos_el, const float* sin_hd,
                                  const float* cos_hd, const float* position, int n, int height,
                                  int width, int channels, float void_class, float depth_scale,
                                  float* xyz1, void* feats_out, int64_t m_total, int64_t m_offset,
                                  void* stream) {
  if (n <= 0 || height <= 0 || width <= 0 || channels <= 0) return SE3DS_E_BADSHAPE;
  // SE3DS_XYZ1_ONES_PRESET: row 3 of the window already holds 1.0 (a memory fills it at allocation)
  const int write_ones = (feat_dtype & SE3DS_XYZ1_ONES_PRESET) ? 0 : 1;
  feat_dtype &= ~SE3DS_XYZ1_ONES_PRESET;
  hipStream_t s = as_stream(stream);
  int64_t p = (int64_t)height * width;
  if (m_offset < 0 || m_offset + p > m_total) return SE3DS_E_BADSHAPE;
  dim3 grid((unsigned)grid_for(p, kBlock), (unsigned)n);
  // four pixels per thread, 16-byte accesses (SE3DS_UNPROJECT_VEC=0: the scalar kernel, A/B and tests)
  const char* e_vec = getenv("SE3DS_UNPROJECT_VEC");   // (read per call: the parity test switches it)
  const bool no_vec = e_vec && atoi(e_vec) == 0;
  const bool aligned = (width % 4) == 0 && (m_total % 4) == 0 && (m_offset % 4) == 0 &&
      (((uintptr_t)feats | (uintptr_t)depth | (uintptr_t)xyz1 | (uintptr_t)feats_out |
        (uintptr_t)sin_hd | (uintptr_t)cos_hd) & 15) == 0 && p < ((int64_t)1 << 31);
  if (!no_vec && aligned && (feat_dtype == SE3DS_F32 || feat_dtype == SE3DS_I32) &&
      (channels == 1 || channels == 3)) {
    dim3 g4((unsigned)grid_for(p / 4, kBlock), (unsigned)n);
#define SE3DS_U4(T, CC)                                                                          \
    hipLaunchKernelGGL((unproject_equirect_vec4_kernel<T, CC>), g4, dim3(kBlock), 0, s,          \
                       (const T*)feats, depth, sin_el, cos_el, sin_hd, cos_hd, position, height, \
                       width, void_class, depth_scale, xyz1, (T*)feats_out, m_total, m_offset, write_ones)
    if (feat_dtype == SE3DS_F32) {
      if (channels == 3) SE3DS_U4(float, 3); else SE3DS_U4(float, 1);
    } else {
      if (channels == 3) SE3DS_U4(int32_t, 3); else SE3DS_U4(int32_t, 1);
    }
#undef SE3DS_U4
    return check_launch("unproject_equirect(vec4)");
  }
  switch (feat_dtype) {
    case SE3DS_F32:
      hipLaunchKernelGGL(unproject_equirect_kernel<float>, grid, dim3(kBlock), 0, s,
                         (const float*)feats, depth, sin_el, cos_el, sin_hd, cos_hd, position,
                         height, width, channels, void_class, depth_scale, xyz1, (float*)feats_out,
                         m_total, m_offset, write_ones);
      break;
    case SE3DS_I32:
      hipLaunchKernelGGL(unproject_equirect_kernel<int32_t>, grid, dim3(kBlock), 0, s,
                         (const int32_t*)feats, depth, sin_el, cos_el, sin_hd, cos_hd, position,
                         height, width, channels, void_class, depth_scale, xyz1,
                         (int32_t*)feats_out, m_total, m_offset, write_ones);
      break;
    case SE3DS_U8:
      if (void_class < 0.0f) return SE3DS_E_BADDTYPE;  // pano_utils.py:193-197
      hipLaunchKernelGGL(unproject_equirect_kernel<uint8_t>, grid, dim3(kBlock), 0, s,
                         (const uint8_t*)feats, depth, sin_el, cos_el, sin_hd, cos_hd, position,
                         height, width, channels, void_class, depth_scale, xyz1,
                         (uint8_t*)feats_out, m_total, m_offset, write_ones);
      break;
    default:
      return SE3DS_E_BADDTYPE;
  }
  return check_launch("unproject_equirect");
}

int se3ds_unproject_equirect(const void* feats, int feat_dtype, const float* depth,
                             const float* sin_el, const float* cos_el, const float* sin_hd,
                             const float* cos_hd, const float* position, int n, int height,
                             int width, int channels, float void_class, float depth_scale,
                             float* xyz1, void* feats_out, void* stream) {
  return se3ds_unproject_equirect_into(feats, feat_dtype, depth, sin_el, cos_el, sin_hd, cos_hd,
                                       position, n, height, width, channels, void_class,
                                       depth_scale, xyz1, feats_out, (int64_t)height * width, 0,
                                       stream);
}

size_t se3ds_splat_workspace_bytes(int n, int64_t m, int height, int width, int channels) {
  const size_t base = splat_hdr_bytes() +
                      (sizeof(int32_t) + sizeof(float)) * (size_t)n * (size_t)(m > 0 ? m : 0);
  const int c = channels > 0 ? channels : 1;
  const size_t three_pass = bin_ws_bytes(n, m, height, width, c);
  const size_t packed = pack_ws_bytes(n, m, height, width);
  const size_t sorted = sort_ws_bytes(n, m, height, width);
  size_t bins = three_pass;
  if (packed > bins) bins = packed;
  if (sorted > bins) bins = sorted;
  return align16(base) + bins + 16;
}

int se3ds_project_equirect(const float* xyz1, const float* offset, const void* feats,
                           int feat_dtype, int n, int64_t m, int channels, int height, int width,
                           float depth_scale, float input_void, float output_void, float* depth,
                           float* feat, float* mask, float mask_void, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return dispatch_splat<true>(xyz1, offset, feats, feat_dtype, n, m, m, channels, height, width,
                              depth_scale, input_void, output_void, depth, feat, mask, mask_void,
                              workspace, workspace_bytes, stream);
}

int se3ds_project_equirect_memory(const float* xyz1, const float* offset, const void* feats,
                                  int feat_dtype, int n, int64_t m, int64_t capacity, int channels,
                                  int height, int width, float depth_scale, float input_void,
                                  float output_void, float* depth, float* feat, float* mask,
                                  float mask_void, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  return dispatch_splat<true>(xyz1, offset, feats, feat_dtype, n, m, capacity, channels, height,
                              width, depth_scale, input_void, output_void, depth, feat, mask,
                              mask_void, workspace, workspace_bytes, stream);
}

int se3ds_warp_views_to_target(const void* const* view_feats, int feat_dtype,
                               const float* const* view_depth, const float* const* view_position,
                               int views, int n, int height, int width, int channels,
                               float void_class, float depth_scale, const float* sin_el,
                               const float* cos_el, const float* sin_hd, const float* cos_hd,
                               float* mem_xyz1, void* mem_feats, int64_t capacity, int64_t m_offset,
                               const float* target, int out_height, int out_width,
                               float output_void, float* depth, float* feat, float* mask,
                               float mask_void, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (views < 0 || n <= 0 || height <= 0 || width <= 0) return SE3DS_E_BADSHAPE;
  const int64_t p = (int64_t)height * width;
  if (m_offset < 0 || m_offset + (int64_t)views * p > capacity) return SE3DS_E_BADSHAPE;
  const int dt = feat_dtype & ~SE3DS_FEAT_BYTE_RANGE;   // (keeps SE3DS_XYZ1_ONES_PRESET for the unprojects)
  feat_dtype &= ~SE3DS_XYZ1_ONES_PRESET;
  for (int v = 0; v < views; ++v) {
    const int rc = se3ds_unproject_equirect_into(
        view_feats[v], dt, view_depth[v], sin_el, cos_el, sin_hd, cos_hd,
        view_position ? view_position[v] : nullptr, n, height, width, channels, void_class,
        depth_scale, mem_xyz1, mem_feats, capacity, m_offset + (int64_t)v * p, stream);
    if (rc != SE3DS_OK) return rc;
  }
  return dispatch_splat<true>(mem_xyz1, target, mem_feats, feat_dtype, n, m_offset + (int64_t)views * p,
                              capacity, channels, out_height, out_width, depth_scale, void_class,
                              output_void, depth, feat, mask, mask_void, workspace, workspace_bytes,
                              stream);
}

int se3ds_project_to_feat(const float* coords, const void* feats, int feat_dtype, int n, int64_t m,
                          int channels, int height, int width, float depth_scale,
                          float input_void, float output_void, float* depth, float* feat,
                          float* mask, float mask_void, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return dispatch_splat<false>(coords, nullptr, feats, feat_dtype, n, m, m, channels, height, width,
                               depth_scale, input_void, output_void, depth, feat, mask, mask_void,
                               workspace, workspace_bytes, stream);
}

int se3ds_feats_byte_range(const void* feats, int feat_dtype, int64_t count, float void_class,
                           uint32_t* bad_out, void* stream) {
  if (count < 0 || !bad_out) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(bad_out, 0, sizeof(uint32_t), s) != hipSuccess) return SE3DS_E_LAUNCH;
  if (count == 0) return SE3DS_OK;
  const dim3 g((unsigned)grid_for(count, kBlock));
  if (feat_dtype == SE3DS_I32)
    hipLaunchKernelGGL(feats_byte_range_kernel<int32_t>, g, dim3(kBlock), 0, s,
                       (const int32_t*)feats, count, void_class, bad_out);
  else if (feat_dtype == SE3DS_U8)
    hipLaunchKernelGGL(feats_byte_range_kernel<uint8_t>, g, dim3(kBlock), 0, s,
                       (const uint8_t*)feats, count, void_class, bad_out);
  else
    return SE3DS_E_BADDTYPE;
  return check_launch("feats_byte_range");
}

int se3ds_splat_promise_broken(const void* workspace, int n, int64_t m, uint32_t* broken_out,
                               void* stream) {
  if (n <= 0 || m <= 0 || !broken_out) return SE3DS_E_BADSHAPE;
  const size_t base = splat_hdr_bytes() + (sizeof(int32_t) + sizeof(float)) * (size_t)n * (size_t)m;
  PackWs pw;
  pw.ctl = (uint32_t*)((char*)const_cast<void*>(workspace) + align16(base));
  hipLaunchKernelGGL(splat_promise_kernel, dim3(1), dim3(1), 0, as_stream(stream), pw, broken_out);
  return check_launch("splat_promise_broken");
}

int se3ds_splat_promise_sticky(void* workspace, uint32_t* broken_out, int clear, void* stream) {
  if (!workspace || !broken_out) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(splat_promise_sticky_kernel, dim3(1), dim3(1), 0, as_stream(stream),
                     (uint32_t*)workspace, broken_out, clear);
  return check_launch("splat_promise_sticky");
}

int se3ds_splat_debug_indices(const void* workspace, int n, int64_t m, int32_t* idx_out,
                              float* z_out, void* stream) {
  if (n <= 0 || m <= 0) return SE3DS_OK;
  SplatWs ws = carve_ws(const_cast<void*>(workspace), n, m);
  int64_t total = (int64_t)n * m;
  hipLaunchKernelGGL(splat_debug_copy_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), ws, total, idx_out, z_out);
  return check_launch("splat_debug_indices");
}

int se3ds_debug_fast_fxy(const float* xyz, int64_t m, int width, int height, float* fx, float* fy,
                         int32_t* verdict, void* stream) {
  if (m <= 0) return SE3DS_OK;
  hipLaunchKernelGGL(debug_fast_fxy_kernel, dim3(grid_for(m, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), xyz, m, width, height, fx, fy, verdict);
  return check_launch("debug_fast_fxy");
}

int se3ds_unproject_perspective(const int32_t* feats, const float* depth, const float* xs,
                                const float* ys, const float* kinv, int n, int height, int width,
                                int channels, float depth_scale, float* xyz, float* feats_out,
                                void* stream) {
  if (n <= 0 || height <= 0 || width <= 0 || channels <= 0) return SE3DS_E_BADSHAPE;
  int64_t p = (int64_t)height * width;
  dim3 grid((unsigned)grid_for(p, kBlock), (unsigned)n);
  hipLaunchKernelGGL(unproject_perspective_kernel, grid, dim3(kBlock), 0, as_stream(stream), feats,
                     depth, xs, ys, kinv, height, width, channels, depth_scale, xyz, feats_out);
  return check_launch("unproject_perspective");
}

int se3ds_interp_bilinear(const float* grid, const float* query, int b, int height, int width,
                          int channels, int64_t q, int indexing_xy, float* out, void* stream) {
  if (b <= 0 || height < 2 || width < 2 || channels <= 0 || q < 0) return SE3DS_E_BADSHAPE;
  if (q == 0) return SE3DS_OK;
  dim3 g((unsigned)grid_for(q * channels, kBlock), (unsigned)b);
  hipLaunchKernelGGL(interp_bilinear_kernel, g, dim3(kBlock), 0, as_stream(stream), grid, query,
                     height, width, channels, q, indexing_xy, out);
  return check_launch("interp_bilinear");
}

int se3ds_rotate_coords(const float* rays, const float* matrix, int n, int64_t q, int src_h,
                        int src_w, float* out, void* stream) {
  if (n <= 0 || q <= 0) return SE3DS_E_BADSHAPE;
  dim3 g((unsigned)grid_for(q, kBlock), (unsigned)n);
  hipLaunchKernelGGL(rotate_coords_kernel, g, dim3(kBlock), 0, as_stream(stream), rays, matrix, q,
                     src_h, src_w, out);
  return check_launch("rotate_coords");
}

int se3ds_perspective_coords(const float* rays, const float* w2i, int64_t q, int round_nearest,
                             float add, float* out, void* stream) {
  if (q <= 0) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(perspective_coords_kernel, dim3(grid_for(q, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), rays, w2i, q, round_nearest, add, out);
  return check_launch("perspective_coords");
}

int se3ds_persp_from_equirect_coords(const float* kinv_t, const float* rot, int height, int width,
                                     int eq_h, int eq_w, float* out, void* stream) {
  if (height <= 0 || width <= 0) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(persp_from_equirect_coords_kernel,
                     dim3(grid_for((int64_t)height * width, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), kinv_t, rot, height, width, eq_h, eq_w, out);
  return check_launch("persp_from_equirect_coords");
}

int se3ds_perspective_to_pointcloud(const float* image, const float* depth, int ih, int iw,
                                    int channels, const float* rays, const float* w2i,
                                    int round_nearest, float pad_value, const float* sin_el,
                                    const float* cos_el, const float* sin_hd, const float* cos_hd,
                                    const float* position, int height, int width, float void_class,
                                    float depth_scale, float* xyz1, int32_t* feats_out, void* stream) {
  if (ih <= 0 || iw <= 0 || channels <= 0 || channels > 4 || height <= 0 || width != 2 * height)
    return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(perspective_to_pointcloud_kernel,
                     dim3(grid_for((int64_t)height * width, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), image, depth, ih, iw, channels, rays, w2i, round_nearest,
                     pad_value, sin_el, cos_el, sin_hd, cos_hd, position, height, width, void_class,
                     depth_scale, xyz1, feats_out);
  return check_launch("perspective_to_pointcloud");
}

int se3ds_perspective_guidance(const float* pred_rgb, const float* pred_depth, int eq_h, int eq_w,
                               const float* kinv_t, const float* rot, int height, int width,
                               float* proj_image, float* proj_depth, float* proj_mask,
                               void* stream) {
  if (eq_h < 2 || eq_w < 2 || height <= 0 || width <= 0) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(perspective_guidance_kernel, dim3(grid_for((int64_t)height * width, kBlock)),
                     dim3(kBlock), 0, as_stream(stream), pred_rgb, pred_depth, eq_h, eq_w, kinv_t,
                     rot, height, width, proj_image, proj_depth, proj_mask);
  return check_launch("perspective_guidance");
}

int se3ds_resize(const void* x, int dtype, int n, int h, int w, int c, int oh, int ow, int method,
                 void* y, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0) return SE3DS_E_BADSHAPE;
  if (method != 0 && method != 1) return SE3DS_E_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const dim3 g((unsigned)grid_for((int64_t)n * oh * ow * c, kBlock));
#define SE3DS_RS(T)                                                                               \
  if (method == 0)                                                                                \
    hipLaunchKernelGGL(resize_nearest_kernel<T>, g, dim3(kBlock), 0, s, (const T*)x, n, h, w, c,  \
                       oh, ow, (T*)y);                                                            \
  else                                                                                            \
    hipLaunchKernelGGL(resize_bilinear_kernel<T>, g, dim3(kBlock), 0, s, (const T*)x, n, h, w, c, \
                       oh, ow, (float*)y)
  switch (dtype) {
    case SE3DS_F32: SE3DS_RS(float); break;
    case SE3DS_I32: SE3DS_RS(int32_t); break;
    case SE3DS_U8: SE3DS_RS(uint8_t); break;
    default: return SE3DS_E_BADDTYPE;
  }
#undef SE3DS_RS
  return check_launch("resize");
}

int se3ds_mean_f32(const float* x, int64_t n, float* out, void* stream) {
  if (n <= 0 || !out) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(mean_f32_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, n, out);
  return check_launch("mean_f32");
}

int se3ds_mask_pano(const void* pano, int dtype, int n, int height, int width, int channels,
                    int masked_height, float value, void* out, void* stream) {
  if (n <= 0 || height <= 0 || width <= 0 || channels <= 0) return SE3DS_E_BADSHAPE;
  int64_t row = (int64_t)width * channels;
  int64_t total = (int64_t)n * height * row;
  dim3 g(grid_for(total, kBlock));
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case SE3DS_F32:
      hipLaunchKernelGGL(mask_pano_kernel<float>, g, dim3(kBlock), 0, s, (const float*)pano, height,
                         row, masked_height, value, total, (float*)out);
      break;
    case SE3DS_I32:
      hipLaunchKernelGGL(mask_pano_kernel<int32_t>, g, dim3(kBlock), 0, s, (const int32_t*)pano,
                         height, row, masked_height, (int32_t)value, total, (int32_t*)out);
      break;
    case SE3DS_U8:
      hipLaunchKernelGGL(mask_pano_kernel<uint8_t>, g, dim3(kBlock), 0, s, (const uint8_t*)pano,
                         height, row, masked_height, (uint8_t)value, total, (uint8_t*)out);
      break;
    default:
      return SE3DS_E_BADDTYPE;
  }
  return check_launch("mask_pano");
}

size_t se3ds_compact_workspace_bytes(int64_t m) {
  return sizeof(uint32_t) * (size_t)(ceil_div(m > 0 ? m : 0, kCTile) + 1) + 16;
}

int se3ds_compact_valid(const float* xyz1, const void* feats, int feat_dtype, int n, int64_t m,
                        int channels, float void_class, float* xyz1_out, void* feats_out,
                        int64_t* count_out, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (n <= 0 || m < 0 || channels <= 0) return SE3DS_E_BADSHAPE;
  if (workspace_bytes < se3ds_compact_workspace_bytes(m)) return SE3DS_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  switch (feat_dtype) {
    case SE3DS_F32:
      return launch_compact<float>(xyz1, (const float*)feats, n, m, channels, void_class, xyz1_out,
                                   (float*)feats_out, count_out, workspace, s);
    case SE3DS_I32:
      return launch_compact<int32_t>(xyz1, (const int32_t*)feats, n, m, channels, void_class,
                                     xyz1_out, (int32_t*)feats_out, count_out, workspace, s);
    case SE3DS_U8:
      return launch_compact<uint8_t>(xyz1, (const uint8_t*)feats, n, m, channels, void_class,
                                     xyz1_out, (uint8_t*)feats_out, count_out, workspace, s);
    default:
      return SE3DS_E_BADDTYPE;
  }
}

#ifdef SE3DS_PROBE
/* probe builds only: the phase stamps of the last sorted splat (host buffer of 2 * 4096 * 12 uint64) */
int se3ds_geom_probe_read(uint64_t* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(se3ds::g_probe), sizeof(uint64_t) * 2 * 4096 * 12) == hipSuccess
             ? SE3DS_OK : SE3DS_E_LAUNCH;
}
#endif

}  // extern "C"
