// Normalisation kernels (HBM-bound): SyncBatchNorm / InstanceNorm statistics, apply with
// fused residual + activation, and the matching backward reductions.  A tensor is viewed
// as [G][R][C] (G = 1 for batch norm over N*H*W rows, G = N samples for instance norm).
// Statistics are accumulated in fp32 from bf16/fp32 activations with 16-byte loads, reduced
// deterministically in two stages (per-block partials -> final), so that the cross-replica
// all-reduce of [2][C] sums (SyncBN, image_models.py:80-123 etc.) sits between two launches.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace se3ds {
namespace {

struct Layout2D {
  int cx;    // threads along channel vectors (power of two <= 256)
  int ry;    // threads along rows = 256 / cx
  int vec;   // elements per thread along C (V or 1)
  int ctiles;
};
inline Layout2D make_layout(int C, int V) {
  Layout2D l;
  l.vec = (C % V == 0) ? V : 1;
  int cvec = C / l.vec;
  int cx = 1;
  while (cx < cvec && cx < 256) cx <<= 1;
  l.cx = cx;
  l.ry = 256 / cx;
  l.ctiles = (int)ceil_div(cvec, cx);
  return l;
}

// raw 16-byte load / unpack of 8 bf16 (the two-row kernels load everything first, THEN convert:
// with VT<T>::load the compiler schedules the conversion -- the first use -- right behind each
// load and waits for every load separately)
__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void unpack8(const uint4& v, float (&o)[8]) {
  const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[2 * e] = __uint_as_float(q[e] << 16);
    o[2 * e + 1] = __uint_as_float(q[e] & 0xffff0000u);
  }
}

// mode 0: sums of (x, x^2)                        [forward statistics / column sums]
// mode 1: sums of (dpre, dpre * xhat), dpre = dy * act'(y)   [backward statistics]
// Optional row_scale (rows of the [G*R] view) multiplies the first operand (x or dy).
// CG (round 5, bf16 MODE 1 only): the launch uses the CHANNEL-GROUP layout -- cx = 8 lanes x 8
// channels = 64 channels = one 128-byte line per row, ry = 32 rows -- and the 32 row partials of a
// block are folded by all 256 threads (two LDS rounds) instead of 8 threads walking 32 rows each.
template <typename T, int VEC, int MODE, bool U2 = false, int ACT = 0, bool CG = false>
__global__ void __launch_bounds__(256)
norm_partial_kernel(const T* __restrict__ a, const T* __restrict__ y, const T* __restrict__ x,
                    const float* __restrict__ mean, const float* __restrict__ rstd,
                    const float* __restrict__ row_scale, int64_t R, int C, int cx, int ry, int act,
                    float alpha, int rblocks, float* __restrict__ partial,
                    const uint8_t* __restrict__ amask, T* __restrict__ sc_out = nullptr,
                    const float* __restrict__ sc_row = nullptr) {
  // grid: (rblocks, ctiles, G)
  // sc_out / sc_row (MODE 0, bf16, VEC 8 only): the same pass also writes a * sc_row[row] -- the
  // partial convs' backward needs dy * (per-pixel renormalisation) for its LDS-DMA kernels AND the
  // column sums of dy * (another per-pixel factor) for the bias gradient: one read of dy, not two.
  // fp32 (parity) path: binary64 accumulators.  Batch-norm statistics enter as
  // var = E[x^2] - E[x]^2 (the reference's formula): with channel means of a few standard
  // deviations every 1e-6 of relative error in the sums becomes 1e-5 in the output, and on a
  // 16 M-element tensor behind a ReLU that flips the derivative of hundreds of elements against
  // any other fp32 implementation.  PyTorch's cascade summation is ~10x more accurate than fp32
  // partial sums of 8..128 rows + a 64-deep second stage; binary64 closes that gap for free (the
  // kernel is bound by its loads).  bf16 keeps fp32 accumulators.
  typedef typename std::conditional<sizeof(T) == 4, double, float>::type ACC;
  __shared__ ACC red[2][256 * (VEC > 1 ? VEC : 1)];
  const int g = blockIdx.z;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const int cv = blockIdx.y * cx + tx;  // channel-vector index
  const int c0 = cv * VEC;
  const bool cok = c0 < C;
  ACC s0[VEC], s1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s0[e] = 0; s1[e] = 0; }
  float mu[VEC], rs[VEC];
  if (MODE == 1 && cok) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mu[e] = mean[(int64_t)g * C + c0 + e]; rs[e] = rstd[(int64_t)g * C + c0 + e]; }
  }
  const int64_t rows_per_block = ceil_div(R, rblocks);
  const int64_t r_lo = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_hi = r_lo + rows_per_block;
  if (r_hi > R) r_hi = R;
  if (cok && MODE == 1 && VEC == 8 && U2) {   // (launched only when ACT matches `act`, no row scale)
    // NR rows per iteration: the 2 * NR 16-byte loads (+ NR mask bytes) are issued as raw loads
    // before anything is converted (load latency x occupancy bounds this kernel, not bytes: one
    // row in flight ran at 3.5 TB/s on a 1 GB tensor, two at 5.5)
    constexpr int NR = 4;
    const uint16_t* A = (const uint16_t*)a;
    const uint16_t* X = (const uint16_t*)x;
    for (int64_t r = r_lo + ty; r < r_hi; r += (int64_t)NR * ry) {
      uint4 qa[NR], qx[NR];
      unsigned mk[NR];
      bool ok[NR];
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        ok[u] = r + (int64_t)u * ry < r_hi;
        const int64_t off = ((int64_t)g * R + (ok[u] ? r + (int64_t)u * ry : r)) * C + c0;
        qa[u] = ld16(A + off);
        qx[u] = ld16(X + off);
        mk[u] = ACT != 0 ? amask[off >> 3] : 0xffu;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        if (!ok[u]) continue;
        float av[8], xv[8];
        unpack8(qa[u], av);
        unpack8(qx[u], xv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          // (ACT is the compile-time activation kind: 0 none, 1 relu, 2 leaky relu)
          const float d = (ACT == 0 || ((mk[u] >> e) & 1u)) ? av[e] : (ACT == 1 ? 0.0f : av[e] * alpha);
          s0[e] += d;
          s1[e] += d * ((xv[e] - mu[e]) * rs[e]);
        }
      }
    }
  } else if (cok && MODE == 0 && VEC == 8 && sizeof(T) == 2) {
    // forward statistics / (row-scaled) column sums, bf16: four rows per iteration, raw loads first
    // (bias-gradient sums of the partial convs and instance-norm statistics: ~170 launches per
    // step that ran one row at a time at ~1 TB/s)
    constexpr int NR = 4;
    const uint16_t* A = (const uint16_t*)a;
    for (int64_t r = r_lo + ty; r < r_hi; r += (int64_t)NR * ry) {
      uint4 qa[NR];
      float rsc[NR], osc[NR];
      bool ok[NR];
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        ok[u] = r + (int64_t)u * ry < r_hi;
        const int64_t row = (int64_t)g * R + (ok[u] ? r + (int64_t)u * ry : r);
        qa[u] = ld16(A + row * C + c0);
        rsc[u] = row_scale ? row_scale[row] : 1.0f;
        osc[u] = sc_row ? sc_row[row] : 1.0f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        if (!ok[u]) continue;
        float av[8];
        unpack8(qa[u], av);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const ACC v = (ACC)(av[e] * rsc[u]);
          s0[e] += v;
          s1[e] += v * v;
        }
        if (sc_out) {
          float o[8];
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = av[e] * osc[u];
          const int64_t row = (int64_t)g * R + r + (int64_t)u * ry;
          VT<T>::store(sc_out + row * C + c0, reinterpret_cast<float(&)[VT<T>::V]>(o));
        }
      }
    }
  } else if (cok) {
    for (int64_t r = r_lo + ty; r < r_hi; r += ry) {
      const int64_t off = ((int64_t)g * R + r) * C + c0;
      float av[VEC];
      if constexpr (VEC > 1) VT<T>::load(a + off, reinterpret_cast<float(&)[VT<T>::V]>(av));
      else av[0] = VT<T>::ld1(a + off);
      const float rsc = row_scale ? row_scale[(int64_t)g * R + r] : 1.0f;
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const ACC v = (ACC)(av[e] * rsc);
          s0[e] += v;
          s1[e] += v * v;
        }
      } else {
        float yv[VEC], xv[VEC];
        const bool use_mask = VEC == 8 && amask != nullptr;
        unsigned mbits = 0;
        if constexpr (VEC > 1) {
          if (act) {
            if (use_mask) mbits = amask[off >> 3];
            else VT<T>::load(y + off, reinterpret_cast<float(&)[VT<T>::V]>(yv));
          }
          VT<T>::load(x + off, reinterpret_cast<float(&)[VT<T>::V]>(xv));
        } else {
          if (act) yv[0] = VT<T>::ld1(y + off);
          xv[0] = VT<T>::ld1(x + off);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float ag = !act ? 1.f
                           : (use_mask ? act_grad_from_bit((mbits >> e) & 1u, act, alpha)
                                       : act_grad_from_out(yv[e], act, alpha));
          float d = av[e] * rsc * ag;
          s0[e] += d;
          s1[e] += d * ((xv[e] - mu[e]) * rs[e]);
        }
      }
    }
  }
  // reduce over ty
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x * VEC + e] = s0[e];
    red[1][threadIdx.x * VEC + e] = s1[e];
  }
  __syncthreads();
  if constexpr (CG && VEC == 8) {
    // red[k][ty][64 channels]: thread t sums rows q*16 .. q*16+15 of (k, channel), q = bit 6 of t
    __shared__ ACC red2[2][2][64];
    const int t = threadIdx.x, k = t >> 7, q = (t >> 6) & 1, ch = t & 63;
    ACC v = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[k][(q * 16 + i) * 64 + ch];
    red2[k][q][ch] = v;
    __syncthreads();
    if (t < 128) {
      const int kk = t >> 6;
      const int c = blockIdx.y * 64 + ch;
      if (c < C)
        partial[(((int64_t)g * rblocks + blockIdx.x) * 2 + kk) * C + c] = (float)(red2[kk][0][ch] + red2[kk][1][ch]);
    }
    return;
  }
  if (ty == 0 && cok) {
    for (int t = 1; t < ry; ++t) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s0[e] += red[0][(t * cx + tx) * VEC + e];
        s1[e] += red[1][(t * cx + tx) * VEC + e];
      }
    }
    float* P = partial + (((int64_t)g * rblocks + blockIdx.x) * 2) * C;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      P[c0 + e] = (float)s0[e];
      P[C + c0 + e] = (float)s1[e];
    }
  }
}

// sums[g][k][c] = sum_b partial[g][b][k][c].  Block = COLS columns x LANES partial-lanes
// (COLS * LANES = 256), four independent accumulators per lane so the (L2-resident) partial loads
// overlap.  Long partial lists use 8 x 32 (a lane then walks rblocks / 32 rows: the kernel is a
// chain of dependent load rounds, and there are ~800 of these launches per step), short ones
// 32 x 8.  Optional direct outputs dst0 / dst1 (length C) receive row 0 / row 1 of group 0 (bias
// / affine gradients written straight into the gradient arena).
template <int COLS, int LANES>
__global__ void __launch_bounds__(256)
norm_final_reduce_kernel(const float* __restrict__ partial, int rblocks, int C, int G,
                         float* __restrict__ sums, float* __restrict__ dst0,
                         float* __restrict__ dst1) {
  static_assert(COLS * LANES == 256, "block shape");
  // (binary64 accumulation: a few hundred adds per launch, see norm_partial_kernel)
  __shared__ double sh[LANES][COLS + 1];
  const int g = blockIdx.y;
  const int cl = threadIdx.x % COLS;
  const int col = blockIdx.x * COLS + cl;   // index into [2][C]
  const int lane_b = threadIdx.x / COLS;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (col < 2 * C) {
    const float* P = partial + (int64_t)g * rblocks * 2 * C + col;
    int b = lane_b;
    for (; b + 3 * LANES < rblocks; b += 4 * LANES) {
      s0 += P[(int64_t)b * 2 * C];
      s1 += P[(int64_t)(b + LANES) * 2 * C];
      s2 += P[(int64_t)(b + 2 * LANES) * 2 * C];
      s3 += P[(int64_t)(b + 3 * LANES) * 2 * C];
    }
    for (; b < rblocks; b += LANES) s0 += P[(int64_t)b * 2 * C];
  }
  sh[lane_b][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (lane_b == 0 && col < 2 * C) {
    double td = 0;
#pragma unroll
    for (int i = 0; i < LANES; ++i) td += sh[i][cl];
    const float t = (float)td;
    sums[(int64_t)g * 2 * C + col] = t;
    if (g == 0) {
      if (dst0 && col < C) dst0[col] = t;
      if (dst1 && col >= C) dst1[col - C] = t;
    }
  }
}

// launch helper: block shape by the length of the partial list
inline void launch_final_reduce(hipStream_t s, const float* partial, int rblocks, int C, int G,
                                float* sums, float* dst0, float* dst1) {
  if (rblocks > 32)
    hipLaunchKernelGGL((norm_final_reduce_kernel<8, 32>), dim3((unsigned)ceil_div(2 * C, 8), (unsigned)G),
                       dim3(256), 0, s, partial, rblocks, C, G, sums, dst0, dst1);
  else
    hipLaunchKernelGGL((norm_final_reduce_kernel<32, 8>), dim3((unsigned)ceil_div(2 * C, 32), (unsigned)G),
                       dim3(256), 0, s, partial, rblocks, C, G, sums, dst0, dst1);
}

// Per-(group, channel) scale/shift from sums.  Keras SyncBatchNormalization: mean = S1/cnt,
// var = S2/cnt - mean^2 (biased), y = x*inv + (beta - mean*inv), inv = gamma*rsqrt(var+eps);
// moving -= (moving - batch) * (1 - momentum).  use_moving: inference (moving statistics).
__global__ void __launch_bounds__(256)
norm_finalize_kernel(const float* __restrict__ sums, float count, int G, int C,
                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                     float momentum, float* __restrict__ moving_mean,
                     float* __restrict__ moving_var, int use_moving, float* __restrict__ scale,
                     float* __restrict__ shift, float* __restrict__ mean_out,
                     float* __restrict__ rstd_out) {
  const int64_t total = (int64_t)G * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * 256) {
    int g = (int)(i / C), c = (int)(i - (int64_t)g * C);
    float mean, var;
    if (use_moving) {
      mean = moving_mean[c];
      var = moving_var[c];
    } else {
      mean = sums[((int64_t)g * 2) * C + c] / count;
      var = sums[((int64_t)g * 2 + 1) * C + c] / count - mean * mean;
      if (moving_mean) {
        moving_mean[c] = moving_mean[c] - (moving_mean[c] - mean) * (1.0f - momentum);
        moving_var[c] = moving_var[c] - (moving_var[c] - var) * (1.0f - momentum);
      }
    }
    float rstd = rsqrtf(var + eps);
    // correctly rounded 1/sqrt: refine rsqrtf (1 ulp) with one Newton step in fp32
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    float inv = rstd * (gamma ? gamma[c] : 1.0f);
    scale[i] = inv;
    shift[i] = (beta ? beta[c] : 0.0f) - mean * inv;
    mean_out[i] = mean;
    rstd_out[i] = rstd;
  }
}

// norm_final_reduce_kernel + norm_finalize_kernel in one launch (single replica, one group: the
// batch norms whose statistics came out of the producing convolution's epilogue -- 309 per step, two
// launch-latency-bound kernels each).  Same arithmetic in the same order as the pair: binary64 column
// sums rounded once to fp32, then the fp32 finalize.  Block = COLS channels x LANES partial lanes.
template <int COLS, int LANES>
__global__ void __launch_bounds__(256)
norm_reduce_finalize_kernel(const float* __restrict__ partial, int rblocks, int C, float count,
                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                            float momentum, float* __restrict__ moving_mean,
                            float* __restrict__ moving_var, float* __restrict__ scale,
                            float* __restrict__ shift, float* __restrict__ mean_out,
                            float* __restrict__ rstd_out) {
  static_assert(COLS * LANES == 256, "block shape");
  __shared__ double sh[2][LANES][COLS + 1];
  const int cl = threadIdx.x % COLS;
  const int c = blockIdx.x * COLS + cl;
  const int lane_b = threadIdx.x / COLS;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  if (c < C) {
    const float* P = partial + c;
    int b = lane_b;
    for (; b + 3 * LANES < rblocks; b += 4 * LANES) {
      a0 += P[(int64_t)b * 2 * C];
      a1 += P[(int64_t)(b + LANES) * 2 * C];
      a2 += P[(int64_t)(b + 2 * LANES) * 2 * C];
      a3 += P[(int64_t)(b + 3 * LANES) * 2 * C];
      b0 += P[(int64_t)b * 2 * C + C];
      b1 += P[(int64_t)(b + LANES) * 2 * C + C];
      b2 += P[(int64_t)(b + 2 * LANES) * 2 * C + C];
      b3 += P[(int64_t)(b + 3 * LANES) * 2 * C + C];
    }
    for (; b < rblocks; b += LANES) {
      a0 += P[(int64_t)b * 2 * C];
      b0 += P[(int64_t)b * 2 * C + C];
    }
  }
  sh[0][lane_b][cl] = (a0 + a1) + (a2 + a3);
  sh[1][lane_b][cl] = (b0 + b1) + (b2 + b3);
  __syncthreads();
  if (lane_b == 0 && c < C) {
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int i = 0; i < LANES; ++i) { t1 += sh[0][i][cl]; t2 += sh[1][i][cl]; }
    const float mean = (float)t1 / count;
    const float var = (float)t2 / count - mean * mean;
    if (moving_mean) {
      moving_mean[c] = moving_mean[c] - (moving_mean[c] - mean) * (1.0f - momentum);
      moving_var[c] = moving_var[c] - (moving_var[c] - var) * (1.0f - momentum);
    }
    float rstd = rsqrtf(var + eps);
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    const float inv = rstd * (gamma ? gamma[c] : 1.0f);
    scale[c] = inv;
    shift[c] = (beta ? beta[c] : 0.0f) - mean * inv;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
  }
}

// Elementwise kernels use the same 2-D thread layout as the statistics kernel: a thread owns
// VEC consecutive channels (scale/shift/mean/rstd live in registers) and strides over rows, so
// the inner loop has no integer divisions.  grid = (row blocks, channel tiles, G).

// y = act(x * scale + shift [+ res]) [+ post]
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
norm_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                  const float* __restrict__ shift, const T* __restrict__ res,
                  const T* __restrict__ post, int64_t R, int C, int cx, int ry, int act,
                  float alpha, T* __restrict__ y, uint8_t* __restrict__ amask) {
  const int g = blockIdx.z;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const int c0 = (blockIdx.y * cx + tx) * VEC;
  if (c0 >= C) return;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = scale[(int64_t)g * C + c0 + e];
    sh[e] = shift[(int64_t)g * C + c0 + e];
  }
  for (int64_t r = (int64_t)blockIdx.x * ry + ty; r < R; r += (int64_t)gridDim.x * ry) {
    const int64_t off = ((int64_t)g * R + r) * C + c0;
    float xv[VEC], rv[VEC], pv[VEC], o[VEC];
    if constexpr (VEC > 1) {
      VT<T>::load(x + off, reinterpret_cast<float(&)[VT<T>::V]>(xv));
      if (res) VT<T>::load(res + off, reinterpret_cast<float(&)[VT<T>::V]>(rv));
      if (post) VT<T>::load(post + off, reinterpret_cast<float(&)[VT<T>::V]>(pv));
    } else {
      xv[0] = VT<T>::ld1(x + off);
      if (res) rv[0] = VT<T>::ld1(res + off);
      if (post) pv[0] = VT<T>::ld1(post + off);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = xv[e] * sc[e] + sh[e];
      if (res) v += rv[e];
      v = act_apply(v, act, alpha);
      if (post) v += pv[e];
      o[e] = v;
    }
    if constexpr (VEC > 1) VT<T>::store(y + off, reinterpret_cast<float(&)[VT<T>::V]>(o));
    else VT<T>::st1(y + off, o[0]);
    if constexpr (VEC == 8 && sizeof(T) == 2) {
      // one bit per element: the stored (rounded) output is > 0.  The backward kernels read
      // this byte instead of the 16 bytes of y.
      if (amask) {
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const uint16_t hb = f32_to_bf16(o[e]);
          bits |= ((hb & 0x7fffu) != 0 && (hb & 0x8000u) == 0) ? (1u << e) : 0u;
        }
        amask[off >> 3] = (uint8_t)bits;
      }
    }
  }
}

// bf16 fast path of norm_apply: activation kind and the residual operand are template
// parameters (branch-free row loop, cf. norm_bwd_apply_fast_kernel); one row per iteration (two
// rows in flight measured slower for this kernel).
template <int ACT, bool HAS_RES>
__global__ void __launch_bounds__(256)
norm_apply_fast_kernel(const uint16_t* __restrict__ x, const float* __restrict__ scale,
                       const float* __restrict__ shift, const uint16_t* __restrict__ res, int64_t R,
                       int C, int cx, int ry, float alpha, uint16_t* __restrict__ y,
                       uint8_t* __restrict__ amask) {
  typedef uint16_t T;
  constexpr int VEC = 8;
  const int g = blockIdx.z;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const int c0 = (blockIdx.y * cx + tx) * VEC;
  if (c0 >= C) return;
  float sc[VEC], sh[VEC];
  VT<float>::load(scale + (int64_t)g * C + c0, reinterpret_cast<float(&)[4]>(sc[0]));
  VT<float>::load(scale + (int64_t)g * C + c0 + 4, reinterpret_cast<float(&)[4]>(sc[4]));
  VT<float>::load(shift + (int64_t)g * C + c0, reinterpret_cast<float(&)[4]>(sh[0]));
  VT<float>::load(shift + (int64_t)g * C + c0 + 4, reinterpret_cast<float(&)[4]>(sh[4]));
  for (int64_t r = (int64_t)blockIdx.x * ry + ty; r < R; r += (int64_t)gridDim.x * ry) {
    const int64_t off = ((int64_t)g * R + r) * C + c0;
    float xv[VEC], rv[VEC], o[VEC];
    VT<T>::load(x + off, xv);
    if (HAS_RES) VT<T>::load(res + off, rv);
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = xv[e] * sc[e] + sh[e];
      if (HAS_RES) v += rv[e];
      if (ACT == 1) v = v > 0.f ? v : 0.f;
      if (ACT == 2) v = v > 0.f ? v : v * alpha;
      o[e] = v;
      const uint16_t hb = f32_to_bf16(v);
      bits |= ((hb & 0x7fffu) != 0 && (hb & 0x8000u) == 0) ? (1u << e) : 0u;
    }
    VT<T>::store(y + off, o);
    if (amask) amask[off >> 3] = (uint8_t)bits;
  }
}

// dpre = dy * act'(y);  dx = gamma*rstd * (dpre - S0/cnt - xhat * S1/cnt);  dres = dpre
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
norm_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                      const float* __restrict__ mean, const float* __restrict__ rstd,
                      const float* __restrict__ gamma, const float* __restrict__ sums, float count,
                      int64_t R, int C, int cx, int ry, int act, float alpha, T* __restrict__ dx,
                      T* __restrict__ dres, const uint8_t* __restrict__ amask, int in_act,
                      float in_alpha) {
  const int g = blockIdx.z;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const int c0 = (blockIdx.y * cx + tx) * VEC;
  if (c0 >= C) return;
  // per-channel constants: dx = gr * d - k1 * x + c0k  with  k1 = gr * rstd * S1 / cnt,
  // c0k = mean * k1 - gr * S0 / cnt  (== gr * (d - S0/cnt - xhat * S1/cnt)).  The prologue runs
  // once per thread and, on the small tensors (2 rows per thread), used to cost more than the
  // rows: no divisions here (1 / count comes from the host), two FMAs per element below.
  float gr[VEC], k1[VEC], c0k[VEC];
  const float inv_count = 1.0f / count;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int64_t gc = (int64_t)g * C + c0 + e;
    const float mu = mean[gc], rs = rstd[gc];
    gr[e] = (gamma ? gamma[c0 + e] : 1.0f) * rs;
    const float s0 = sums[((int64_t)g * 2) * C + c0 + e] * inv_count;
    const float s1 = sums[((int64_t)g * 2 + 1) * C + c0 + e] * inv_count;
    k1[e] = gr[e] * rs * s1;
    c0k[e] = mu * k1[e] - gr[e] * s0;
  }
  for (int64_t r = (int64_t)blockIdx.x * ry + ty; r < R; r += (int64_t)gridDim.x * ry) {
    const int64_t off = ((int64_t)g * R + r) * C + c0;
    float dv[VEC], yv[VEC], xv[VEC], o[VEC], dr[VEC];
    const bool use_mask = VEC == 8 && amask != nullptr;
    unsigned mbits = 0;
    if constexpr (VEC > 1) {
      VT<T>::load(dy + off, reinterpret_cast<float(&)[VT<T>::V]>(dv));
      if (act) {
        if (use_mask) mbits = amask[off >> 3];
        else VT<T>::load(y + off, reinterpret_cast<float(&)[VT<T>::V]>(yv));
      }
      VT<T>::load(x + off, reinterpret_cast<float(&)[VT<T>::V]>(xv));
    } else {
      dv[0] = VT<T>::ld1(dy + off);
      if (act) yv[0] = VT<T>::ld1(y + off);
      xv[0] = VT<T>::ld1(x + off);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float ag = !act ? 1.f
                       : (use_mask ? act_grad_from_bit((mbits >> e) & 1u, act, alpha)
                                   : act_grad_from_out(yv[e], act, alpha));
      float d = dv[e] * ag;
      dr[e] = d;
      o[e] = fmaf(gr[e], d, fmaf(-k1[e], xv[e], c0k[e]));
      // x is the output of an activation whose producer left its derivative to us
      if (in_act) o[e] *= act_grad_from_out(xv[e], in_act, in_alpha);
    }
    if constexpr (VEC > 1) {
      VT<T>::store(dx + off, reinterpret_cast<float(&)[VT<T>::V]>(o));
      if (dres) VT<T>::store(dres + off, reinterpret_cast<float(&)[VT<T>::V]>(dr));
    } else {
      VT<T>::st1(dx + off, o[0]);
      if (dres) VT<T>::st1(dres + off, dr[0]);
    }
  }
}

// bf16 fast path of norm_bwd_apply: activation kinds are TEMPLATE parameters.  With them as
// runtime values the generic kernel above compiles to ~300 scalar branches per row and waits for
// every load separately -- 41 us for a 17 MB tensor where the forward apply takes 10 us (ISA:
// tools/isa_scan.py; measured with tools/norm_bwd_apply_probe.py).  Here the row loop is
// branch-free and two rows are in flight.  ACT != 0 requires the activation bit mask.
// ROWS (batch norm, one group; x is the output of a PARTIAL convolution with a bias and this norm
// its only consumer): the kernel hands that convolution's backward pass what it would otherwise
// take from a pass of its own over dx (se3ds_colsum_row_scale, ~144 launches per step):
//   dx is stored PRE-SCALED by out_row[row] (ratio * update_mask: the operand of the partial
//   conv's weight / data gradients), computed from the bf16-rounded dx as that pass would;
//   colpart[blockIdx.x][2][C] (second row zero: the layout of the statistics partials, so the
//   same final reduction applies) = column sums of rounded dx * sum_row[row] -- the bias gradient.
// PRO (round 5; channel-group layout cx = 8, ry = 32, one group, C % 64 == 0): `sums` is not the
// reduced [2][C] array but the statistics kernel's PARTIAL rows [prows][2][C]; every workgroup folds
// the columns of its own 64 channels in a prologue (prows x 512 bytes, L2-resident) -- the stand-
// alone norm_final_reduce launch between the two passes (255 per step, 7 us alone / 17 us inside the
// step, plus two dependent-launch boundaries each) is gone.  Every workgroup of a channel group adds
// the same numbers in the same order: identical sums.  Row block 0 also writes the folded sums
// (dbeta = sum dz, dgamma = sum dz * xhat: straight into the gradient arena) and sums_out[2][C].
template <int ACT, int IN_ACT, bool ROWS = false, bool PRO = false>
__global__ void __launch_bounds__(256)
norm_bwd_apply_fast_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                           const float* __restrict__ mean, const float* __restrict__ rstd,
                           const float* __restrict__ gamma, const float* __restrict__ sums,
                           float count, int64_t R, int C, int cx, int ry, float alpha,
                           uint16_t* __restrict__ dx, uint16_t* __restrict__ dres,
                           const uint8_t* __restrict__ amask, float in_alpha,
                           const float* __restrict__ sum_row = nullptr,
                           const float* __restrict__ out_row = nullptr,
                           float* __restrict__ colpart = nullptr, int prows = 0,
                           float* __restrict__ dbeta = nullptr, float* __restrict__ dgamma = nullptr,
                           float* __restrict__ sums_out = nullptr) {
  typedef uint16_t T;
  constexpr int VEC = 8;
  const int g = blockIdx.z;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const int c0 = (blockIdx.y * cx + tx) * VEC;
  if (!ROWS && !PRO && c0 >= C) return;
  const bool live = c0 < C;
  float bsum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bsum[e] = 0.f;
  float gr[VEC], k1[VEC], c0k[VEC];
  __shared__ float pro_sums[PRO ? 2 : 1][PRO ? 64 : 1];
  // one LDS buffer for the prologue's fold (PRO: 2 x 32 x 64 floats) and, after it, the bias
  // partials of ROWS (256 x 8 floats)
  __shared__ float lds_buf[PRO ? 4096 : (ROWS ? 256 * VEC : 1)];
  if constexpr (PRO) {
    // (cx == 8, ry == 32, C % 64 == 0: every thread is live)
    float (*pro_red)[32][64] = reinterpret_cast<float (*)[32][64]>(lds_buf);
    float a0[VEC], a1[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    for (int b = ty; b < prows; b += 32) {
      const float* P = sums + (int64_t)b * 2 * C + c0;
      float v0[VEC], v1[VEC];
      VT<float>::load(P, reinterpret_cast<float(&)[4]>(v0[0]));
      VT<float>::load(P + 4, reinterpret_cast<float(&)[4]>(v0[4]));
      VT<float>::load(P + C, reinterpret_cast<float(&)[4]>(v1[0]));
      VT<float>::load(P + C + 4, reinterpret_cast<float(&)[4]>(v1[4]));
#pragma unroll
      for (int e = 0; e < VEC; ++e) { a0[e] += v0[e]; a1[e] += v1[e]; }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      pro_red[0][ty][tx * VEC + e] = a0[e];
      pro_red[1][ty][tx * VEC + e] = a1[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int k = threadIdx.x >> 6, ch = threadIdx.x & 63;
      double t = 0;   // (binary64 second stage, as norm_final_reduce_kernel)
#pragma unroll 8
      for (int i = 0; i < 32; ++i) t += pro_red[k][i][ch];
      const float tf = (float)t;
      pro_sums[k][ch] = tf;
      if (blockIdx.x == 0) {
        const int c = blockIdx.y * 64 + ch;
        if (sums_out) sums_out[k * C + c] = tf;
        if (k == 0 && dbeta) dbeta[c] = tf;
        if (k == 1 && dgamma) dgamma[c] = tf;
      }
    }
    __syncthreads();
  }
  if (live) {
    const int64_t gc = (int64_t)g * C + c0;
    float mu[VEC], rs[VEC], gm[VEC], s0[VEC], s1[VEC];
    VT<float>::load(mean + gc, reinterpret_cast<float(&)[4]>(mu[0]));
    VT<float>::load(mean + gc + 4, reinterpret_cast<float(&)[4]>(mu[4]));
    VT<float>::load(rstd + gc, reinterpret_cast<float(&)[4]>(rs[0]));
    VT<float>::load(rstd + gc + 4, reinterpret_cast<float(&)[4]>(rs[4]));
    VT<float>::load(gamma + c0, reinterpret_cast<float(&)[4]>(gm[0]));
    VT<float>::load(gamma + c0 + 4, reinterpret_cast<float(&)[4]>(gm[4]));
    if constexpr (PRO) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { s0[e] = pro_sums[0][tx * VEC + e]; s1[e] = pro_sums[1][tx * VEC + e]; }
    } else {
    VT<float>::load(sums + (int64_t)g * 2 * C + c0, reinterpret_cast<float(&)[4]>(s0[0]));
    VT<float>::load(sums + (int64_t)g * 2 * C + c0 + 4, reinterpret_cast<float(&)[4]>(s0[4]));
    VT<float>::load(sums + ((int64_t)g * 2 + 1) * C + c0, reinterpret_cast<float(&)[4]>(s1[0]));
    VT<float>::load(sums + ((int64_t)g * 2 + 1) * C + c0 + 4, reinterpret_cast<float(&)[4]>(s1[4]));
    }
    const float inv_count = 1.0f / count;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      gr[e] = gm[e] * rs[e];
      k1[e] = gr[e] * rs[e] * (s1[e] * inv_count);
      c0k[e] = mu[e] * k1[e] - gr[e] * (s0[e] * inv_count);
    }
  }
  const int64_t stride = (int64_t)gridDim.x * ry;
  for (int64_t r = (int64_t)blockIdx.x * ry + ty; live && r < R; r += 2 * stride) {
    const bool two = r + stride < R;
    const int64_t off0 = ((int64_t)g * R + r) * C + c0;
    const int64_t off1 = two ? off0 + stride * C : off0;
    float d0[VEC], d1[VEC], x0[VEC], x1[VEC], o[VEC];
    unsigned m0 = 0xffu, m1 = 0xffu;
    const uint4 qd0 = ld16(dy + off0), qx0 = ld16(x + off0);
    const uint4 qd1 = ld16(dy + off1), qx1 = ld16(x + off1);
    if (ACT != 0) m0 = amask[off0 >> 3];
    if (ACT != 0) m1 = amask[off1 >> 3];
    float sr0 = 1.f, or0 = 1.f, sr1 = 1.f, or1 = 1.f;
    if (ROWS) {
      const int64_t r1 = two ? r + stride : r;
      sr0 = sum_row[r]; or0 = out_row[r]; sr1 = sum_row[r1]; or1 = out_row[r1];
    }
    __builtin_amdgcn_sched_barrier(0);   // all six loads are issued before the first use
    unpack8(qd0, d0); unpack8(qx0, x0); unpack8(qd1, d1); unpack8(qx1, x1);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (ACT == 1) d0[e] = ((m0 >> e) & 1u) ? d0[e] : 0.0f;
      if (ACT == 2) d0[e] = ((m0 >> e) & 1u) ? d0[e] : d0[e] * alpha;
      float v = fmaf(gr[e], d0[e], fmaf(-k1[e], x0[e], c0k[e]));
      if (IN_ACT == 1) v = x0[e] > 0.0f ? v : 0.0f;
      if (IN_ACT == 2) v = x0[e] > 0.0f ? v : v * in_alpha;
      if (ROWS) {
        const float vr = bf16_to_f32(f32_to_bf16(v));   // what a pass over the stored dx reads
        bsum[e] += vr * sr0;
        v = vr * or0;
      }
      o[e] = v;
    }
    VT<T>::store(dx + off0, o);
    if (dres) VT<T>::store(dres + off0, d0);
    if (two) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (ACT == 1) d1[e] = ((m1 >> e) & 1u) ? d1[e] : 0.0f;
        if (ACT == 2) d1[e] = ((m1 >> e) & 1u) ? d1[e] : d1[e] * alpha;
        float v = fmaf(gr[e], d1[e], fmaf(-k1[e], x1[e], c0k[e]));
        if (IN_ACT == 1) v = x1[e] > 0.0f ? v : 0.0f;
        if (IN_ACT == 2) v = x1[e] > 0.0f ? v : v * in_alpha;
        if (ROWS) {
          const float vr = bf16_to_f32(f32_to_bf16(v));
          bsum[e] += vr * sr1;
          v = vr * or1;
        }
        o[e] = v;
      }
      VT<T>::store(dx + off1, o);
      if (dres) VT<T>::store(dres + off1, d1);
    }
  }
  if (ROWS) {
    float* red = lds_buf;
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = bsum[e];
    __syncthreads();
    if (PRO) {
      // compact rows [gridDim.x][C] (the slab layout of se3ds_wgrad_reduce_multi: the bias sums of a
      // module are folded with its weight-gradient slabs in ONE deferred launch); 64 threads fold
      // the 32 row partials of one channel each
      if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) t += red[q * 64 + threadIdx.x];
        colpart[(int64_t)blockIdx.x * C + blockIdx.y * 64 + threadIdx.x] = t;
      }
    } else
    if (ty == 0 && live) {
      const int ny = 256 / cx;
      float* row = colpart + (int64_t)blockIdx.x * 2 * C + c0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float t = 0.f;
        for (int q = 0; q < ny; ++q) t += red[(q * cx + tx) * VEC + e];
        row[e] = t;
        row[C + e] = 0.f;
      }
    }
  }
}

// inference-mode backward (statistics are constants): dx = dpre * scale
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
affine_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                  const float* __restrict__ scale, int64_t R, int C, int cx, int ry, int act,
                  float alpha, T* __restrict__ dx, T* __restrict__ dres,
                  const uint8_t* __restrict__ amask) {
  const int g = blockIdx.z;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const int c0 = (blockIdx.y * cx + tx) * VEC;
  if (c0 >= C) return;
  float sc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) sc[e] = scale[(int64_t)g * C + c0 + e];
  for (int64_t r = (int64_t)blockIdx.x * ry + ty; r < R; r += (int64_t)gridDim.x * ry) {
    const int64_t off = ((int64_t)g * R + r) * C + c0;
    float dv[VEC], yv[VEC], o[VEC], dr[VEC];
    const bool use_mask = VEC == 8 && amask != nullptr;
    unsigned mbits = 0;
    if constexpr (VEC > 1) {
      VT<T>::load(dy + off, reinterpret_cast<float(&)[VT<T>::V]>(dv));
      if (act) {
        if (use_mask) mbits = amask[off >> 3];
        else VT<T>::load(y + off, reinterpret_cast<float(&)[VT<T>::V]>(yv));
      }
    } else {
      dv[0] = VT<T>::ld1(dy + off);
      if (act) yv[0] = VT<T>::ld1(y + off);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float ag = !act ? 1.f
                       : (use_mask ? act_grad_from_bit((mbits >> e) & 1u, act, alpha)
                                   : act_grad_from_out(yv[e], act, alpha));
      float d = dv[e] * ag;
      dr[e] = d;
      o[e] = d * sc[e];
    }
    if constexpr (VEC > 1) {
      VT<T>::store(dx + off, reinterpret_cast<float(&)[VT<T>::V]>(o));
      if (dres) VT<T>::store(dres + off, reinterpret_cast<float(&)[VT<T>::V]>(dr));
    } else {
      VT<T>::st1(dx + off, o[0]);
      if (dres) VT<T>::st1(dres + off, dr[0]);
    }
  }
}

// row blocks for the elementwise kernels: ~8 blocks per CU in total
constexpr int kEwBlocksPerCu = 8;
inline dim3 ew_grid(const Layout2D& l, int64_t R, int G) {
  int64_t want = (256 * (int64_t)kEwBlocksPerCu) / ((int64_t)l.ctiles * G);
  if (want < 1) want = 1;
  int64_t max_rb = ceil_div(R, l.ry);
  if (want > max_rb) want = max_rb;
  return dim3((unsigned)want, (unsigned)l.ctiles, (unsigned)G);
}

int pick_rblocks(int64_t R, int ry, int ctiles, int G) {
  // ~2048 blocks in total (8 per CU) so the streaming reads have enough waves in flight, at
  // least 4 row-iterations per thread, at most 1024 partials per column.
  constexpr int total = 2048;
  int64_t want = total / ((int64_t)ctiles * G);
  if (want < 1) want = 1;
  int64_t max_rb = ceil_div(R, (int64_t)ry * 4);
  if (max_rb < 1) max_rb = 1;
  if (want > max_rb) want = max_rb;
  if (want > 512) want = 512;
  return (int)want;
}

template <typename T, int MODE>
int launch_partial(const T* a, const T* y, const T* x, const float* mean, const float* rstd,
                   const float* row_scale, int G, int64_t R, int C, int act, float alpha,
                   float* sums, float* dst0, float* dst1, float* ws, size_t ws_bytes,
                   hipStream_t s, const uint8_t* amask = nullptr, T* sc_out = nullptr,
                   const float* sc_row = nullptr) {
  Layout2D l = make_layout(C, VT<T>::V);
  if (sc_out != nullptr && !(MODE == 0 && sizeof(T) == 2 && l.vec == 8)) return SE3DS_E_UNSUPPORTED;
  int rb = pick_rblocks(R, l.ry, l.ctiles, G);
  if (ws_bytes < sizeof(float) * (size_t)G * rb * 2 * C) return SE3DS_E_WORKSPACE;
  dim3 grid((unsigned)rb, (unsigned)l.ctiles, (unsigned)G);
  // two rows per iteration where the specialised kernel exists
  if (l.vec > 1 && MODE == 1 && sizeof(T) == 2 && row_scale == nullptr &&
      (act == 0 || ((act == 1 || act == 2) && amask != nullptr))) {
    if (act == 0)
      hipLaunchKernelGGL((norm_partial_kernel<T, VT<T>::V, MODE, true, 0>), grid, dim3(256), 0, s, a, y,
                         x, mean, rstd, row_scale, R, C, l.cx, l.ry, act, alpha, rb, ws, amask);
    else if (act == 1)
      hipLaunchKernelGGL((norm_partial_kernel<T, VT<T>::V, MODE, true, 1>), grid, dim3(256), 0, s, a, y,
                         x, mean, rstd, row_scale, R, C, l.cx, l.ry, act, alpha, rb, ws, amask);
    else
      hipLaunchKernelGGL((norm_partial_kernel<T, VT<T>::V, MODE, true, 2>), grid, dim3(256), 0, s, a, y,
                         x, mean, rstd, row_scale, R, C, l.cx, l.ry, act, alpha, rb, ws, amask);
  } else if (l.vec > 1)
    hipLaunchKernelGGL((norm_partial_kernel<T, VT<T>::V, MODE>), grid, dim3(256), 0, s, a, y, x,
                       mean, rstd, row_scale, R, C, l.cx, l.ry, act, alpha, rb, ws, amask, sc_out,
                       sc_row);
  else
    hipLaunchKernelGGL((norm_partial_kernel<T, 1, MODE>), grid, dim3(256), 0, s, a, y, x, mean,
                       rstd, row_scale, R, C, l.cx, l.ry, act, alpha, rb, ws, amask);
  launch_final_reduce(s, ws, rb, C, G, sums, dst0, dst1);
  return check_launch("norm_partial");
}

}  // namespace
}  // namespace se3ds

using namespace se3ds;

constexpr int kCgMaxRows = 64;
// smallest channel count that takes the channel-group path (256 measured slower: DESIGN 3.2)
static int cg_min_c() { return 512; }

extern "C" {

size_t se3ds_norm_workspace_bytes(int g, int c) {
  return sizeof(float) * (size_t)g * 1024 * 2 * (size_t)c + 16;
}

int se3ds_norm_stats(const void* x, int dtype, int g, int64_t r, int c, const float* row_scale,
                     float* sums, float* colsum_out, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (g <= 0 || r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  if (dtype == SE3DS_F32)
    return launch_partial<float, 0>((const float*)x, nullptr, nullptr, nullptr, nullptr, row_scale,
                                    g, r, c, 0, 0.f, sums, colsum_out, nullptr, (float*)workspace,
                                    workspace_bytes, s);
  if (dtype == SE3DS_BF16)
    return launch_partial<uint16_t, 0>((const uint16_t*)x, nullptr, nullptr, nullptr, nullptr,
                                       row_scale, g, r, c, 0, 0.f, sums, colsum_out, nullptr,
                                       (float*)workspace, workspace_bytes, s);
  return SE3DS_E_BADDTYPE;
}

int se3ds_colsum_row_scale(const void* x, int dtype, int64_t r, int c, const float* sum_row_scale,
                           const float* out_row_scale, void* scaled_out, float* sums,
                           float* colsum_out, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  if (dtype != SE3DS_BF16 || (c % 8) != 0 || scaled_out == nullptr) return SE3DS_E_UNSUPPORTED;
  return launch_partial<uint16_t, 0>((const uint16_t*)x, nullptr, nullptr, nullptr, nullptr,
                                     sum_row_scale, 1, r, c, 0, 0.f, sums, colsum_out, nullptr,
                                     (float*)workspace, workspace_bytes, as_stream(stream), nullptr,
                                     (uint16_t*)scaled_out, out_row_scale);
}

int se3ds_norm_reduce_rows_dst(const float* partial, int64_t rows, int c, float* sums, float* dst0,
                               float* dst1, void* workspace, size_t workspace_bytes, void* stream);

int se3ds_norm_reduce_rows(const float* partial, int64_t rows, int c, float* sums, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return se3ds_norm_reduce_rows_dst(partial, rows, c, sums, nullptr, nullptr, workspace,
                                    workspace_bytes, stream);
}

// ... and optionally the two column sums also to dst0 / dst1 (c floats each): the beta / gamma
// gradients of a batch norm whose backward statistics came out of a conv epilogue
int se3ds_norm_reduce_rows_dst(const float* partial, int64_t rows, int c, float* sums, float* dst0,
                               float* dst1, void* workspace, size_t workspace_bytes, void* stream) {
  if (rows <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  if (rows <= 2048) {
    launch_final_reduce(s, partial, (int)rows, c, 1, sums, dst0, dst1);
    return check_launch("norm_reduce_rows");
  }
  // two levels: groups of kChunk rows -> workspace[groups][2][c] -> sums
  constexpr int kChunk = 512;
  const int64_t full = rows / kChunk, tail = rows - full * kChunk;
  const int64_t groups = full + (tail ? 1 : 0);
  if (groups > 1024 * 64 || workspace_bytes < sizeof(float) * (size_t)groups * 2 * c)
    return SE3DS_E_WORKSPACE;
  float* mid = (float*)workspace;
  if (full)
    launch_final_reduce(s, partial, kChunk, c, (int)full, mid, nullptr, nullptr);
  if (tail)
    launch_final_reduce(s, partial + full * kChunk * 2 * c, (int)tail, c, 1, mid + full * 2 * c,
                        nullptr, nullptr);
  launch_final_reduce(s, mid, (int)groups, c, 1, sums, dst0, dst1);
  return check_launch("norm_reduce_rows");
}

int se3ds_norm_reduce_rows_finalize(const float* partial, int64_t rows, int c, float count,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* moving_mean, float* moving_var, float* scale, float* shift,
                                    float* mean, float* rstd, void* stream) {
  if (rows <= 0 || rows > 2048 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  if (rows > 32)
    hipLaunchKernelGGL((norm_reduce_finalize_kernel<8, 32>), dim3((unsigned)ceil_div(c, 8)), dim3(256), 0, s,
                       partial, (int)rows, c, count, gamma, beta, eps, momentum, moving_mean, moving_var,
                       scale, shift, mean, rstd);
  else
    hipLaunchKernelGGL((norm_reduce_finalize_kernel<32, 8>), dim3((unsigned)ceil_div(c, 32)), dim3(256), 0, s,
                       partial, (int)rows, c, count, gamma, beta, eps, momentum, moving_mean, moving_var,
                       scale, shift, mean, rstd);
  return check_launch("norm_reduce_rows_finalize");
}

int se3ds_norm_finalize(const float* sums, float count, int g, int c, const float* gamma,
                        const float* beta, float eps, float momentum, float* moving_mean,
                        float* moving_var, int use_moving, float* scale, float* shift, float* mean,
                        float* rstd, void* stream) {
  if (g <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(grid_for((int64_t)g * c, 256)), dim3(256), 0,
                     as_stream(stream), sums, count, g, c, gamma, beta, eps, momentum, moving_mean,
                     moving_var, use_moving, scale, shift, mean, rstd);
  return check_launch("norm_finalize");
}

int se3ds_norm_apply(const void* x, int dtype, int g, int64_t r, int c, const float* scale,
                     const float* shift, const void* res, const void* post, int act, float alpha,
                     void* y, void* act_mask, void* stream) {
  if (g <= 0 || r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
#define LAUNCH_APPLY(T, V)                                                                        \
  hipLaunchKernelGGL((norm_apply_kernel<T, V>), ew_grid(l, r, g), dim3(256), 0, s, (const T*)x,   \
                     scale, shift, (const T*)res, (const T*)post, r, c, l.cx, l.ry, act, alpha,   \
                     (T*)y, (uint8_t*)act_mask)
  if (dtype == SE3DS_F32) {
    Layout2D l = make_layout(c, 4);
    if (l.vec > 1) LAUNCH_APPLY(float, 4); else LAUNCH_APPLY(float, 1);
  } else if (dtype == SE3DS_BF16) {
    Layout2D l = make_layout(c, 8);
    // (the channel-group layout of se3ds_norm_bwd_cg was tried for this forward apply too: 198.2 /
    // 197.1 ms per step with, 196.8 / 196.9 without on one box -- not kept)
    if (l.vec > 1 && post == nullptr && act >= 0 && act <= 2) {
#define LAUNCH_FAST(A, RES)                                                                      \
  hipLaunchKernelGGL((norm_apply_fast_kernel<A, RES>), ew_grid(l, r, g), dim3(256), 0, s,        \
                     (const uint16_t*)x, scale, shift, (const uint16_t*)res, r, c, l.cx, l.ry,   \
                     alpha, (uint16_t*)y, (uint8_t*)act_mask)
      if (res) {
        if (act == 0) LAUNCH_FAST(0, true); else if (act == 1) LAUNCH_FAST(1, true); else LAUNCH_FAST(2, true);
      } else {
        if (act == 0) LAUNCH_FAST(0, false); else if (act == 1) LAUNCH_FAST(1, false); else LAUNCH_FAST(2, false);
      }
#undef LAUNCH_FAST
    } else if (l.vec > 1) LAUNCH_APPLY(uint16_t, 8); else LAUNCH_APPLY(uint16_t, 1);
  } else {
    return SE3DS_E_BADDTYPE;
  }
#undef LAUNCH_APPLY
  return check_launch("norm_apply");
}

int se3ds_norm_bwd_stats(const void* dy, const void* y, const void* x, int dtype, int g, int64_t r,
                         int c, const float* mean, const float* rstd, int act, float alpha,
                         float* sums, float* dbeta_out, float* dgamma_out, const void* act_mask,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (g <= 0 || r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  if (dtype == SE3DS_F32)
    return launch_partial<float, 1>((const float*)dy, (const float*)y, (const float*)x, mean, rstd,
                                    nullptr, g, r, c, act, alpha, sums, dbeta_out, dgamma_out,
                                    (float*)workspace, workspace_bytes, s);
  if (dtype == SE3DS_BF16)
    return launch_partial<uint16_t, 1>((const uint16_t*)dy, (const uint16_t*)y, (const uint16_t*)x,
                                       mean, rstd, nullptr, g, r, c, act, alpha, sums, dbeta_out,
                                       dgamma_out, (float*)workspace, workspace_bytes, s,
                                       (const uint8_t*)act_mask);
  return SE3DS_E_BADDTYPE;
}

int se3ds_norm_bwd_apply(const void* dy, const void* y, const void* x, int dtype, int g, int64_t r,
                         int c, const float* mean, const float* rstd, const float* gamma,
                         const float* sums, float count, int act, float alpha, void* dx,
                         void* dres, const void* act_mask, int in_act, float in_alpha,
                         void* stream) {
  if (g <= 0 || r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
#define LAUNCH_BWD(T, V)                                                                         \
  hipLaunchKernelGGL((norm_bwd_apply_kernel<T, V>), ew_grid(l, r, g), dim3(256), 0, s,           \
                     (const T*)dy, (const T*)y, (const T*)x, mean, rstd, gamma, sums, count, r,  \
                     c, l.cx, l.ry, act, alpha, (T*)dx, (T*)dres, (const uint8_t*)act_mask,       \
                     in_act, in_alpha)
  if (dtype == SE3DS_F32) {
    Layout2D l = make_layout(c, 4);
    if (l.vec > 1) LAUNCH_BWD(float, 4); else LAUNCH_BWD(float, 1);
  } else if (dtype == SE3DS_BF16) {
    Layout2D l = make_layout(c, 8);
    const bool fast = l.vec > 1 && gamma != nullptr && (act == 0 || act_mask != nullptr) &&
                      act >= 0 && act <= 2 && (in_act == 0 || in_act == 2);
    if (fast) {
#define LAUNCH_FAST(A, I)                                                                        \
  hipLaunchKernelGGL((norm_bwd_apply_fast_kernel<A, I>), ew_grid(l, r, g), dim3(256), 0, s,      \
                     (const uint16_t*)dy, (const uint16_t*)x, mean, rstd, gamma, sums, count, r, \
                     c, l.cx, l.ry, alpha, (uint16_t*)dx, (uint16_t*)dres,                        \
                     (const uint8_t*)act_mask, in_alpha)
      if (in_act == 0) {
        if (act == 0) LAUNCH_FAST(0, 0); else if (act == 1) LAUNCH_FAST(1, 0); else LAUNCH_FAST(2, 0);
      } else {
        if (act == 0) LAUNCH_FAST(0, 2); else if (act == 1) LAUNCH_FAST(1, 2); else LAUNCH_FAST(2, 2);
      }
#undef LAUNCH_FAST
    } else if (l.vec > 1) LAUNCH_BWD(uint16_t, 8); else LAUNCH_BWD(uint16_t, 1);
  } else {
    return SE3DS_E_BADDTYPE;
  }
#undef LAUNCH_BWD
  return check_launch("norm_bwd_apply");
}

// se3ds_norm_bwd_apply for a batch norm (one group, bf16, c % 8 == 0) whose input x is the output
// of a partial convolution with a bias (and this norm its only consumer): dx is stored pre-scaled
// by out_row[row] and colsum_dst[c] = sum over rows of rounded dx * sum_row[row] (the bias
// gradient) -- what se3ds_colsum_row_scale would take from its own pass over dx.
// workspace: se3ds_norm_workspace_bytes(3, c).  SE3DS_E_UNSUPPORTED where the fast kernel does not
// apply (the caller then takes se3ds_norm_bwd_apply and leaves the rest to the convolution).
int se3ds_norm_bwd_apply_rows(const void* dy, const void* x, int dtype, int64_t r, int c,
                              const float* mean, const float* rstd, const float* gamma,
                              const float* sums, float count, int act, float alpha, void* dx,
                              void* dres, const void* act_mask, const float* sum_row,
                              const float* out_row, float* colsum_dst, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  if (dtype != SE3DS_BF16 || colsum_dst == nullptr || gamma == nullptr || sum_row == nullptr ||
      out_row == nullptr)
    return SE3DS_E_UNSUPPORTED;
  Layout2D l = make_layout(c, 8);
  if (!(l.vec > 1 && (act == 0 || act_mask != nullptr) && act >= 0 && act <= 2))
    return SE3DS_E_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const dim3 grid = ew_grid(l, r, 1);
  if (workspace_bytes < sizeof(float) * ((size_t)grid.x * 2 * c + 2 * (size_t)c))
    return SE3DS_E_WORKSPACE;
  float* part = (float*)workspace;
  float* scratch_sums = part + (size_t)grid.x * 2 * c;   // [2][c], discarded
#define LAUNCH_ROWS(A)                                                                           \
  hipLaunchKernelGGL((norm_bwd_apply_fast_kernel<A, 0, true>), grid, dim3(256), 0, s,            \
                     (const uint16_t*)dy, (const uint16_t*)x, mean, rstd, gamma, sums, count, r, \
                     c, l.cx, l.ry, alpha, (uint16_t*)dx, (uint16_t*)dres,                        \
                     (const uint8_t*)act_mask, 0.0f, sum_row, out_row, part)
  if (act == 0) LAUNCH_ROWS(0); else if (act == 1) LAUNCH_ROWS(1); else LAUNCH_ROWS(2);
#undef LAUNCH_ROWS
  launch_final_reduce(s, part, (int)grid.x, c, 1, scratch_sums, colsum_dst, nullptr);
  return check_launch("norm_bwd_apply_rows");
}

// Round 5: batch-norm backward in TWO launches (statistics partials, apply) instead of three: the
// channel-group layout (64 channels = one 128-byte line per row and workgroup) lets every apply
// workgroup fold the partial rows of its own channels in a prologue.  One group, bf16, c % 64 == 0,
// c >= kCgMinC (enough channel groups to fill the chip with <= kCgMaxRows partial rows).
static int cg_stat_blocks(int64_t r, int c) {
  constexpr int total = 768;    // 3 workgroups per CU (1 024 and 2 048 measured no faster)
  int64_t rb = total / (c / 64);
  const int64_t max_rb = ceil_div(r, (int64_t)32 * 4);   // >= 4 rows per thread
  if (rb > max_rb) rb = max_rb;
  if (rb > kCgMaxRows) rb = kCgMaxRows;
  if (rb < 1) rb = 1;
  return (int)rb;
}
static int cg_apply_blocks(int64_t r, int c) {
  constexpr int total = 2048;   // 8 workgroups per CU
  int64_t rb = total / (c / 64);
  const int64_t max_rb = ceil_div(r, (int64_t)32);
  if (rb > max_rb) rb = max_rb;
  if (rb < 1) rb = 1;
  return (int)rb;
}

int se3ds_norm_bwd_cg_supported(int dtype, int64_t r, int c, int act, int has_mask, int in_act) {
  // SE3DS_NORM_CG=0: the three-launch backward of rounds 2-4 (read per call: the parity test of
  // the two forms in tests/test_blocks_gpu.py switches it)
  const char* e_cg = getenv("SE3DS_NORM_CG");
  const bool off = e_cg && atoi(e_cg) == 0;
  if (off || dtype != SE3DS_BF16 || r <= 0 || c < cg_min_c() || (c % 64) != 0) return 0;
  if (act < 0 || act > 2 || (act != 0 && !has_mask) || !(in_act == 0 || in_act == 2)) return 0;
  return 1;
}

size_t se3ds_norm_bwd_cg_workspace_bytes(int c) { return sizeof(float) * (size_t)kCgMaxRows * 2 * (size_t)c; }

// rows of the [rows][c] bias partials se3ds_norm_bwd_cg writes when sum_row / out_row are given
int se3ds_norm_bwd_cg_col_rows(int64_t r, int c) { return cg_apply_blocks(r, c); }

int se3ds_norm_bwd_cg(const void* dy, const void* x, int dtype, int64_t r, int c, const float* mean,
                      const float* rstd, const float* gamma, float count, int act, float alpha,
                      void* dx, void* dres, const void* act_mask, int in_act, float in_alpha,
                      float* dbeta_out, float* dgamma_out, float* sums_out, const float* sum_row,
                      const float* out_row, float* colpart, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!se3ds_norm_bwd_cg_supported(dtype, r, c, act, act_mask != nullptr, in_act) || gamma == nullptr)
    return SE3DS_E_UNSUPPORTED;
  const bool rows = sum_row != nullptr;
  if (rows && (out_row == nullptr || colpart == nullptr || in_act != 0)) return SE3DS_E_UNSUPPORTED;
  if (workspace_bytes < se3ds_norm_bwd_cg_workspace_bytes(c)) return SE3DS_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* part = (float*)workspace;
  const int rb = cg_stat_blocks(r, c);
  const dim3 gs((unsigned)rb, (unsigned)(c / 64), 1);
  const uint16_t* A = (const uint16_t*)dy;
  const uint16_t* X = (const uint16_t*)x;
#define LAUNCH_ST(AC)                                                                             \
  hipLaunchKernelGGL((norm_partial_kernel<uint16_t, 8, 1, true, AC, true>), gs, dim3(256), 0, s, A,  \
                     (const uint16_t*)nullptr, X, mean, rstd, (const float*)nullptr, r, c, 8, 32,    \
                     act, alpha, rb, part, (const uint8_t*)act_mask)
  if (act == 0) LAUNCH_ST(0); else if (act == 1) LAUNCH_ST(1); else LAUNCH_ST(2);
#undef LAUNCH_ST
  const dim3 ga((unsigned)cg_apply_blocks(r, c), (unsigned)(c / 64), 1);
#define LAUNCH_AP(AC, IA, RW)                                                                     \
  hipLaunchKernelGGL((norm_bwd_apply_fast_kernel<AC, IA, RW, true>), ga, dim3(256), 0, s, A, X, mean, \
                     rstd, gamma, part, count, r, c, 8, 32, alpha, (uint16_t*)dx, (uint16_t*)dres,     \
                     (const uint8_t*)act_mask, in_alpha, sum_row, out_row, colpart, rb, dbeta_out,     \
                     dgamma_out, sums_out)
  if (rows) {
    if (act == 0) LAUNCH_AP(0, 0, true); else if (act == 1) LAUNCH_AP(1, 0, true); else LAUNCH_AP(2, 0, true);
  } else if (in_act == 0) {
    if (act == 0) LAUNCH_AP(0, 0, false); else if (act == 1) LAUNCH_AP(1, 0, false); else LAUNCH_AP(2, 0, false);
  } else {
    if (act == 0) LAUNCH_AP(0, 2, false); else if (act == 1) LAUNCH_AP(1, 2, false); else LAUNCH_AP(2, 2, false);
  }
#undef LAUNCH_AP
  return check_launch("norm_bwd_cg");
}

int se3ds_affine_bwd(const void* dy, const void* y, int dtype, int g, int64_t r, int c,
                     const float* scale, int act, float alpha, void* dx, void* dres,
                     const void* act_mask, void* stream) {
  if (g <= 0 || r <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
#define LAUNCH_AFF(T, V)                                                                      \
  hipLaunchKernelGGL((affine_bwd_kernel<T, V>), ew_grid(l, r, g), dim3(256), 0, s,            \
                     (const T*)dy, (const T*)y, scale, r, c, l.cx, l.ry, act, alpha, (T*)dx,  \
                     (T*)dres, (const uint8_t*)act_mask)
  if (dtype == SE3DS_F32) {
    Layout2D l = make_layout(c, 4);
    if (l.vec > 1) LAUNCH_AFF(float, 4); else LAUNCH_AFF(float, 1);
  } else if (dtype == SE3DS_BF16) {
    Layout2D l = make_layout(c, 8);
    if (l.vec > 1) LAUNCH_AFF(uint16_t, 8); else LAUNCH_AFF(uint16_t, 1);
  } else {
    return SE3DS_E_BADDTYPE;
  }
#undef LAUNCH_AFF
  return check_launch("affine_bwd");
}

}  // extern "C"
