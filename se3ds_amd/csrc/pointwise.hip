// Pointwise / pooling / small-reduction kernels of the SE3DS G+D step (all HBM-bound):
// partial-conv mask window statistics, activation backward, 2x2 max pool, 3x3/s2 average
// pool, nearest x2 upsample, channel-slice copies (concat / split with dtype conversion),
// accumulate, output heads (tanh -> [0,1], clip) and the loss reductions / gradients of
// trainers/se3ds_trainer.py:39-71,148-234.
#include "common.h"
#include <type_traits>

namespace se3ds {
namespace {

constexpr int kB = 256;

// ---------------------------------------------------------------- partial-conv mask window
// layers.py:153-163: cnt = conv(mask, ones(k,k)); ratio = k*k/(cnt+1e-6) * clip(cnt,0,1);
// um = clip(cnt,0,1).  Also emits ru = ratio*um and bu = (1-ratio)*um for the backward.
__global__ void __launch_bounds__(kB)
mask_window_kernel(const float* __restrict__ mask, int N, int H, int W, int Ho, int Wo, int kh,
                   int kw, int stride, int pad_t, int pad_l, int wrap_w, float* __restrict__ ratio,
                   float* __restrict__ um, float* __restrict__ ru, float* __restrict__ bu) {
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int ox = (int)(i % Wo);
    int oy = (int)((i / Wo) % Ho);
    int n = (int)(i / ((int64_t)Wo * Ho));
    float cnt = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      int sy = oy * stride - pad_t + ky;
      if (sy < 0 || sy >= H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        int sx = ox * stride - pad_l + kx;
        if (wrap_w) sx = sx < 0 ? sx + W : (sx >= W ? sx - W : sx);
        if (sx < 0 || sx >= W) continue;
        cnt += mask[((int64_t)n * H + sy) * W + sx];
      }
    }
    float u = fminf(fmaxf(cnt, 0.f), 1.f);
    float r = ((float)(kh * kw) / (cnt + 1e-6f)) * u;
    ratio[i] = r;
    um[i] = u;
    if (ru) ru[i] = r * u;
    if (bu) bu[i] = (1.0f - r) * u;
  }
}

// ----------------------------------------------------------------------- activation backward
template <typename T, int VEC>
__global__ void __launch_bounds__(kB)
act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, int64_t nvec, int act,
               float alpha, T* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kB) {
    float d[VEC], v[VEC];
    if constexpr (VEC > 1) {
      VT<T>::load(dy + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(d));
      VT<T>::load(y + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(v));
    } else {
      d[0] = VT<T>::ld1(dy + i);
      v[0] = VT<T>::ld1(y + i);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) d[e] *= act_grad_from_out(v[e], act, alpha);
    if constexpr (VEC > 1) VT<T>::store(dx + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(d));
    else VT<T>::st1(dx + i, d[0]);
  }
}

// out[r, :] = x[r, :] * scale[r]   (partial-conv backward: dy * ratio * update_mask)
template <typename T, int VEC>
__global__ void __launch_bounds__(kB)
row_scale_kernel(const T* __restrict__ x, const float* __restrict__ scale, int64_t rows, int cvec,
                 T* __restrict__ out) {
  const int64_t total = rows * cvec;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    const float sc = scale[i / cvec];
    float v[VEC];
    if constexpr (VEC > 1) VT<T>::load(x + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(v));
    else v[0] = VT<T>::ld1(x + i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] *= sc;
    if constexpr (VEC > 1) VT<T>::store(out + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(v));
    else VT<T>::st1(out + i, v[0]);
  }
}

// dst (+)= src, elementwise, same dtype
template <typename T, int VEC>
__global__ void __launch_bounds__(kB)
accumulate_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t nvec,
                  T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kB) {
    float x[VEC], y[VEC];
    if constexpr (VEC > 1) {
      VT<T>::load(a + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(x));
      VT<T>::load(b + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(y));
    } else {
      x[0] = VT<T>::ld1(a + i);
      y[0] = VT<T>::ld1(b + i);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) x[e] += y[e];
    if constexpr (VEC > 1) VT<T>::store(out + i * VEC, reinterpret_cast<float(&)[VT<T>::V]>(x));
    else VT<T>::st1(out + i, x[0]);
  }
}

// --------------------------------------------------------------------------- 2x2/s2 max pool
// Keras MaxPool2D(padding='SAME'), image_models.py:267,289: window rows 2oy..2oy+1 (clipped).
template <typename T>
__global__ void __launch_bounds__(kB)
maxpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                   T* __restrict__ y) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((int64_t)Wo * Ho));
    float m = -INFINITY;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        int sy = 2 * oy + dy, sx = 2 * ox + dx;
        if (sy < H && sx < W) m = fmaxf(m, VT<T>::ld1(x + (((int64_t)n * H + sy) * W + sx) * C + c));
      }
    VT<T>::st1(y + i, m);
  }
}
// gradient goes to the first maximal element of the window (TF MaxPoolGrad)
template <typename T>
__global__ void __launch_bounds__(kB)
maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                   int N, int H, int W, int C, int Ho, int Wo, T* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int sx = (int)(p % W), sy = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    int oy = sy >> 1, ox = sx >> 1;
    int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C + c;
    float v = VT<T>::ld1(x + i), m = VT<T>::ld1(y + o);
    float g = 0.f;
    if (v == m) {
      bool first = true;  // any earlier element of the window equal to the max?
      for (int dyy = 0; dyy < 2 && first; ++dyy)
        for (int dxx = 0; dxx < 2; ++dxx) {
          int qy = 2 * oy + dyy, qx = 2 * ox + dxx;
          if (qy == sy && qx == sx) { dyy = 2; break; }
          if (qy < H && qx < W &&
              VT<T>::ld1(x + (((int64_t)n * H + qy) * W + qx) * C + c) == m) { first = false; break; }
        }
      if (first) g = VT<T>::ld1(dy + o);
    }
    VT<T>::st1(dx + i, g);
  }
}

// bf16, C % 8 == 0: one thread per (output pixel, 8 channels): its 2x2 window is read and written
// with 16-byte accesses and no per-element index arithmetic (the scalar kernel above spends
// ~0.87 ms on the generator's 8 x 256 x 512 x 128 stem tensor; this one is bound by its 1.2 GB).
__global__ void __launch_bounds__(kB)
maxpool_bwd_vec8_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                        const uint16_t* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                        uint16_t* __restrict__ dx) {
  const int cv = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    const int c0 = (int)(i % cv) * 8;
    int64_t p = i / cv;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho), n = (int)(p / Ho);
    const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C + c0;
    float m[8], g[8], v[4][8];
    bool in[4];
    int64_t off[4];
    VT<uint16_t>::load(y + o, m);
    VT<uint16_t>::load(dy + o, g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int qy = 2 * oy + (q >> 1), qx = 2 * ox + (q & 1);
      in[q] = qy < H && qx < W;
      off[q] = (((int64_t)n * H + (in[q] ? qy : 2 * oy)) * W + (in[q] ? qx : 2 * ox)) * C + c0;
      VT<uint16_t>::load(x + off[q], v[q]);
    }
    float out[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bool taken = false;   // gradient goes to the FIRST maximal element (row-major window order)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool hit = in[q] && !taken && v[q][e] == m[e];
        out[q][e] = hit ? g[e] : 0.f;
        taken = taken || hit;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (in[q]) VT<uint16_t>::store(dx + off[q], out[q]);
  }
}

// ----------------------------------------------------------------- 3x3 / s2 SAME average pool
// tf.nn.avg_pool (image_models.py:617): the divisor counts in-bounds taps only.
template <typename T>
__global__ void __launch_bounds__(kB)
avgpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo, int pad_t,
                   int pad_l, T* __restrict__ y) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((int64_t)Wo * Ho));
    float s = 0.f;
    int cnt = 0;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        int sy = oy * 2 - pad_t + ky, sx = ox * 2 - pad_l + kx;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
          s += VT<T>::ld1(x + (((int64_t)n * H + sy) * W + sx) * C + c);
          ++cnt;
        }
      }
    VT<T>::st1(y + i, s / (float)cnt);
  }
}
template <typename T>
__global__ void __launch_bounds__(kB)
avgpool_bwd_kernel(const T* __restrict__ dy, int N, int H, int W, int C, int Ho, int Wo, int pad_t,
                   int pad_l, T* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int sx = (int)(p % W), sy = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float g = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      int ty = sy + pad_t - ky;
      if (ty < 0 || (ty & 1)) continue;
      int oy = ty >> 1;
      if (oy >= Ho) continue;
      int y0 = oy * 2 - pad_t;
      int ny = (y0 + 3 <= H ? y0 + 3 : H) - (y0 < 0 ? 0 : y0);
      for (int kx = 0; kx < 3; ++kx) {
        int tx = sx + pad_l - kx;
        if (tx < 0 || (tx & 1)) continue;
        int ox = tx >> 1;
        if (ox >= Wo) continue;
        int x0 = ox * 2 - pad_l;
        int nx = (x0 + 3 <= W ? x0 + 3 : W) - (x0 < 0 ? 0 : x0);
        g += VT<T>::ld1(dy + (((int64_t)n * Ho + oy) * Wo + ox) * C + c) / (float)(ny * nx);
      }
    }
    VT<T>::st1(dx + i, g);
  }
}

// ------------------------------------------------------------------------ nearest x2 upsample
template <typename T>
__global__ void __launch_bounds__(kB)
upsample2x_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, T* __restrict__ y) {
  const int64_t total = (int64_t)N * 2 * H * 2 * W * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int ox = (int)(p % (2 * W)), oy = (int)((p / (2 * W)) % (2 * H));
    int n = (int)(p / ((int64_t)4 * W * H));
    y[i] = x[(((int64_t)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
  }
}
template <typename T>
__global__ void __launch_bounds__(kB)
upsample2x_bwd_kernel(const T* __restrict__ dy, int N, int H, int W, int C, T* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int sx = (int)(p % W), sy = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    const int64_t base = (((int64_t)n * 2 * H + 2 * sy) * 2 * W + 2 * sx) * C + c;
    float g = VT<T>::ld1(dy + base) + VT<T>::ld1(dy + base + C);
    g += VT<T>::ld1(dy + base + (int64_t)2 * W * C) + VT<T>::ld1(dy + base + (int64_t)2 * W * C + C);
    VT<T>::st1(dx + i, g);
  }
}

// ------------------------------------------------- channel-slice copy with dtype conversion
template <typename TS, typename TD>
__global__ void __launch_bounds__(kB)
copy_channels_kernel(const TS* __restrict__ src, int src_c, int src_c0, TD* __restrict__ dst,
                     int dst_c, int dst_c0, int ncopy, int64_t rows) {
  const int64_t total = rows * ncopy;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % ncopy);
    int64_t r = i / ncopy;
    VT<TD>::st1(dst + r * dst_c + dst_c0 + c, VT<TS>::ld1(src + r * src_c + src_c0 + c));
  }
}

template <typename T>
__global__ void __launch_bounds__(kB)
fill_kernel(T* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB)
    VT<T>::st1(p + i, v);
}


// ------------------------------------------------------------------- standalone PadLayer
// models/layers.py:22-97.  mode 0 CONSTANT(value), 1 REFLECT, 2 SYMMETRIC for H (and for W when
// wrap_w == 0); wrap_w: W is padded circularly (taken from the already H-padded tensor, which
// is the same as wrapping the column index).
template <typename T>
__global__ void __launch_bounds__(kB)
pad2d_kernel(const T* __restrict__ x, int N, int H, int W, int C, int pad, int mode, int wrap_w,
             float value, T* __restrict__ y) {
  const int Ho = H + 2 * pad, Wo = W + 2 * pad;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((int64_t)Wo * Ho));
    int sy = oy - pad, sx = ox - pad;
    bool inb = true;
    auto fold = [&](int v, int size) -> int {
      if (v >= 0 && v < size) return v;
      if (mode == 1) return v < 0 ? -v : 2 * (size - 1) - v;         // REFLECT
      if (mode == 2) return v < 0 ? -v - 1 : 2 * size - 1 - v;       // SYMMETRIC
      inb = false;
      return 0;
    };
    if (wrap_w) sx = sx < 0 ? sx + W : (sx >= W ? sx - W : sx);
    else sx = fold(sx, W);
    sy = fold(sy, H);
    if (inb) y[i] = x[(((int64_t)n * H + sy) * W + sx) * C + c];
    else VT<T>::st1(y + i, value);
  }
}

// ---------------------------------------------------------------------------------- heads
// image_models.py:187-190: rgb = (tanh(x) + 1) / 2 ; depth = clip(x, 0, 1).  Outputs fp32.
template <typename T>
__global__ void __launch_bounds__(kB)
head_fwd_kernel(const T* __restrict__ x, int64_t n, int kind, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) {
    float v = VT<T>::ld1(x + i);
    y[i] = kind == 0 ? (tanhf(v) + 1.0f) / 2.0f : fminf(fmaxf(v, 0.f), 1.f);
  }
}
// d/dx: rgb: (1 - t^2)/2 with t = 2*rgb - 1 ; depth: 1 where 0 <= x <= 1 (tf.clip_by_value)
template <typename T>
__global__ void __launch_bounds__(kB)
head_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                const T* __restrict__ x, int64_t n, int kind, T* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) {
    float g = dy[i];
    if (kind == 0) {
      float t = 2.0f * y[i] - 1.0f;
      g = g * (1.0f - t * t) * 0.5f;
    } else {
      float v = VT<T>::ld1(x + i);
      g = (v >= 0.f && v <= 1.f) ? g : 0.f;
    }
    VT<T>::st1(dx + i, g);
  }
}

// ---------------------------------------------------------------------------------- losses
// One block per sample: out[n] = sum over (p, c) of f(...)
__device__ __forceinline__ float block_sum(float v) {
  __shared__ float sh[kB / 64];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < kB / 64; ++i) t += sh[i];
  return t;
}

// mode 0: sum(a)                     (a: (N, P*C))
// mode 1: sum(|a - b| * m[p])        (a, b: (N,P,C); m: (N,P))
// mode 2: sum(1[0 < a < 1])          (valid-depth pixel count, se3ds_trainer.py:148-152)
// mode 3: sum(m[p] * (1 - m2[p]))    (wc mask, :176-178) a=m, b=m2
// mode 4: sum((a - b)^2 * 1[0 < b < 1]) (depth RMSE numerator, utils/eval_metric.py:225-231)
__global__ void __launch_bounds__(kB)
sample_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                  const float* __restrict__ m, int64_t P, int C, int mode,
                  float* __restrict__ partial) {
  // grid (chunks, N): partial[n][chunk]; summed by sample_sum_final_kernel (deterministic)
  const int n = blockIdx.y;
  const int64_t total = P * C;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    const int64_t idx = (int64_t)n * total + i;
    if (mode == 0) s += a[idx];
    else if (mode == 1) s += fabsf(a[idx] - b[idx]) * (m ? m[(int64_t)n * P + i / C] : 1.0f);
    else if (mode == 2) s += (a[idx] > 0.f && a[idx] < 1.f) ? 1.f : 0.f;
    else if (mode == 4) {
      const float d = a[idx] - b[idx];
      s += (d * d) * ((b[idx] > 0.f && b[idx] < 1.f) ? 1.f : 0.f);
    }
    else s += a[idx] * (1.0f - b[idx]);
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[(int64_t)n * gridDim.x + blockIdx.x] = s;
}

__global__ void __launch_bounds__(kB)
sample_sum_final_kernel(const float* __restrict__ partial, int chunks, float* __restrict__ out) {
  const int n = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < chunks; i += kB) s += partial[(int64_t)n * chunks + i];
  s = block_sum(s);
  if (threadIdx.x == 0) out[n] = s;
}

// grad[n,p,c] = coef[n] * sign(a - b) * w[n,p]; w = valid-depth mask (mode 0: 0 < t < 1 of
// target b) or m*(1-m2) (mode 1).  coef is a device vector (already holds lambda / norms).
// modes 2 / 3 just materialise w (the masks of modes 0 / 1) for the loss-value reductions.
__global__ void __launch_bounds__(kB)
l1_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
               const float* __restrict__ m, const float* __restrict__ m2,
               const float* __restrict__ coef, int64_t P, int C, int N, int mode,
               float* __restrict__ grad) {
  const int64_t total = (int64_t)N * P * C;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kB) {
    int64_t np = i / C;
    int n = (int)(np / P);
    float w;
    if (mode == 0 || mode == 2) w = (b[i] > 0.f && b[i] < 1.f) ? 1.f : 0.f;
    else w = m[np] * (1.0f - m2[np]);
    if (mode >= 2) { grad[i] = w; continue; }
    float d = a[i] - b[i];
    float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    grad[i] = coef[n] * sgn * w;
  }
}

// coef[i] = scale / max(sums[i], 1)   (per-sample loss normalisers, se3ds_trainer.py:151-152)
__global__ void __launch_bounds__(kB)
recip_clamp_kernel(const float* __restrict__ sums, int n, float scale, float* __restrict__ out) {
  for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
    out[i] = scale / fmaxf(sums[i], 1.0f);
}

// Hinge terms on one logits map (2N, P) [fake first, real second], se3ds_trainer.py:58-71:
// sums[0] = sum(-fake), sums[1] = sum(relu(1-real) + relu(1+fake)).
// dlog_d = cd * d(disc)/dlogit, dlog_g = cg * d(gen)/dlogit (fake half; real half zero).
template <typename T>
__global__ void __launch_bounds__(kB)
hinge_kernel(const T* __restrict__ logits, int64_t half, float cd, float cg,
             float* __restrict__ sums, T* __restrict__ dlog_d, T* __restrict__ dlog_g) {
  float sg = 0.f, sd = 0.f;
  for (int64_t i = threadIdx.x; i < half; i += kB) {
    float f = VT<T>::ld1(logits + i), r = VT<T>::ld1(logits + half + i);
    sg += -f;
    sd += fmaxf(1.0f - r, 0.f) + fmaxf(1.0f + f, 0.f);
    if (dlog_d) {
      VT<T>::st1(dlog_d + i, (1.0f + f > 0.f) ? cd : 0.f);
      VT<T>::st1(dlog_d + half + i, (1.0f - r > 0.f) ? -cd : 0.f);
    }
    if (dlog_g) {
      VT<T>::st1(dlog_g + i, -cg);
      VT<T>::st1(dlog_g + half + i, 0.f);
    }
  }
  sg = block_sum(sg);
  sd = block_sum(sd);
  if (threadIdx.x == 0) { sums[0] = sg; sums[1] = sd; }
}

// fp32 master HWIO [K][Cout] -> bf16 operand copies wt [Cout][K] and wn [K][Cout] for MANY layers
// in one launch (64 x 64 LDS-tiled transpose per block, as weight_prep_vec_kernel in conv.hip;
// blockIdx -> layer by binary search over the layers' first-tile column of the table).
__global__ void __launch_bounds__(256)
weight_prep_multi_kernel(const int64_t* __restrict__ table, int nlayers) {
  __shared__ float tile[64][65];
  const int64_t blk = blockIdx.x;
  int lo = 0, hi = nlayers - 1;
  while (lo < hi) {   // last layer whose first tile <= blk
    const int mid = (lo + hi + 1) >> 1;
    if (table[(int64_t)mid * 6 + 5] <= blk) lo = mid; else hi = mid - 1;
  }
  const int64_t* L = table + (int64_t)lo * 6;
  const float* __restrict__ w = (const float*)L[0];
  const int64_t K = L[1];
  const int Cout = (int)L[2];
  uint16_t* __restrict__ wt = (uint16_t*)L[3];
  uint16_t* __restrict__ wn = (uint16_t*)L[4];
  const int64_t local = blk - L[5];
  const int64_t tiles_k = (K + 63) / 64;
  const int64_t k0 = (local % tiles_k) * 64;
  const int c0 = (int)(local / tiles_k) * 64;
  const int tid = threadIdx.x;
  {
    const int cq = (tid & 15) * 4, r0 = tid >> 4;   // 16 lanes x float4 cover 64 columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + 16 * i;
      const int64_t k = k0 + r;
      const int c = c0 + cq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K && c < Cout) {
        v = *reinterpret_cast<const float4*>(w + k * Cout + c);
        uint16_t o[4] = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
        *reinterpret_cast<uint2*>(wn + k * Cout + c) = *reinterpret_cast<const uint2*>(o);
      }
      tile[r][cq] = v.x; tile[r][cq + 1] = v.y; tile[r][cq + 2] = v.z; tile[r][cq + 3] = v.w;
    }
  }
  __syncthreads();
  {
    const int kq = (tid & 7) * 8, cr0 = tid >> 3;   // 8 lanes x 8 bf16 cover 64 k of one channel
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cr = cr0 + 32 * j;
      const int c = c0 + cr;
      const int64_t k = k0 + kq;
      if (c < Cout && k < K) {
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(tile[kq + e][cr]);
        *reinterpret_cast<uint4*>(wt + (int64_t)c * K + k) = *reinterpret_cast<const uint4*>(o);
      }
    }
  }
}

}  // namespace
}  // namespace se3ds

using namespace se3ds;

// ---------------------------------------------------------- SE3DSModel quantisation steps
// models/models.py:289-291 (proj_rgb / 255, clip), :325-331 (int32(g * 255), clip(-1, 255);
// int32(clip(g, 0, 1) * 255)), :198 (int / 255), :353 (cast to uint8): one rounding per reference
// op -- optional pre-clamp, * mul, / div (IEEE division), then either a float clamp or
// truncation toward zero (tf.cast float -> int) followed by an integer clamp.  Float clamps are
// tf.clip_by_value: min(max(v, lo), hi) in TF's NaN-PROPAGATING sense (a NaN prediction stays NaN
// instead of turning into `lo`, so a diverged roll-out is visible).  lo > hi selects a PURE CAST for
// integer outputs: no clamp, the value wraps modulo 2^bits as tf.cast(int32 -> uint8) does (a -1
// void class becomes 255).
__device__ inline float clip_by_value(float v, float lo, float hi) {
  return v < lo ? lo : (v > hi ? hi : v);   // comparisons are false for NaN: NaN passes through
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(kB)
quantize_kernel(const TI* __restrict__ in, int64_t n, int pre_clamp, float pre_lo, float pre_hi,
                float mul, float div, float lo, float hi, TO* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) {
    float v = (float)in[i];
    if (pre_clamp) v = clip_by_value(v, pre_lo, pre_hi);
    v = v * mul;
    v = v / div;
    if constexpr (std::is_floating_point<TO>::value) {
      out[i] = (TO)clip_by_value(v, lo, hi);
    } else {
      // float -> int: truncation toward zero; NaN and values outside int32 are undefined in TF
      // (and in C): pinned here to 0 / saturation so that the result is at least deterministic
      int t = (v != v) ? 0 : (v >= 2147483648.f ? 2147483647 : (v <= -2147483648.f ? (-2147483647 - 1) : (int)v));
      if (lo <= hi) {
        const int ilo = (int)lo, ihi = (int)hi;
        t = t < ilo ? ilo : (t > ihi ? ihi : t);
        out[i] = (TO)t;
      } else {
        out[i] = (TO)(unsigned int)t;   // pure cast: wraps modulo 2^bits
      }
    }
  }
}


#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
  if ((dtype) == SE3DS_F32) { CALL_F32; } else if ((dtype) == SE3DS_BF16) { CALL_BF16; } \
  else return SE3DS_E_BADDTYPE;

extern "C" {

int se3ds_mask_window(const float* mask, int n, int h, int w, int ho, int wo, int kh, int kw,
                      int stride, int pad_t, int pad_l, int wrap_w, float* ratio, float* um,
                      float* ru, float* bu, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || ho <= 0 || wo <= 0) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(mask_window_kernel, dim3(grid_for((int64_t)n * ho * wo, kB)), dim3(kB), 0,
                     as_stream(stream), mask, n, h, w, ho, wo, kh, kw, stride, pad_t, pad_l, wrap_w,
                     ratio, um, ru, bu);
  return check_launch("mask_window");
}

int se3ds_act_bwd(const void* dy, const void* y, int dtype, int64_t n, int act, float alpha,
                  void* dx, void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
#define L(T, V) hipLaunchKernelGGL((act_bwd_kernel<T, V>), dim3(grid_for(n / V, kB)), dim3(kB), 0, s, \
                                   (const T*)dy, (const T*)y, n / V, act, alpha, (T*)dx)
  DISPATCH_T(dtype, if (n % 4 == 0) L(float, 4); else L(float, 1),
             if (n % 8 == 0) L(uint16_t, 8); else L(uint16_t, 1))
#undef L
  return check_launch("act_bwd");
}

int se3ds_row_scale(const void* x, int dtype, int64_t rows, int c, const float* scale, void* out,
                    void* stream) {
  if (rows <= 0 || c <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
#define L(T, V) hipLaunchKernelGGL((row_scale_kernel<T, V>), dim3(grid_for(rows * (c / V), kB)), dim3(kB), \
                                   0, s, (const T*)x, scale, rows, c / V, (T*)out)
  DISPATCH_T(dtype, if (c % 4 == 0) L(float, 4); else L(float, 1),
             if (c % 8 == 0) L(uint16_t, 8); else L(uint16_t, 1))
#undef L
  return check_launch("row_scale");
}

int se3ds_add(const void* a, const void* b, int dtype, int64_t n, void* out, void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
#define L(T, V) hipLaunchKernelGGL((accumulate_kernel<T, V>), dim3(grid_for(n / V, kB)), dim3(kB), 0, s, \
                                   (const T*)a, (const T*)b, n / V, (T*)out)
  DISPATCH_T(dtype, if (n % 4 == 0) L(float, 4); else L(float, 1),
             if (n % 8 == 0) L(uint16_t, 8); else L(uint16_t, 1))
#undef L
  return check_launch("add");
}

int se3ds_maxpool2x2_fwd(const void* x, int dtype, int n, int h, int w, int c, void* y,
                         void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  int ho = (h + 1) / 2, wo = (w + 1) / 2;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for((int64_t)n * ho * wo * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_fwd_kernel<float>, g, dim3(kB), 0, s, (const float*)x, n, h, w, c, ho, wo, (float*)y),
             hipLaunchKernelGGL(maxpool_fwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)x, n, h, w, c, ho, wo, (uint16_t*)y))
  return check_launch("maxpool2x2_fwd");
}

int se3ds_maxpool2x2_bwd(const void* dy, const void* x, const void* y, int dtype, int n, int h,
                         int w, int c, void* dx, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  int ho = (h + 1) / 2, wo = (w + 1) / 2;
  hipStream_t s = as_stream(stream);
  if (dtype == SE3DS_BF16 && (c % 8) == 0) {
    hipLaunchKernelGGL(maxpool_bwd_vec8_kernel, dim3(grid_for((int64_t)n * ho * wo * (c / 8), kB)), dim3(kB), 0, s,
                       (const uint16_t*)dy, (const uint16_t*)x, (const uint16_t*)y, n, h, w, c, ho, wo,
                       (uint16_t*)dx);
    return check_launch("maxpool2x2_bwd");
  }
  dim3 g(grid_for((int64_t)n * h * w * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_bwd_kernel<float>, g, dim3(kB), 0, s, (const float*)dy, (const float*)x, (const float*)y, n, h, w, c, ho, wo, (float*)dx),
             hipLaunchKernelGGL(maxpool_bwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)dy, (const uint16_t*)x, (const uint16_t*)y, n, h, w, c, ho, wo, (uint16_t*)dx))
  return check_launch("maxpool2x2_bwd");
}

static void avg_geom(int h, int w, int* ho, int* wo, int* pt, int* pl) {
  *ho = (h + 1) / 2; *wo = (w + 1) / 2;
  int ph = (*ho - 1) * 2 + 3 - h; if (ph < 0) ph = 0;
  int pw = (*wo - 1) * 2 + 3 - w; if (pw < 0) pw = 0;
  *pt = ph / 2; *pl = pw / 2;
}

int se3ds_avgpool3s2_fwd(const void* x, int dtype, int n, int h, int w, int c, void* y,
                         void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  int ho, wo, pt, pl; avg_geom(h, w, &ho, &wo, &pt, &pl);
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for((int64_t)n * ho * wo * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(avgpool_fwd_kernel<float>, g, dim3(kB), 0, s, (const float*)x, n, h, w, c, ho, wo, pt, pl, (float*)y),
             hipLaunchKernelGGL(avgpool_fwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)x, n, h, w, c, ho, wo, pt, pl, (uint16_t*)y))
  return check_launch("avgpool3s2_fwd");
}

int se3ds_avgpool3s2_bwd(const void* dy, int dtype, int n, int h, int w, int c, void* dx,
                         void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  int ho, wo, pt, pl; avg_geom(h, w, &ho, &wo, &pt, &pl);
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for((int64_t)n * h * w * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(avgpool_bwd_kernel<float>, g, dim3(kB), 0, s, (const float*)dy, n, h, w, c, ho, wo, pt, pl, (float*)dx),
             hipLaunchKernelGGL(avgpool_bwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)dy, n, h, w, c, ho, wo, pt, pl, (uint16_t*)dx))
  return check_launch("avgpool3s2_bwd");
}

int se3ds_upsample2x_fwd(const void* x, int dtype, int n, int h, int w, int c, void* y,
                         void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for((int64_t)n * 4 * h * w * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, g, dim3(kB), 0, s, (const float*)x, n, h, w, c, (float*)y),
             hipLaunchKernelGGL(upsample2x_fwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)x, n, h, w, c, (uint16_t*)y))
  return check_launch("upsample2x_fwd");
}

int se3ds_upsample2x_bwd(const void* dy, int dtype, int n, int h, int w, int c, void* dx,
                         void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for((int64_t)n * h * w * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, g, dim3(kB), 0, s, (const float*)dy, n, h, w, c, (float*)dx),
             hipLaunchKernelGGL(upsample2x_bwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)dy, n, h, w, c, (uint16_t*)dx))
  return check_launch("upsample2x_bwd");
}

int se3ds_copy_channels(const void* src, int src_dtype, int src_c, int src_c0, void* dst,
                        int dst_dtype, int dst_c, int dst_c0, int ncopy, int64_t rows,
                        void* stream) {
  if (rows <= 0 || ncopy <= 0) return SE3DS_OK;
  if (src_c0 + ncopy > src_c || dst_c0 + ncopy > dst_c) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for(rows * ncopy, kB));
#define L(TS, TD) hipLaunchKernelGGL((copy_channels_kernel<TS, TD>), g, dim3(kB), 0, s, (const TS*)src, \
                                     src_c, src_c0, (TD*)dst, dst_c, dst_c0, ncopy, rows)
  if (src_dtype == SE3DS_F32 && dst_dtype == SE3DS_F32) L(float, float);
  else if (src_dtype == SE3DS_F32 && dst_dtype == SE3DS_BF16) L(float, uint16_t);
  else if (src_dtype == SE3DS_BF16 && dst_dtype == SE3DS_F32) L(uint16_t, float);
  else if (src_dtype == SE3DS_BF16 && dst_dtype == SE3DS_BF16) L(uint16_t, uint16_t);
  else return SE3DS_E_BADDTYPE;
#undef L
  return check_launch("copy_channels");
}

int se3ds_fill(void* p, int dtype, int64_t n, float value, void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for(n, kB));
  DISPATCH_T(dtype, hipLaunchKernelGGL(fill_kernel<float>, g, dim3(kB), 0, s, (float*)p, n, value),
             hipLaunchKernelGGL(fill_kernel<uint16_t>, g, dim3(kB), 0, s, (uint16_t*)p, n, value))
  return check_launch("fill");
}

int se3ds_quantize(const void* in, int in_dtype, int64_t n, int pre_clamp, float pre_lo,
                   float pre_hi, float mul, float div, float lo, float hi, void* out, int out_dtype,
                   void* stream) {
  if (n <= 0) return SE3DS_OK;
  if (div == 0.f) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for(n, kB));
#define Q(TI, TO) hipLaunchKernelGGL((quantize_kernel<TI, TO>), g, dim3(kB), 0, s, (const TI*)in, n, \
                                     pre_clamp, pre_lo, pre_hi, mul, div, lo, hi, (TO*)out)
  if (in_dtype == SE3DS_F32) {
    if (out_dtype == SE3DS_F32) Q(float, float);
    else if (out_dtype == SE3DS_I32) Q(float, int32_t);
    else if (out_dtype == SE3DS_U8) Q(float, uint8_t);
    else return SE3DS_E_BADDTYPE;
  } else if (in_dtype == SE3DS_I32) {
    if (out_dtype == SE3DS_F32) Q(int32_t, float);
    else if (out_dtype == SE3DS_I32) Q(int32_t, int32_t);
    else if (out_dtype == SE3DS_U8) Q(int32_t, uint8_t);
    else return SE3DS_E_BADDTYPE;
  } else if (in_dtype == SE3DS_U8) {
    if (out_dtype == SE3DS_F32) Q(uint8_t, float);
    else if (out_dtype == SE3DS_I32) Q(uint8_t, int32_t);
    else return SE3DS_E_BADDTYPE;
  } else return SE3DS_E_BADDTYPE;
#undef Q
  return check_launch("quantize");
}

int se3ds_weight_prep_multi(const int64_t* table, int nlayers, int64_t total_tiles, void* stream) {
  if (nlayers <= 0 || total_tiles <= 0) return SE3DS_OK;
  if (total_tiles >= ((int64_t)1 << 31)) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(weight_prep_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0,
                     as_stream(stream), table, nlayers);
  return check_launch("weight_prep_multi");
}

int se3ds_pad2d(const void* x, int dtype, int n, int h, int w, int c, int pad, int mode, int wrap_w,
                float value, void* y, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || pad < 0 || pad > h || pad > w) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for((int64_t)n * (h + 2 * pad) * (w + 2 * pad) * c, kB));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pad2d_kernel<float>, g, dim3(kB), 0, s, (const float*)x, n, h, w, c, pad, mode, wrap_w, value, (float*)y),
             hipLaunchKernelGGL(pad2d_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)x, n, h, w, c, pad, mode, wrap_w, value, (uint16_t*)y))
  return check_launch("pad2d");
}

int se3ds_head_fwd(const void* x, int dtype, int64_t n, int kind, float* y, void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for(n, kB));
  DISPATCH_T(dtype, hipLaunchKernelGGL(head_fwd_kernel<float>, g, dim3(kB), 0, s, (const float*)x, n, kind, y),
             hipLaunchKernelGGL(head_fwd_kernel<uint16_t>, g, dim3(kB), 0, s, (const uint16_t*)x, n, kind, y))
  return check_launch("head_fwd");
}

int se3ds_head_bwd(const float* dy, const float* y, const void* x, int dtype, int64_t n, int kind,
                   void* dx, void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for(n, kB));
  DISPATCH_T(dtype, hipLaunchKernelGGL(head_bwd_kernel<float>, g, dim3(kB), 0, s, dy, y, (const float*)x, n, kind, (float*)dx),
             hipLaunchKernelGGL(head_bwd_kernel<uint16_t>, g, dim3(kB), 0, s, dy, y, (const uint16_t*)x, n, kind, (uint16_t*)dx))
  return check_launch("head_bwd");
}

int se3ds_sample_sum(const float* a, const float* b, const float* m, int n, int64_t p, int c,
                     int mode, float* out, float* workspace, void* stream) {
  if (n <= 0 || p <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  if (!workspace) return SE3DS_E_WORKSPACE;   // n * 256 floats
  constexpr int kChunks = 256;
  int chunks = (int)ceil_div(p * c, (int64_t)kB * 8);
  if (chunks > kChunks) chunks = kChunks;
  if (chunks < 1) chunks = 1;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(sample_sum_kernel, dim3(chunks, n), dim3(kB), 0, s, a, b, m, p, c, mode,
                     workspace);
  hipLaunchKernelGGL(sample_sum_final_kernel, dim3(n), dim3(kB), 0, s, workspace, chunks, out);
  return check_launch("sample_sum");
}

int se3ds_l1_grad(const float* a, const float* b, const float* m, const float* m2,
                  const float* coef, int n, int64_t p, int c, int mode, float* grad, void* stream) {
  if (n <= 0 || p <= 0 || c <= 0) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(l1_grad_kernel, dim3(grid_for((int64_t)n * p * c, kB)), dim3(kB), 0,
                     as_stream(stream), a, b, m, m2, coef, p, c, n, mode, grad);
  return check_launch("l1_grad");
}

int se3ds_recip_clamp(const float* sums, int n, float scale, float* out, void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipLaunchKernelGGL(recip_clamp_kernel, dim3(grid_for(n, kB)), dim3(kB), 0, as_stream(stream), sums,
                     n, scale, out);
  return check_launch("recip_clamp");
}

int se3ds_hinge(const void* logits, int dtype, int64_t half, float cd, float cg, float* sums,
                void* dlog_d, void* dlog_g, void* stream) {
  if (half <= 0) return SE3DS_E_BADSHAPE;
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(hinge_kernel<float>, dim3(1), dim3(kB), 0, s, (const float*)logits, half, cd, cg, sums, (float*)dlog_d, (float*)dlog_g),
             hipLaunchKernelGGL(hinge_kernel<uint16_t>, dim3(1), dim3(kB), 0, s, (const uint16_t*)logits, half, cd, cg, sums, (uint16_t*)dlog_d, (uint16_t*)dlog_g))
  return check_launch("hinge");
}

}  // extern "C"
