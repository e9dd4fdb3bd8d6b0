// Multi-tensor optimiser kernels over flat fp32 parameter arenas (per-tensor clip-by-norm,
// Keras-form Adam, EMA) and the layer-batched spectral-norm power iteration / backward
// fix-up.  All HBM-bound streaming; reductions are two-stage and deterministic.
//   clip   trainers/se3ds_trainer.py:27-32  (tf.clip_by_norm per tensor, before aggregation)
//   adam   trainers/gan_manager.py:175-183  (Keras Adam -> ResourceApplyAdam)
//   ema    utils/ema.py:54-64
//   SN     models/layers.py:312-331 (power iteration), gradient through sigma = v W u^T
#include "common.h"
#include <cmath>

namespace se3ds {
namespace {

constexpr int kB = 256;

// chunk table: int64 triples (tensor id, start element, length)
__global__ void __launch_bounds__(kB)
chunk_sqsum_kernel(const float* __restrict__ g, const int64_t* __restrict__ chunks,
                   float* __restrict__ partial, const int64_t* __restrict__ tensor_sn) {
  __shared__ float sh[kB / 64];
  const int64_t start = chunks[blockIdx.x * 3 + 1], len = chunks[blockIdx.x * 3 + 2];
  if (tensor_sn && tensor_sn[chunks[blockIdx.x * 3]] != 0) {   // (block-uniform)
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.f;
    return;
  }
  float s = 0.f;
  // 16-byte loads (round 5: the dword loop ran this pass at ~1.5 TB/s next to the backward pass);
  // tensors start on 16-byte boundaries and chunks are 65536 elements, so only a tensor's last
  // chunk has a tail
  const float* gp = g + start;
  const int64_t n4 = (((uintptr_t)gp & 15) == 0) ? (len >> 2) : 0;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int64_t i = threadIdx.x; i < n4; i += kB) {
    const float4 v = reinterpret_cast<const float4*>(gp)[i];
    s += v.x * v.x;
    s1 += v.y * v.y;
    s2 += v.z * v.z;
    s3 += v.w * v.w;
  }
  for (int64_t i = 4 * n4 + threadIdx.x; i < len; i += kB) {
    const float v = gp[i];
    s += v * v;
  }
  s = (s + s1) + (s2 + s3);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kB / 64; ++i) t += sh[i];
    partial[blockIdx.x] = t;
  }
}

// tensor_chunk_start: (T+1) prefix of chunk indices per tensor
__global__ void __launch_bounds__(kB)
tensor_sqsum_kernel(const float* __restrict__ partial, const int64_t* __restrict__ tensor_chunk_start,
                    int T, float* __restrict__ sqnorm, const int64_t* __restrict__ tensor_sn) {
  for (int t = blockIdx.x * kB + threadIdx.x; t < T; t += gridDim.x * kB) {
    if (tensor_sn && tensor_sn[t] != 0) continue;   // written by sn_sqnorm_kernel
    float s = 0.f;
    for (int64_t c = tensor_chunk_start[t]; c < tensor_chunk_start[t + 1]; ++c) s += partial[c];
    sqnorm[t] = s;
  }
}

// tf.clip_by_norm: (g * clip) / max(l2norm, clip), l2norm = sqrt(sum g^2) (0 if the sum is 0)
constexpr int SN_RB = 64;   // row blocks for v = W u and for dot products (vpart holds 3 * SN_RB + 2)

// <G, W> of one spectral layer from its SN_RB partials (sn_dots_kernel / sn_dot_kernel): ONE
// summation order for the fix-up (sn_fix_kernel), the fused fix-up inside clip_kernel and the
// closed-form squared norm (sn_sqnorm_kernel), so the clip denominator always matches the
// gradient it is applied to.
__device__ inline float sn_dot_gw(const float* __restrict__ vpart) {
  float dot = 0.f;
  for (int i = 0; i < SN_RB; ++i) dot += vpart[i];
  return dot;
}

// Per-element arithmetic shared by the separate passes (clip_kernel, adam_kernel, adam_ema_kernel)
// and the fused one (clip_adam_kernel): one expression each, so that both compile to the same
// operation sequence (contractions included) and the fused update is bit-identical.
// (Every operation is pinned with the non-contractable intrinsics: left to -ffp-contract=fast the
// compiler fused `inv * g - coef * v * u` one way in clip_kernel and the other way in the fused
// kernel, where coef * v is shared by four elements -- a 1-ulp difference in the update.)
__device__ __forceinline__ float clip_plain(float g, float clip, float den) {
  return __fdiv_rn(__fmul_rn(g, clip), den);
}
__device__ __forceinline__ float clip_sn(float g, float inv, float coef, float vk, float uc,
                                         float clip, float den) {
  const float f = __fmaf_rn(inv, g, -__fmul_rn(__fmul_rn(coef, vk), uc));
  return __fdiv_rn(__fmul_rn(f, clip), den);
}
__device__ __forceinline__ void adam_elem(float gi, float& p, float& m, float& v, float alpha,
                                          float b1, float b2, float eps) {
  const float mi = __fmaf_rn(__fsub_rn(gi, m), __fsub_rn(1.0f, b1), m);
  const float vi = __fmaf_rn(__fmaf_rn(gi, gi, -v), __fsub_rn(1.0f, b2), v);
  m = mi;
  v = vi;
  p = __fsub_rn(p, __fdiv_rn(__fmul_rn(mi, alpha), __fadd_rn(__fsqrt_rn(vi), eps)));
}
__device__ __forceinline__ float ema_elem(float e, float pn, float omd) {
  return __fmaf_rn(-__fsub_rn(e, pn), omd, e);
}

__global__ void __launch_bounds__(kB)
clip_kernel(float* __restrict__ g, const int64_t* __restrict__ chunks,
            const float* __restrict__ sqnorm, float clip, const int64_t* __restrict__ tensor_sn) {
  const int64_t t = chunks[blockIdx.x * 3], start = chunks[blockIdx.x * 3 + 1],
                len = chunks[blockIdx.x * 3 + 2];
  const float sq = sqnorm[t];
  const float norm = sq > 0.f ? sqrtf(sq) : sq;
  const float den = fmaxf(norm, clip);
  const int64_t* L = tensor_sn ? (const int64_t*)tensor_sn[t] : nullptr;
  if (L == nullptr) {
    for (int64_t i = threadIdx.x; i < len; i += kB) g[start + i] = clip_plain(g[start + i], clip, den);
    return;
  }
  // spectral layer whose gradient still is dL/d(W/sigma): the fix-up of sn_fix_kernel
  // (G := inv*G - inv^2 <G,W> vhat uhat^T) is applied here, in the clip pass (one read-modify-write
  // of the arena instead of two).  Table fields as in the SN section below.
  const float* v = (const float*)L[2];
  const float* uhat = (const float*)L[3];
  const float* sig = (const float*)L[4];
  const float* vpart = (const float*)L[6];
  const int C = (int)L[8];
  const float dot = sn_dot_gw(vpart);
  const float inv = sig[1];
  const float coef = inv * inv * dot;
  const int64_t e0 = (g + start) - (const float*)L[9] + threadIdx.x;   // element of the tensor
  int64_t k = e0 / C;
  int c = (int)(e0 - k * C);
  const int dq = kB / C, dr = kB - dq * C;
  for (int64_t i = threadIdx.x; i < len; i += kB) {
    g[start + i] = clip_sn(g[start + i], inv, coef, v[k], uhat[c], clip, den);
    k += dq;
    c += dr;
    if (c >= C) { c -= C; ++k; }
  }
}

// metric: mean over tensors of ||clipped g|| = norm * clip / max(norm, clip)
__global__ void __launch_bounds__(kB)
mean_clipped_norm_kernel(const float* __restrict__ sqnorm, int T, float clip,
                         float* __restrict__ out) {
  __shared__ float sh[kB / 64];
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += kB) {
    float sq = sqnorm[t];
    float norm = sq > 0.f ? sqrtf(sq) : sq;
    s += (norm * clip) / fmaxf(norm, clip);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kB / 64; ++i) t += sh[i];
    t = t / (float)T;
    out[0] = (t != t) ? 0.f : t;  // NaN -> 0 (se3ds_trainer.py:240-242)
  }
}

// ResourceApplyAdam: m += (g - m)(1-b1); v += (g^2 - v)(1-b2); p -= (m * alpha)/(sqrt(v)+eps)
__global__ void __launch_bounds__(kB)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float alpha, float b1, float b2, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_elem(g[i], pi, mi, vi, alpha, b1, b2, eps);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// The same with the generator's EMA copy advanced in the same pass (saves re-reading theta):
// ema -= (ema - p_new) * omd, exactly what ema_kernel computes afterwards.
__global__ void __launch_bounds__(kB)
adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                float* __restrict__ v, int64_t n, float alpha, float b1, float b2, float eps,
                float* __restrict__ ema, float omd) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_elem(g[i], pi, mi, vi, alpha, b1, b2, eps);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    ema[i] = ema_elem(ema[i], pi, omd);
  }
}

// Round 4: per-tensor clip (+ the spectral fix-up) applied INSIDE the Adam (+ EMA) pass -- one
// replica only, where nobody else needs the clipped arena: the gradient is read once and never
// rewritten (clip_kernel's read-modify-write of the arena, 8.9 GB per generator step, is gone).
// One workgroup per chunk row (tensor id, start, length) as clip_kernel; 16-byte accesses where a
// chunk allows them (chunks start 16-byte aligned: tensors are, and the chunk length is a multiple
// of 4 except at a tensor's end).  Same arithmetic per element as clip_kernel followed by
// adam_ema_kernel (shared inline functions above): bit-identical update.
template <bool EMA>
__global__ void __launch_bounds__(kB)
clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, const int64_t* __restrict__ chunks,
                 const float* __restrict__ sqnorm, float clip,
                 const int64_t* __restrict__ tensor_sn, float alpha, float b1, float b2, float eps,
                 float* __restrict__ ema, float omd) {
  const int64_t t = chunks[blockIdx.x * 3], start = chunks[blockIdx.x * 3 + 1],
                len = chunks[blockIdx.x * 3 + 2];
  const float sq = sqnorm[t];
  const float norm = sq > 0.f ? sqrtf(sq) : sq;
  const float den = fmaxf(norm, clip);
  const int64_t* L = tensor_sn ? (const int64_t*)tensor_sn[t] : nullptr;
  const float* sv = nullptr;
  const float* su = nullptr;
  float inv = 0.f, coef = 0.f;
  int C = 4;
  int64_t e_base = 0;
  if (L != nullptr) {
    sv = (const float*)L[2];
    su = (const float*)L[3];
    const float* sig = (const float*)L[4];
    const float* vpart = (const float*)L[6];
    C = (int)L[8];
    const float dot = sn_dot_gw(vpart);
    inv = sig[1];
    coef = inv * inv * dot;
    e_base = (g + start) - (const float*)L[9];   // element of the tensor this chunk starts at
  }
  auto one = [&](int64_t i, float gi) {   // clipped gradient of element start + i
    if (L == nullptr) return clip_plain(gi, clip, den);
    const int64_t e = e_base + i;
    const int64_t k = e / C;
    const int c = (int)(e - k * C);
    return clip_sn(gi, inv, coef, sv[k], su[c], clip, den);
  };
  const int64_t len4 = ((C & 3) == 0) ? (len & ~(int64_t)3) : 0;   // (rows of C: a float4 stays in one)
  for (int64_t i = (int64_t)threadIdx.x * 4; i < len4; i += (int64_t)kB * 4) {
    const int64_t o = start + i;
    const float4 g4 = *reinterpret_cast<const float4*>(g + o);
    float4 p4 = *reinterpret_cast<const float4*>(p + o);
    float4 m4 = *reinterpret_cast<const float4*>(m + o);
    float4 v4 = *reinterpret_cast<const float4*>(v + o);
    float gc[4];
    if (L == nullptr) {
      gc[0] = clip_plain(g4.x, clip, den); gc[1] = clip_plain(g4.y, clip, den);
      gc[2] = clip_plain(g4.z, clip, den); gc[3] = clip_plain(g4.w, clip, den);
    } else {
      const int64_t e = e_base + i;
      const int64_t k = e / C;
      const int c = (int)(e - k * C);
      const float vk = sv[k];
      const float4 u4 = *reinterpret_cast<const float4*>(su + c);
      gc[0] = clip_sn(g4.x, inv, coef, vk, u4.x, clip, den);
      gc[1] = clip_sn(g4.y, inv, coef, vk, u4.y, clip, den);
      gc[2] = clip_sn(g4.z, inv, coef, vk, u4.z, clip, den);
      gc[3] = clip_sn(g4.w, inv, coef, vk, u4.w, clip, den);
    }
    adam_elem(gc[0], p4.x, m4.x, v4.x, alpha, b1, b2, eps);
    adam_elem(gc[1], p4.y, m4.y, v4.y, alpha, b1, b2, eps);
    adam_elem(gc[2], p4.z, m4.z, v4.z, alpha, b1, b2, eps);
    adam_elem(gc[3], p4.w, m4.w, v4.w, alpha, b1, b2, eps);
    *reinterpret_cast<float4*>(m + o) = m4;
    *reinterpret_cast<float4*>(v + o) = v4;
    *reinterpret_cast<float4*>(p + o) = p4;
    if (EMA) {
      float4 e4 = *reinterpret_cast<const float4*>(ema + o);
      e4.x = ema_elem(e4.x, p4.x, omd); e4.y = ema_elem(e4.y, p4.y, omd);
      e4.z = ema_elem(e4.z, p4.z, omd); e4.w = ema_elem(e4.w, p4.w, omd);
      *reinterpret_cast<float4*>(ema + o) = e4;
    }
  }
  for (int64_t i = len4 + threadIdx.x; i < len; i += kB) {
    const int64_t o = start + i;
    float pi = p[o], mi = m[o], vi = v[o];
    adam_elem(one(i, g[o]), pi, mi, vi, alpha, b1, b2, eps);
    m[o] = mi;
    v[o] = vi;
    p[o] = pi;
    if (EMA) ema[o] = ema_elem(ema[o], pi, omd);
  }
}

__global__ void __launch_bounds__(kB)
ema_kernel(float* __restrict__ ema, const float* __restrict__ var, int64_t n, float omd) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB)
    ema[i] = ema[i] - (ema[i] - var[i]) * omd;
}

// ------------------------------------------------------------------ spectral norm (batched)
// Layer table: 10 int64 per layer:
//   0 W (float*)  1 u (float*)  2 v (float*, K)  3 uhat (float*, Cout)  4 sig (float*, 2)
//   5 part (float*, SN_KB*Cout)  6 vpart (float*, SN_RB)  7 K  8 Cout  9 grad (float*)
constexpr int SN_F = 10;
constexpr int SN_KB = 32;   // K slabs for u' = v^T W
constexpr float SN_EPS = 1e-10f;

// v[k] = sum_c W[k][c] u[c]; one wave per row; vpart[block] = sum of v[k]^2 over its rows
__global__ void __launch_bounds__(kB)
sn_v_kernel(const int64_t* __restrict__ tab) {
  const int64_t* L = tab + (int64_t)blockIdx.y * SN_F;
  const float* W = (const float*)L[0];
  const float* u = (const float*)L[1];
  float* v = (float*)L[2];
  float* vpart = (float*)L[6];
  const int64_t K = L[7];
  const int C = (int)L[8];
  __shared__ float sh[kB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sq = 0.f;
  for (int64_t k = (int64_t)blockIdx.x * 4 + wave; k < K; k += (int64_t)SN_RB * 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += W[k * C + c] * u[c];
    s = wave_sum(s);
    if (lane == 0) { v[k] = s; sq += s * s; }
  }
  if (lane == 0) sh[wave] = sq;
  __syncthreads();
  if (threadIdx.x == 0) vpart[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// partial[kb][c] = sum_{k in slab kb} (v[k] * inv_vnorm) * W[k][c]
__global__ void __launch_bounds__(kB)
sn_u_kernel(const int64_t* __restrict__ tab) {
  const int64_t* L = tab + (int64_t)blockIdx.y * SN_F;
  const float* W = (const float*)L[0];
  const float* v = (const float*)L[2];
  float* part = (float*)L[5];
  const float* vpart = (const float*)L[6];
  const int64_t K = L[7];
  const int C = (int)L[8];
  float vn = 0.f;
  for (int i = 0; i < SN_RB; ++i) vn += vpart[i];
  const float inv_vn = 1.0f / (sqrtf(vn) + SN_EPS);
  const int64_t per = ceil_div(K, SN_KB);
  const int64_t k_lo = (int64_t)blockIdx.x * per;
  int64_t k_hi = k_lo + per;
  if (k_hi > K) k_hi = K;
  for (int c = threadIdx.x; c < C; c += kB) {
    float s = 0.f;
    for (int64_t k = k_lo; k < k_hi; ++k) s += (v[k] * inv_vn) * W[k * C + c];
    part[(int64_t)blockIdx.x * C + c] = s;
  }
}

// u' -> uhat, sigma = u' . uhat, inv = 1/(sigma+eps); v := vhat; u := uhat if training
__global__ void __launch_bounds__(kB)
sn_finish_kernel(const int64_t* __restrict__ tab, int training) {
  const int64_t* L = tab + (int64_t)blockIdx.x * SN_F;
  float* u = (float*)L[1];
  float* v = (float*)L[2];
  float* uhat = (float*)L[3];
  float* sig = (float*)L[4];
  const float* part = (const float*)L[5];
  const float* vpart = (const float*)L[6];
  const int64_t K = L[7];
  const int C = (int)L[8];
  __shared__ float sh[kB / 64];
  __shared__ float s_bcast;
  float sq = 0.f;
  for (int c = threadIdx.x; c < C; c += kB) {
    float s = 0.f;
    for (int b = 0; b < SN_KB; ++b) s += part[(int64_t)b * C + c];
    uhat[c] = s;  // u' for now
    sq += s * s;
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) s_bcast = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  const float un = sqrtf(s_bcast);
  const float inv_un = 1.0f / (un + SN_EPS);
  float dot = 0.f;
  for (int c = threadIdx.x; c < C; c += kB) {
    float up = uhat[c];
    float uh = up * inv_un;
    dot += up * uh;
    uhat[c] = uh;
    if (training) u[c] = uh;
  }
  dot = wave_sum(dot);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    float sigma = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    sig[0] = sigma;
    sig[1] = 1.0f / (sigma + SN_EPS);
  }
  float vn = 0.f;
  for (int i = 0; i < SN_RB; ++i) vn += vpart[i];
  const float inv_vn = 1.0f / (sqrtf(vn) + SN_EPS);
  for (int64_t k = threadIdx.x; k < K; k += kB) v[k] = v[k] * inv_vn;
}

// vpart[block] = partial of <G, W>
__global__ void __launch_bounds__(kB)
sn_dot_kernel(const int64_t* __restrict__ tab) {
  const int64_t* L = tab + (int64_t)blockIdx.y * SN_F;
  const float* W = (const float*)L[0];
  float* vpart = (float*)L[6];
  const float* G = (const float*)L[9];
  const int64_t n = L[7] * L[8];
  __shared__ float sh[kB / 64];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)SN_RB * kB)
    s += G[i] * W[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) vpart[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// G := inv*G - inv^2 * <G,W> * vhat[k] * uhat[c]
__global__ void __launch_bounds__(kB)
sn_fix_kernel(const int64_t* __restrict__ tab) {
  const int64_t* L = tab + (int64_t)blockIdx.y * SN_F;
  const float* v = (const float*)L[2];
  const float* uhat = (const float*)L[3];
  const float* sig = (const float*)L[4];
  const float* vpart = (const float*)L[6];
  float* G = (float*)L[9];
  const int C = (int)L[8];
  const int64_t n = L[7] * C;
  const float dot = sn_dot_gw(vpart);
  const float inv = sig[1];
  const float coef = inv * inv * dot;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)SN_RB * kB) {
    int64_t k = i / C;
    int c = (int)(i - k * C);
    G[i] = inv * G[i] - coef * v[k] * uhat[c];
  }
}

// Fused variant of the fix-up (round 2): pass 1 of the backward fix-up with everything the clip
// pass needs.  Per layer and block: partials of <G,W>, <G,G> and <G, vhat uhat^T>; block 0 also
// |vhat|^2 and |uhat|^2.  vpart layout: [0,RB) dot, [RB,2RB) gg, [2RB,3RB) gvu, [3RB] |v|^2, [3RB+1] |u|^2.
// One wave per row of W (rows of C floats are contiguous), no integer division per element.
__global__ void __launch_bounds__(kB)
sn_dots_kernel(const int64_t* __restrict__ tab) {
  const int64_t* L = tab + (int64_t)blockIdx.y * SN_F;
  const float* W = (const float*)L[0];
  const float* v = (const float*)L[2];
  const float* uhat = (const float*)L[3];
  float* vpart = (float*)L[6];
  const float* G = (const float*)L[9];
  const int64_t K = L[7];
  const int C = (int)L[8];
  __shared__ float sh[3][kB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dot = 0.f, gg = 0.f, gvu = 0.f;
  // 16-byte loads where the rows allow (C % 4 == 0, aligned bases -- every layer of the models: the
  // dword loop ran a module's launch at 1.8 TB/s, 332 us next to the backward pass, round 5)
  const bool vec = (C & 3) == 0 && ((((uintptr_t)G | (uintptr_t)W | (uintptr_t)uhat) & 15) == 0);
  for (int64_t k = (int64_t)blockIdx.x * 4 + wave; k < K; k += (int64_t)SN_RB * 4) {
    const float vk = v[k];
    float row = 0.f;
    if (vec) {
      const float4* G4 = reinterpret_cast<const float4*>(G + k * C);
      const float4* W4 = reinterpret_cast<const float4*>(W + k * C);
      const float4* U4 = reinterpret_cast<const float4*>(uhat);
      for (int c = lane; c < (C >> 2); c += 64) {
        const float4 gv = G4[c], wv = W4[c], uv = U4[c];
        dot += (gv.x * wv.x + gv.y * wv.y) + (gv.z * wv.z + gv.w * wv.w);
        gg += (gv.x * gv.x + gv.y * gv.y) + (gv.z * gv.z + gv.w * gv.w);
        row += (gv.x * uv.x + gv.y * uv.y) + (gv.z * uv.z + gv.w * uv.w);
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float gv = G[k * C + c];
        dot += gv * W[k * C + c];
        gg += gv * gv;
        row += gv * uhat[c];
      }
    }
    gvu += row * vk;
  }
  dot = wave_sum(dot);
  gg = wave_sum(gg);
  gvu = wave_sum(gvu);
  if (lane == 0) { sh[0][wave] = dot; sh[1][wave] = gg; sh[2][wave] = gvu; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float* q = sh[threadIdx.x];
    vpart[threadIdx.x * SN_RB + blockIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
  }
  if (blockIdx.x == 0) {
    __syncthreads();
    float nv = 0.f, nu = 0.f;
    for (int64_t k = threadIdx.x; k < K; k += kB) nv += v[k] * v[k];
    for (int c = threadIdx.x; c < C; c += kB) nu += uhat[c] * uhat[c];
    nv = wave_sum(nv);
    nu = wave_sum(nu);
    if (lane == 0) { sh[0][wave] = nv; sh[1][wave] = nu; }
    __syncthreads();
    if (threadIdx.x < 2) {
      const float* q = sh[threadIdx.x];
      vpart[3 * SN_RB + threadIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
    }
  }
}

// |inv*G - coef*v u^T|^2 = inv^2 <G,G> - 2 inv coef <G, v u^T> + coef^2 |v|^2 |u|^2, coef = inv^2 <G,W>
// (combined in binary64: the three terms cancel when G is nearly parallel to v u^T).  One thread
// per tensor of [tensor_base, tensor_base + T); sqnorm is indexed relative to tensor_base.
__global__ void __launch_bounds__(kB)
sn_sqnorm_kernel(const int64_t* __restrict__ tensor_sn, int tensor_base, int T,
                 float* __restrict__ sqnorm) {
  for (int t = blockIdx.x * kB + threadIdx.x; t < T; t += gridDim.x * kB) {
    const int64_t* L = (const int64_t*)tensor_sn[tensor_base + t];
    if (L == nullptr) continue;
    const float* sig = (const float*)L[4];
    const float* vpart = (const float*)L[6];
    const float dot = sn_dot_gw(vpart);   // (fp32, the order sn_fix_kernel / clip_kernel use)
    double gg = 0.0, gvu = 0.0;
    for (int i = 0; i < SN_RB; ++i) {
      gg += (double)vpart[SN_RB + i];
      gvu += (double)vpart[2 * SN_RB + i];
    }
    const float inv = sig[1];
    const float coef = inv * inv * dot;
    const double nvu = (double)vpart[3 * SN_RB] * (double)vpart[3 * SN_RB + 1];
    double sq = (double)inv * inv * gg - 2.0 * (double)inv * coef * gvu + (double)coef * coef * nvu;
    sqnorm[t] = sq > 0.0 ? (float)sq : 0.f;
  }
}

}  // namespace
}  // namespace se3ds

using namespace se3ds;

// Keras Adam step size: lr * sqrt(1 - beta2^t) / (1 - beta1^t) with the powers taken by ONE pow
// each (tf.pow in optimizer_v2/adam.py), not by a t-step product: O(1) host work at any step
// count and no accumulated rounding drift.  Evaluated in double, rounded once per fp32 op.
static float adam_alpha(float lr, float beta1, float beta2, int64_t step) {
  float b1p = (float)pow((double)beta1, (double)step);
  float b2p = (float)pow((double)beta2, (double)step);
  return lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}

extern "C" {

int se3ds_multi_sqnorm_sn(const float* grads, const int64_t* chunks, int64_t nchunks,
                          const int64_t* tensor_chunk_start, int ntensors, float* partial,
                          float* sqnorm, const int64_t* tensor_sn, int tensor_base, void* stream) {
  if (nchunks <= 0 || ntensors <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(chunk_sqsum_kernel, dim3((unsigned)nchunks), dim3(kB), 0, s, grads, chunks,
                     partial, tensor_sn);
  hipLaunchKernelGGL(tensor_sqsum_kernel, dim3(grid_for(ntensors, kB)), dim3(kB), 0, s, partial,
                     tensor_chunk_start, ntensors, sqnorm,
                     tensor_sn ? tensor_sn + tensor_base : nullptr);
  if (tensor_sn)
    hipLaunchKernelGGL(sn_sqnorm_kernel, dim3(grid_for(ntensors, kB)), dim3(kB), 0, s, tensor_sn,
                       tensor_base, ntensors, sqnorm);
  return check_launch("multi_sqnorm");
}

int se3ds_multi_sqnorm(const float* grads, const int64_t* chunks, int64_t nchunks,
                       const int64_t* tensor_chunk_start, int ntensors, float* partial,
                       float* sqnorm, void* stream) {
  return se3ds_multi_sqnorm_sn(grads, chunks, nchunks, tensor_chunk_start, ntensors, partial, sqnorm,
                               nullptr, 0, stream);
}

int se3ds_multi_clip_by_norm(float* grads, const int64_t* chunks, int64_t nchunks,
                             const float* sqnorm, int ntensors, float clip_norm,
                             float* mean_norm_out, void* stream) {
  return se3ds_multi_clip_by_norm_sn(grads, chunks, nchunks, sqnorm, ntensors, clip_norm,
                                     mean_norm_out, nullptr, stream);
}

int se3ds_multi_clip_by_norm_sn(float* grads, const int64_t* chunks, int64_t nchunks,
                                const float* sqnorm, int ntensors, float clip_norm,
                                float* mean_norm_out, const int64_t* tensor_sn, void* stream) {
  if (nchunks <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(clip_kernel, dim3((unsigned)nchunks), dim3(kB), 0, s, grads, chunks, sqnorm,
                     clip_norm, tensor_sn);
  if (mean_norm_out)
    hipLaunchKernelGGL(mean_clipped_norm_kernel, dim3(1), dim3(kB), 0, s, sqnorm, ntensors,
                       clip_norm, mean_norm_out);
  return check_launch("multi_clip_by_norm");
}

int se3ds_mean_clipped_norm(const float* sqnorm, int ntensors, float clip_norm, float* out,
                            void* stream) {
  if (ntensors <= 0) return SE3DS_OK;
  hipLaunchKernelGGL(mean_clipped_norm_kernel, dim3(1), dim3(kB), 0, as_stream(stream), sqnorm,
                     ntensors, clip_norm, out);
  return check_launch("mean_clipped_norm");
}

int se3ds_multi_adam_keras(float* params, const float* grads, float* m, float* v, int64_t n,
                           float lr, float beta1, float beta2, float eps, int64_t step,
                           void* stream) {
  if (n <= 0) return SE3DS_OK;
  float alpha = adam_alpha(lr, beta1, beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, kB)), dim3(kB), 0, as_stream(stream), params,
                     grads, m, v, n, alpha, beta1, beta2, eps);
  return check_launch("multi_adam_keras");
}

int se3ds_multi_adam_keras_ema(float* params, const float* grads, float* m, float* v, int64_t n,
                               float lr, float beta1, float beta2, float eps, int64_t step,
                               float* ema, float one_minus_decay, void* stream) {
  if (ema == nullptr)
    return se3ds_multi_adam_keras(params, grads, m, v, n, lr, beta1, beta2, eps, step, stream);
  if (n <= 0) return SE3DS_OK;
  float alpha = adam_alpha(lr, beta1, beta2, step);
  hipLaunchKernelGGL(adam_ema_kernel, dim3(grid_for(n, kB)), dim3(kB), 0, as_stream(stream), params,
                     grads, m, v, n, alpha, beta1, beta2, eps, ema, one_minus_decay);
  return check_launch("multi_adam_keras_ema");
}

int se3ds_multi_clip_adam_keras_ema(float* params, const float* grads, float* m, float* v,
                                    const int64_t* chunks, int64_t nchunks, const float* sqnorm,
                                    float clip_norm, const int64_t* tensor_sn, float lr, float beta1,
                                    float beta2, float eps, int64_t step, float* ema,
                                    float one_minus_decay, void* stream) {
  if (nchunks <= 0) return SE3DS_OK;
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0)
    return SE3DS_E_UNSUPPORTED;
  const float alpha = adam_alpha(lr, beta1, beta2, step);
  if (ema)
    hipLaunchKernelGGL(clip_adam_kernel<true>, dim3((unsigned)nchunks), dim3(kB), 0, as_stream(stream),
                       params, grads, m, v, chunks, sqnorm, clip_norm, tensor_sn, alpha, beta1, beta2,
                       eps, ema, one_minus_decay);
  else
    hipLaunchKernelGGL(clip_adam_kernel<false>, dim3((unsigned)nchunks), dim3(kB), 0, as_stream(stream),
                       params, grads, m, v, chunks, sqnorm, clip_norm, tensor_sn, alpha, beta1, beta2,
                       eps, ema, one_minus_decay);
  return check_launch("multi_clip_adam_keras_ema");
}

int se3ds_multi_ema(float* ema, const float* vars, int64_t n, float one_minus_decay,
                    void* stream) {
  if (n <= 0) return SE3DS_OK;
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, kB)), dim3(kB), 0, as_stream(stream), ema, vars,
                     n, one_minus_decay);
  return check_launch("multi_ema");
}

int se3ds_spectral_power_iter(const int64_t* table, int nlayers, int training, void* stream) {
  if (nlayers <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(sn_v_kernel, dim3(SN_RB, (unsigned)nlayers), dim3(kB), 0, s, table);
  hipLaunchKernelGGL(sn_u_kernel, dim3(SN_KB, (unsigned)nlayers), dim3(kB), 0, s, table);
  hipLaunchKernelGGL(sn_finish_kernel, dim3((unsigned)nlayers), dim3(kB), 0, s, table, training);
  return check_launch("spectral_power_iter");
}

int se3ds_spectral_bwd_fixup(const int64_t* table, int nlayers, void* stream) {
  if (nlayers <= 0) return SE3DS_OK;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(sn_dot_kernel, dim3(SN_RB, (unsigned)nlayers), dim3(kB), 0, s, table);
  hipLaunchKernelGGL(sn_fix_kernel, dim3(SN_RB, (unsigned)nlayers), dim3(kB), 0, s, table);
  return check_launch("spectral_bwd_fixup");
}

int se3ds_spectral_bwd_dots(const int64_t* table, int nlayers, void* stream) {
  if (nlayers <= 0) return SE3DS_OK;
  hipLaunchKernelGGL(sn_dots_kernel, dim3(SN_RB, (unsigned)nlayers), dim3(kB), 0, as_stream(stream),
                     table);
  return check_launch("spectral_bwd_dots");
}

int se3ds_spectral_table_fields(void) { return SN_F; }
int se3ds_spectral_part_rows(void) { return SN_KB; }
int se3ds_spectral_vpart_len(void) { return 3 * SN_RB + 2; }

}  // extern "C"
