// Implicit-GEMM convolution family for gfx950 (CDNA4), NHWC, bf16 or fp32 operands with
// fp32 MFMA accumulation.  Three kernels cover every contraction on the SE3DS path:
//
//   igemm<FWD>    y[n,oy,ox,co]  = sum_{ky,kx,ci} x[n, oy*s-pt+ky, ox*s-pl+kx, ci] * W[ky,kx,ci,co]
//                 conv forward (layers.py:193-198,334-339) and ConvT input-gradient
//   igemm<DGRAD>  dx[n,iy,ix,ci] = sum_{ky,kx,co} dy[n,(iy+pt-ky)/s,(ix+pl-kx)/s,co] * W[ky,kx,ci,co]
//                 conv input-gradient and Conv2DTranspose forward (layers.py:417-423,475-480;
//                 the Keras ConvT kernel (kh,kw,Cout_T,Cin_T) IS the HWIO kernel of the
//                 associated forward conv).  Strided: output pixels are grouped by parity
//                 class so that every tile only visits the taps that hit it (no zero work).
//   wgrad         dW[ky,kx,ci,co] = sum_{n,oy,ox} x[..] * dy[n,oy,ox,co]   (fp32, split over L)
//
// Tiling (wave64): 128x128 output tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32
// tiles (64 accumulator VGPRs).  K step = one 64-byte row chunk (32 bf16 / 16 fp32).  Operand
// tiles are staged global -> registers -> LDS (double buffered; the next tile's global loads
// are in flight under the current tile's MFMAs), rows of 64 B with an XOR swizzle of the
// 16-byte chunk index so ds_read_b128 fragment reads are bank-conflict free.  Zero / circular
// padding (PadLayer, layers.py:62-97) and the partial-conv input mask are folded into the
// gather; the partial-conv renormalisation, spectral 1/sigma scale, bias and activation are
// the epilogue.  The weight operand is the MFMA "A" side so each lane ends up holding 4
// consecutive output channels of one pixel (8/16-byte stores).
#include <stdlib.h>

#include <cmath>
#include <type_traits>
#include "common.h"

namespace se3ds {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BM = 128;     // pixels per tile
constexpr int BN = 128;     // output channels per tile
constexpr int kThreads = 256;
constexpr int ROWB = 64;    // bytes per LDS row (one K step)
constexpr int TILE_BYTES = 128 * ROWB;  // one operand tile: 8 KiB

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
  static constexpr int BK = 16;   // elements per K step
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct TT<uint16_t> {  // bf16 raw bits
  static constexpr int EPC = 8;
  static constexpr int BK = 32;
  static __device__ __forceinline__ float to_f(uint16_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ uint16_t from_f(float v) { return f32_to_bf16(v); }
};

// Division of a 32-bit unsigned value by a runtime constant, multiplier and shifts from the host
// (Granlund-Montgomery): q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(n, m); exact for every
// n < 2^32.  Tile row -> (image, row, column) is four of these pairs per thread at the head of a
// workgroup and four more in front of its epilogue; as 64-bit / 32-bit hardware-less divisions they
// were ~1 000 VALU instructions per tile on the critical path (~2 us of a 25 us 1x1 tile).
struct FastDiv { uint32_t m, s1, s2; };
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;
  uint32_t l = 0;
  while (l < 32 && ((uint64_t)1 << l) < d) ++l;
  f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l - f.s1;
  return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  const uint32_t t = __umulhi(n, f.m);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct IgemmParams {
  // source tensor of the gather (x for FWD, dy for DGRAD) and its dims
  const void* src; int sH, sW, sC;
  // weights: element (tap, n_out, c_red) at w + tap*w_tap + n_out*w_n + c_red
  const void* w; int64_t w_tap, w_n;
  void* out;              // output tensor, rows of oC channels
  int N, oH, oW, oC;      // output spatial dims / channels
  int kh, kw, stride, pad_t, pad_l, wrap_w;
  // DGRAD parity classes (stride^2 of them); FWD uses class 0 only
  int n_classes; int cls_tile_start[5]; int cls_py[4], cls_px[4];
  FastDiv fd_hw[4], fd_w[4];   // division by a class's cH * cW and cW (fill_classes)
  // gather-side per-pixel multiplier (partial conv: x * mask), (N,sH,sW) fp32 or null
  const float* src_mask;
  int mask_binary;        // src_mask holds only {0,1}: masked rows may be fetched from the zero page
  // epilogue
  const float* scale;     // device scalar (1/(sigma+eps)) or null
  const float* bias;      // (oC) or null
  const float* row_a;     // (N*oH*oW) ratio / mask or null
  const float* row_b;     // (N*oH*oW) update_mask or null (only with bias)
  int act; float act_alpha;  // 0 none, 1 relu, 2 leaky relu
  int vec;                // reduction channels % BK == 0 -> vector gather
  int halo_ty, halo_tx;   // igemm_halo_kernel: output tiles per image (rows of 8, columns of 32)
  // igemm_halo_kernel, forward: per (pixel tile, pixel quarter) column sums of the stored
  // outputs, stats[row][2][oC] (sum, sum of squares), row = tile * 4 + quarter; or NULL
  float* stats;
  // dx = conv result + addend (same layout and dtype as out; may alias out): the second
  // contribution to a tensor with two consumers (ResNet block input) without a separate pass
  const void* addend;
  // DGRAD with fused batch-norm backward statistics (se3ds_conv2d_dgrad_bnstats): `out` is the
  // gradient of y = act(norm(bn_x)) + ..., and the epilogue also emits, per 64-pixel wave tile,
  // stats[row][2][oC] = (sum dz, sum dz * xhat) with dz = out * act'(y) (activation bit mask
  // bn_mask, one bit per element, or null without activation) and xhat = (bn_x - mean) * rstd,
  // taken from the STORED (rounded, addend included) gradient -- what norm_partial_kernel<MODE 1>
  // would read back.  bn_x == nullptr: off.
  const uint16_t* bn_x; const uint8_t* bn_mask; const float* bn_mean; const float* bn_rstd;
  int bn_act; float bn_alpha;
  // Output pixel of (image n, row a, column b) = (n * oH + a) * o_pitch + o_off + b (store_tile /
  // store_tile16; default o_pitch = oW, o_off = 0) and bias index = channel % bias_mod (0: plain).
  // A 2x2 stride-2 Conv2DTranspose is two 1x1 convolutions with 2 * Cout "channels" (px, co), one
  // per output row parity py: the (px, co) pairs of input pixel (i, j) are the 2 * Cout CONTIGUOUS
  // elements of output pixels (2i + py, 2j) and (2i + py, 2j + 1) -- virtual pixels of 2 * Cout
  // channels with o_pitch = 2 * W and o_off = py * W (se3ds_conv_transpose2x2_fwd).
  int64_t o_pitch, o_off;
  int bias_mod;
#ifdef SE3DS_PROBE
  // timing-only builds (tools/probes/conv_phases.sh; never the shipped library): 1 = return in front
  // of the epilogue, 2 = skip the K loop -- what the phases of a kernel cost on their own
  int probe;
#endif
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

// row m of a class's pixel list -> (image, row, column) of the class sub-grid (m < N * cH * cW < 2^31)
__device__ __forceinline__ void pix_of(const IgemmParams& p, int cls, int64_t m, int cH, int cW, int& n,
                                       int& a, int& b) {
  const uint32_t mu = (uint32_t)m;
  const uint32_t nn = fd_div(mu, p.fd_hw[cls]);
  const uint32_t rem = mu - nn * (uint32_t)(cH * cW);
  const uint32_t aa = fd_div(rem, p.fd_w[cls]);
  n = (int)nn; a = (int)aa; b = (int)(rem - aa * (uint32_t)cW);
}

template <typename T>
__device__ __forceinline__ void mfma_tile(f32x16_t (&acc)[2][2], const uint4 (&wf)[2][2],
                                          const uint4 (&xf)[2][2]);

template <>
__device__ __forceinline__ void mfma_tile<uint16_t>(f32x16_t (&acc)[2][2],
                                                    const uint4 (&wf)[2][2],
                                                    const uint4 (&xf)[2][2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8_t, wf[i][ks]), __builtin_bit_cast(bf16x8_t, xf[j][ks]),
            acc[i][j], 0, 0, 0);
}

template <>
__device__ __forceinline__ void mfma_tile<float>(f32x16_t (&acc)[2][2], const uint4 (&wf)[2][2],
                                                 const uint4 (&xf)[2][2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t wa = e == 0 ? wf[i][ks].x : e == 1 ? wf[i][ks].y
                            : e == 2 ? wf[i][ks].z : wf[i][ks].w;
          const uint32_t xb = e == 0 ? xf[j][ks].x : e == 1 ? xf[j][ks].y
                            : e == 2 ? xf[j][ks].z : xf[j][ks].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wa),
                                                           __uint_as_float(xb), acc[i][j], 0, 0, 0);
        }
}

// fp32 (parity) path: two-level accumulation.  A 32x32x2 MFMA chain adds the K products strictly in
// sequence, so its rounding error grows like sqrt(K) (measured 1.6e-6 relative at K = 9216, 6.5x the
// PyTorch-CPU oracle whose vector lanes hold 16 interleaved partial sums).  Flushing the chain into
// a second accumulator every kFlushSteps K steps brings the two to the same level; deep
// batch-normalised nets amplify that difference ~1e4 x (tools/noise_probe.py).  bf16
// instantiations compile to nothing.
constexpr int kFlushSteps = 4;
template <typename T>
__device__ __forceinline__ void flush_acc(f32x16_t (&tot)[2][2], f32x16_t (&acc)[2][2]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tot[i][j][r] += acc[i][j][r];
          acc[i][j][r] = 0.f;
        }
  }
}

// Decoded coordinates of one gather row (an output pixel of this kernel).
struct RowInfo {
  int n, a, b;   // FWD: (n, oy, ox)   DGRAD: (n, iy, ix)
  bool valid;
};

// Epilogue shared by the implicit-GEMM kernels.
// acc[i][j][r]: channel = co_base + i*32 + (r&3) + 8*(r>>2) + 4*half, pixel row = m_base + j*32 + l32
// store_pixel writes the NI*16 channels this lane holds for pixel column j to output pixel opix.
template <typename T, int NI>
__device__ __forceinline__ void store_pixel(const IgemmParams& p, f32x16_t (&acc)[NI][2], int j,
                                            int64_t opix, int co_base, int half, float scale) {
  using tt = TT<T>;
  T* __restrict__ out = (T*)p.out;
  const float ra = p.row_a ? p.row_a[opix] : 1.0f;
  const float rb = p.row_b ? p.row_b[opix] : 1.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co_base + i * 32 + g * 8 + half * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[i][j][g * 4 + e] * scale;
        const float bv = (p.bias && co + e < p.oC) ? p.bias[p.bias_mod ? (co + e) % p.bias_mod : co + e] : 0.0f;
        if (p.row_a) {
          if (p.bias) t = ((t - bv) * ra + bv) * rb;
          else t = t * ra;
        } else if (p.bias) {
          t = t + bv;
        }
        if (p.act == 1) t = t > 0.f ? t : 0.f;
        else if (p.act == 2) t = t > 0.f ? t : t * p.act_alpha;
        v[e] = t;
      }
      T* o = out + opix * p.oC + co;
      if (p.addend) {
        const T* ad = (const T*)p.addend + opix * p.oC + co;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < p.oC) v[e] = tt::to_f(tt::from_f(v[e])) + tt::to_f(ad[e]);
      }
      if (co + 3 < p.oC && (p.oC & 3) == 0) {
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 pk;
          pk.x = pack2_bf16(v[0], v[1]);
          pk.y = pack2_bf16(v[2], v[3]);
          *reinterpret_cast<uint2*>(o) = pk;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < p.oC) o[e] = tt::from_f(v[e]);
      }
    }
}

// bf16 epilogue through LDS.  In the MFMA layout a lane owns 4 consecutive channels of one
// pixel, so direct stores are 8-byte pieces scattered over 32 pixel rows per instruction: the
// 3x3 128->128 @512x1024 layer spent 36 % of its time in them (1.07 GB written at 1.5 TB/s).
// Here the wave parks 32 pixels x NI*32 channels in its own LDS scratch ([pixel][channel], rows
// padded by 16 B) and writes them back with 16-byte stores in which NI*4 consecutive lanes cover
// one pixel's contiguous NI*64 bytes.  LDS instructions of one wave execute in order, so the
// wave needs no barrier between its own ds_write and ds_read.
// opix[j]: output pixel index of (fragment j, lane & 31), or -1.  scratch: kEpiScratch<NI> bytes.
template <int NI, int PXC = 32> constexpr int kEpiScratch = PXC * (NI * 64 + 16) + 256;

// PXC: pixels parked per pass (32, or 16 to halve the scratch).  bias4(cl) returns the bias of
// channels co_base + cl .. + 3 (zeros without bias); `scale` is the spectral 1/(sigma+eps).
template <int NI, int PXC, bool BNB = false, typename BiasFn>
__device__ __forceinline__ void store_wave_lds_impl(const IgemmParams& p, f32x16_t (&acc)[NI][2],
                                                    const int64_t (&opix)[2], int co_base, int lane,
                                                    unsigned char* scratch, float scale,
                                                    BiasFn bias4, float* stats_row = nullptr) {
  constexpr int RB = NI * 64 + 16;   // padded row bytes
  constexpr int LPP = NI * 4;        // lanes per pixel in the write-back
  constexpr int PPI = 64 / LPP;      // pixels per store instruction
  constexpr int NPASS = 32 / PXC;
  const int half = lane >> 5, l32 = lane & 31;
  // activation as one select: none -> slope 1, relu -> slope 0, leaky relu -> alpha
  const float slope = p.act == 0 ? 1.0f : (p.act == 1 ? 0.0f : p.act_alpha);
  uint16_t* __restrict__ out = (uint16_t*)p.out;
  int64_t* offs = reinterpret_cast<int64_t*>(scratch + PXC * RB);
  float4 bv[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[i][g] = bias4(i * 32 + g * 8 + half * 4);
  // epilogue form (wave-uniform): 0 plain, 1 + bias, 2 * ratio, 3 partial conv with bias
  const int form = p.row_a ? (p.bias ? 3 : 2) : (p.bias ? 1 : 0);
  float cs1 = 0.f, cs2 = 0.f;   // column sums of channel co_base + lane (NI == 2 only)
  // fused batch-norm backward statistics (wave-uniform switch): in the write-back a lane owns
  // channels co_base + (lane % 8) * 8 .. + 7 of pixel lane / 8
  const bool bnb = BNB && NI == 2 && NPASS == 1 && stats_row != nullptr && p.bn_x != nullptr;
  // (sum dz * (x - mean) is accumulated and scaled by rstd once at the end: 24 live registers
  // instead of 32 in an epilogue that is already at the register limit)
  float bs1[8], bs2[8], bmu[8];
  if (BNB && NI == 2 && bnb) {
    const int c0 = co_base + (lane % LPP) * 8;
#pragma unroll
    for (int h4 = 0; h4 < 2; ++h4) {
      const float4 m4 = *reinterpret_cast<const float4*>(p.bn_mean + c0 + h4 * 4);
      bmu[h4 * 4] = m4.x; bmu[h4 * 4 + 1] = m4.y; bmu[h4 * 4 + 2] = m4.z; bmu[h4 * 4 + 3] = m4.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t o = opix[j];
    const int64_t oc = o < 0 ? 0 : o;
    const float ra = p.row_a ? p.row_a[oc] : 1.0f;
    const float rb = p.row_b ? p.row_b[oc] : 1.0f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      // this pass parks pixels ps*PXC .. +PXC-1 of fragment j (lanes owning other pixels idle)
      const bool mine = NPASS == 1 || (l32 / PXC) == ps;
      const int lp = l32 % PXC;
      if (mine && half == 0) offs[lp] = o;
      auto emit = [&](auto form_c) {
        constexpr int F = decltype(form_c)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float b4[4] = {bv[i][g].x, bv[i][g].y, bv[i][g].z, bv[i][g].w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = acc[i][j][g * 4 + e] * scale;
              if (F == 1) t = t + b4[e];
              else if (F == 2) t = t * ra;
              else if (F == 3) t = ((t - b4[e]) * ra + b4[e]) * rb;
              v[e] = t > 0.f ? t : t * slope;
            }
            uint2 pk;
            pk.x = pack2_bf16(v[0], v[1]);
            pk.y = pack2_bf16(v[2], v[3]);
            if (o < 0) pk = make_uint2(0u, 0u);   // (keeps the column sums clean)
            if (mine)
              *reinterpret_cast<uint2*>(scratch + lp * RB + (i * 32 + g * 8 + half * 4) * 2) = pk;
          }
      };
      if (form == 0) emit(std::integral_constant<int, 0>());
      else if (form == 1) emit(std::integral_constant<int, 1>());
      else if (form == 2) emit(std::integral_constant<int, 2>());
      else emit(std::integral_constant<int, 3>());
      __builtin_amdgcn_wave_barrier();
      if (NI == 2 && NPASS == 1 && stats_row != nullptr && !bnb) {
        // batch-norm statistics of the stored (rounded) outputs: lane = channel
#pragma unroll 8
        for (int px = 0; px < PXC; ++px) {
          const float v = bf16_to_f32(*reinterpret_cast<const uint16_t*>(scratch + px * RB + lane * 2));
          cs1 += v;
          cs2 += v * v;
        }
      }
#pragma unroll
      for (int k = 0; k < PXC / PPI; ++k) {
        const int px = k * PPI + lane / LPP, c16 = lane % LPP;
        uint4 v = *reinterpret_cast<const uint4*>(scratch + px * RB + c16 * 16);
        const int64_t po = offs[px];
        if (po >= 0) {
          if (p.addend) {
            // (requesting the four pieces of a fragment before the parking phase instead of one
            // exposed round trip per store was measured 0.8 % slower per step: 16 more live
            // registers in an epilogue at the limit)
            const uint4 a = *reinterpret_cast<const uint4*>((const uint16_t*)p.addend + po * p.oC +
                                                            co_base + c16 * 8);
            uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
            const uint32_t* aw = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float lo = __uint_as_float(vw[q] << 16) + __uint_as_float(aw[q] << 16);
              const float hi = __uint_as_float(vw[q] & 0xffff0000u) + __uint_as_float(aw[q] & 0xffff0000u);
              vw[q] = pack2_bf16(lo, hi);
            }
          }
          *reinterpret_cast<uint4*>(out + po * p.oC + co_base + c16 * 8) = v;
          if (BNB && NI == 2 && bnb) {
            // (fetching these before the parking phase instead -- 40 more live registers, spills
            // in the 256-channel kernel -- was measured slower still)
            const int64_t e0 = po * p.oC + co_base + c16 * 8;
            const uint4 xr = *reinterpret_cast<const uint4*>(p.bn_x + e0);
            const unsigned mb = p.bn_mask ? p.bn_mask[e0 >> 3] : 0xffu;
            const uint32_t* vw = reinterpret_cast<const uint32_t*>(&v);
            const uint32_t* xw = reinterpret_cast<const uint32_t*>(&xr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float d0 = __uint_as_float(vw[q] << 16) *
                               act_grad_from_bit((mb >> (2 * q)) & 1u, p.bn_act, p.bn_alpha);
              const float d1 = __uint_as_float(vw[q] & 0xffff0000u) *
                               act_grad_from_bit((mb >> (2 * q + 1)) & 1u, p.bn_act, p.bn_alpha);
              const float x0 = __uint_as_float(xw[q] << 16), x1 = __uint_as_float(xw[q] & 0xffff0000u);
              bs1[2 * q] += d0;
              bs2[2 * q] += d0 * (x0 - bmu[2 * q]);
              bs1[2 * q + 1] += d1;
              bs2[2 * q + 1] += d1 * (x1 - bmu[2 * q + 1]);
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (BNB && NI == 2 && bnb) {
    // the eight lanes with the same lane % 8 hold the same channels: butterfly over lane bits 3-5
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {
        bs1[e] += __shfl_xor(bs1[e], m, 64);
        bs2[e] += __shfl_xor(bs2[e], m, 64);
      }
    }
    if (lane < 8) {
      float* r1 = stats_row + co_base + lane * 8;
      float* r2 = stats_row + p.oC + co_base + lane * 8;
      const float4 ra4 = *reinterpret_cast<const float4*>(p.bn_rstd + co_base + lane * 8);
      const float4 rb4 = *reinterpret_cast<const float4*>(p.bn_rstd + co_base + lane * 8 + 4);
      bs2[0] *= ra4.x; bs2[1] *= ra4.y; bs2[2] *= ra4.z; bs2[3] *= ra4.w;
      bs2[4] *= rb4.x; bs2[5] *= rb4.y; bs2[6] *= rb4.z; bs2[7] *= rb4.w;
      *reinterpret_cast<float4*>(r1) = make_float4(bs1[0], bs1[1], bs1[2], bs1[3]);
      *reinterpret_cast<float4*>(r1 + 4) = make_float4(bs1[4], bs1[5], bs1[6], bs1[7]);
      *reinterpret_cast<float4*>(r2) = make_float4(bs2[0], bs2[1], bs2[2], bs2[3]);
      *reinterpret_cast<float4*>(r2 + 4) = make_float4(bs2[4], bs2[5], bs2[6], bs2[7]);
    }
  } else if (NI == 2 && stats_row != nullptr) {
    stats_row[co_base + lane] = cs1;
    stats_row[p.oC + co_base + lane] = cs2;
  }
}

// The same epilogue for accumulators of v_mfma_f32_16x16x32_bf16 (igemm_halo_kernel<..., M16>): 64
// channels x 64 pixels of one wave as acc[i][j], i = 16-channel block, j = 16-pixel block (j >> 1 =
// the 32-pixel fragment, j & 1 its half); lane (g = lane / 16, c = lane % 16) holds channels
// i*16 + g*4 .. +3 of pixel c of block j.  Only the parking phase differs from
// store_wave_lds_impl<2, 32>: scratch layout, statistics and write-back are shared.
// opix[j]: output pixel of (block j, lane % 16) or -1.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// HOIST (the 128 x 128 and 256-pixel macro tiles; not the halo kernels, which are at their register
// limit): the per-pixel renormalisation factors of a partial conv (row_a / row_b, two global loads per
// 16-pixel block) are requested for all four blocks up front -- in the loop each pair was an exposed
// memory round trip in front of the block's arithmetic (the epilogue is ~50 % of a 1x1 layer).
// BATCH (round 5; not the 256-channel halo kernel, which has no registers to spare): the write-back
// of a 32-pixel pass issues ALL of its LDS reads (four 16-byte pieces + their pixel offsets), then
// the addend loads, then the four stores.  The loop form compiled to a strictly serial chain per
// piece -- ds_read offs, wait, branch, ds_read_b96 + ds_read_b32 (the compiler did not know the
// 16-byte alignment), wait, store: eight exposed LDS round trips per pass, 32 per 128-channel wave
// tile -- and the epilogue IS the kernel for the layers with few K steps (1x1, 2x2 transposed, the
// 128-channel 3x3 layers: tools/conv_phases.py).
template <bool BNB = false, bool HOIST = false, bool BATCH = false>
__device__ __forceinline__ void store_wave_lds16(const IgemmParams& p, f32x4_t (&acc)[4][4],
                                                 const int64_t (&opix)[4], int co_base, int lane,
                                                 unsigned char* scratch, float* stats_row = nullptr) {
  constexpr int NI = 2, PXC = 32;
  // Rows of exactly 128 bytes, 16-byte chunk index ^= pixel & 7 (no padding): the padded 144-byte
  // rows of store_wave_lds_impl run both the parking writes and the write-back reads at twice their
  // conflict-free cycle count (tools/probes/lds_epi.hip: 33 conflict cycles per 67 active for either
  // layout; 128-byte rows: write-back 0 / 34); the statistics reads are a permutation of one row.
  constexpr int RB = NI * 64;
  constexpr int LPP = NI * 4;        // lanes per pixel in the write-back
  constexpr int PPI = 64 / LPP;      // pixels per store instruction
  const float scale = p.scale ? *p.scale : 1.0f;
  const int g = lane >> 4, c16 = lane & 15;
  const float slope = p.act == 0 ? 1.0f : (p.act == 1 ? 0.0f : p.act_alpha);
  uint16_t* __restrict__ out = (uint16_t*)p.out;
  int64_t* offs = reinterpret_cast<int64_t*>(scratch + PXC * RB);
  float4 bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    bv[i] = p.bias ? *reinterpret_cast<const float4*>(
                         p.bias + (p.bias_mod ? (co_base + i * 16 + g * 4) % p.bias_mod : co_base + i * 16 + g * 4))
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  const int form = p.row_a ? (p.bias ? 3 : 2) : (p.bias ? 1 : 0);
  float ra4[4], rb4[4];
  if (HOIST) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t oc = opix[j] < 0 ? 0 : opix[j];
      ra4[j] = p.row_a ? p.row_a[oc] : 1.0f;
      rb4[j] = p.row_b ? p.row_b[oc] : 1.0f;
    }
  }
  float cs1 = 0.f, cs2 = 0.f;
  const bool bnb = BNB && stats_row != nullptr && p.bn_x != nullptr;
  float bs1[8], bs2[8], bmu[8];
  if (BNB && bnb) {
    const int c0 = co_base + (lane % LPP) * 8;
#pragma unroll
    for (int h4 = 0; h4 < 2; ++h4) {
      const float4 m4 = *reinterpret_cast<const float4*>(p.bn_mean + c0 + h4 * 4);
      bmu[h4 * 4] = m4.x; bmu[h4 * 4 + 1] = m4.y; bmu[h4 * 4 + 2] = m4.z; bmu[h4 * 4 + 3] = m4.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
  }
#pragma unroll
  for (int J = 0; J < 2; ++J) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = J * 2 + jj;
      const int64_t o = opix[j];
      const int64_t oc = o < 0 ? 0 : o;
      const float ra = HOIST ? ra4[j] : (p.row_a ? p.row_a[oc] : 1.0f);
      const float rb = HOIST ? rb4[j] : (p.row_b ? p.row_b[oc] : 1.0f);
      const int lp = jj * 16 + c16;
      if (g == 0) offs[lp] = o;
      auto emit = [&](auto form_c) {
        constexpr int F = decltype(form_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float b4[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = acc[i][j][e] * scale;
            if (F == 1) t = t + b4[e];
            else if (F == 2) t = t * ra;
            else if (F == 3) t = ((t - b4[e]) * ra + b4[e]) * rb;
            v[e] = t > 0.f ? t : t * slope;
          }
          uint2 pk;
          pk.x = pack2_bf16(v[0], v[1]);
          pk.y = pack2_bf16(v[2], v[3]);
          if (o < 0) pk = make_uint2(0u, 0u);   // (keeps the column sums clean)
          // channels i*16 + g*4 .. +3: chunk 2 i + g / 2, half (g & 1)
          *reinterpret_cast<uint2*>(scratch + lp * RB + (((2 * i + (g >> 1)) ^ (lp & 7)) << 4) + (g & 1) * 8) = pk;
        }
      };
      if (form == 0) emit(std::integral_constant<int, 0>());
      else if (form == 1) emit(std::integral_constant<int, 1>());
      else if (form == 2) emit(std::integral_constant<int, 2>());
      else emit(std::integral_constant<int, 3>());
    }
    __builtin_amdgcn_wave_barrier();
    if (stats_row != nullptr && !bnb) {
      // batch-norm statistics of the stored (rounded) outputs: lane = channel
#pragma unroll 8
      for (int px = 0; px < PXC; ++px) {
        const float v = bf16_to_f32(*reinterpret_cast<const uint16_t*>(
            scratch + px * RB + (((lane >> 3) ^ (px & 7)) << 4) + (lane & 7) * 2));
        cs1 += v;
        cs2 += v * v;
      }
    }
#ifdef SE3DS_PROBE
    const bool batch_rt = p.probe != 4;   // (timing probe: 4 = the serial write-back loop)
#else
    constexpr bool batch_rt = true;
#endif
    if (BATCH && !BNB && batch_rt) {
      constexpr int NK = PXC / PPI;   // 4
      const int c8 = lane % LPP, pxl = lane / LPP;
      const unsigned char* sbase =
          static_cast<const unsigned char*>(__builtin_assume_aligned(scratch, 16));
      int64_t po[NK];
      uint4 vv[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) po[k] = offs[k * PPI + pxl];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int px = k * PPI + pxl;
        vv[k] = *reinterpret_cast<const uint4*>(sbase + px * RB + ((c8 ^ (px & 7)) << 4));
      }
      if (p.addend) {   // (wave-uniform)
        uint4 av[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const int64_t pk = po[k] < 0 ? 0 : po[k];
          av[k] = *reinterpret_cast<const uint4*>((const uint16_t*)p.addend + pk * p.oC + co_base + c8 * 8);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          uint32_t* vw = reinterpret_cast<uint32_t*>(&vv[k]);
          const uint32_t* aw = reinterpret_cast<const uint32_t*>(&av[k]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(vw[q] << 16) + __uint_as_float(aw[q] << 16);
            const float hi = __uint_as_float(vw[q] & 0xffff0000u) + __uint_as_float(aw[q] & 0xffff0000u);
            vw[q] = pack2_bf16(lo, hi);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (po[k] >= 0) *reinterpret_cast<uint4*>(out + po[k] * p.oC + co_base + c8 * 8) = vv[k];
    } else {
#pragma unroll
    for (int k = 0; k < PXC / PPI; ++k) {
      const int px = k * PPI + lane / LPP, c8 = lane % LPP;
      uint4 v = *reinterpret_cast<const uint4*>(scratch + px * RB + ((c8 ^ (px & 7)) << 4));
      const int64_t po = offs[px];
      if (po >= 0) {
        if (p.addend) {
          const uint4 a = *reinterpret_cast<const uint4*>((const uint16_t*)p.addend + po * p.oC +
                                                          co_base + c8 * 8);
          uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
          const uint32_t* aw = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(vw[q] << 16) + __uint_as_float(aw[q] << 16);
            const float hi = __uint_as_float(vw[q] & 0xffff0000u) + __uint_as_float(aw[q] & 0xffff0000u);
            vw[q] = pack2_bf16(lo, hi);
          }
        }
#ifdef SE3DS_PROBE
        if (p.probe == 3) {   // (timing probe: non-temporal output stores)
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
          __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v),
                                      reinterpret_cast<u32x4_t*>(out + po * p.oC + co_base + c8 * 8));
        } else
#endif
        *reinterpret_cast<uint4*>(out + po * p.oC + co_base + c8 * 8) = v;
        if (BNB && bnb) {
          const int64_t e0 = po * p.oC + co_base + c8 * 8;
          const uint4 xr = *reinterpret_cast<const uint4*>(p.bn_x + e0);
          const unsigned mb = p.bn_mask ? p.bn_mask[e0 >> 3] : 0xffu;
          const uint32_t* vw = reinterpret_cast<const uint32_t*>(&v);
          const uint32_t* xw = reinterpret_cast<const uint32_t*>(&xr);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d0 = __uint_as_float(vw[q] << 16) *
                             act_grad_from_bit((mb >> (2 * q)) & 1u, p.bn_act, p.bn_alpha);
            const float d1 = __uint_as_float(vw[q] & 0xffff0000u) *
                             act_grad_from_bit((mb >> (2 * q + 1)) & 1u, p.bn_act, p.bn_alpha);
            const float x0 = __uint_as_float(xw[q] << 16), x1 = __uint_as_float(xw[q] & 0xffff0000u);
            bs1[2 * q] += d0;
            bs2[2 * q] += d0 * (x0 - bmu[2 * q]);
            bs1[2 * q + 1] += d1;
            bs2[2 * q + 1] += d1 * (x1 - bmu[2 * q + 1]);
          }
        }
      }
    }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (BNB && bnb) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {
        bs1[e] += __shfl_xor(bs1[e], m, 64);
        bs2[e] += __shfl_xor(bs2[e], m, 64);
      }
    }
    if (lane < 8) {
      float* r1 = stats_row + co_base + lane * 8;
      float* r2 = stats_row + p.oC + co_base + lane * 8;
      const float4 ra4 = *reinterpret_cast<const float4*>(p.bn_rstd + co_base + lane * 8);
      const float4 rb4 = *reinterpret_cast<const float4*>(p.bn_rstd + co_base + lane * 8 + 4);
      bs2[0] *= ra4.x; bs2[1] *= ra4.y; bs2[2] *= ra4.z; bs2[3] *= ra4.w;
      bs2[4] *= rb4.x; bs2[5] *= rb4.y; bs2[6] *= rb4.z; bs2[7] *= rb4.w;
      *reinterpret_cast<float4*>(r1) = make_float4(bs1[0], bs1[1], bs1[2], bs1[3]);
      *reinterpret_cast<float4*>(r1 + 4) = make_float4(bs1[4], bs1[5], bs1[6], bs1[7]);
      *reinterpret_cast<float4*>(r2) = make_float4(bs2[0], bs2[1], bs2[2], bs2[3]);
      *reinterpret_cast<float4*>(r2 + 4) = make_float4(bs2[4], bs2[5], bs2[6], bs2[7]);
    }
  } else if (stats_row != nullptr) {
    stats_row[co_base + lane] = cs1;
    stats_row[p.oC + co_base + lane] = cs2;
  }
}

template <int NI, bool BNB = false>
__device__ __forceinline__ void store_wave_lds(const IgemmParams& p, f32x16_t (&acc)[NI][2],
                                               const int64_t (&opix)[2], int co_base, int lane,
                                               unsigned char* scratch, float* stats_row = nullptr) {
  const float scale = p.scale ? *p.scale : 1.0f;
  // vector loads up front (element-wise loads each paid a full vmcnt(0) round trip behind the
  // prefetch DMA)
  auto bias4 = [&](int cl) {
    return p.bias ? *reinterpret_cast<const float4*>(
                        p.bias + (p.bias_mod ? (co_base + cl) % p.bias_mod : co_base + cl))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  store_wave_lds_impl<NI, 32, BNB>(p, acc, opix, co_base, lane, scratch, scale, bias4, stats_row);
}

// m_base / co_base: first pixel row / output channel of this wave's sub-tile.
// scratch: this wave's kEpiScratch<NI> bytes of LDS (no longer read by anyone) or null.
template <typename T, int MODE, int NI = 2, bool BNB = false>
__device__ __forceinline__ void store_tile(const IgemmParams& p, f32x16_t (&acc)[NI][2],
                                           int64_t m_base, int co_base, int64_t Mc, int cH, int cW,
                                           int py, int px, int half, int l32,
                                           unsigned char* scratch = nullptr,
                                           float* stats_row = nullptr, int cls = 0) {
  const int s = p.stride;
  const float scale = p.scale ? *p.scale : 1.0f;
  int64_t opix[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int64_t m = m_base + j * 32 + l32;
    opix[j] = -1;
    if (m >= Mc) continue;
    int n, a, b;
    pix_of(p, cls, m, cH, cW, n, a, b);
    if (MODE == MODE_DGRAD) { a = py + a * s; b = px + b * s; }
    opix[j] = ((int64_t)n * p.oH + a) * p.o_pitch + p.o_off + b;
  }
  if (sizeof(T) == 2 && scratch != nullptr && (p.oC & 7) == 0 && co_base + NI * 32 <= p.oC) {
    if (NI == 4 && stats_row != nullptr) {
      // column sums are kept per 64-channel half (lane = channel)
      store_wave_lds<2, BNB>(p, *reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), opix, co_base,
                             half * 32 + l32, scratch, stats_row);
      store_wave_lds<2, BNB>(p, *reinterpret_cast<f32x16_t(*)[2][2]>(&acc[NI - 2]), opix,
                             co_base + 64, half * 32 + l32, scratch, stats_row);
    } else {
      store_wave_lds<NI, BNB>(p, acc, opix, co_base, half * 32 + l32, scratch, stats_row);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (opix[j] >= 0) store_pixel<T, NI>(p, acc, j, opix[j], co_base, half, scale);
}

// store_tile for the 16x16x32 accumulator layout (bf16, LDS epilogue only): CB 16-channel blocks
// of one wave as acc[CB][4]; pixel block j holds tile rows m_base + j*16 + lane % 16.
template <int MODE, int CB, bool BNB = false>
__device__ __forceinline__ void store_tile16(const IgemmParams& p, f32x4_t (&acc)[CB][4], int64_t m_base,
                                             int co_base, int64_t Mc, int cH, int cW, int py, int px,
                                             int lane, unsigned char* scratch, float* stats_row,
                                             int cls = 0) {
  const int s = p.stride;
  int64_t opix[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = m_base + j * 16 + (lane & 15);
    opix[j] = -1;
    if (m >= Mc) continue;
    int n, a, b;
    pix_of(p, cls, m, cH, cW, n, a, b);
    if (MODE == MODE_DGRAD) { a = py + a * s; b = px + b * s; }
    opix[j] = ((int64_t)n * p.oH + a) * p.o_pitch + p.o_off + b;
  }
  store_wave_lds16<BNB, true, true>(p, *reinterpret_cast<f32x4_t(*)[4][4]>(&acc[0]), opix, co_base, lane,
                                    scratch, stats_row);
  if (CB == 8)
    store_wave_lds16<BNB, true, true>(p, *reinterpret_cast<f32x4_t(*)[4][4]>(&acc[CB - 4]), opix, co_base + 64,
                                      lane, scratch, stats_row);
}

template <typename T, int MODE>
__global__ void __launch_bounds__(kThreads, 2)
igemm_kernel(const IgemmParams p) {
  using tt = TT<T>;
  constexpr int EPC = tt::EPC, BK = tt::BK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  // stage s: W tile at smem + s*2*TILE, X tile at + TILE
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // wave position: wm over channels, wn over pixels

  // ---- which tile
  int cls = 0;
  if (MODE == MODE_DGRAD) {
#pragma unroll
    for (int c = 1; c < 4; ++c)
      if (c < p.n_classes && (int)blockIdx.x >= p.cls_tile_start[c]) cls = c;
  }
  const int tile_m = blockIdx.x - p.cls_tile_start[cls];
  const int n0 = blockIdx.y * BN;           // first output channel of the tile
  const int s = p.stride;
  int py = 0, px = 0, cH = p.oH, cW = p.oW;  // class sub-grid
  int ky0 = 0, kx0 = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (MODE == MODE_DGRAD) {
    py = p.cls_py[cls]; px = p.cls_px[cls];
    cH = (p.oH - py + s - 1) / s;
    cW = (p.oW - px + s - 1) / s;
    ky0 = (py + p.pad_t) % s; kx0 = (px + p.pad_l) % s; kstep = s;
    nky = ky0 < p.kh ? (p.kh - ky0 + s - 1) / s : 0;
    nkx = kx0 < p.kw ? (p.kw - kx0 + s - 1) / s : 0;
  }
  const int64_t Mc = (int64_t)p.N * cH * cW;
  const int ntaps = nky * nkx;
  const int Cr = p.sC;  // reduction channels per tap

  // ---- this thread's two staging rows (row, row+64) and chunk
  const int chunk = tid & 3;
  const int srow = tid >> 2;
  RowInfo ri[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int64_t m = (int64_t)tile_m * BM + srow + h * 64;
    ri[h].valid = m < Mc;
    int64_t mm = ri[h].valid ? m : 0;
    int n, a, b;
    pix_of(p, cls, mm, cH, cW, n, a, b);
    ri[h].n = n;
    ri[h].a = MODE == MODE_DGRAD ? py + a * s : a;
    ri[h].b = MODE == MODE_DGRAD ? px + b * s : b;
  }

  const T* __restrict__ src = (const T*)p.src;
  const T* __restrict__ wp = (const T*)p.w;

  // source pixel of (row, tap); returns false when the tap falls in the zero padding
  auto src_pixel = [&](const RowInfo& r, int ky, int kx, int64_t& pix) -> bool {
    int sy, sx;
    if (MODE == MODE_FWD) {
      sy = r.a * s - p.pad_t + ky;
      sx = r.b * s - p.pad_l + kx;
      if (p.wrap_w) { sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx); }
    } else {
      int ty = r.a + p.pad_t - ky, tx = r.b + p.pad_l - kx;
      if (p.wrap_w) { tx = tx < 0 ? tx + p.sW : (tx >= p.sW ? tx - p.sW : tx); }
      if (ty < 0 || tx < 0) return false;
      sy = ty / s; sx = tx / s;  // exact by construction of the class tap list
    }
    if (sy < 0 || sy >= p.sH || sx < 0 || sx >= p.sW) return false;
    pix = ((int64_t)r.n * p.sH + sy) * p.sW + sx;
    return true;
  };

  const int ksteps_per_tap = p.vec ? Cr / BK : 0;
  const int64_t Ktot = (int64_t)ntaps * Cr;
  const int nk = p.vec ? ntaps * ksteps_per_tap : (int)((Ktot + BK - 1) / BK);

  uint4 rx[2], rw[2];  // staged chunks: X rows (srow, srow+64), W rows (srow, srow+64)

  // Vector path state: everything that depends on the tap (source pixel of each staging row,
  // its mask value, the weight row) is decoded ONCE per tap; the inner channel loop only
  // advances an element offset, so a K step costs a handful of VALU instructions.
  const T* xptr[2] = {nullptr, nullptr};
  const T* wptr[2] = {nullptr, nullptr};
  float xmk[2] = {1.0f, 1.0f};
  int tap_i = 0, cstep = 0;
  auto setup_tap = [&](int t) {
    const int ky = ky0 + kstep * (t / nkx), kx = kx0 + kstep * (t % nkx);
    const int tap_lin = ky * p.kw + kx;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t pix;
      xptr[h] = nullptr;
      if (ri[h].valid && src_pixel(ri[h], ky, kx, pix)) {
        xptr[h] = src + pix * Cr + chunk * EPC;
        if (p.src_mask) xmk[h] = p.src_mask[pix];
      }
      const int co = n0 + srow + h * 64;
      wptr[h] = co < p.oC ? wp + (int64_t)tap_lin * p.w_tap + (int64_t)co * p.w_n + chunk * EPC
                          : nullptr;
    }
  };
  if (p.vec && ntaps > 0) setup_tap(0);

  auto load_step = [&](int kstep_i) {
    if (p.vec) {
      const int c_off = cstep * BK;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (xptr[h]) {
          v = *reinterpret_cast<const uint4*>(xptr[h] + c_off);
          if (p.src_mask && xmk[h] != 1.0f) {
            const float mk = xmk[h];
            if (mk == 0.0f) {
              v = make_uint4(0, 0, 0, 0);
            } else if (sizeof(T) == 4) {
              v.x = __float_as_uint(__uint_as_float(v.x) * mk);
              v.y = __float_as_uint(__uint_as_float(v.y) * mk);
              v.z = __float_as_uint(__uint_as_float(v.z) * mk);
              v.w = __float_as_uint(__uint_as_float(v.w) * mk);
            } else {
              uint32_t* q = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float lo = __uint_as_float(q[e] << 16) * mk;
                float hi = __uint_as_float(q[e] & 0xffff0000u) * mk;
                q[e] = pack2_bf16(lo, hi);
              }
            }
          }
        }
        rx[h] = v;
        rw[h] = wptr[h] ? *reinterpret_cast<const uint4*>(wptr[h] + c_off) : make_uint4(0, 0, 0, 0);
      }
      if (++cstep == ksteps_per_tap) {
        cstep = 0;
        if (++tap_i < ntaps) setup_tap(tap_i);
      }
    } else {
      // generic gather: K is the linear (tap, channel) index, decoded per element
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        T xe[EPC], we[EPC];
        int co = n0 + srow + h * 64;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          int64_t kk = (int64_t)kstep_i * BK + chunk * EPC + e;
          T xv = tt::from_f(0.f), wv = tt::from_f(0.f);
          if (kk < Ktot) {
            int tap = (int)(kk / Cr);
            int c = (int)(kk - (int64_t)tap * Cr);
            int ky = ky0 + kstep * (tap / nkx), kx = kx0 + kstep * (tap % nkx);
            int64_t pix;
            if (ri[h].valid && src_pixel(ri[h], ky, kx, pix)) {
              xv = src[pix * Cr + c];
              if (p.src_mask) xv = tt::from_f(tt::to_f(xv) * p.src_mask[pix]);
            }
            if (co < p.oC)
              wv = wp[(int64_t)(ky * p.kw + kx) * p.w_tap + (int64_t)co * p.w_n + c];
          }
          xe[e] = xv; we[e] = wv;
        }
        rx[h] = *reinterpret_cast<uint4*>(xe);
        rw[h] = *reinterpret_cast<uint4*>(we);
      }
    }
  };

  auto store_step = [&](int stage) {
    unsigned char* wt = smem + stage * 2 * TILE_BYTES;
    unsigned char* xt = wt + TILE_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int row = srow + h * 64;
      int off = row * ROWB + swz(row, chunk) * 16;
      *reinterpret_cast<uint4*>(wt + off) = rw[h];
      *reinterpret_cast<uint4*>(xt + off) = rx[h];
    }
  };

  f32x16_t acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

  if (nk > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  const int half = lane >> 5, l32 = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) load_step(kt + 1);  // global loads in flight under the MFMAs below
    const unsigned char* wt = smem + stage * 2 * TILE_BYTES;
    const unsigned char* xt = wt + TILE_BYTES;
    uint4 wf[2][2], xf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        int wrow = wm * 64 + i * 32 + l32;
        int xrow = wn * 64 + i * 32 + l32;
        int c = ks * 2 + half;
        wf[i][ks] = *reinterpret_cast<const uint4*>(wt + wrow * ROWB + swz(wrow, c) * 16);
        xf[i][ks] = *reinterpret_cast<const uint4*>(xt + xrow * ROWB + swz(xrow, c) * 16);
      }
    mfma_tile<T>(acc, wf, xf);
    if ((kt % kFlushSteps) == kFlushSteps - 1) flush_acc<T>(tot, acc);
    if (kt + 1 < nk) store_step(stage ^ 1);
    __syncthreads();
  }
  if constexpr (sizeof(T) == 4) {
    flush_acc<T>(tot, acc);
    flush_acc<T>(acc, tot);   // result back in acc
  }

  store_tile<T, MODE>(p, acc, (int64_t)tile_m * BM + wn * 64, n0 + wm * 64, Mc, cH, cW, py, px,
                      half, l32, nullptr, nullptr, cls);
}

// ------------------------------------------------------------------ LDS-DMA implicit GEMM
// Same tiling, but operand tiles go HBM -> LDS directly (global_load_lds, 16 B per lane), rows
// are full 128-byte lines (K step = 64 bf16 / 32 fp32) and no staging registers or ds_write
// pass exist.  The LDS image is lane-linear per wave instruction (8 rows x 8 chunks = 1 KiB),
// so the bank-conflict swizzle is applied to the SOURCE chunk each lane fetches and to the
// fragment read address (chunk ^= (row >> 1) & 7).  Rows that fall in the zero padding read a
// 128-byte zero page.  Used whenever the reduction channels are a multiple of the K step and
// no gather-side mask is needed (decoder, heads, discriminator: ~88 % of the step's FLOPs).
__device__ __attribute__((aligned(128))) unsigned char g_zero_page[128];

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// LDS-DMA issued from inline assembly (16 bytes per lane to lds + lane * 16; `lds` wave-uniform).
// The compiler's waitcnt pass treats every __builtin_amdgcn_global_load_lds as a store that later
// LDS reads may alias and -- depending on register allocation, in one K step out of two to four --
// puts `s_waitcnt vmcnt(0)` in front of fragment reads of a DIFFERENT stage: the wave then waits
// for the tiles it has just requested.  Behind asm it sees neither the store nor the counter; the
// kernels that use this helper order DMA and reads themselves (counted vmcnt + barrier).  The
// counters complete in order, so the unknown transfers can only make a compiler-computed
// vmcnt(N) stricter, never laxer.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(gsrc), "s"(l) : "memory");   // (m0: reserved, cannot be listed; nothing else in
                                                    // these kernels uses it on gfx950)
}

// M16 (bf16 only): v_mfma_f32_16x16x32_bf16 (see igemm_halo_kernel).
template <typename T, int MODE, bool BNB = false, bool M16 = false>
__global__ void __launch_bounds__(kThreads, 2)
igemm_glds_kernel(const IgemmParams p) {
  static_assert(!M16 || sizeof(T) == 2, "16x16x32: bf16");
  using tt = TT<T>;
  constexpr int EPC = tt::EPC, BK2 = 2 * tt::BK;
  constexpr int ROW2 = 128, TILE2 = 128 * ROW2;
  // one LDS object per stage: the compiler's waitcnt pass tells LDS-DMA targets apart by the
  // variable's alias scope; with a single array every fragment read waited (vmcnt(0)) for the
  // DMA of the NEXT tile issued just before it, serialising fill and compute inside a wave
  __shared__ __attribute__((aligned(16))) unsigned char stage0[2 * TILE2];
  __shared__ __attribute__((aligned(16))) unsigned char stage1[2 * TILE2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int cls = 0;
  if (MODE == MODE_DGRAD) {
#pragma unroll
    for (int c = 1; c < 4; ++c)
      if (c < p.n_classes && (int)blockIdx.x >= p.cls_tile_start[c]) cls = c;
  }
  const int tile_m = blockIdx.x - p.cls_tile_start[cls];
  const int n0 = blockIdx.y * BN;
  const int s = p.stride;
  int py = 0, px = 0, cH = p.oH, cW = p.oW;
  int ky0 = 0, kx0 = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (MODE == MODE_DGRAD) {
    py = p.cls_py[cls]; px = p.cls_px[cls];
    cH = (p.oH - py + s - 1) / s;
    cW = (p.oW - px + s - 1) / s;
    ky0 = (py + p.pad_t) % s; kx0 = (px + p.pad_l) % s; kstep = s;
    nky = ky0 < p.kh ? (p.kh - ky0 + s - 1) / s : 0;
    nkx = kx0 < p.kw ? (p.kw - kx0 + s - 1) / s : 0;
  }
  const int64_t Mc = (int64_t)p.N * cH * cW;
  const int ntaps = nky * nkx;
  const int Cr = p.sC;

  // staging slots: instruction j of this wave fills rows (j*4 + wave)*8 .. +7
  RowInfo ri[4];
  int lch[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    lch[j] = (lane & 7) ^ ((row >> 1) & 7);
    int64_t m = (int64_t)tile_m * BM + row;
    ri[j].valid = m < Mc;
    int64_t mm = ri[j].valid ? m : 0;
    int n, a, b;
    pix_of(p, cls, mm, cH, cW, n, a, b);
    ri[j].n = n;
    ri[j].a = MODE == MODE_DGRAD ? py + a * s : a;
    ri[j].b = MODE == MODE_DGRAD ? px + b * s : b;
  }
  const T* __restrict__ src = (const T*)p.src;
  const T* __restrict__ wp = (const T*)p.w;
  const T* zero = reinterpret_cast<const T*>(g_zero_page);


  const int ksteps_per_tap = Cr / BK2;
#ifdef SE3DS_PROBE
  const int nk = p.probe == 2 ? 0 : ntaps * ksteps_per_tap;
#else
  const int nk = ntaps * ksteps_per_tap;
#endif
  const T* xsrc[4];
  const T* wsrc[4];
  int xm[4], wmk[4];   // -1: advance with the channel offset, 0: parked on the zero page
  int tap_i = 0, cstep = 0;
  // Per-slot constants so that a tap change costs a few adds / compares (32-bit pixel math):
  //   FWD  : source row/col = a0 + ky, b0 + kx        with a0 = a*s - pad_t, b0 = b*s - pad_l
  //   DGRAD: numerators       a0 - ky, b0 - kx        with a0 = a + pad_t,   b0 = b + pad_l
  int a0[4], b0[4], img0[4];
  const T* wrow[4];
  const int sshift = s == 2 ? 1 : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a0[j] = MODE == MODE_FWD ? ri[j].a * s - p.pad_t : ri[j].a + p.pad_t;
    b0[j] = MODE == MODE_FWD ? ri[j].b * s - p.pad_l : ri[j].b + p.pad_l;
    img0[j] = ri[j].n * p.sH * p.sW;
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int co = n0 + row;
    if (co < p.oC) {
      wrow[j] = wp + (int64_t)co * p.w_n + lch[j] * EPC;
      wmk[j] = -1;
    } else {
      wrow[j] = nullptr;
      wmk[j] = 0;
    }
  }
  int tky = 0, tkx = 0;   // running tap counters (index inside the class tap list)
  auto setup_tap = [&](int) {
    const int ky = ky0 + kstep * tky, kx = kx0 + kstep * tkx;
    const int64_t wtap = (int64_t)(ky * p.kw + kx) * p.w_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int sy, sx;
      bool ok = ri[j].valid;
      if (MODE == MODE_FWD) {
        sy = a0[j] + ky;
        sx = b0[j] + kx;
        if (p.wrap_w) sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx);
      } else {
        int ty = a0[j] - ky, tx = b0[j] - kx;
        if (p.wrap_w) tx = tx < 0 ? tx + p.sW : (tx >= p.sW ? tx - p.sW : tx);
        ok = ok && ty >= 0 && tx >= 0;
        sy = ty >> sshift;
        sx = tx >> sshift;
      }
      ok = ok && (unsigned)sy < (unsigned)p.sH && (unsigned)sx < (unsigned)p.sW;
      const int pix = img0[j] + sy * p.sW + sx;
      if (ok && p.src_mask) ok = p.src_mask[pix] != 0.0f;
      if (ok) {
        xsrc[j] = src + (int64_t)pix * Cr + lch[j] * EPC;
        xm[j] = -1;
      } else {
        xsrc[j] = zero + lch[j] * EPC;
        xm[j] = 0;
      }
      wsrc[j] = wmk[j] ? wrow[j] + wtap : zero + lch[j] * EPC;
    }
    if (++tkx == nkx) { tkx = 0; ++tky; }
  };
  auto issue = [&](unsigned char* wt) {
    unsigned char* xt = wt + TILE2;
    const int c_off = cstep * BK2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int slab = (j * 4 + wave) * 8 * ROW2;   // wave-uniform LDS byte offset
      __builtin_amdgcn_global_load_lds((gas_ptr)(wsrc[j] + (c_off & wmk[j])), (las_ptr)(wt + slab),
                                       16, 0, 0);
      __builtin_amdgcn_global_load_lds((gas_ptr)(xsrc[j] + (c_off & xm[j])), (las_ptr)(xt + slab),
                                       16, 0, 0);
    }
    if (++cstep == ksteps_per_tap) {
      cstep = 0;
      if (++tap_i < ntaps) setup_tap(tap_i);
    }
  };

  f32x16_t acc[2][2], tot[2][2];
  f32x4_t acc16[M16 ? 4 : 1][4];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  } else {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  }

  if (nk > 0) {
    setup_tap(0);
    issue(stage0);
  }
  __syncthreads();   // drains the LDS-DMA of stage 0 (vmcnt(0)) before anyone reads it
  const int half = lane >> 5, l32 = lane & 31;
  auto k_step = [&](unsigned char* cur, unsigned char* nxt, bool has_next) {
    if (has_next) issue(nxt);   // DMA for the next tile flies under the MFMAs below
    const unsigned char* wt = cur;
    const unsigned char* xt = cur + TILE2;
    if constexpr (M16) {
      // two k32 steps per K step; lane (g, r): chunk 4 * step + g of row r of a 16-row block
      const int g16 = lane >> 4, r16 = lane & 15;
      const int sw16 = (g16 ^ ((r16 >> 1) & 7)) << 4;
      const int wo16 = (wm * 64 + r16) * ROW2 + sw16, xo16 = TILE2 + (wn * 64 + r16) * ROW2 + sw16;
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
        uint4 wq[4], xq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          wq[i] = *reinterpret_cast<const uint4*>(cur + i * 16 * ROW2 + (wo16 ^ (kp << 6)));
          xq[i] = *reinterpret_cast<const uint4*>(cur + i * 16 * ROW2 + (xo16 ^ (kp << 6)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, wq[i]), __builtin_bit_cast(bf16x8_t, xq[j]), acc16[i][j], 0, 0, 0);
      }
      __syncthreads();
      return;
    }
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      uint4 wf[2][2], xf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int wrow = wm * 64 + i * 32 + l32;
          const int xrow = wn * 64 + i * 32 + l32;
          const int c = (kp * 2 + ks) * 2 + half;
          wf[i][ks] = *reinterpret_cast<const uint4*>(wt + wrow * ROW2 + ((c ^ ((wrow >> 1) & 7)) * 16));
          xf[i][ks] = *reinterpret_cast<const uint4*>(xt + xrow * ROW2 + ((c ^ ((xrow >> 1) & 7)) * 16));
        }
      mfma_tile<T>(acc, wf, xf);
    }
    __syncthreads();   // next tile landed (vmcnt(0)) and every wave is done with `cur`
  };
  for (int kt = 0; kt < nk; kt += 2) {
    k_step(stage0, stage1, kt + 1 < nk);
    if (kt + 1 < nk) k_step(stage1, stage0, kt + 2 < nk);
    if ((kt & 2) != 0) flush_acc<T>(tot, acc);   // every 4 K steps (128 fp32 products per output)
  }
  if constexpr (sizeof(T) == 4) {
    flush_acc<T>(tot, acc);
    flush_acc<T>(acc, tot);
  }
#ifdef SE3DS_PROBE
  if (p.probe == 1) {
    if (acc16[0][0][0] == 12345.678f) ((float*)p.out)[0] = acc16[0][0][0] + acc[0][0][0];   // (keeps the loop alive)
    return;
  }
#endif
  // (the loop's closing __syncthreads leaves both stages idle: stage0 is the epilogue scratch)
  // (DGRAD: only in the instantiation with fused batch-norm backward statistics)
  float* stats_row = ((MODE == MODE_FWD || BNB) && p.stats)
                         ? p.stats + ((int64_t)(tile_m * (BM / 64) + wn) * 2) * p.oC : nullptr;
  if constexpr (M16) {
    store_tile16<MODE, 4, BNB>(p, acc16, (int64_t)tile_m * BM + wn * 64, n0 + wm * 64, Mc, cH, cW, py, px,
                               lane, stage0 + wave * kEpiScratch<2>, stats_row, cls);
  } else {
  store_tile<T, MODE, 2, BNB>(p, acc, (int64_t)tile_m * BM + wn * 64, n0 + wm * 64, Mc, cH, cW, py,
                              px, half, l32, stage0 + wave * kEpiScratch<2>, stats_row, cls);
  }
}

// ------------------------------------------------------------------ 256-pixel macro tiles
// bf16 only.  256 pixels x CO (256 or 128) output channels per workgroup, 8 waves (2 channel
// halves x 4 pixel quarters), K step 64, both operands by LDS-DMA into two stages (2 x 64 KiB
// or 2 x 48 KiB of LDS): one workgroup per CU.  Measured on the 3x3 1024->1024 @32x64 layer the
// LDS-DMA path delivers ~37 B/clk/CU, so the 128 x 128 tile (32 KiB per 2.1 MFLOP K step) is
// fill-bound; this tile doubles (CO = 128: x1.33) the FLOPs per byte filled and, with a
// 128 x 64 wave tile, needs 0.75 fragment reads per MFMA instead of 1.  Requires oC % CO == 0
// and reduction channels % 64 == 0, and the same gather restrictions as igemm_glds_kernel.
// M16: v_mfma_f32_16x16x32_bf16 (see igemm_halo_kernel).
template <int MODE, int CO, bool BNB = false, bool M16 = false>
__global__ void __launch_bounds__(512)
igemm_big_kernel(const IgemmParams p) {
  constexpr int CB16 = CO / 32;
  typedef uint16_t T;
  constexpr int EPC = 8, BK2 = 64, ROW2 = 128, PIX = 256;
  constexpr int WT = CO * ROW2, XT = PIX * ROW2, STAGE = WT + XT;
  constexpr int NI = CO / 64;   // 32-channel MFMA blocks per wave
  constexpr int WS = CO / 64;   // weight LDS-DMA instructions per wave and K step
  // two separate LDS objects: the compiler's waitcnt pass tells LDS-DMA targets apart by the
  // alias scope of the variable, so fragment reads of one stage do not wait (vmcnt(0)) for the
  // DMA still filling the other
  __shared__ __attribute__((aligned(16))) unsigned char stage0[STAGE];
  __shared__ __attribute__((aligned(16))) unsigned char stage1[STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  int cls = 0;
  if (MODE == MODE_DGRAD) {
#pragma unroll
    for (int c = 1; c < 4; ++c)
      if (c < p.n_classes && (int)blockIdx.x >= p.cls_tile_start[c]) cls = c;
  }
  const int tile_m = blockIdx.x - p.cls_tile_start[cls];
  const int n0 = blockIdx.y * CO;
  const int s = p.stride;
  int py = 0, px = 0, cH = p.oH, cW = p.oW;
  int ky0 = 0, kx0 = 0, kstep = 1, nky = p.kh, nkx = p.kw;
  if (MODE == MODE_DGRAD) {
    py = p.cls_py[cls]; px = p.cls_px[cls];
    cH = (p.oH - py + s - 1) / s;
    cW = (p.oW - px + s - 1) / s;
    ky0 = (py + p.pad_t) % s; kx0 = (px + p.pad_l) % s; kstep = s;
    nky = ky0 < p.kh ? (p.kh - ky0 + s - 1) / s : 0;
    nkx = kx0 < p.kw ? (p.kw - kx0 + s - 1) / s : 0;
  }
  const int64_t Mc = (int64_t)p.N * cH * cW;
  const int ntaps = nky * nkx;
  const int Cr = p.sC;
  const T* __restrict__ src = (const T*)p.src;
  const T* zero = reinterpret_cast<const T*>(g_zero_page);

  // staging: instruction j of this wave fills rows (j*8 + wave)*8 .. +7 of an operand tile;
  // the swizzled chunk a lane fetches is the same for every j ((row >> 1) & 7 has period 16 rows)
  const int lrow = lane >> 3;
  const int lch = (lane & 7) ^ (((wave * 8 + lrow) >> 1) & 7);
  // per-slot gather constants (see igemm_glds_kernel); rows past the end of the pixel list get
  // an a0 that fails every bounds test
  int a0[4], b0[4], img0[4];
  const int sshift = s == 2 ? 1 : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 8 + wave) * 8 + lrow;
    const int64_t m = (int64_t)tile_m * PIX + row;
    const bool valid = m < Mc;
    const int64_t mm = valid ? m : 0;
    int n, a, b;
    pix_of(p, cls, mm, cH, cW, n, a, b);
    if (MODE == MODE_DGRAD) { a = py + a * s; b = px + b * s; }
    a0[j] = MODE == MODE_FWD ? a * s - p.pad_t : a + p.pad_t;
    b0[j] = MODE == MODE_FWD ? b * s - p.pad_l : b + p.pad_l;
    if (!valid) a0[j] = MODE == MODE_FWD ? (1 << 28) : -(1 << 28);
    img0[j] = n * p.sH * p.sW;
  }
  // weights: every channel row of the tile exists (oC % CO == 0)
  const T* wbase = (const T*)p.w + (int64_t)(n0 + wave * 8 + lrow) * p.w_n + lch * EPC;
  const int64_t wjs = 64 * p.w_n;

  const int ksteps_per_tap = Cr / BK2;
#ifdef SE3DS_PROBE
  const int nk = p.probe == 2 ? 0 : ntaps * ksteps_per_tap;
#else
  const int nk = ntaps * ksteps_per_tap;
#endif
  const T* xsrc[4];
  int xm[4];
  int64_t wtap = 0;
  int tap_i = 0, cstep = 0, tky = 0, tkx = 0;
  auto setup_tap = [&]() {
    const int ky = ky0 + kstep * tky, kx = kx0 + kstep * tkx;
    wtap = (int64_t)(ky * p.kw + kx) * p.w_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int sy, sx;
      bool ok = true;
      if (MODE == MODE_FWD) {
        sy = a0[j] + ky;
        sx = b0[j] + kx;
        if (p.wrap_w) sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx);
      } else {
        int ty = a0[j] - ky, tx = b0[j] - kx;
        if (p.wrap_w) tx = tx < 0 ? tx + p.sW : (tx >= p.sW ? tx - p.sW : tx);
        ok = ty >= 0 && tx >= 0;
        sy = ty >> sshift;
        sx = tx >> sshift;
      }
      ok = ok && (unsigned)sy < (unsigned)p.sH && (unsigned)sx < (unsigned)p.sW;
      const int pix = img0[j] + sy * p.sW + sx;
      if (ok && p.src_mask) ok = p.src_mask[pix] != 0.0f;
      if (ok) {
        xsrc[j] = src + (int64_t)pix * Cr + lch * EPC;
        xm[j] = -1;
      } else {
        xsrc[j] = zero + lch * EPC;
        xm[j] = 0;
      }
    }
    if (++tkx == nkx) { tkx = 0; ++tky; }
  };
  // One staging slot = one weight piece (if j < WS) + one pixel piece of the next tile.
  auto issue_slot = [&](unsigned char* wt, int j) {
    unsigned char* xt = wt + WT;
    const int c_off = cstep * BK2;
    const int slab = (j * 8 + wave) * 8 * ROW2;   // wave-uniform LDS byte offset
    if (j < WS)
      __builtin_amdgcn_global_load_lds((gas_ptr)(wbase + wtap + c_off + j * wjs),
                                       (las_ptr)(wt + slab), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gas_ptr)(xsrc[j] + (c_off & xm[j])), (las_ptr)(xt + slab),
                                     16, 0, 0);
  };
  auto issue_advance = [&]() {
    if (++cstep == ksteps_per_tap) {
      cstep = 0;
      if (++tap_i < ntaps) setup_tap();
    }
  };

  f32x16_t acc[M16 ? 1 : NI][2];
  f32x4_t acc16[M16 ? CB16 : 1][4];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < CB16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  } else {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  if (nk > 0) {
    setup_tap();
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_slot(stage0, j);
    issue_advance();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  const int half = lane >> 5, l32 = lane & 31;
  const int wrow0 = wm * (CO / 2) + l32, xrow0 = wn * 64 + l32;
  // (row >> 1) & 7 is unchanged by adding multiples of 32 rows: one swizzle term per operand
  const int wsw = (wrow0 >> 1) & 7, xsw = (xrow0 >> 1) & 7;
  // M16: byte offsets of this lane's chunk in k32 step 0 (step 1: ^ 64); blocks are 16 rows apart
  const int g16 = lane >> 4, r16 = lane & 15;
  const int sw16 = (g16 ^ ((r16 >> 1) & 7)) << 4;
  const int wo16 = (wm * (CO / 2) + r16) * ROW2 + sw16;
  const int xo16 = WT + (wn * 64 + r16) * ROW2 + sw16;

  // Ping-pong schedule.  A K step is NP phases; a phase is a READ slot (fragment ds_reads for 8
  // MFMAs, plus LDS-DMA issue for the next tile) and an MFMA slot (8 MFMAs = 256 pipe cycles),
  // each closed by a workgroup barrier.  Channel half 1 (waves 4-7) starts one barrier late, so
  // on every SIMD one wave is in its MFMA slot while the other reads / issues: the MFMA pipe is
  // fed continuously although an LDS-DMA piece blocks its wave's issue for 60-180 cycles.
  // (With all 8 waves in lock step the pipe idles ~800 cycles per K step during DMA issue and
  // again while every wave waits for its fragments.)
  //   stage hand-off: tile t+1 is DMA'd into the other stage during the first read slots of
  //   K step t (that stage was last read in K step t-1; the lagging half's last reads are
  //   retired by the lgkmcnt(0) in front of the barrier that ends its final read slot).  Every
  //   wave waits for its own pieces (vmcnt(0)) in the last read slot of K step t, in front of a
  //   barrier that all readers of tile t+1 pass before their first read of it.
  constexpr int QP = 8 / (2 * NI);   // 16-deep MFMA steps per phase: 1 (CO 256) or 2 (CO 128)
  constexpr int NP = 4 / QP;         // phases per K step
  if (wm == 1) __builtin_amdgcn_s_barrier();
  auto k_step = [&](unsigned char* cur, unsigned char* nxt, const bool has_next) {
    const unsigned char* wt = cur + wrow0 * ROW2;
    const unsigned char* xt = cur + WT + xrow0 * ROW2;
    uint4 xq16[M16 ? 4 : 1];
#pragma unroll
    for (int ph = 0; ph < NP; ++ph) {
      // ---- read slot
      uint4 wf[QP][NI], xf[QP][2];
      uint4 wq16[M16 ? 4 : 1];
      if constexpr (M16) {
        constexpr int ks = CO == 256 ? 1 : 0;   // ph >> ks = k32 step
        const int b0 = CO == 256 ? (ph & 1) * 4 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wq16[i] = *reinterpret_cast<const uint4*>(cur + (b0 + i) * 16 * ROW2 + (wo16 ^ ((ph >> ks) << 6)));
        if (CO == 128 || (ph & 1) == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            xq16[j] = *reinterpret_cast<const uint4*>(cur + j * 16 * ROW2 + (xo16 ^ ((ph >> ks) << 6)));
        }
      } else {
#pragma unroll
      for (int qq = 0; qq < QP; ++qq) {
        const int c = (ph * QP + qq) * 2 + half;
#pragma unroll
        for (int i = 0; i < NI; ++i)
          wf[qq][i] = *reinterpret_cast<const uint4*>(wt + i * 32 * ROW2 + ((c ^ wsw) * 16));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          xf[qq][j] = *reinterpret_cast<const uint4*>(xt + j * 32 * ROW2 + ((c ^ xsw) * 16));
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
        if (NP == 4 && ph < 2) {
          issue_slot(nxt, 2 * ph);
          issue_slot(nxt, 2 * ph + 1);
        }
        if (NP == 2 && ph == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) issue_slot(nxt, j);
        }
        if (ph == NP / 2) issue_advance();
      }
      if (ph == NP - 1) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA slot
      if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc16[(CO == 256 ? (ph & 1) * 4 : 0) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, wq16[i]), __builtin_bit_cast(bf16x8_t, xq16[j]),
                acc16[(CO == 256 ? (ph & 1) * 4 : 0) + i][j], 0, 0, 0);
      } else {
#pragma unroll
      for (int qq = 0; qq < QP; ++qq)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, wf[qq][i]), __builtin_bit_cast(bf16x8_t, xf[qq][j]),
                acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  for (int kt = 0; kt < nk; kt += 2) {
    k_step(stage0, stage1, kt + 1 < nk);
    if (kt + 1 < nk) k_step(stage1, stage0, kt + 2 < nk);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // balance the late start of channel half 1
#ifdef SE3DS_PROBE
  if (p.probe == 1) {
    if (acc16[0][0][0] == 12345.678f) ((float*)p.out)[0] = acc16[0][0][0] + acc[0][0][0];
    return;
  }
#endif
  // every wave is past its last fragment read: the stages become the epilogue scratch
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  float* stats_row = ((MODE == MODE_FWD || BNB) && p.stats)
                         ? p.stats + ((int64_t)(tile_m * (PIX / 64) + wn) * 2) * p.oC : nullptr;
  if constexpr (M16) {
    if ((p.oC & 7) == 0) {   // (always: oC % CO == 0)
      store_tile16<MODE, CB16, BNB>(p, acc16, (int64_t)tile_m * PIX + wn * 64, n0 + wm * (CO / 2), Mc, cH,
                                    cW, py, px, lane,
                                    (wave < 4 ? stage0 : stage1) + (wave & 3) * kEpiScratch<NI>, stats_row,
                                    cls);
    }
  } else {
  store_tile<T, MODE, NI, BNB>(p, acc, (int64_t)tile_m * PIX + wn * 64, n0 + wm * (CO / 2), Mc, cH, cW,
                          py, px, half, l32,
                          (wave < 4 ? stage0 : stage1) + (wave & 3) * kEpiScratch<NI>, stats_row, cls);
  }
}

// ------------------------------------------------------------------ halo-resident 3x3 tiles
// Stride-1 3x3 convolutions (forward and data gradient), bf16.  The 256 output pixels of a
// workgroup are an 8 x 32 patch of one image, so the nine taps read nine shifted views of ONE
// (8+2) x (32+2) source patch.  Per 64-channel slab the patch (340 pixels x 128 B = 43 KiB) is
// DMA'd into LDS once and stays resident for the nine K steps of the slab; only the weight tile
// (CO x 128 B) changes per K step.  Against the generic macro tile this cuts the LDS-DMA bytes
// per slab from 9 x (32 + CO/8) KiB to 43 + 9 x CO/8 KiB (CO 256: 576 -> 331 KiB, CO 128:
// 432 -> 187 KiB) and the DMA instructions per wave and K step from 8 (6) to about 5 (3), which
// is what the ping-pong read slots have to hide.  LDS: two weight stages + two patch buffers
// (CO 256: 150 KiB).  Same wave layout, MFMA order and epilogue as igemm_big_kernel.
//   patch row r = pr * 34 + pc holds source pixel (oy0 + pr, ox0 + pc); tap (ky, kx) of output
//   pixel (y0 + a, x0 + b) is patch pixel (a + dy, b + dx) with (dy, dx) = (ky, kx) forward and
//   (2 - ky, 2 - kx) for the data gradient.  16-byte chunks are XOR-swizzled with (r >> 1) & 7;
//   the 32 consecutive rows of a fragment read stay conflict-free for any start row.
// M16 (round 4): the MFMAs are v_mfma_f32_16x16x32_bf16 -- twice as many as of the 32x32x16 shape
// for the same fragment bytes; the register-only probe
// (tools/probes/mfma_ceiling.hip) sustains 1.80-1.86 PFLOP/s of them on random data against
// 1.65-1.67 under the package power limit.  Fragments: lane (g = lane / 16, r = lane % 16) reads the
// 16-byte chunk 4 * phase + g of row r of a 16-row block (conflict-free under the same swizzle);
// accumulators acc16[i][j], i = 16-channel block (8 / 4), j = 16-pixel block (4); epilogue
// store_wave_lds16.
template <int MODE, int CO, int WST, bool BNB = false, bool M16 = false>
__global__ void __launch_bounds__(512)
igemm_halo_kernel(const IgemmParams p) {
  constexpr int CB16 = CO / 32;   // M16: 16-channel blocks of a wave (8 / 4)
  typedef uint16_t T;
  constexpr int EPC = 8, ROW2 = 128, TH = 8, TW = 32, PC = TW + 2, XROWS = (TH + 2) * PC;
  constexpr int XPIECES = (XROWS + 7) / 8;          // 43 LDS-DMA pieces of 8 rows
  constexpr int XS = (XPIECES + 7) / 8;             // patch pieces per wave: up to 6
  constexpr int XBUF = XPIECES * 8 * ROW2;
  constexpr int WT = CO * ROW2;
  constexpr int NI = CO / 64, WS = CO / 64;
  // phases per K step: NP read / MFMA slot pairs of QP 16-channel fragments each (8 MFMAs per
  // slot and wave).  One 16-MFMA slot pair per K step for the 128-channel tile (half the
  // barriers) was measured and changed nothing: 761 / 799 vs 769 / 782 TFLOP/s forward / data
  // gradient on 3x3 128 -> 128 @512x1024 -- barriers are not what holds that tile at 0.36-0.40
  // MFMA-busy (its fragment reads book ~50 % of the 256 B/clk LDS pipe, one 1 KiB read per MFMA
  // against 0.75 for the 256-channel tile; the tap-fused weight gradient of the same layer,
  // with 1.2 half-size reads per MFMA and no activation-sized output, runs at 1 190 TFLOP/s).
  constexpr int QP = 8 / (2 * NI), NP = 4 / QP;   // (M16: phase = (k32 step, channel half): 16 MFMAs)
  constexpr int DIST = WST - 1;   // weight tiles are fetched DIST K steps ahead
  static_assert(WST == 2 || WST == 3, "weight stages");
  __shared__ __attribute__((aligned(16))) unsigned char wst0[WT];
  __shared__ __attribute__((aligned(16))) unsigned char wst1[WT];
  __shared__ __attribute__((aligned(16))) unsigned char wst2[WST == 3 ? WT : 16];
  // Every K step issues the same number of LDS-DMA instructions (idle ones copy the zero page
  // into `sink`): with a path-dependent count the compiler's waitcnt pass falls back to
  // vmcnt(0) in front of the fragment reads and the prefetch distance collapses.
  __shared__ __attribute__((aligned(16))) unsigned char sink[1024];
  __shared__ __attribute__((aligned(16))) unsigned char xb0[XBUF];
  __shared__ __attribute__((aligned(16))) unsigned char xb1[XBUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5, l32 = lane & 31;

  const int Cr = p.sC;
  const int nslabs = Cr / 64;
  const T* __restrict__ src = (const T*)p.src;
  const T* zero = reinterpret_cast<const T*>(g_zero_page);
  const int lrow = lane >> 3;
  const int wch = (lane & 7) ^ (((wave * 8 + lrow) >> 1) & 7);
  const int64_t wjs = 64 * p.w_n;
  const int nco = p.oC / CO;
  const int nitems = p.N * p.halo_ty * p.halo_tx * nco;

  // ---- per work item (output patch x channel tile) state
  struct ItemPos { int img, y0, x0, n0, tile; };
  ItemPos cur = {0, 0, 0, 0, 0};
  const T* xptr[XS];    // patch pieces of this wave: piece s*8 + wave, rows 8*piece + lane/8
  int xmk[XS];
  const T* wbase = nullptr;   // weight pieces: rows (j*8 + wave)*8 + lane/8 of the CO-row tile
  auto setup_item = [&](int item, ItemPos& pos, const T* (&xp)[XS], int (&xm)[XS], const T*& wb) {
    pos.n0 = (item % nco) * CO;
    int bt = item / nco;
    pos.tile = bt;
    const int tx = bt % p.halo_tx; bt /= p.halo_tx;
    const int ty = bt % p.halo_ty;
    pos.img = bt / p.halo_ty;
    pos.y0 = ty * TH; pos.x0 = tx * TW;
    const int oy0 = pos.y0 - (MODE == MODE_FWD ? p.pad_t : 2 - p.pad_t);
    const int ox0 = pos.x0 - (MODE == MODE_FWD ? p.pad_l : 2 - p.pad_l);
#pragma unroll
    for (int sl = 0; sl < XS; ++sl) {
      const int r = (sl * 8 + wave) * 8 + lrow;
      // M16: the patch rows are swizzled with r & 7 -- a 16-row x 4-chunk fragment that starts at
      // an arbitrary patch row (tap offsets) is conflict-free only under that one of eight
      // candidates (tools/probes/lds_swz.hip); (r >> 1) & 7 conflicts for 12 of 16 start rows
      const int ch = (lane & 7) ^ (M16 ? (r & 7) : ((r >> 1) & 7));
      const int pr = r / PC, pc = r - pr * PC;
      const int sy = oy0 + pr;
      int sx = ox0 + pc;
      if (p.wrap_w) sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx);
      bool ok = r < XROWS && (unsigned)sy < (unsigned)p.sH && (unsigned)sx < (unsigned)p.sW;
      const int pix = (pos.img * p.sH + (ok ? sy : 0)) * p.sW + (ok ? sx : 0);
      if (ok && p.src_mask) ok = p.src_mask[pix] != 0.0f;
      xp[sl] = ok ? src + (int64_t)pix * Cr + ch * EPC : zero + ch * EPC;
      xm[sl] = ok ? -1 : 0;
    }
    wb = (const T*)p.w + (int64_t)(pos.n0 + wave * 8 + lrow) * p.w_n + wch * EPC;
  };

  const T* zlane = zero + (lane & 7) * EPC;
  // piece sl of the patch whose pointers are (xp, xm), channel slab `slab`
  auto issue_x = [&](unsigned char* xb, int sl, const T* xp, int xm, int slab, bool real) {
    const bool live = real && sl * 8 + wave < XPIECES;   // wave-uniform
    if (live)
      __builtin_amdgcn_global_load_lds((gas_ptr)(xp + ((slab * 64) & xm)),
                                       (las_ptr)(xb + (sl * 8 + wave) * 8 * ROW2), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gas_ptr)zlane, (las_ptr)sink, 16, 0, 0);
  };
  auto issue_w = [&](unsigned char* wt, int j, const T* wb0, int tap, int slab, bool real) {
    // (wb is laundered so that the 18 x WS addresses of the unrolled K steps are recomputed
    // from scalars instead of being hoisted and spilled)
    const T* wb = wb0;
    asm volatile("" : "+v"(wb));
    const int64_t soff = (int64_t)tap * p.w_tap + slab * 64 + j * wjs;   // wave-uniform
    if (real)
      __builtin_amdgcn_global_load_lds((gas_ptr)(wb + soff),
                                       (las_ptr)(wt + (j * 8 + wave) * 8 * ROW2), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gas_ptr)zlane, (las_ptr)sink, 16, 0, 0);
  };

  f32x16_t acc[M16 ? 1 : NI][2];
  f32x4_t acc16[M16 ? CB16 : 1][M16 ? 4 : 1];
  // first patch slab and the first DIST weight tiles of the current item
  auto issue_prologue = [&]() {
#pragma unroll
    for (int sl = 0; sl < XS; ++sl) issue_x(xb0, sl, xptr[sl], xmk[sl], 0, true);
#pragma unroll
    for (int j = 0; j < WS; ++j) issue_w(wst0, j, wbase, 0, 0, true);
    if (WST == 3) {
#pragma unroll
      for (int j = 0; j < WS; ++j) issue_w(wst1, j, wbase, 1, 0, true);
    }
  };
  int item = blockIdx.x;
  setup_item(item, cur, xptr, xmk, wbase);
  issue_prologue();

  const int wrow0 = wm * (CO / 2) + l32;
  const int wsw = (wrow0 >> 1) & 7;
  int rb[2];   // patch row of this lane's pixel (tap offset 0) for the two 32-pixel fragments
#pragma unroll
  for (int j = 0; j < 2; ++j) rb[j] = (wn * 2 + j) * PC + l32;
  // M16: 16-row blocks.  Byte offset of this lane's chunk in phase 0; phase 1 is the same ^ 64
  // (chunk index 4 * phase + g under the XOR swizzle); pixel block j sits j / 2 patch rows and
  // (j & 1) * 16 columns behind block 0.
  const int g16 = lane >> 4, r16 = lane & 15;
  const int wo16 = (wm * (CO / 2) + r16) * ROW2 + ((g16 ^ ((r16 >> 1) & 7)) << 4);
  int rb16 = wn * 2 * PC + r16;

  // Persistent over work items (grid = one workgroup per CU): the next item's prologue DMA is
  // issued before this item's epilogue, and the epilogue's stores drain under the next item's
  // K loop (the 128-channel layers have only 18 K steps per item: prologue + epilogue were 40 %
  // of their time as one workgroup per item).  Streaming the next item's first slab through the
  // regular K-step prefetch instead was measured slower (1.63 vs 1.43 ms on 3x3 128->128
  // @512x1024) and was dropped.
  for (;;) {
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < CB16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  } else {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): prologue landed, older stores retired
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // ping-pong: channel half 1 runs one slot behind
  // One K step = one tap of one slab (see igemm_big_kernel for the slot / hand-off rules).
  // Additional hand-off: the patch of slab+1 is DMA'd one piece per K step (taps 0..XS-1, in
  // the read slot of the second-to-last phase, after this K step's weight pieces), so the
  // closing wait may leave that one piece in flight: vmcnt(1); tap 8 drains everything.
  auto k_step = [&](auto tap_c, unsigned char* wcur, unsigned char* wnxt, unsigned char* xcur,
                    unsigned char* xnxt, int slab, bool next_slab) {
    // wnxt: the stage that receives the weight tile of K step (this + DIST)
    constexpr int tap = decltype(tap_c)::value;
    constexpr int ky = tap / 3, kx = tap - ky * 3;
    constexpr int toff = MODE == MODE_FWD ? ky * PC + kx : (2 - ky) * PC + (2 - kx);
    constexpr int ntap = (tap + DIST) % 9;
    const bool has_next = tap + DIST < 9 || next_slab;
    const int nslab = tap + DIST >= 9 ? slab + 1 : slab;
    const T* wfar = wbase;
    const bool x_piece = tap < XS && next_slab;
    const int xsl = tap < XS ? tap : 0;
    const T* xfar = xptr[xsl];
    const int xfm = xmk[xsl];
    const int xslab = slab + 1;
    const unsigned char* wt = wcur + wrow0 * ROW2;
    const unsigned char* xr[2];
    int xsw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // opaque to the optimiser: otherwise the 72 per-tap fragment addresses are hoisted out of
      // the slab loop and spilled
      asm volatile("" : "+v"(rb[j]));
      const int r = rb[j] + toff;
      xr[j] = xcur + r * ROW2;
      xsw[j] = (r >> 1) & 7;
    }
    int xo16[4];
    uint4 xq16[M16 ? 4 : 1];
    if constexpr (M16) {
      asm volatile("" : "+v"(rb16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb16 + ((j >> 1) * PC + (j & 1) * 16 + toff);
        xo16[j] = (r << 7) | ((g16 ^ (r & 7)) << 4);
      }
    }
#pragma unroll
    for (int ph = 0; ph < NP; ++ph) {
      // ---- read slot
      uint4 wf[QP][NI], xf[QP][2];
      uint4 wq16[M16 ? 4 : 1];
      if constexpr (M16) {
        // CO 256: phase = (k32 step ph / 2, 64-channel half ph % 2); the pixel fragments of a
        // k32 step are read with its first half and stay in registers for the second.
        // CO 128: phase = k32 step, all four 16-channel blocks.
        constexpr int ks = CO == 256 ? 1 : 0;   // ph >> ks = k32 step
        const int b0 = CO == 256 ? (ph & 1) * 4 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wq16[i] = *reinterpret_cast<const uint4*>(wcur + (b0 + i) * 16 * ROW2 +
                                                    (wo16 ^ ((ph >> ks) << 6)));
        if (CO == 128 || (ph & 1) == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            xq16[j] = *reinterpret_cast<const uint4*>(xcur + (xo16[j] ^ ((ph >> ks) << 6)));
        }
      } else {
#pragma unroll
      for (int qq = 0; qq < QP; ++qq) {
        const int c = (ph * QP + qq) * 2 + half;
#pragma unroll
        for (int i = 0; i < NI; ++i)
          wf[qq][i] = *reinterpret_cast<const uint4*>(wt + i * 32 * ROW2 + ((c ^ wsw) * 16));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          xf[qq][j] = *reinterpret_cast<const uint4*>(xr[j] + ((c ^ xsw[j]) * 16));
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NP == 4 && ph < 2) {
        issue_w(wnxt, 2 * ph, wfar, ntap, nslab, has_next);
        issue_w(wnxt, 2 * ph + 1, wfar, ntap, nslab, has_next);
      }
      if (NP == 2 && ph == 0) {
#pragma unroll
        for (int j = 0; j < WS; ++j) issue_w(wnxt, j, wfar, ntap, nslab, has_next);
      }
      if (ph == NP - 2 + (NP == 2)) issue_x(xnxt, xsl, xfar, xfm, xslab, x_piece);
      if (ph == NP - 1) {
        // the weight tile of the NEXT K step must have landed; younger DMA stays in flight:
        // this K step's patch piece and, with three weight stages, this K step's weight pieces
        if (WST == 2) __builtin_amdgcn_s_waitcnt(0x0071);   // vmcnt(1) lgkmcnt(0)
        else if (WS == 2) __builtin_amdgcn_s_waitcnt(0x0073);   // vmcnt(3) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0075);   // vmcnt(5) lgkmcnt(0)
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA slot
      if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc16[(CO == 256 ? (ph & 1) * 4 : 0) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, wq16[i]), __builtin_bit_cast(bf16x8_t, xq16[j]),
                acc16[(CO == 256 ? (ph & 1) * 4 : 0) + i][j], 0, 0, 0);
      } else {
#pragma unroll
      for (int qq = 0; qq < QP; ++qq)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, wf[qq][i]), __builtin_bit_cast(bf16x8_t, xf[qq][j]),
                acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  // nine taps per slab, written out with compile-time stage / patch-buffer identities.  Two
  // weight stages: the stage parity flips from one slab to the next (two slab bodies); three
  // stages: every slab starts on stage 0, only the patch buffers alternate.
  auto wsel = [&](int i) -> unsigned char* { return i == 0 ? wst0 : (i == 1 ? wst1 : wst2); };
  auto slab_body = [&](auto par_c, int slab, bool next_slab) {
    constexpr int P = decltype(par_c)::value;
    unsigned char* xc = P ? xb1 : xb0;
    unsigned char* xn = P ? xb0 : xb1;
#define SE3DS_TAP(T_)                                                                      \
    k_step(std::integral_constant<int, T_>(),                                              \
           wsel(WST == 2 ? ((T_ + P) & 1) : (T_ % 3)),                                     \
           wsel(WST == 2 ? ((T_ + P + 1) & 1) : ((T_ + 2) % 3)), xc, xn, slab, next_slab)
    SE3DS_TAP(0); SE3DS_TAP(1); SE3DS_TAP(2); SE3DS_TAP(3); SE3DS_TAP(4);
    SE3DS_TAP(5); SE3DS_TAP(6); SE3DS_TAP(7); SE3DS_TAP(8);
#undef SE3DS_TAP
  };
  auto slab_even = [&](int slab, bool next_slab) {
    slab_body(std::integral_constant<int, 0>(), slab, next_slab);
  };
  auto slab_odd = [&](int slab, bool next_slab) {
    slab_body(std::integral_constant<int, 1>(), slab, next_slab);
  };
  for (int slab = 0; slab < nslabs; slab += 2) {
    slab_even(slab, slab + 1 < nslabs);
    if (slab + 1 < nslabs) slab_odd(slab + 1, slab + 2 < nslabs);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // balance the late start of channel half 1

  int64_t opix[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int y = cur.y0 + wn * 2 + j, x = cur.x0 + l32;
    opix[j] = (y < p.oH && x < p.oW) ? ((int64_t)cur.img * p.oH + y) * p.oW + x : -1;
  }
  const int co_base = cur.n0 + wm * (CO / 2);
  const int opy0 = cur.y0, opx0 = cur.x0, opimg = cur.img;
  // Every wave is past its last fragment read.  xb1 is the epilogue scratch (the last K steps'
  // idle copies only touch `sink`); xb0 and the first weight stages hold / take the next item's
  // first slab and weight tiles.
  float* stats_row = ((MODE == MODE_FWD || BNB) && p.stats)
                         ? p.stats + ((int64_t)(cur.tile * 4 + wn) * 2) * p.oC : nullptr;
  // The next item's pointers and prologue DMA are set up BEFORE the stores (they drain under the
  // next K loop) -- except in the 16x16x32 variant of the 256-channel tile, where the pointer set
  // on top of 128 live accumulators and the epilogue's temporaries spilled ~45 registers per item
  // (20 MB of scratch writes per launch of the dominant layer, whose workgroups have ONE item).
  constexpr bool LATE = M16 && CO == 256;
  bool have = false;
  auto next_item = [&]() {
    item += gridDim.x;
    have = item < nitems;
    if (have) {
      setup_item(item, cur, xptr, xmk, wbase);
      issue_prologue();
    }
  };
  if (!LATE) next_item();
  // (an epilogue scratch of its own that is not an LDS-DMA target, with the bias staged in LDS
  // so that the epilogue issues no global load, was measured slower: 1.62 vs 1.44 ms on the
  // 3x3 128->128 @512x1024 layer)
  unsigned char* scratch = xb1 + wave * kEpiScratch<2>;
  if constexpr (M16) {
    // (the lane index is laundered: everything the epilogue derives from it -- scratch addresses,
    // pixel offsets, bias pointers -- is otherwise hoisted above the K loop as loop-invariant and
    // spilled there: 45 registers, 20 MB of scratch traffic per launch of the dominant layer)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    int64_t opix16[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = opy0 + wn * 2 + (j >> 1), x = opx0 + (j & 1) * 16 + (lane_e & 15);
      opix16[j] = (y < p.oH && x < p.oW) ? ((int64_t)opimg * p.oH + y) * p.oW + x : -1;
    }
    // (batched write-back for the 128-channel tile: 181 VGPRs; the 256-channel tile is at 256 and spills)
    store_wave_lds16<BNB, false, CO == 128>(p, *reinterpret_cast<f32x4_t(*)[4][4]>(&acc16[0]), opix16, co_base,
                                            lane_e, scratch, stats_row);
    if (CO == 256)
      store_wave_lds16<BNB>(p, *reinterpret_cast<f32x4_t(*)[4][4]>(&acc16[CB16 - 4]), opix16,
                            co_base + 64, lane_e, scratch, stats_row);
  } else {
  store_wave_lds<2, BNB>(p, *reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), opix, co_base, lane,
                         scratch, stats_row);
  if (NI == 4)
    store_wave_lds<2, BNB>(p, *reinterpret_cast<f32x16_t(*)[2][2]>(&acc[NI - 2]), opix, co_base + 64,
                           lane, scratch, stats_row);
  }
  if (LATE) next_item();
  if (!have) break;
  }
}

// (Round 4's 4-wave variant for the 128-output-channel layers -- igemm_halo4w_kernel: a wave owns
// all 128 channels of its 64 pixels, two workgroups per CU; 13-43 % faster alone, 2-3 ms SLOWER in
// the power-limited step, DESIGN.md section 3.1 -- was removed from the library in round 5; it is
// in the history at commit 07a2034.)

// ------------------------------------------------------------------------------- wgrad
// dW[(tap,ci), co] = sum_l xg[l,(tap,ci)] * dy[l, co].  Tile: 128 (ci of one tap) x 128 (co),
// reduction over pixels l in steps of 32.  Both operands have the reduction dim as the SLOW
// memory dim ([l][c]); bf16 fragments (8 consecutive l per lane) come from
// ds_read_b64_tr_b16 transposing reads, fp32 fragments from plain ds_read_b32.
struct WgradParams {
  const void* x; int H, W, Cin;       // gathered operand (image side)
  const void* dy; int Ho, Wo, Cout;   // output-side operand
  int N, kh, kw, stride, pad_t, pad_l, wrap_w;
  const float* src_mask;              // (N,H,W) or null: x * mask (partial conv)
  int mask_binary;                    // src_mask is {0,1}
  const float* row_scale;             // (N*Ho*Wo) or null: dy * row_scale (partial renorm)
  float* dw;                          // [split][kh*kw*Cin][Cout] fp32 partial sums
  int splits; int64_t l_per_split;    // pixels per split (multiple of 32)
  int ci_tiles;                       // ceil(Cin / 128)
  int linear_k;                       // thin Cin: tile rows are the linear (tap, ci) index
};

constexpr int WG_BL = 32;               // reduction pixels per step
constexpr int WG_ROWB_BF16 = 320;       // 128 ch * 2 B + 64 B pad (conflict-free tr reads)
constexpr int WG_ROWB_F32 = 528;        // 128 ch * 4 B + 16 B pad

template <typename T>
__global__ void __launch_bounds__(kThreads, 2)
wgrad_kernel(const WgradParams p) {
  using tt = TT<T>;
  constexpr int ROWBYTES = sizeof(T) == 2 ? WG_ROWB_BF16 : WG_ROWB_F32;
  constexpr int TILE = WG_BL * ROWBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // linear_k (thin Cin, e.g. the 5- and 4-channel input convs): the 128 tile rows are 128
  // consecutive values of the linear index k = tap*Cin + ci, so ceil(K/128) tiles replace
  // kh*kw tiles that would each use Cin of their 128 rows.
  const int tap = p.linear_k ? 0 : blockIdx.x / p.ci_tiles;
  const int ci0 = p.linear_k ? blockIdx.x * 128 : (blockIdx.x - tap * p.ci_tiles) * 128;
  const int co0 = blockIdx.y * 128;
  const int split = blockIdx.z;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int Ktot = p.kh * p.kw * p.Cin;
  const int64_t L = (int64_t)p.N * p.Ho * p.Wo;
  const int64_t l_begin = (int64_t)split * p.l_per_split;
  int64_t l_end = l_begin + p.l_per_split;
  if (l_end > L) l_end = L;
  const int nsteps = l_begin < l_end ? (int)((l_end - l_begin + WG_BL - 1) / WG_BL) : 0;

  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ dy = (const T*)p.dy;
  constexpr int EPC = tt::EPC;
  constexpr int CHUNKS_PER_ROW = 128 / EPC;                 // 16 (bf16) / 32 (f32)
  constexpr int CH_PER_THREAD = WG_BL * CHUNKS_PER_ROW / kThreads;  // 2 (bf16) / 4 (f32)
  const bool vec_x = (p.Cin % EPC) == 0, vec_y = (p.Cout % EPC) == 0;

  uint4 rx[CH_PER_THREAD], ry[CH_PER_THREAD];

  // Each thread stages CH_PER_THREAD fixed (row, chunk) slots of every 32-pixel step; the pixel
  // of a slot advances by 32 per step, so (n, oy, ox) is updated incrementally (no divisions
  // in the loop).
  int pn[CH_PER_THREAD], poy[CH_PER_THREAD], pox[CH_PER_THREAD];
#pragma unroll
  for (int q = 0; q < CH_PER_THREAD; ++q) {
    int id = tid + q * kThreads;
    int r = id / CHUNKS_PER_ROW;
    int64_t l = l_begin + r;
    if (l >= L) l = L - 1;   // clamped rows are never loaded (l < l_end is re-checked)
    pn[q] = (int)(l / ((int64_t)p.Ho * p.Wo));
    int rem = (int)(l - (int64_t)pn[q] * p.Ho * p.Wo);
    poy[q] = rem / p.Wo;
    pox[q] = rem - poy[q] * p.Wo;
  }

  auto load_step = [&](int st) {
#pragma unroll
    for (int q = 0; q < CH_PER_THREAD; ++q) {
      int id = tid + q * kThreads;
      int r = id / CHUNKS_PER_ROW, c = (id - r * CHUNKS_PER_ROW) * EPC;
      int64_t l = l_begin + (int64_t)st * WG_BL + r;
      T xe[EPC], ye[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { xe[e] = tt::from_f(0.f); ye[e] = tt::from_f(0.f); }
      if (l < l_end && p.linear_k) {
        const int n = pn[q], oy = poy[q], ox = pox[q];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const int kk = ci0 + c + e;
          if (kk < Ktot) {
            const int t = kk / p.Cin, ci = kk - t * p.Cin;
            const int tky = t / p.kw, tkx = t - tky * p.kw;
            int sy = oy * p.stride - p.pad_t + tky, sx = ox * p.stride - p.pad_l + tkx;
            if (p.wrap_w) sx = sx < 0 ? sx + p.W : (sx >= p.W ? sx - p.W : sx);
            if ((unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W) {
              const int64_t pix = ((int64_t)n * p.H + sy) * p.W + sx;
              float v = tt::to_f(x[pix * p.Cin + ci]);
              if (p.src_mask) v *= p.src_mask[pix];
              xe[e] = tt::from_f(v);
            }
          }
        }
      } else if (l < l_end) {
        const int n = pn[q], oy = poy[q], ox = pox[q];
        int sy = oy * p.stride - p.pad_t + ky, sx = ox * p.stride - p.pad_l + kx;
        if (p.wrap_w) sx = sx < 0 ? sx + p.W : (sx >= p.W ? sx - p.W : sx);
        if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) {
          int64_t pix = ((int64_t)n * p.H + sy) * p.W + sx;
          const float mk = p.src_mask ? p.src_mask[pix] : 1.0f;
          const T* xs = x + pix * p.Cin + ci0 + c;
          if (vec_x && ci0 + c + EPC <= p.Cin) {
            *reinterpret_cast<uint4*>(xe) = *reinterpret_cast<const uint4*>(xs);
          } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              if (ci0 + c + e < p.Cin) xe[e] = xs[e];
          }
          if (p.src_mask && mk != 1.0f) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) xe[e] = tt::from_f(tt::to_f(xe[e]) * mk);
          }
        }
      }
      if (l < l_end) {
        const T* ys = dy + l * p.Cout + co0 + c;
        if (vec_y && co0 + c + EPC <= p.Cout) {
          *reinterpret_cast<uint4*>(ye) = *reinterpret_cast<const uint4*>(ys);
        } else {
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            if (co0 + c + e < p.Cout) ye[e] = ys[e];
        }
        if (p.row_scale) {
          const float rs = p.row_scale[l];
          if (rs != 1.0f) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) ye[e] = tt::from_f(tt::to_f(ye[e]) * rs);
          }
        }
      }
      rx[q] = *reinterpret_cast<uint4*>(xe);
      ry[q] = *reinterpret_cast<uint4*>(ye);
      // advance this slot's pixel by one step (32 pixels)
      pox[q] += WG_BL;
      while (pox[q] >= p.Wo) {
        pox[q] -= p.Wo;
        if (++poy[q] == p.Ho) { poy[q] = 0; ++pn[q]; }
      }
    }
  };
  auto store_step = [&](int stage) {
    unsigned char* xt = smem + stage * 2 * TILE;
    unsigned char* yt = xt + TILE;
#pragma unroll
    for (int q = 0; q < CH_PER_THREAD; ++q) {
      int id = tid + q * kThreads;
      int r = id / CHUNKS_PER_ROW, cb = (id - r * CHUNKS_PER_ROW) * 16;
      *reinterpret_cast<uint4*>(xt + r * ROWBYTES + cb) = rx[q];
      *reinterpret_cast<uint4*>(yt + r * ROWBYTES + cb) = ry[q];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) { load_step(0); store_step(0); }
  __syncthreads();
  const int half = lane >> 5, l32 = lane & 31;
  for (int st = 0; st < nsteps; ++st) {
    const int stage = st & 1;
    if (st + 1 < nsteps) load_step(st + 1);
    const unsigned char* xt = smem + stage * 2 * TILE;
    const unsigned char* yt = xt + TILE;
    if (sizeof(T) == 2) {
      // A fragment (32 channels x 16 l): lane (g16 = (lane>>4)&1, i = lane&15, half):
      // two transposing reads give l = lb .. lb+3 and lb+4 .. lb+7 for channel cbase+16*g16+i,
      // with lb = ks*16 + half*8.  Address lane (4j+q) -> row lb+j, channels 4q..4q+3.
      const int g16 = (lane >> 4) & 1, i16 = lane & 15;
      const int jrow = i16 >> 2, qcol = i16 & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 xf[2], yf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int lb = ks * 16 + half * 8;
          const int xc = (wm * 64 + t * 32 + g16 * 16 + qcol * 4) * 2;
          const int yc = (wn * 64 + t * 32 + g16 * 16 + qcol * 4) * 2;
          s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xt + (lb + jrow) * ROWBYTES + xc));
          s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xt + (lb + 4 + jrow) * ROWBYTES + xc));
          s16x4_t y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yt + (lb + jrow) * ROWBYTES + yc));
          s16x4_t y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yt + (lb + 4 + jrow) * ROWBYTES + yc));
          uint2 a0 = __builtin_bit_cast(uint2, x0), a1 = __builtin_bit_cast(uint2, x1);
          uint2 b0 = __builtin_bit_cast(uint2, y0), b1 = __builtin_bit_cast(uint2, y1);
          xf[t] = make_uint4(a0.x, a0.y, a1.x, a1.y);
          yf[t] = make_uint4(b0.x, b0.y, b1.x, b1.y);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, yf[j]), __builtin_bit_cast(bf16x8_t, xf[i]),
                acc[i][j], 0, 0, 0);
      }
    } else {
      // fp32: MFMA 32x32x2 takes A[i=l32][k=half]: one float per lane, read [l][c] directly
#pragma unroll 4
      for (int k2 = 0; k2 < WG_BL / 2; ++k2) {
        const int l = k2 * 2 + half;
        float xa[2], yb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          xa[t] = *reinterpret_cast<const float*>(xt + l * ROWBYTES + (wm * 64 + t * 32 + l32) * 4);
          yb[t] = *reinterpret_cast<const float*>(yt + l * ROWBYTES + (wn * 64 + t * 32 + l32) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(yb[j], xa[i], acc[i][j], 0, 0, 0);
      }
    }
    if (st + 1 < nsteps) store_step(stage ^ 1);
    __syncthreads();
  }
  // acc[i][j][r]: co = co0 + wn*64 + j*32 + (r&3) + 8*(r>>2) + 4*half ; ci = ci0 + wm*64 + i*32 + l32
  const int64_t K = (int64_t)p.kh * p.kw * p.Cin;
  float* __restrict__ dw = p.dw + (int64_t)split * K * p.Cout;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = ci0 + wm * 64 + i * 32 + l32;
    if (ci >= (p.linear_k ? Ktot : p.Cin)) continue;
    float* row = dw + (p.linear_k ? (int64_t)ci : (int64_t)tap * p.Cin + ci) * p.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + wn * 64 + j * 32 + g * 8 + half * 4;
        if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
          *reinterpret_cast<float4*>(row + co) = make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1],
                                                             acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) row[co + e] = acc[i][j][g * 4 + e];
        }
      }
  }
}

// ------------------------------------------------------------------------ LDS-DMA wgrad
// Same contraction as wgrad_kernel with both operand tiles DMA'd HBM -> LDS.  bf16: 64-pixel
// steps, rows of 256 B kept unpadded (the DMA image is lane-linear) and made conflict-free
// for the transposing reads by XOR-ing the 16-byte chunk index with (row & 3) << 2 on the
// source side and on the read side.  fp32: 32-pixel steps, plain ds_read_b32 (no swizzle).
// Used when no gather-side mask / row scale is needed.
__device__ __attribute__((aligned(256))) unsigned char g_zero_page_w[512];

template <typename T>
__global__ void __launch_bounds__(kThreads, 2)
wgrad_glds_kernel(const WgradParams p) {
  using tt = TT<T>;
  constexpr int EPC = tt::EPC;
  constexpr int BL = sizeof(T) == 2 ? 64 : 32;          // pixels per step
  constexpr int ROWBYTES = 128 * sizeof(T);             // 256 (bf16) / 512 (f32)
  constexpr int CPR = 128 / EPC;                        // 16-byte chunks per row: 16 / 32
  constexpr int RPI = 64 / CPR;                         // rows per wave instruction: 4 / 2
  constexpr int NI = BL / RPI / 4;                      // instructions per wave per operand: 4
  constexpr int TILE = BL * ROWBYTES;                   // 16 KiB
  // one LDS object per stage (see igemm_glds_kernel)
  __shared__ __attribute__((aligned(16))) unsigned char stage0[2 * TILE];
  __shared__ __attribute__((aligned(16))) unsigned char stage1[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tap = blockIdx.x / p.ci_tiles;
  const int ci0 = (blockIdx.x - tap * p.ci_tiles) * 128;
  const int co0 = blockIdx.y * 128;
  const int split = blockIdx.z;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int64_t L = (int64_t)p.N * p.Ho * p.Wo;
  const int64_t l_begin = (int64_t)split * p.l_per_split;
  int64_t l_end = l_begin + p.l_per_split;
  if (l_end > L) l_end = L;
  const int nsteps = l_begin < l_end ? (int)((l_end - l_begin + BL - 1) / BL) : 0;
  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ dy = (const T*)p.dy;
  const T* zero = reinterpret_cast<const T*>(g_zero_page_w);

  // slot j: rows (j*4 + wave)*RPI + lane/CPR; physical chunk lane%CPR
  int srow[NI], lch[NI];
  bool xc_ok[NI], yc_ok[NI];
  int pn[NI], poy[NI], pox[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    srow[j] = (j * 4 + wave) * RPI + lane / CPR;
    const int pch = lane % CPR;
    lch[j] = sizeof(T) == 2 ? (pch ^ ((srow[j] & 3) << 2)) : pch;
    xc_ok[j] = ci0 + lch[j] * EPC + EPC <= p.Cin;
    yc_ok[j] = co0 + lch[j] * EPC + EPC <= p.Cout;
    int64_t l = l_begin + srow[j];
    if (l >= L) l = L - 1;
    pn[j] = (int)(l / ((int64_t)p.Ho * p.Wo));
    int rem = (int)(l - (int64_t)pn[j] * p.Ho * p.Wo);
    poy[j] = rem / p.Wo;
    pox[j] = rem - poy[j] * p.Wo;
  }

  // Per-slot running state (all 32-bit): pixel coordinates of the slot's row, advanced by BL
  // pixels per step without loops (BL = adv_y rows + adv_x columns), and -- one step AHEAD --
  // the gathered pixel's index, its bounds flag and its mask value: the mask load is issued
  // right behind a step's LDS-DMA and is consumed a step later, when those DMAs have landed
  // anyway (a load + wait inside the issue path made every step pay a memory latency:
  // measured 97 us against 74 us without mask on the 1x1 2048->512 @32x64 layer).
  // 1x1 / stride 1 / no padding (`lin`): gathered pixel == output pixel, no coordinates at all.
  const int dyoff = ky - p.pad_t, dxoff = kx - p.pad_l;
  const int st_ = p.stride;
  const int l_end_rel = (int)(l_end - l_begin);
  const bool lin = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0;
  const int adv_y = BL / p.Wo, adv_x = BL - adv_y * p.Wo;
  int pixn[NI];
  bool okn[NI];
  float mnext[NI];
  auto look_ahead = [&](int st) {   // pixel / bounds / mask of step `st` at the current coordinates
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int lrel = st * BL + srow[j];
      bool ok = xc_ok[j] && lrel < l_end_rel;
      int pix;
      if (lin) {
        pix = (int)l_begin + lrel;
      } else {
        int sy = poy[j] * st_ + dyoff, sx = pox[j] * st_ + dxoff;
        if (p.wrap_w) sx = sx < 0 ? sx + p.W : (sx >= p.W ? sx - p.W : sx);
        ok = ok && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
        pix = (pn[j] * p.H + sy) * p.W + sx;
      }
      pixn[j] = pix;
      okn[j] = ok;
      mnext[j] = (p.src_mask && ok) ? p.src_mask[pix] : 1.0f;
    }
  };
  auto issue = [&](int st, unsigned char* xt) {
    unsigned char* yt = xt + TILE;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int lrel = st * BL + srow[j];
      const T* xs = zero + lch[j] * EPC;
      const T* ys = xs;
      if (okn[j] && mnext[j] != 0.0f) xs = x + (int64_t)pixn[j] * p.Cin + ci0 + lch[j] * EPC;
      if (yc_ok[j] && lrel < l_end_rel) ys = dy + (l_begin + lrel) * p.Cout + co0 + lch[j] * EPC;
      const int slab = (j * 4 + wave) * RPI * ROWBYTES;   // wave-uniform
      __builtin_amdgcn_global_load_lds((gas_ptr)xs, (las_ptr)(xt + slab), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gas_ptr)ys, (las_ptr)(yt + slab), 16, 0, 0);
      if (!lin) {
        pox[j] += adv_x;
        poy[j] += adv_y;
        if (pox[j] >= p.Wo) { pox[j] -= p.Wo; ++poy[j]; }
        if (poy[j] >= p.Ho) {
          const int t = poy[j] / p.Ho;
          pn[j] += t;
          poy[j] -= t * p.Ho;
        }
      }
    }
    look_ahead(st + 1);
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  look_ahead(0);
  if (nsteps > 0) issue(0, stage0);
  __syncthreads();
  const int half = lane >> 5, l32 = lane & 31;
  auto l_step = [&](int st, unsigned char* cur, unsigned char* nxt) {
    if (st + 1 < nsteps) issue(st + 1, nxt);
    const unsigned char* xt = cur;
    const unsigned char* yt = xt + TILE;
    if (sizeof(T) == 2) {
      const int g16 = (lane >> 4) & 1, i16 = lane & 15;
      const int jrow = i16 >> 2, qcol = i16 & 3;
#pragma unroll
      for (int ks = 0; ks < BL / 16; ++ks) {
        uint4 xf[2], yf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int lb = ks * 16 + half * 8;
          // channel byte offset -> (chunk, 8-byte half); rows lb+jrow and lb+4+jrow share row&3
          const int xcol = wm * 64 + t * 32 + g16 * 16 + qcol * 4;
          const int ycol = wn * 64 + t * 32 + g16 * 16 + qcol * 4;
          const int sw = (jrow & 3) << 2;
          const int xo = (((xcol >> 3) ^ sw) << 4) + ((xcol & 4) << 1);
          const int yo = (((ycol >> 3) ^ sw) << 4) + ((ycol & 4) << 1);
          s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xt + (lb + jrow) * ROWBYTES + xo));
          s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xt + (lb + 4 + jrow) * ROWBYTES + xo));
          s16x4_t y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yt + (lb + jrow) * ROWBYTES + yo));
          s16x4_t y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yt + (lb + 4 + jrow) * ROWBYTES + yo));
          uint2 a0 = __builtin_bit_cast(uint2, x0), a1 = __builtin_bit_cast(uint2, x1);
          uint2 b0 = __builtin_bit_cast(uint2, y0), b1 = __builtin_bit_cast(uint2, y1);
          xf[t] = make_uint4(a0.x, a0.y, a1.x, a1.y);
          yf[t] = make_uint4(b0.x, b0.y, b1.x, b1.y);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, yf[j]), __builtin_bit_cast(bf16x8_t, xf[i]),
                acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int k2 = 0; k2 < BL / 2; ++k2) {
        const int l = k2 * 2 + half;
        float xa[2], yb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          xa[t] = *reinterpret_cast<const float*>(xt + l * ROWBYTES + (wm * 64 + t * 32 + l32) * 4);
          yb[t] = *reinterpret_cast<const float*>(yt + l * ROWBYTES + (wn * 64 + t * 32 + l32) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(yb[j], xa[i], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  for (int st = 0; st < nsteps; st += 2) {
    l_step(st, stage0, stage1);
    if (st + 1 < nsteps) l_step(st + 1, stage1, stage0);
  }
  const int64_t K = (int64_t)p.kh * p.kw * p.Cin;
  float* __restrict__ dw = p.dw + (int64_t)split * K * p.Cout;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = ci0 + wm * 64 + i * 32 + l32;
    if (ci >= p.Cin) continue;
    float* row = dw + ((int64_t)tap * p.Cin + ci) * p.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + wn * 64 + j * 32 + g * 8 + half * 4;
        if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
          *reinterpret_cast<float4*>(row + co) = make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1],
                                                             acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cout) row[co + e] = acc[i][j][g * 4 + e];
        }
      }
  }
}

// ------------------------------------------------------------------- tap-fused 3x3 wgrad
// Stride-1 3x3 weight gradients, bf16.  One workgroup owns a 64-channel slice of the input, a
// 128-channel slice of the output gradient and ALL NINE taps: dW[tap][ci][co] for 9 x 64 x 128
// outputs (8 waves x 9 accumulator tiles of 32 x 32).  The reduction runs over "steps" of 64
// output pixels (2 rows x 32 columns of one image); a step needs the 64 x 128 dy tile and the
// (2+2) x (32+2) x 64 halo of x, and the nine taps are nine shifted views of that one halo.
// Per step a CU fills 33 KiB for 9.4 MFLOP (the per-tap 128 x 128 tile: 9 x 32 KiB for the same
// work), so the kernel is MFMA-bound instead of fill-bound.  Schedule: as igemm_halo_kernel
// (ping-pong read / MFMA slots of one 16-pixel step = 9 MFMAs, LDS-DMA of the next step in the
// first two read slots, two LDS stages).  Both operands are [pixel][channel] in LDS and reach
// the MFMA layout (8 consecutive pixels per lane) through ds_read_b64_tr_b16:
//   dy  rows of 256 B, 16-byte chunk index ^= (row & 3) << 2   (as wgrad_glds_kernel)
//   x   rows of 128 B, 16-byte chunk index ^= (row & 2) << 1 -- any four consecutive rows then
//       cover the four 64-byte windows of the 256-byte bank line, for every tap shift.
// Partial sums go to dw[split][(tap, ci)][co]; wgrad_reduce_kernel adds the splits in order.
struct WgradTapsParams {
  const uint16_t* x; int H, W, Cin;
  const uint16_t* dy; int Ho, Wo, Cout;
  int N, pad_t, pad_l, wrap_w;
  const float* src_mask;     // binary (N,H,W) or null
  float* dw;
  int steps_y, steps_x;      // steps per image: ceil(Ho/2) x ceil(Wo/32)
  int total_steps, steps_per_split;
  // thin output (Cout <= 8: the 128->3 / 128->1 heads): dy is a zero-padded copy with
  // dy_cstride = 8 channels per pixel, only the first co_valid channels are real; the waves of
  // the other three 32-channel blocks skip their MFMAs and dw rows have Cout (= co_valid) floats
  int dy_cstride, co_valid;
};

__global__ void __launch_bounds__(512)
wgrad_taps_kernel(const WgradTapsParams p) {
  typedef uint16_t T;
  constexpr int YROW = 256, XROW = 128, PC = 34;
  constexpr int YT = 64 * YROW;              // 16 KiB
  constexpr int XR = 4 * PC;                 // 136 halo rows
  constexpr int XPIECES = (XR + 7) / 8;      // 17
  constexpr int XT = XPIECES * 8 * XROW + 1024;   // 17 KiB + slack: the 12-pixel window of the last
                                                  // halo row overshoots by two rows (never used)
  // NST stages: the tiles of step st + NST - 1 are fetched during step st.  (Three stages
  // measured slower than two on every shape: 0.43 vs 0.36 ms on 3x3 1024->1024 @32x64.)
  constexpr int NST = 2;
  __shared__ __attribute__((aligned(16))) unsigned char stage0[YT + XT];
  __shared__ __attribute__((aligned(16))) unsigned char stage1[YT + XT];
  __shared__ __attribute__((aligned(16))) unsigned char stage2[NST == 3 ? YT + XT : 16];
  __shared__ __attribute__((aligned(16))) unsigned char sink[1024];   // target of idle copies
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int iw = wave >> 2, cw = wave & 3;   // 32-channel block of x / of dy owned by this wave
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 128;
  const int split = blockIdx.z;
  const int st_begin = split * p.steps_per_split;
  int st_end = st_begin + p.steps_per_split;
  if (st_end > p.total_steps) st_end = p.total_steps;
  const int nsteps = st_end > st_begin ? st_end - st_begin : 0;
  const T* zero = reinterpret_cast<const T*>(g_zero_page_w);

  // ---- LDS-DMA pieces of this wave
  // dy: pieces 2*wave, 2*wave+1; piece = 4 rows x 256 B; lane -> row piece*4 + lane/16
  int ya[2], yb[2], ych[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 4 + (lane >> 4);
    ya[j] = r >> 5; yb[j] = r & 31;
    ych[j] = (lane & 15) ^ ((r & 3) << 2);
  }
  // x: pieces wave, wave + 8 (and 16 on wave 0); piece = 8 rows x 128 B; lane -> row piece*8 + lane/8
  int xa[3], xb[3], xch[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = (wave + 8 * j) * 8 + (lane >> 3);
    xa[j] = r / PC; xb[j] = r - xa[j] * PC;
    if (r >= XR) xa[j] = 1 << 20;   // rows past the halo: always out of bounds
    xch[j] = (lane & 7) ^ ((r & 2) << 1);
  }

  // Every wave issues 2 dy + 3 x LDS-DMA instructions per step (idle ones copy the zero page
  // into `sink`), so the counted waits below are exact on every path.
  struct StepPos { int img, y0, x0; };
  auto step_pos = [&](int st) {
    int t = st_begin + st;   // step -> (image, first row, first column); wave-uniform
    StepPos sp;
    const int sx_i = t % p.steps_x; t /= p.steps_x;
    const int sy_i = t % p.steps_y;
    sp.img = t / p.steps_y;
    sp.y0 = sy_i * 2; sp.x0 = sx_i * 32;
    return sp;
  };
  auto issue_y = [&](const StepPos& sp, unsigned char* stg, bool real) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = sp.y0 + ya[j], ox = sp.x0 + yb[j];
      const T* src = zero + ych[j] * 8;
      if (oy < p.Ho && ox < p.Wo && co0 + ych[j] * 8 < p.dy_cstride)
        src = p.dy + ((int64_t)(sp.img * p.Ho + oy) * p.Wo + ox) * p.dy_cstride + co0 + ych[j] * 8;
      if (real)
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(stg + (wave * 2 + j) * 4 * YROW), 16,
                                         0, 0);
      else
        __builtin_amdgcn_global_load_lds((gas_ptr)zero, (las_ptr)sink, 16, 0, 0);
    }
  };
  auto issue_x = [&](const StepPos& sp, unsigned char* stg, bool real) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int sy = sp.y0 - p.pad_t + xa[j];
      int sx = sp.x0 - p.pad_l + xb[j];
      if (p.wrap_w) sx = sx < 0 ? sx + p.W : (sx >= p.W ? sx - p.W : sx);
      const T* src = zero + xch[j] * 8;
      // (`real` guards the mask read: past the last step the position is outside the tensor)
      if (real && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W) {
        const int pix = (sp.img * p.H + sy) * p.W + sx;
        if (!(p.src_mask && p.src_mask[pix] == 0.0f))
          src = p.x + (int64_t)pix * p.Cin + ci0 + xch[j] * 8;
      }
      if (real && (j < 2 || wave == 0))
        __builtin_amdgcn_global_load_lds((gas_ptr)src,
                                         (las_ptr)(stg + YT + (wave + 8 * j) * 8 * XROW), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gas_ptr)zero, (las_ptr)sink, 16, 0, 0);
    }
  };

  f32x16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // position of the step being prefetched, advanced by one step per l_step (no divisions in
  // the loop)
  StepPos fp = step_pos(0);
  auto advance = [&](StepPos& sp) {
    sp.x0 += 32;
    if (sp.x0 >= p.steps_x * 32) {
      sp.x0 = 0;
      sp.y0 += 2;
      if (sp.y0 >= p.steps_y * 2) { sp.y0 = 0; ++sp.img; }
    }
  };
  issue_y(fp, stage0, nsteps > 0); issue_x(fp, stage0, nsteps > 0);
  advance(fp);
  if (NST == 3) {
    issue_y(fp, stage1, nsteps > 1); issue_x(fp, stage1, nsteps > 1);
    advance(fp);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __builtin_amdgcn_s_barrier();

  const int half = lane >> 5, l32 = lane & 31;
  const int g16 = (lane >> 4) & 1, i16 = lane & 15;
  const int jrow = i16 >> 2, qcol = i16 & 3;
  // per-lane constants of the transposing reads
  const int ycol = cw * 32 + g16 * 16 + qcol * 4;                         // dy channel of the segment
  const int yo = (((ycol >> 3) ^ ((jrow & 3) << 2)) << 4) + ((ycol & 4) << 1);
  const int xcol = iw * 32 + g16 * 16 + qcol * 4;
  const int xlo = (xcol & 4) << 1;
  typedef __attribute__((address_space(3))) s16x4_t* lds_seg;
  // x fragment address = lane part + compile-time row offset.  The swizzle bit of halo row
  // r = 34*(a+ky) + kx + lb + jrow is bit 1 of r = ((a+ky) & 1) ^ bit1(kx + jrow): six lane
  // constants cover all 36 (step quarter, tap) combinations, the rest is an immediate offset.
  int xlane[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int swz = pp ^ (((kx + jrow) >> 1) & 1);
      xlane[kx][pp] = (YT + (half * 8 + jrow) * XROW + ((xcol >> 3) << 4) + xlo) ^ (swz << 6);
    }
  const int ylane = (half * 8 + jrow) * YROW + yo;

  if (iw == 1) __builtin_amdgcn_s_barrier();   // ping-pong: waves 4-7 run one slot behind
  // `far` receives the tiles of step st+2 (it held step st-1, whose last reads every wave has
  // retired before the barrier that opened this step)
  const bool work = cw * 32 < p.co_valid;   // wave-uniform: this wave's channel block is real
  auto l_step = [&](unsigned char* cur, unsigned char* far, int st) {
    const bool has_far = st + NST - 1 < nsteps;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      // ---- read slot: 16 pixels (row a, columns b0 .. b0+15); this lane: pixels lb .. lb+7
      const int a = kq >> 1;
      // x: per tap row ky ONE window of 12 pixels (three transposing reads); the fragments of
      // kx = 0, 1, 2 are pixels [0,8), [1,9), [2,10) of it (kx = 1 by four v_alignbit) -- 11 LDS
      // reads per 9 MFMAs instead of 20.
      uint4 yf;
      uint2 xr[3][3];
      if (work) {
        const unsigned char* yp = cur + ylane + (a * 32 + (kq & 1) * 16) * YROW;
        uint2 v0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)yp));
        uint2 v1 = __builtin_bit_cast(
            uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(yp + 4 * YROW)));
        yf = make_uint4(v0.x, v0.y, v1.x, v1.y);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned char* xp =
              cur + xlane[0][(a + ky) & 1] + ((a + ky) * PC + (kq & 1) * 16) * XROW;
#pragma unroll
          for (int w = 0; w < 3; ++w)
            xr[ky][w] = __builtin_bit_cast(
                uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(xp + w * 4 * XROW)));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kq == 0) issue_y(fp, far, has_far);
      if (kq == 1) { issue_x(fp, far, has_far); advance(fp); }
      // step st+1 must have landed; with three stages this step's five pieces stay in flight
      if (kq == 3) {
        if (NST == 3) __builtin_amdgcn_s_waitcnt(0x0075);   // vmcnt(5) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0)
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA slot
      if (work) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const uint2 v0 = xr[ky][0], v1 = xr[ky][1], v2 = xr[ky][2];
          const uint4 f0 = make_uint4(v0.x, v0.y, v1.x, v1.y);
          const uint4 f1 = make_uint4(__builtin_amdgcn_alignbit(v0.y, v0.x, 16),
                                      __builtin_amdgcn_alignbit(v1.x, v0.y, 16),
                                      __builtin_amdgcn_alignbit(v1.y, v1.x, 16),
                                      __builtin_amdgcn_alignbit(v2.x, v1.y, 16));
          const uint4 f2 = make_uint4(v0.y, v1.x, v1.y, v2.x);
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8_t, yf), __builtin_bit_cast(bf16x8_t, f0), acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8_t, yf), __builtin_bit_cast(bf16x8_t, f1), acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8_t, yf), __builtin_bit_cast(bf16x8_t, f2), acc[ky * 3 + 2], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  if (NST == 3) {
    for (int st = 0; st < nsteps; st += 3) {
      l_step(stage0, stage2, st);
      if (st + 1 < nsteps) l_step(stage1, stage0, st + 1);
      if (st + 2 < nsteps) l_step(stage2, stage1, st + 2);
    }
  } else {
    for (int st = 0; st < nsteps; st += 2) {
      l_step(stage0, stage1, st);
      if (st + 1 < nsteps) l_step(stage1, stage0, st + 1);
    }
  }
  if (iw == 0) __builtin_amdgcn_s_barrier();

  // acc[t][r]: ci = ci0 + iw*32 + l32, co = co0 + cw*32 + (r&3) + 8*(r>>2) + 4*half
  const int64_t K = (int64_t)9 * p.Cin;
  float* __restrict__ dw = p.dw + (int64_t)split * K * p.Cout;
  const int ci = ci0 + iw * 32 + l32;
  if (!work) return;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* row = dw + ((int64_t)t * p.Cin + ci) * p.Cout + co0 + cw * 32 + half * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (p.co_valid >= 128) {
        *reinterpret_cast<float4*>(row + g * 8) =
            make_float4(acc[t][g * 4], acc[t][g * 4 + 1], acc[t][g * 4 + 2], acc[t][g * 4 + 3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cw * 32 + g * 8 + half * 4 + e < p.co_valid) row[g * 8 + e] = acc[t][g * 4 + e];
      }
    }
  }
}

// ------------------------------------------------- tap-fused 3x3 wgrad, step-pipelined schedule
// Same tiling, LDS layout and fragment reads as wgrad_taps_kernel, different schedule (round 2;
// measured on 3x3 1024->1024 @32x64 batch 8 with compile-time switches: MFMAs alone 126-143 us,
// the ping-pong kernel 346 us of which the per-step DMA issue code, the barrier bubble at every
// step boundary and the unhidden LDS-DMA account for ~150 us):
//  * three LDS stages; ONE barrier per step, placed after the third of the four 16-pixel slots:
//    it publishes the tiles of step st+1 (fetched a whole step earlier) and frees the stage of
//    step st-1 for the tiles of step st+2, whose LDS-DMA is issued right behind it.  The last
//    slot then already reads the first fragments of step st+1, so no wave meets the barrier
//    with an empty MFMA queue;
//  * inside a step: wait(k) -> issue the reads of slot k+1 -> nine MFMAs of slot k.  At every
//    wait only the fragments it needs are outstanding (lgkmcnt(0) is exact);
//  * the DMA source addresses are a per-step scalar base + per-lane constants (no divisions,
//    no per-piece branches); stage pointers are compile-time constants in the unrolled loop so
//    that the compiler knows DMA targets and fragment reads never alias (it otherwise puts
//    vmcnt(0) in front of the reads).
// Restrictions (the launcher falls back to wgrad_taps_kernel): no gather mask, no wrap, full
// 128-channel dy blocks.
// M16 (round 4): v_mfma_f32_16x16x32_bf16 (1.80-1.86 vs 1.65-1.67 PFLOP/s sustained under the
// power limit, tools/probes/mfma_ceiling.hip).  A slot is one 32-pixel row of the step x one
// 16-channel half of the wave's input channels: 18 MFMAs (3 x 3 taps x 2 output-channel blocks);
// lane (g = lane / 16, i = lane % 16) holds pixels 8g .. 8g+7 of the row for channel i of a block
// (the transposing reads give a 16-lane group 4 pixels x 16 channels); accumulators
// acc16[tap][output block][input block], rows g*4 .. +3 = output channels, column i = input channel.
template <bool M16>
__global__ void __launch_bounds__(512)
wgrad_taps3_kernel(const WgradTapsParams p) {
  typedef uint16_t T;
  constexpr int YROW = 256, XROW = 128, PC = 34;
  constexpr int YT = 64 * YROW;              // 16 KiB
  constexpr int XR = 4 * PC;                 // 136 halo rows
  constexpr int XPIECES = (XR + 7) / 8;      // 17
  constexpr int XT = XPIECES * 8 * XROW + 1024;
  __shared__ __attribute__((aligned(16))) unsigned char stage0[YT + XT];
  __shared__ __attribute__((aligned(16))) unsigned char stage1[YT + XT];
  __shared__ __attribute__((aligned(16))) unsigned char stage2[YT + XT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int iw = wave >> 2, cw = wave & 3;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 128;
  const int split = blockIdx.z;
  const int st_begin = split * p.steps_per_split;
  int st_end = st_begin + p.steps_per_split;
  if (st_end > p.total_steps) st_end = p.total_steps;
  const int nsteps = st_end > st_begin ? st_end - st_begin : 0;
  const T* zero = reinterpret_cast<const T*>(g_zero_page_w);

  // ---- LDS-DMA pieces of this wave: lane constants
  // dy: pieces 2*wave, 2*wave+1 (4 rows x 256 B); x: pieces wave, wave+8 (and 16 on wave 0; 8 rows
  // x 128 B).  Offsets in elements relative to the step's first pixel.
  int ya[2], yb[2];
  int64_t yoff[2];
  const T* yzero[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 4 + (lane >> 4);
    ya[j] = r >> 5; yb[j] = r & 31;
    // (M16: chunk bit 1 ^= bit 3 of the row -- the four 16-lane groups of a transposing read then
    // address pixel octets 8 rows apart, which would otherwise share their banks: 2 x the LDS
    // cycles, tools/probes/lds_conflict.hip patterns 7 / 8)
    const int ch = (lane & 15) ^ ((r & 3) << 2) ^ (M16 ? ((r >> 3) & 1) << 1 : 0);
    yoff[j] = ((int64_t)ya[j] * p.Wo + yb[j]) * p.Cout + co0 + ch * 8;
    yzero[j] = zero + ch * 8;
  }
  int xa[3], xb[3];
  int64_t xoff[3];
  const T* xzero[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = (wave + 8 * j) * 8 + (lane >> 3);
    xa[j] = r / PC; xb[j] = r - xa[j] * PC;
    if (r >= XR) xa[j] = 1 << 20;   // rows past the halo: always out of bounds
    const int ch = (lane & 7) ^ ((r & 2) << 1) ^ (M16 ? ((r >> 3) & 1) << 1 : 0);
    xoff[j] = ((int64_t)(r >= XR ? 0 : xa[j]) * p.W + xb[j]) * p.Cin + ci0 + ch * 8;
    xzero[j] = zero + ch * 8;
  }

  struct StepPos { int img, y0, x0; };
  StepPos fp;
  {
    int t = st_begin;
    const int sx_i = t % p.steps_x; t /= p.steps_x;
    const int sy_i = t % p.steps_y;
    fp.img = t / p.steps_y;
    fp.y0 = sy_i * 2; fp.x0 = sx_i * 32;
  }
  auto advance = [&](StepPos& sp) {
    sp.x0 += 32;
    if (sp.x0 >= p.steps_x * 32) {
      sp.x0 = 0;
      sp.y0 += 2;
      if (sp.y0 >= p.steps_y * 2) { sp.y0 = 0; ++sp.img; }
    }
  };
  // all five (four) pieces of one step; `real` is wave-uniform (past the last step nothing is
  // issued: the waits below are vmcnt(0))
  // binary gather mask (partial convs): the three mask values of a step are loaded one issue
  // AHEAD, right behind the previous step's LDS-DMA, and are in registers when they are needed
  float mxn[3] = {1.0f, 1.0f, 1.0f};
  auto mask_ahead = [&](const StepPos& sp, bool real) {
    if (p.src_mask == nullptr) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int sy = sp.y0 - p.pad_t + xa[j], sx = sp.x0 - p.pad_l + xb[j];
      const bool ok = real && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
      mxn[j] = ok ? p.src_mask[(sp.img * p.H + sy) * p.W + sx] : 1.0f;
    }
  };
  auto issue = [&](const StepPos& sp, unsigned char* stg, bool real) {
    if (!real) return;
    const T* ybase = p.dy + ((int64_t)(sp.img * p.Ho + sp.y0) * p.Wo + sp.x0) * p.Cout;
    const T* xbase = p.x + ((int64_t)(sp.img * p.H + sp.y0 - p.pad_t) * p.W + (sp.x0 - p.pad_l)) * p.Cin;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = sp.y0 + ya[j] < p.Ho && sp.x0 + yb[j] < p.Wo;
      const T* src = ok ? ybase + yoff[j] : yzero[j];
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(stg + (wave * 2 + j) * 4 * YROW), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j == 2 && wave != 0) break;
      const bool ok = (unsigned)(sp.y0 - p.pad_t + xa[j]) < (unsigned)p.H &&
                      (unsigned)(sp.x0 - p.pad_l + xb[j]) < (unsigned)p.W && mxn[j] != 0.0f;
      const T* src = ok ? xbase + xoff[j] : xzero[j];
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(stg + YT + (wave + 8 * j) * 8 * XROW), 16,
                                       0, 0);
    }
  };

  f32x16_t acc[M16 ? 1 : 9];
  f32x4_t acc16[M16 ? 9 : 1][2][2];
  if constexpr (M16) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc16[t][cb][b][r] = 0.f;
  } else {
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  }

  mask_ahead(fp, nsteps > 0);
  issue(fp, stage0, nsteps > 0);
  advance(fp);
  mask_ahead(fp, nsteps > 1);
  issue(fp, stage1, nsteps > 1);
  advance(fp);
  mask_ahead(fp, nsteps > 2);

  const int half = lane >> 5, l32 = lane & 31;
  const int g16 = (lane >> 4) & 1, i16 = lane & 15;
  const int jrow = i16 >> 2, qcol = i16 & 3;
  const int ycol = cw * 32 + g16 * 16 + qcol * 4;
  const int yo = (((ycol >> 3) ^ ((jrow & 3) << 2)) << 4) + ((ycol & 4) << 1);
  const int xcol = iw * 32 + g16 * 16 + qcol * 4;
  const int xlo = (xcol & 4) << 1;
  typedef __attribute__((address_space(3))) s16x4_t* lds_seg;
  int xlane[2];   // swizzle parity of the halo row (see wgrad_taps_kernel)
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int swz = pp ^ ((jrow >> 1) & 1);
    xlane[pp] = (YT + (half * 8 + jrow) * XROW + ((xcol >> 3) << 4) + xlo) ^ (swz << 6);
  }
  const int ylane = (half * 8 + jrow) * YROW + yo;

  if constexpr (M16) {
    const int g4 = lane >> 4;                                    // pixel octet of the 32-pixel row
    const int ycol6 = cw * 32 + qcol * 4;                        // (+ 16 per output block)
    const int yo6 = (((ycol6 >> 3) ^ ((jrow & 3) << 2)) << 4) + ((ycol6 & 4) << 1);
    const int ylane6 = ((g4 * 8 + jrow) * YROW + yo6) ^ ((g4 & 1) << 5);   // (row bit 3 = g4 & 1)
    const int xcol6 = iw * 32 + qcol * 4;                        // (+ 16 per input block)
    int xlane6[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int swz = pp ^ ((jrow >> 1) & 1);
      xlane6[pp] = (YT + (g4 * 8 + jrow) * XROW + ((xcol6 >> 3) << 4) + ((xcol6 & 4) << 1)) ^ (swz << 6);
    }
    // halo row r = 34 m + 8 g4 + (2 m + 4 w + jrow): bit 3 = (g4 & 1) ^ bit 3 of t = 2 m + 4 w + jrow,
    // which is 0 for 2 m + 4 w < 6, 1 for 8 .. 12, (jrow >= 2) for 6 and (jrow < 2) for 14
    const int j6 = (jrow >> 1) & 1;
    const int xs[4] = {(g4 & 1) << 5, ((g4 & 1) ^ 1) << 5, ((g4 & 1) ^ j6) << 5, ((g4 & 1) ^ j6 ^ 1) << 5};
    // dy fragments of row a: two 16-channel blocks (block 1 = chunk index + 2: bit 5, untouched by
    // the row swizzle of bits 6-7)
    auto read_y = [&](const unsigned char* cur, int a, uint4 (&yf)[2]) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const unsigned char* yp = cur + (ylane6 ^ (cb << 5)) + a * 32 * YROW;
        uint2 v0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)yp));
        uint2 v1 = __builtin_bit_cast(
            uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(yp + 4 * YROW)));
        yf[cb] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
    };
    // x windows (12 pixels) of row a, input block b, for the three tap rows
    auto read_x = [&](const unsigned char* cur, int a, int b, uint2 (&xr)[3][3]) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const int K = 2 * (a + ky) + 4 * w;
          const int sel = K < 6 ? 0 : (K == 6 ? 2 : (K == 14 ? 3 : 1));
          const unsigned char* xp = cur + (xlane6[(a + ky) & 1] ^ (b << 5) ^ xs[sel]) + (a + ky) * PC * XROW;
          xr[ky][w] = __builtin_bit_cast(
              uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(xp + w * 4 * XROW)));
        }
      }
    };
    auto mfma18 = [&](auto b_c, const uint4 (&yf)[2], const uint2 (&xr)[3][3]) {
      constexpr int B = decltype(b_c)::value;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const uint2 v0 = xr[ky][0], v1 = xr[ky][1], v2 = xr[ky][2];
        const uint4 f[3] = {make_uint4(v0.x, v0.y, v1.x, v1.y),
                            make_uint4(__builtin_amdgcn_alignbit(v0.y, v0.x, 16),
                                       __builtin_amdgcn_alignbit(v1.x, v0.y, 16),
                                       __builtin_amdgcn_alignbit(v1.y, v1.x, 16),
                                       __builtin_amdgcn_alignbit(v2.x, v1.y, 16)),
                            make_uint4(v0.y, v1.x, v1.y, v2.x)};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc16[ky * 3 + kx][cb][B] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, yf[cb]), __builtin_bit_cast(bf16x8_t, f[kx]),
                acc16[ky * 3 + kx][cb][B], 0, 0, 0);
      }
    };
#define TAPS_WAIT_LDS() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); \
                             __builtin_amdgcn_sched_barrier(0); } while (0)
    uint4 yE[2], yO[2];
    uint2 xA[3][3], xB[3][3];
    const std::integral_constant<int, 0> B0;
    const std::integral_constant<int, 1> B1;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): steps 0 and 1 landed
    __builtin_amdgcn_s_barrier();
    if (nsteps > 0) { read_y(stage0, 0, yE); read_x(stage0, 0, 0, xA); }
    auto step = [&](unsigned char* cur, unsigned char* nxt, unsigned char* far, int st) {
      TAPS_WAIT_LDS();
      read_x(cur, 0, 1, xB);
      __builtin_amdgcn_sched_barrier(0);
      mfma18(B0, yE, xA);
      TAPS_WAIT_LDS();
      read_y(cur, 1, yO);
      read_x(cur, 1, 0, xA);
      __builtin_amdgcn_sched_barrier(0);
      mfma18(B1, yE, xB);
      TAPS_WAIT_LDS();
      read_x(cur, 1, 1, xB);
      __builtin_amdgcn_sched_barrier(0);
      mfma18(B0, yO, xA);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of step st+1 landed
      __builtin_amdgcn_s_barrier();         // ... everyone's did; every wave is done with `far`
      __builtin_amdgcn_sched_barrier(0);
      issue(fp, far, st + 2 < nsteps);
      advance(fp);
      mask_ahead(fp, st + 3 < nsteps);
      TAPS_WAIT_LDS();
      if (st + 1 < nsteps) { read_y(nxt, 0, yE); read_x(nxt, 0, 0, xA); }
      __builtin_amdgcn_sched_barrier(0);
      mfma18(B1, yO, xB);
    };
#undef TAPS_WAIT_LDS
    for (int st = 0; st < nsteps; st += 3) {
      step(stage0, stage1, stage2, st);
      if (st + 1 < nsteps) step(stage1, stage2, stage0, st + 1);
      if (st + 2 < nsteps) step(stage2, stage0, stage1, st + 2);
    }
    // acc16[t][cb][b][r]: ci = ci0 + iw*32 + b*16 + lane%16, co = co0 + cw*32 + cb*16 + 4*(lane/16) + r
    const int64_t K6 = (int64_t)9 * p.Cin;
    float* __restrict__ dw6 = p.dw + (int64_t)split * K6 * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ci6 = ci0 + iw * 32 + b * 16 + (lane & 15);
        float* row = dw6 + ((int64_t)t * p.Cin + ci6) * p.Cout + co0 + cw * 32 + g4 * 4;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          *reinterpret_cast<float4*>(row + cb * 16) =
              make_float4(acc16[t][cb][b][0], acc16[t][cb][b][1], acc16[t][cb][b][2], acc16[t][cb][b][3]);
      }
    return;
  }

  auto read_frag = [&](const unsigned char* cur, int kq, uint4& yf, uint2 (&xr)[3][3]) {
    const int a = kq >> 1;
    const unsigned char* yp = cur + ylane + (a * 32 + (kq & 1) * 16) * YROW;
    uint2 v0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)yp));
    uint2 v1 = __builtin_bit_cast(
        uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(yp + 4 * YROW)));
    yf = make_uint4(v0.x, v0.y, v1.x, v1.y);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const unsigned char* xp = cur + xlane[(a + ky) & 1] + ((a + ky) * PC + (kq & 1) * 16) * XROW;
#pragma unroll
      for (int w = 0; w < 3; ++w)
        xr[ky][w] = __builtin_bit_cast(
            uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(xp + w * 4 * XROW)));
    }
  };
  auto mfma9 = [&](const uint4& yf, const uint2 (&xr)[3][3]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const uint2 v0 = xr[ky][0], v1 = xr[ky][1], v2 = xr[ky][2];
      const uint4 f0 = make_uint4(v0.x, v0.y, v1.x, v1.y);
      const uint4 f1 = make_uint4(__builtin_amdgcn_alignbit(v0.y, v0.x, 16),
                                  __builtin_amdgcn_alignbit(v1.x, v0.y, 16),
                                  __builtin_amdgcn_alignbit(v1.y, v1.x, 16),
                                  __builtin_amdgcn_alignbit(v2.x, v1.y, 16));
      const uint4 f2 = make_uint4(v0.y, v1.x, v1.y, v2.x);
      acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8_t, yf), __builtin_bit_cast(bf16x8_t, f0), acc[ky * 3 + 0], 0, 0, 0);
      acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8_t, yf), __builtin_bit_cast(bf16x8_t, f1), acc[ky * 3 + 1], 0, 0, 0);
      acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          __builtin_bit_cast(bf16x8_t, yf), __builtin_bit_cast(bf16x8_t, f2), acc[ky * 3 + 2], 0, 0, 0);
    }
  };
#define TAPS_WAIT_LDS() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); \
                             __builtin_amdgcn_sched_barrier(0); } while (0)
  uint4 yA, yB;
  uint2 xA[3][3], xB[3][3];
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): steps 0 and 1 landed
  __builtin_amdgcn_s_barrier();
  if (nsteps > 0) read_frag(stage0, 0, yA, xA);
  // cur: step st, nxt: step st+1, far: held step st-1, receives step st+2
  auto step = [&](unsigned char* cur, unsigned char* nxt, unsigned char* far, int st) {
    TAPS_WAIT_LDS();
    read_frag(cur, 1, yB, xB);
    __builtin_amdgcn_sched_barrier(0);
    mfma9(yA, xA);
    TAPS_WAIT_LDS();
    read_frag(cur, 2, yA, xA);
    __builtin_amdgcn_sched_barrier(0);
    mfma9(yB, xB);
    TAPS_WAIT_LDS();
    read_frag(cur, 3, yB, xB);
    __builtin_amdgcn_sched_barrier(0);
    mfma9(yA, xA);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of step st+1 landed
    __builtin_amdgcn_s_barrier();         // ... everyone's did; every wave is done with `far`
    __builtin_amdgcn_sched_barrier(0);
    issue(fp, far, st + 2 < nsteps);
    advance(fp);
    mask_ahead(fp, st + 3 < nsteps);
    TAPS_WAIT_LDS();
    if (st + 1 < nsteps) read_frag(nxt, 0, yA, xA);
    __builtin_amdgcn_sched_barrier(0);
    mfma9(yB, xB);
  };
#undef TAPS_WAIT_LDS
  for (int st = 0; st < nsteps; st += 3) {
    step(stage0, stage1, stage2, st);
    if (st + 1 < nsteps) step(stage1, stage2, stage0, st + 1);
    if (st + 2 < nsteps) step(stage2, stage0, stage1, st + 2);
  }

  // acc[t][r]: ci = ci0 + iw*32 + l32, co = co0 + cw*32 + (r&3) + 8*(r>>2) + 4*half
  const int64_t K = (int64_t)9 * p.Cin;
  float* __restrict__ dw = p.dw + (int64_t)split * K * p.Cout;
  const int ci = ci0 + iw * 32 + l32;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* row = dw + ((int64_t)t * p.Cin + ci) * p.Cout + co0 + cw * 32 + half * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(row + g * 8) =
          make_float4(acc[t][g * 4], acc[t][g * 4 + 1], acc[t][g * 4 + 2], acc[t][g * 4 + 3]);
  }
}

// dy[px][cout] (cout <= 8) -> dst[px][8], zero padded (input of the thin tap-fused wgrad)
__global__ void __launch_bounds__(256)
pad_channels8_kernel(const uint16_t* __restrict__ src, int cout, int64_t px,
                     uint16_t* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < px; i += (int64_t)gridDim.x * 256) {
    uint16_t v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < cout ? src[i * cout + c] : (uint16_t)0;
    *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(v);
  }
}

// sum the split partials: out[i] (+)= sum_s part[s][i]
__global__ void __launch_bounds__(256)
wgrad_reduce_scalar_kernel(const float* __restrict__ part, int splits, int64_t n, int accumulate,
                           const float* __restrict__ out_scale, float* __restrict__ out) {
  const float sc = out_scale ? *out_scale : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // four independent partial sums keep the loads of a long split list in flight
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      s0 += part[(int64_t)k * n + i];
      s1 += part[(int64_t)(k + 1) * n + i];
      s2 += part[(int64_t)(k + 2) * n + i];
      s3 += part[(int64_t)(k + 3) * n + i];
    }
    for (; k < splits; ++k) s0 += part[(int64_t)k * n + i];
    float s = (s0 + s1) + (s2 + s3);
    s *= sc;
    out[i] = accumulate ? out[i] + s : s;
  }
}

// The same with 16-byte accesses (n % 4 == 0, 16-byte aligned slabs and output): the reduce is
// pure HBM traffic (108 MB for a 3x3 1024 -> 1024 layer with two slabs) and ran at a third of
// the bandwidth with dword loads (295 launches, 8.3 ms per step in profiles/r03_*).  Per element
// the summation order is the scalar kernel's, so the results are bit-identical.
__global__ void __launch_bounds__(256)
wgrad_reduce_vec_kernel(const float* __restrict__ part, int splits, int64_t n4, int accumulate,
                        const float* __restrict__ out_scale, float* __restrict__ out) {
  const float sc = out_scale ? *out_scale : 1.0f;
  const float4* __restrict__ P = reinterpret_cast<const float4*>(part);
  float4* __restrict__ O = reinterpret_cast<float4*>(out);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int k = 0;
    auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    for (; k + 3 < splits; k += 4) {
      const float4 a = P[(int64_t)k * n4 + i], b = P[(int64_t)(k + 1) * n4 + i];
      const float4 c = P[(int64_t)(k + 2) * n4 + i], d = P[(int64_t)(k + 3) * n4 + i];
      add(s0, a); add(s1, b); add(s2, c); add(s3, d);
    }
    for (; k < splits; ++k) add(s0, P[(int64_t)k * n4 + i]);
    float4 r = make_float4(((s0.x + s1.x) + (s2.x + s3.x)) * sc, ((s0.y + s1.y) + (s2.y + s3.y)) * sc,
                           ((s0.z + s1.z) + (s2.z + s3.z)) * sc, ((s0.w + s1.w) + (s2.w + s3.w)) * sc);
    if (accumulate) {
      const float4 o = O[i];
      r.x = o.x + r.x; r.y = o.y + r.y; r.z = o.z + r.z; r.w = o.w + r.w;
    }
    O[i] = r;
  }
}

// Round 4: the split reductions of MANY layers in one launch (se3ds_wgrad_reduce_multi): the
// weight-gradient kernels of a module leave their partial slabs in per-layer scratch and the module's
// reductions run as one table-driven launch when the backward pass has left the module (on the
// optimiser's side stream in the one-replica trainer) -- 295 bandwidth-sized launches per step leave
// the backward pass's critical path.  Table rows (device, int64 x 5): partial slabs, splits, n / 4,
// destination, first workgroup of the row; a workgroup owns 1024 float4 (four per thread, independent
// load chains).  Per element the summation order is wgrad_reduce_vec_kernel's: bit-identical.
constexpr int kRedTile = 1024;
__global__ void __launch_bounds__(256)
wgrad_reduce_multi_kernel(const int64_t* __restrict__ tab, int rows) {
  int lo = 0, hi = rows - 1;
  while (lo < hi) {   // last row whose first workgroup is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid * 5 + 4] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* R = tab + lo * 5;
  const float4* __restrict__ P = reinterpret_cast<const float4*>(R[0]);
  const int splits = (int)R[1];
  const int64_t n4 = R[2];
  float4* __restrict__ O = reinterpret_cast<float4*>(R[3]);
  const int64_t base = ((int64_t)blockIdx.x - R[4]) * kRedTile + threadIdx.x;
  auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  if (splits == 2) {   // (the common case: all four elements' loads in flight together)
    float4 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = base + j * 256;
      const int64_t q = i < n4 ? i : 0;
      a[j] = P[q];
      b[j] = P[n4 + q];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = base + j * 256;
      if (i >= n4) continue;
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
      add(s0, a[j]);
      add(s0, b[j]);
      O[i] = make_float4(((s0.x + s1.x) + (s2.x + s3.x)) * 1.0f, ((s0.y + s1.y) + (s2.y + s3.y)) * 1.0f,
                         ((s0.z + s1.z) + (s2.z + s3.z)) * 1.0f, ((s0.w + s1.w) + (s2.w + s3.w)) * 1.0f);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 256;
    if (i >= n4) continue;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      const float4 a = P[(int64_t)k * n4 + i], b = P[(int64_t)(k + 1) * n4 + i];
      const float4 c = P[(int64_t)(k + 2) * n4 + i], d = P[(int64_t)(k + 3) * n4 + i];
      add(s0, a); add(s1, b); add(s2, c); add(s3, d);
    }
    for (; k < splits; ++k) add(s0, P[(int64_t)k * n4 + i]);
    O[i] = make_float4(((s0.x + s1.x) + (s2.x + s3.x)) * 1.0f, ((s0.y + s1.y) + (s2.y + s3.y)) * 1.0f,
                       ((s0.z + s1.z) + (s2.z + s3.z)) * 1.0f, ((s0.w + s1.w) + (s2.w + s3.w)) * 1.0f);
  }
}

// dW[(ky,kx),ci,co] = tmp[(k-1-ky, k-1-kx), co, ci]: finishes the role-swapped weight gradient
// of thin-Cout stride-1 'same' convs (see se3ds_conv2d_wgrad_swapped in the header).
__global__ void __launch_bounds__(256)
wgrad_swap_fixup_kernel(const float* __restrict__ tmp, int k, int cin, int cout, int accumulate,
                        float* __restrict__ out) {
  const int total = k * k * cin * cout;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int co = i % cout;
    int ci = (i / cout) % cin;
    int t = i / (cout * cin);
    int ky = t / k, kx = t - ky * k;
    float v = tmp[(((int64_t)(k - 1 - ky) * k + (k - 1 - kx)) * cout + co) * cin + ci];
    out[i] = accumulate ? out[i] + v : v;
  }
}

// ------------------------------------------------------------------ thin-Cout 3x3 data gradient
// dx (Cin channels) from dy with <= 4 channels: K = 9 * Cout <= 36 reduction elements per output,
// 1 GB of dx to write per call -- store bound.  A workgroup covers 8 rows x 32 columns; the dy
// patch (10 x 34 pixels x Cout) and the transposed weights W'[ci][tap * Cout + co] sit in LDS,
// each wave builds the im2col fragments of its two rows element by element (16-bit LDS reads,
// a few dozen per lane) and writes through the LDS-transposed full-line epilogue.
constexpr int kThinDRows = 8, kThinCols = 32;
constexpr int kThinDCinMax = 256;
__global__ void __launch_bounds__(256)
thin_cout_dgrad_kernel(const IgemmParams p) {
  constexpr int NI = 2;   // 64 output channels per pass (32 accumulator registers per lane less than 128 channels per pass)
  constexpr int PH = kThinDRows + 2, PW = kThinCols + 2, KP = 48;   // K padded to 3 MFMA steps
  __shared__ __attribute__((aligned(16))) unsigned char scratch[4][kEpiScratch<NI>];
  __shared__ __attribute__((aligned(16))) uint16_t dys[PH * PW * 4];
  __shared__ __attribute__((aligned(16))) uint16_t wk[kThinDCinMax][KP + 8];   // (+8: bank spread)
  __shared__ int16_t koff[KP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int Co = p.sC, Ci = p.oC, K = 9 * Co;
  const int tiles_x = ceil_div(p.oW, kThinCols), tiles_y = ceil_div(p.oH, kThinDRows);
  const int64_t n_tiles = (int64_t)p.N * tiles_y * tiles_x;
  const uint16_t* __restrict__ dy = (const uint16_t*)p.src;
  const uint16_t* __restrict__ wn = (const uint16_t*)p.w;   // wn [(tap * Ci + ci)][Co]
  // element k = tap * Co + co reads the patch at (y + 2 - ky, x + 2 - kx, co)
  if (tid < KP) {
    int o = -1;
    if (tid < K) {
      const int tap = tid / Co, co = tid - tap * Co;
      const int ky = tap / 3, kx = tap - ky * 3;
      o = ((2 - ky) * PW + (2 - kx)) * Co + co;
    }
    koff[tid] = (int16_t)o;
  }
  // W'[ci][k] for every output channel (zero beyond K), loaded once per persistent workgroup
  for (int i = tid; i < Ci * KP; i += 256) {
    const int ci = i / KP, k = i - ci * KP;
    uint16_t v = 0;
    if (k < K) {
      const int tap = k / Co, co = k - tap * Co;
      v = wn[((int64_t)tap * Ci + ci) * Co + co];
    }
    wk[ci][k] = v;
  }
  // dy patch: row r <-> sy = oy0 + pad_t - 2 + r, column q <-> sx = ox0 + pad_l - 2 + q; the
  // next tile's patch (PH * PW * Cout <= 1360 values) is fetched into registers during the
  // current tile's MFMAs and stores
  constexpr int NV = (PH * PW * 4 + 255) / 256;
  const int total = PH * PW * Co;
  uint16_t nxt[NV];
  auto fetch = [&](int64_t tile) {
    int64_t b = tile;
    const int tx = (int)(b % tiles_x);
    b /= tiles_x;
    const int ty = (int)(b % tiles_y), n = (int)(b / tiles_y);
    const int oy0 = ty * kThinDRows, ox0 = tx * kThinCols;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int i = tid + u * 256;
      nxt[u] = 0;
      if (i < total) {
        const int pix = i / Co, c = i - pix * Co;
        const int r = pix / PW, q = pix - r * PW;
        const int sy = oy0 + p.pad_t - 2 + r;
        int sx = ox0 + p.pad_l - 2 + q;
        if (p.wrap_w) sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx);
        if (sy >= 0 && sy < p.sH && sx >= 0 && sx < p.sW)
          nxt[u] = dy[(((int64_t)n * p.sH + sy) * p.sW + sx) * Co + c];
      }
    }
  };
  if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  int64_t b = tile;
  const int tx = (int)(b % tiles_x);
  b /= tiles_x;
  const int ty = (int)(b % tiles_y), n = (int)(b / tiles_y);
  const int oy0 = ty * kThinDRows, ox0 = tx * kThinCols;
  __syncthreads();   // the previous tile's patch is consumed
#pragma unroll
  for (int u = 0; u < NV; ++u)
    if (tid + u * 256 < total) dys[tid + u * 256] = nxt[u];
  __syncthreads();   // patch (and, the first time, the weights) are in place
  if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
  for (int g = 0; g < Ci; g += NI * 32) {
    f32x16_t acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KP / 16; ++ks) {
      if (ks * 16 >= K) break;
      uint4 xf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int base = ((wave * 2 + j) * PW + l32) * Co;
        uint16_t e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int o = koff[ks * 16 + half * 8 + q];
          e[q] = o >= 0 ? dys[base + o] : (uint16_t)0;
        }
        xf[j] = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                           (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint4 wf = *reinterpret_cast<const uint4*>(&wk[g + i * 32 + l32][ks * 16 + half * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                              __builtin_bit_cast(bf16x8_t, xf[j]),
                                                              acc[i][j], 0, 0, 0);
      }
    }
    int64_t opix[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = oy0 + wave * 2 + j, ox = ox0 + l32;
      opix[j] = (oy < p.oH && ox < p.oW) ? ((int64_t)n * p.oH + oy) * p.oW + ox : -1;
    }
    store_wave_lds<NI>(p, acc, opix, g, lane, scratch[wave]);
  }
  }
}

// ------------------------------------------------------------------ thin data gradient, 4x4 stride 2
// dx (<= 4 channels) of the discriminator's first layer from dy (128 channels): 540 MB read,
// 67 MB written per call.  The four output parity classes each see 2 x 2 of the 16 taps; for a
// class the dy pixels under consecutive outputs are consecutive, so a class row is a plain
// "thin-Cout forward" over a dy patch in LDS: MFMA A = the <= 4 weight rows W[(tap, ci)][co],
// B = 16-byte patch reads.  Workgroup tile: 8 x 64 output pixels = 4 classes x 4 rows x 32
// columns; wave w takes class row w of every class.
constexpr int kThinSRows = 8, kThinSCols = 64, kThinSPH = 5, kThinSPW = 33;
__global__ void __launch_bounds__(256)
thin_s2_dgrad_kernel(const IgemmParams p) {
  constexpr int C = 128, PB = C * 2 + 16;   // dy channels, bytes per patch pixel
  __shared__ __attribute__((aligned(16))) unsigned char ds[kThinSPH * kThinSPW * PB];
  __shared__ __attribute__((aligned(16))) uint16_t wl[16 * 4 * C + C];   // wn [(tap * Ci + ci)][co] + a zero row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int Ci = p.oC;
  const int tiles_x = ceil_div(p.oW, kThinSCols), tiles_y = ceil_div(p.oH, kThinSRows);
  const int64_t n_tiles = (int64_t)p.N * tiles_y * tiles_x;
  const uint16_t* __restrict__ dy = (const uint16_t*)p.src;
  for (int i = tid; i < 16 * Ci * C / 8; i += 256)
    reinterpret_cast<uint4*>(wl)[i] = reinterpret_cast<const uint4*>(p.w)[i];
  // (the MFMA rows beyond Ci read the zero row instead of zeroing every fragment with v_cndmask)
  if (tid < C / 8) reinterpret_cast<uint4*>(wl + 16 * 4 * C)[tid] = make_uint4(0u, 0u, 0u, 0u);
  const float scale = p.scale ? *p.scale : 1.0f;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int64_t b = tile;
    const int tx = (int)(b % tiles_x);
    b /= tiles_x;
    const int ty = (int)(b % tiles_y), n = (int)(b / tiles_y);
    const int y0 = ty * kThinSRows, x0 = tx * kThinSCols;
    // first dy row / column any output of the tile reads (numerators y + pad - ky are even)
    const int oyb = (y0 + p.pad_t - 2) >> 1, oxb = (x0 + p.pad_l - 2) >> 1;
    __syncthreads();   // the previous tile's patch is consumed
    {
      constexpr int cpp = C / 8, total = kThinSPH * kThinSPW * cpp, kB = 6;
      for (int i0 = tid; i0 < total; i0 += kB * 256) {
        uint4 v[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int i = i0 + u * 256;
          v[u] = make_uint4(0u, 0u, 0u, 0u);
          if (i < total) {
            const int pix = i / cpp, c = i - pix * cpp;
            const int r = pix / kThinSPW, q = pix - r * kThinSPW;
            const int sy = oyb + r, sx = oxb + q;
            if (sy >= 0 && sy < p.sH && sx >= 0 && sx < p.sW)
              v[u] = *reinterpret_cast<const uint4*>(dy + (((int64_t)n * p.sH + sy) * p.sW + sx) * C + c * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int i = i0 + u * 256;
          if (i < total) {
            const int pix = i / cpp, c = i - pix * cpp;
            *reinterpret_cast<uint4*>(ds + pix * PB + c * 16) = v[u];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int py = cls >> 1, px = cls & 1;
      // taps of the class: ky = ky0, ky0 + 2 with (py + pad - ky0) even; the same for kx
      const int ky0 = (py + p.pad_t) & 1, kx0 = (px + p.pad_l) & 1;
      f32x16_t acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ky = ky0 + 2 * (t >> 1), kx = kx0 + 2 * (t & 1);
        // output (y0 + py + 2 a, x0 + px + 2 b), a = wave, b = l32 reads dy pixel
        // ((y + pad - ky) / 2, (x + pad - kx) / 2)
        const int r = ((y0 + py + 2 * wave + p.pad_t - ky) >> 1) - oyb;
        const int q = ((x0 + px + p.pad_l - kx) >> 1) - oxb;
        const unsigned char* xrow = ds + (r * kThinSPW + q + l32) * PB + half * 16;
        const uint16_t* wrow = (l32 < Ci ? wl + ((ky * 4 + kx) * Ci + l32) * C : wl + 16 * 4 * C) + half * 8;
#pragma unroll
        for (int kc = 0; kc < C / 16; ++kc) {
          const uint4 xf = *reinterpret_cast<const uint4*>(xrow + kc * 32);
          const uint4 wf = *reinterpret_cast<const uint4*>(wrow + kc * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                        __builtin_bit_cast(bf16x8_t, xf), acc, 0, 0,
                                                        0);
        }
      }
      const int oy = y0 + py + 2 * wave, ox = x0 + px + 2 * l32;
      if (half == 0 && oy < p.oH && ox < p.oW) {
        const int64_t o = (((int64_t)n * p.oH + oy) * p.oW + ox) * Ci;
        uint16_t* out = (uint16_t*)p.out + o;
        const uint16_t* add = p.addend ? (const uint16_t*)p.addend + o : nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < Ci) {
            float v = acc[c] * scale;
            if (add) v = bf16_to_f32(f32_to_bf16(v)) + bf16_to_f32(add[c]);
            out[c] = f32_to_bf16(v);
          }
      }
    }
  }
}

// ------------------------------------------------------------------ thin-Cin weight gradient
// dW[k][co] = sum over pixels of im2col(x)[px][k] * dy[px][co] for the same first layers
// (K = kh * kw * Cin <= 256, Cout = 128): 32 K outputs, hundreds of MB of dy to read.  The
// scalar-gather kernel ran them at 1.2 - 1.6 ms.  Persistent workgroups (one per CU, 8 waves)
// walk 8 x 32 pixel tiles: the dy tile (256 px x 128 ch) and the (masked) x patch go to LDS, both
// MFMA operands are gathered with 16-bit LDS reads (pixels are the reduction dimension, so both
// need 8 pixels per lane), the K/32 x 4 accumulator tiles stay in registers across all tiles of
// the workgroup and leave as one partial slab per workgroup for wgrad_reduce_kernel.
constexpr int kThinKMax = 256, kThinCinMax = 8;
constexpr int kThinWRows = 8, kThinWThreads = 512, kThinWDyStride = 128 * 2 + 8;   // bytes per dy pixel
struct ThinCinWgradParams {
  const uint16_t* x; const uint16_t* dy; float* part;
  int N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad_t, pad_l, wrap_w;
  const float* src_mask; const float* row_scale;
};
__host__ __device__ inline size_t thin_cin_wgrad_lds(int kh, int kw, int cin, int stride) {
  const int ph = (kThinWRows - 1) * stride + kh, pw = (kThinCols - 1) * stride + kw;
  return (size_t)kThinWRows * kThinCols * kThinWDyStride + ((size_t)ph * pw * cin * 2 + 15) / 16 * 16 +
         kThinKMax * 2;
}
__global__ void __launch_bounds__(kThinWThreads)
thin_cin_wgrad_kernel(const ThinCinWgradParams p) {
  constexpr int NT = kThinWThreads, PX = kThinWRows * kThinCols, DYS = kThinWDyStride;
  extern __shared__ __attribute__((aligned(16))) unsigned char tw_smem[];
  const int Ci = p.Cin, K = p.kh * p.kw * Ci, MB = (K + 31) / 32;
  const int st = p.stride;
  const int PH = (kThinWRows - 1) * st + p.kh, PW = (kThinCols - 1) * st + p.kw;
  unsigned char* dys = tw_smem;                                   // [PX][DYS]
  uint16_t* xs = reinterpret_cast<uint16_t*>(tw_smem + (size_t)PX * DYS);
  int16_t* koff = reinterpret_cast<int16_t*>(tw_smem + (size_t)PX * DYS +
                                             ((size_t)PH * PW * Ci * 2 + 15) / 16 * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  // (rows k >= K of the im2col operand read the pixel's first element: their accumulator rows are
  // never stored -- no per-element select in the gather)
  for (int k = tid; k < kThinKMax; k += NT) {
    int o = 0;
    if (k < K) {
      const int tap = k / Ci, ci = k - tap * Ci;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      o = (ky * PW + kx) * Ci + ci;
    }
    koff[k] = (int16_t)o;
  }
  __syncthreads();
  // accumulator tile i of this wave: t = wave + 8 i -> (m block t / 4, n block t % 4)
  const int nb = wave & 3;
  int ko[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mb = (wave + 8 * i) >> 2;
    ko[i] = mb < MB ? koff[mb * 32 + l32] : 0;
  }
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int tiles_x = ceil_div(p.Wo, kThinCols), tiles_y = ceil_div(p.Ho, kThinWRows);
  const int64_t n_tiles = (int64_t)p.N * tiles_y * tiles_x;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int64_t b = tile;
    const int tx = (int)(b % tiles_x);
    b /= tiles_x;
    const int ty = (int)(b % tiles_y), n = (int)(b / tiles_y);
    const int oy0 = ty * kThinWRows, ox0 = tx * kThinCols;
    __syncthreads();   // the previous tile's operands are consumed
    // dy tile (zero outside the image; optional per-pixel scale)
    {
      uint4 v[PX * 16 / NT];
#pragma unroll
      for (int u = 0; u < PX * 16 / NT; ++u) {
        const int i = tid + u * NT;
        const int px = i >> 4, c = i & 15;
        const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (oy < p.Ho && ox < p.Wo) {
          const int64_t op = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
          v[u] = *reinterpret_cast<const uint4*>(p.dy + op * 128 + c * 8);
          if (p.row_scale) {
            const float rs = p.row_scale[op];
            uint32_t* w = reinterpret_cast<uint32_t*>(&v[u]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float lo = __uint_as_float(w[q] << 16) * rs;
              const float hi = __uint_as_float(w[q] & 0xffff0000u) * rs;
              w[q] = pack2_bf16(lo, hi);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < PX * 16 / NT; ++u) {
        const int i = tid + u * NT;
        // (pixel rows are 8 bytes off 16-byte alignment: two 8-byte stores)
        uint2* d = reinterpret_cast<uint2*>(dys + (i >> 4) * DYS + (i & 15) * 16);
        d[0] = make_uint2(v[u].x, v[u].y);
        d[1] = make_uint2(v[u].z, v[u].w);
      }
    }
    // x patch, one pixel (<= 8 channels) per thread and trip
    {
      constexpr int kB = 2;
      const int npix = PH * PW;
      for (int q0 = tid; q0 < npix; q0 += kB * NT) {
        uint16_t v[kB][kThinCinMax];
        float mk[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int pix = q0 + u * NT;
          mk[u] = 1.0f;
#pragma unroll
          for (int c = 0; c < kThinCinMax; ++c) v[u][c] = 0;
          if (pix < npix) {
            const int r = pix / PW, q = pix - r * PW;
            const int sy = oy0 * st - p.pad_t + r;
            int sx = ox0 * st - p.pad_l + q;
            if (p.wrap_w) sx = sx < 0 ? sx + p.W : (sx >= p.W ? sx - p.W : sx);
            if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) {
              const int64_t sp = ((int64_t)n * p.H + sy) * p.W + sx;
              const uint16_t* px = p.x + sp * Ci;
#pragma unroll
              for (int c = 0; c < kThinCinMax; ++c)
                if (c < Ci) v[u][c] = px[c];
              if (p.src_mask) mk[u] = p.src_mask[sp];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int pix = q0 + u * NT;
          if (pix < npix) {
#pragma unroll
            for (int c = 0; c < kThinCinMax; ++c)
              if (c < Ci)
                xs[pix * Ci + c] = p.src_mask ? f32_to_bf16(bf16_to_f32(v[u][c]) * mk[u]) : v[u][c];
          }
        }
      }
    }
    __syncthreads();
    // 16 reduction steps of 16 pixels: step s covers row s / 2, columns (s & 1) * 16 .. + 15
    for (int sidx = 0; sidx < PX / 16; ++sidx) {
      const int row = sidx >> 1, col0 = (sidx & 1) * 16 + half * 8;
      // B: dy[px][co], 8 pixels of this lane's channel
      uint16_t eb[8];
      const unsigned char* dp = dys + (row * kThinCols + col0) * DYS + (nb * 32 + l32) * 2;
#pragma unroll
      for (int j = 0; j < 8; ++j) eb[j] = *reinterpret_cast<const uint16_t*>(dp + j * DYS);
      const uint4 bf = make_uint4((uint32_t)eb[0] | ((uint32_t)eb[1] << 16), (uint32_t)eb[2] | ((uint32_t)eb[3] << 16),
                                  (uint32_t)eb[4] | ((uint32_t)eb[5] << 16), (uint32_t)eb[6] | ((uint32_t)eb[7] << 16));
      const int xbase = (row * st * PW + col0 * st) * Ci;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (((wave + 8 * i) >> 2) >= MB) break;   // wave-uniform
        uint16_t ea[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ea[j] = xs[xbase + ko[i] + j * st * Ci];
        const uint4 af = make_uint4((uint32_t)ea[0] | ((uint32_t)ea[1] << 16), (uint32_t)ea[2] | ((uint32_t)ea[3] << 16),
                                    (uint32_t)ea[4] | ((uint32_t)ea[5] << 16), (uint32_t)ea[6] | ((uint32_t)ea[7] << 16));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af),
                                                         __builtin_bit_cast(bf16x8_t, bf), acc[i], 0, 0,
                                                         0);
      }
    }
  }
  // partial slab of this workgroup: part[blockIdx][k][co]
  float* slab = p.part + (int64_t)blockIdx.x * K * 128;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mb = (wave + 8 * i) >> 2;
    if (mb >= MB) break;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = mb * 32 + (e >> 2) * 8 + half * 4 + (e & 3);
      if (k < K) slab[(int64_t)k * 128 + nb * 32 + l32] = acc[i][e];
    }
  }
}

// ------------------------------------------------------------------ thin-Cout 3x3 weight gradient
// dW[(tap, ci)][co] for the heads (Cout <= 4, stride 1, 'same'): the mirror image of the kernel
// above -- dy is the thin operand (MFMA rows = output channels, only Cout of 32 populated), the
// 9 * Cin / 32 column tiles (tap, 32 input channels) are dealt to the 8 waves, the x patch
// (10 x 34 pixels, all channels) is staged once per 8 x 32 pixel tile and its fragments (8
// pixels of one channel per lane) come from transposing LDS reads (ds_read_b64_tr_b16), as in the
// tap-fused kernel.  1 GB of x per call.
struct ThinCoutWgradParams {
  const uint16_t* x; const uint16_t* dy; float* part;
  int N, H, W, Cin, Cout, pad;
};
constexpr int kThinQRows = 8, kThinQSlabs = 256;   // 8 x 32 pixel tiles, one 8-wave workgroup per CU
// (4-row tiles with two workgroups per CU measured slower: 895 vs 787 us)
__host__ __device__ inline int thin_cw_pixel_bytes(int cin) { return cin * 2 + 32; }
__host__ __device__ inline size_t thin_cout_wgrad_lds(int cin) {
  return (size_t)(kThinQRows + 2) * (kThinCols + 2) * thin_cw_pixel_bytes(cin) +
         (size_t)kThinQRows * kThinCols * 4 * 2;
}
// MAXT = column tiles per wave = ceil(9 * Cin / 32 / 8): 5 for the heads (Cin 128).  Every wave runs
// MAXT tiles per pixel slot -- a wave with one tile fewer repeats its last one into an accumulator
// nobody stores (it would otherwise wait at the tile's barrier) -- so the slot's body has no
// branches: all its transposing LDS reads are issued before the first MFMA.  (With a wave-uniform
// `break` per tile every MFMA waited for its own two reads: 80 serial LDS round trips per tile.)
template <int MAXT>
__global__ void __launch_bounds__(kThinWThreads)
thin_cout_wgrad_kernel(const ThinCoutWgradParams p) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_seg;
  constexpr int NT = kThinWThreads, PX = kThinQRows * kThinCols;
  constexpr int PH = kThinQRows + 2, PW = kThinCols + 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char tq_smem[];
  const int Ci = p.Cin, Co = p.Cout, PB = thin_cw_pixel_bytes(Ci);
  unsigned char* xs = tq_smem;                                                  // [PH * PW][PB]
  uint16_t* dys = reinterpret_cast<uint16_t*>(tq_smem + (size_t)PH * PW * PB);  // [PX][Co]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int g16 = (lane >> 4) & 1, i16 = lane & 15, jrow = i16 >> 2, qcol = i16 & 3;
  const int cblocks = Ci / 32, ntile = 9 * cblocks;
  f32x16_t acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // LDS offset of tile i's fragment reads relative to the pixel slot (tile = (tap, 32 input channels))
  int xoff[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    int t = wave + 8 * i;
    t = t < ntile ? t : ntile - 1;
    const int tap = t / cblocks, cb = t - tap * cblocks;
    const int ky = tap / 3, kx = tap - ky * 3;
    xoff[i] = (ky * PW + kx + half * 8 + jrow) * PB + (cb * 32 + g16 * 16 + qcol * 4) * 2;
  }
  const int tiles_x = ceil_div(p.W, kThinCols), tiles_y = ceil_div(p.H, kThinQRows);
  const int64_t n_tiles = (int64_t)p.N * tiles_y * tiles_x;
  // the next tile's x patch is fetched into registers while the current one is computed.  Thread ->
  // (pixel, 16-byte chunk) without runtime divisions where Cin / 8 is a power of two (the heads: 16):
  // chunk = tid % cpp, pixels tid / cpp + u * (NT / cpp); the patch row / column of a pixel is a
  // division by the constant PW.  (The generic form cost ~100 VALU instructions per 16-byte load,
  // 1 100 per tile and wave -- more than the tile's MFMAs.)
  constexpr int NV = (PH * PW * 16 + NT - 1) / NT;   // Cin <= 128: 16 chunks of 16 bytes per pixel
  const int cpp = Ci / 8, total = PH * PW * cpp;
  const bool pow2 = (cpp & (cpp - 1)) == 0;
  const int csh = 31 - __builtin_clz((unsigned)cpp);
  const int fc = tid & (cpp - 1), fp0 = tid >> csh, fpp = NT >> csh;
  auto chunk_of = [&](int u, int& pix, int& c) -> bool {   // false: nothing to do for this thread
    if (pow2) {
      pix = fp0 + u * fpp; c = fc;
      return pix < PH * PW;
    }
    const int i = tid + u * NT;
    pix = i / cpp; c = i - pix * cpp;
    return i < total;
  };
  const uint32_t utiles_x = (uint32_t)tiles_x, utiles_y = (uint32_t)tiles_y;
  auto tile_pos = [&](uint32_t t, int& n, int& oy0, int& ox0) {
    const uint32_t ty_n = t / utiles_x;
    ox0 = (int)(t - ty_n * utiles_x) * kThinCols;
    const uint32_t nn = ty_n / utiles_y;
    oy0 = (int)(ty_n - nn * utiles_y) * kThinQRows;
    n = (int)nn;
  };
  uint4 nxt[NV];
  // ... and so is its dy tile ([8 rows][32 * Co] values, rows contiguous in memory: <= 2 values per
  // thread; fetched synchronously it exposed one global round trip per tile, ~1 us of 7)
  constexpr int ND = (PX * 4 + NT - 1) / NT;
  uint16_t nxd[ND];
  const int rowlen = kThinCols * Co;
  auto fetch = [&](int64_t tile) {
    int n, oy0, ox0;
    tile_pos((uint32_t)tile, n, oy0, ox0);
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * NT;
      nxd[u] = 0;
      if (i < kThinQRows * rowlen) {
        const int ry = i / rowlen, e = i - ry * rowlen;
        const int oy = oy0 + ry, ox_e = ox0 * Co + e;   // element index inside the image row
        if (oy < p.H && ox_e < p.W * Co) nxd[u] = p.dy[((int64_t)n * p.H + oy) * p.W * Co + ox_e];
      }
    }
    const int sy0 = oy0 - p.pad, sx0 = ox0 - p.pad;
    const bool interior = sy0 >= 0 && sy0 + PH <= p.H && sx0 >= 0 && sx0 + PW <= p.W;   // (uniform)
    const int64_t base = (((int64_t)n * p.H + sy0) * p.W + sx0) * Ci;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int pix, c;
      nxt[u] = make_uint4(0u, 0u, 0u, 0u);
      if (chunk_of(u, pix, c)) {
        const int r = pix / PW, q = pix - r * PW;
        if (interior || ((unsigned)(sy0 + r) < (unsigned)p.H && (unsigned)(sx0 + q) < (unsigned)p.W))
          nxt[u] = *reinterpret_cast<const uint4*>(p.x + (base + (int64_t)((r * p.W + q) * Ci + c * 8)));
      }
    }
  };
  if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int n, oy0, ox0;
    tile_pos((uint32_t)tile, n, oy0, ox0);
    __syncthreads();   // the previous tile's operands are consumed
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * NT;
      if (i < kThinQRows * rowlen) {
        const int ry = i / rowlen, e = i - ry * rowlen;
        // stored channel-major [Co][PX]: a lane's 8 consecutive pixels of one output channel are one
        // 16-byte LDS read (pixel-major needed eight 2-byte reads and their packing per fragment)
        int cx, c;
        switch (Co) {
          case 1: cx = e; c = 0; break;
          case 2: cx = e >> 1; c = e & 1; break;
          case 3: cx = e / 3; c = e - cx * 3; break;
          default: cx = e >> 2; c = e & 3; break;
        }
        dys[c * PX + ry * kThinCols + cx] = nxd[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      int pix, c;
      if (chunk_of(u, pix, c)) *reinterpret_cast<uint4*>(xs + pix * PB + c * 16) = nxt[u];
    }
    __syncthreads();
    if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
#pragma unroll 2
    for (int sidx = 0; sidx < PX / 16; ++sidx) {
      const int row = sidx >> 1, cs = (sidx & 1) * 16;
      // A: dy[co][px], 8 pixels of this lane's output channel (rows co >= Cout are zero)
      uint4 af = make_uint4(0u, 0u, 0u, 0u);
      if (l32 < Co)
        af = *reinterpret_cast<const uint4*>(dys + l32 * PX + row * kThinCols + cs + half * 8);
      // B: x[pixel][ci], pixels (row + ky, cs + kx + half * 8 .. + 7), channel cb * 32 + l32:
      // address lane (4 j + q) -> pixel + j, channels 4 q .. 4 q + 3 of the 16-channel group
      const unsigned char* sp = xs + (row * PW + cs) * PB;
      uint4 bf[MAXT];
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        const unsigned char* xp = sp + xoff[i];
        const uint2 b0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)xp));
        const uint2 b1 = __builtin_bit_cast(
            uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_seg)(xp + 4 * PB)));
        bf[i] = make_uint4(b0.x, b0.y, b1.x, b1.y);
      }
#pragma unroll
      for (int i = 0; i < MAXT; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af),
                                                         __builtin_bit_cast(bf16x8_t, bf[i]), acc[i], 0,
                                                         0, 0);
    }
  }
  // lanes of half 0 hold co 0..3 of column (tap, cb * 32 + l32) in acc[.][0..3]
  float* slab = p.part + (int64_t)blockIdx.x * 9 * Ci * Co;
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = wave + 8 * i;
      if (t >= ntile) break;
      const int tap = t / cblocks, cb = t - tap * cblocks;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < Co) slab[((int64_t)tap * Ci + cb * 32 + l32) * Co + c] = acc[i][c];
    }
  }
}

// ------------------------------------------------------------------ thin-Cin forward
// First layers (5 -> 128 7x7 s2 of the generator, 4 -> 128 4x4 s2 of the discriminator): K =
// kh * kw * Cin <= 256 reduction elements, hundreds of MB of output -- store bound, but the
// scalar-gather fallback ran them at 1.1 / 0.7 ms.  One persistent workgroup per CU keeps the
// zero-padded weight rows W[co][k] of 128 output channels in LDS; per 16 x 32 output tile it loads
// the (masked) input patch once and each of the 8 waves builds the im2col fragments of its two
// rows with 16-bit LDS gathers through an offset table; full-line stores through the LDS
// epilogue (two 64-channel halves per wave to keep its scratch small).  LDS is sized per layer.
constexpr int kThinFRows = 16, kThinFThreads = 512;
struct ThinCinLds {
  int kp, ws, ph, pw;
  size_t xs_off, wk_off, koff_off, bytes;
};
__host__ __device__ inline ThinCinLds thin_cin_lds(int kh, int kw, int cin, int stride) {
  ThinCinLds l;
  l.kp = (kh * kw * cin + 15) / 16 * 16;
  l.ws = l.kp + 8;
  l.ph = (kThinFRows - 1) * stride + kh;
  l.pw = (kThinCols - 1) * stride + kw;
  l.xs_off = (size_t)(kThinFThreads / 64) * kEpiScratch<2>;
  l.wk_off = l.xs_off + ((size_t)l.ph * l.pw * cin * 2 + 15) / 16 * 16;
  l.koff_off = l.wk_off + (size_t)128 * l.ws * 2;
  l.bytes = l.koff_off + (size_t)l.kp * 2;
  return l;
}
__global__ void __launch_bounds__(kThinFThreads)
thin_cin_fwd_kernel(const IgemmParams p) {
  constexpr int NI = 4, NT = kThinFThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
  const ThinCinLds L = thin_cin_lds(p.kh, p.kw, p.sC, p.stride);
  unsigned char* scratch = tc_smem;
  uint16_t* xs = reinterpret_cast<uint16_t*>(tc_smem + L.xs_off);
  uint16_t* wk = reinterpret_cast<uint16_t*>(tc_smem + L.wk_off);
  int16_t* koff = reinterpret_cast<int16_t*>(tc_smem + L.koff_off);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int Ci = p.sC, K = p.kh * p.kw * Ci, KP = L.kp, WS = L.ws;
  const int st = p.stride;
  const int PH = L.ph, PW = L.pw;
  const int tiles_x = ceil_div(p.oW, kThinCols), tiles_y = ceil_div(p.oH, kThinFRows);
  const int64_t n_tiles = (int64_t)p.N * tiles_y * tiles_x;
  const uint16_t* __restrict__ src = (const uint16_t*)p.src;
  const uint16_t* __restrict__ wt = (const uint16_t*)p.w;   // wt [oC][K], k = (ky * kw + kx) * Ci + ci
  // (padding elements k >= K read the pixel's first element: their weights are zero, the product
  // vanishes -- no per-element select in the gather)
  for (int k = tid; k < KP; k += NT) {
    int o = 0;
    if (k < K) {
      const int tap = k / Ci, ci = k - tap * Ci;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      o = (ky * PW + kx) * Ci + ci;
    }
    koff[k] = (int16_t)o;
  }
  auto load_weights = [&](int g) {
    for (int i = tid; i < NI * 32 * KP; i += NT) {
      const int co = i / KP, k = i - co * KP;
      wk[co * WS + k] = k < K ? wt[(int64_t)(g + co) * K + k] : (uint16_t)0;
    }
  };
  const bool one_group = p.oC == NI * 32;
  if (one_group) load_weights(0);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int64_t b = tile;
    const int tx = (int)(b % tiles_x);
    b /= tiles_x;
    const int ty = (int)(b % tiles_y), n = (int)(b / tiles_y);
    const int oy0 = ty * kThinFRows, ox0 = tx * kThinCols;
    __syncthreads();   // the previous tile's patch is consumed
    {
      // one pixel (<= 8 channels) per thread and trip, two pixels in flight
      constexpr int kB = 2;
      const int npix = PH * PW;
      for (int q0 = tid; q0 < npix; q0 += kB * NT) {
        uint16_t v[kB][kThinCinMax];
        float mk[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int pix = q0 + u * NT;
          mk[u] = 1.0f;
#pragma unroll
          for (int c = 0; c < kThinCinMax; ++c) v[u][c] = 0;
          if (pix < npix) {
            const int r = pix / PW, q = pix - r * PW;
            const int sy = oy0 * st - p.pad_t + r;
            int sx = ox0 * st - p.pad_l + q;
            if (p.wrap_w) sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx);
            if (sy >= 0 && sy < p.sH && sx >= 0 && sx < p.sW) {
              const int64_t sp = ((int64_t)n * p.sH + sy) * p.sW + sx;
              const uint16_t* px = src + sp * Ci;
#pragma unroll
              for (int c = 0; c < kThinCinMax; ++c)
                if (c < Ci) v[u][c] = px[c];
              if (p.src_mask) mk[u] = p.src_mask[sp];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int pix = q0 + u * NT;
          if (pix < npix) {
#pragma unroll
            for (int c = 0; c < kThinCinMax; ++c)
              if (c < Ci)
                xs[pix * Ci + c] = p.src_mask ? f32_to_bf16(bf16_to_f32(v[u][c]) * mk[u]) : v[u][c];
          }
        }
      }
    }
    for (int g = 0; g < p.oC; g += NI * 32) {
      if (!one_group) {
        __syncthreads();
        load_weights(g);
      }
      __syncthreads();
      f32x16_t acc[NI][2];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      const int base0 = ((wave * 2) * st * PW + l32 * st) * Ci;
      const int base1 = base0 + st * PW * Ci;
      for (int ks = 0; ks < KP; ks += 16) {
        uint4 xf[2];
        int o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = koff[ks + half * 8 + q];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int base = j ? base1 : base0;
          uint16_t e[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) e[q] = xs[base + o[q]];
          xf[j] = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                             (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const uint4 wf = *reinterpret_cast<const uint4*>(&wk[(i * 32 + l32) * WS + ks + half * 8]);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                                __builtin_bit_cast(bf16x8_t, xf[j]),
                                                                acc[i][j], 0, 0, 0);
        }
      }
      int64_t opix[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int oy = oy0 + wave * 2 + j, ox = ox0 + l32;
        opix[j] = (oy < p.oH && ox < p.oW) ? ((int64_t)n * p.oH + oy) * p.oW + ox : -1;
      }
      unsigned char* scr = scratch + wave * kEpiScratch<2>;
      store_wave_lds<2>(p, *reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), opix, g, lane, scr);
      store_wave_lds<2>(p, *reinterpret_cast<f32x16_t(*)[2][2]>(&acc[2]), opix, g + 64, lane, scr);
    }
  }
}

// ------------------------------------------------------------------ thin-Cout 3x3 forward
// Stride-1 3x3 convolutions onto <= 4 output channels (the generator's RGB / depth heads at full
// resolution): 1 GB of input and 10-30 GFLOP of real work per call, HBM bound.  Through the
// 128-channel MFMA tiles they ran at 1.75 ms (42x padded FLOPs).  Here a workgroup stages the
// (4+2) x (32+2) pixel input patch of a 4 x 32 output tile in LDS once (all Cin channels), the
// <= 4 weight rows next to it, and each of its 4 waves runs one output row through 32x32x16
// MFMAs whose A operand has only those rows populated: 9 * Cin / 16 MFMAs per wave, patch
// overhead 1.6x (mostly L2 hits), two workgroups per CU so that one fills while the other
// computes.
constexpr int kThinRows = 4, kThinPH = kThinRows + 2, kThinPW = kThinCols + 2;
__host__ __device__ inline int thin_pixel_bytes(int cin) { return cin * 2 + 16; }
__host__ __device__ inline int thin_weight_bytes(int cin) { return 9 * cin * 2 + 16; }
__host__ __device__ inline size_t thin_lds_bytes(int cin, int cout) {
  // (+ one all-zero weight row: what the MFMA rows beyond Cout read)
  return (size_t)kThinPH * kThinPW * thin_pixel_bytes(cin) + (size_t)(cout + 1) * thin_weight_bytes(cin);
}

__global__ void __launch_bounds__(256)
thin_cout_fwd_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char t_smem[];
  const int C = p.sC, K = 9 * C;
  const int pstride = thin_pixel_bytes(C), wstride = thin_weight_bytes(C);
  unsigned char* xs = t_smem;                                    // [PH * PW][pstride]
  unsigned char* ws = t_smem + kThinPH * kThinPW * pstride;      // [oC][wstride]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int tiles_x = ceil_div(p.oW, kThinCols), tiles_y = ceil_div(p.oH, kThinRows);
  const int64_t n_tiles = (int64_t)p.N * tiles_y * tiles_x;
  const uint16_t* __restrict__ src = (const uint16_t*)p.src;
  {
    const uint16_t* __restrict__ w = (const uint16_t*)p.w;   // wt [oC][K]
    const int chunks = K / 8;
    for (int i = tid; i < p.oC * chunks; i += 256) {
      const int co = i / chunks, c = i - co * chunks;
      *reinterpret_cast<uint4*>(ws + co * wstride + c * 16) =
          *reinterpret_cast<const uint4*>(w + (int64_t)co * K + c * 8);
    }
    // row oC: zeros.  The MFMA's A operand has 32 rows and only oC <= 4 carry weights; the other lanes
    // read this row (one address: a broadcast) instead of zeroing their fragments with four
    // v_cndmask per read -- that was 570 of the 670 VALU instructions per tile and wave
    for (int i = tid; i < chunks; i += 256)
      *reinterpret_cast<uint4*>(ws + p.oC * wstride + i * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  // Persistent workgroups; the next tile's patch is fetched into registers (NV 16-byte chunks
  // per thread, Cin <= 128) while the current one is computed, and parked in LDS afterwards.
  constexpr int NV = (kThinPH * kThinPW * 16 + 255) / 256;
  // thread -> (pixel, 16-byte chunk): chunk = tid % cpp, pixels tid / cpp + u * (256 / cpp) -- Cin is
  // 64 or 128 here, so cpp is a power of two and nothing in the loop divides by a runtime value
  // (the generic form cost ~100 VALU instructions per load, twice the tile's MFMA cycles)
  const int cpp = C / 8;   // 16-byte chunks per pixel
  const int csh = 31 - __builtin_clz((unsigned)cpp);
  const int fc = tid & (cpp - 1), fp0 = tid >> csh, fpp = 256 >> csh;
  const uint32_t utiles_x = (uint32_t)tiles_x, utiles_y = (uint32_t)tiles_y;
  auto tile_pos = [&](uint32_t t, int& n, int& oy0, int& ox0) {
    const uint32_t ty_n = t / utiles_x;
    ox0 = (int)(t - ty_n * utiles_x) * kThinCols;
    const uint32_t nn = ty_n / utiles_y;
    oy0 = (int)(ty_n - nn * utiles_y) * kThinRows;
    n = (int)nn;
  };
  uint4 nxt[NV];
  auto fetch = [&](int64_t tile) {
    int n, oy0, ox0;
    tile_pos((uint32_t)tile, n, oy0, ox0);
    const int sy0 = oy0 - p.pad_t, sx0 = ox0 - p.pad_l;
    const bool interior = sy0 >= 0 && sy0 + kThinPH <= p.sH && sx0 >= 0 && sx0 + kThinPW <= p.sW;
    const int64_t rowbase = ((int64_t)n * p.sH + sy0) * p.sW;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int pix = fp0 + u * fpp;
      nxt[u] = make_uint4(0u, 0u, 0u, 0u);
      if (pix < kThinPH * kThinPW) {
        const int r = pix / kThinPW, q = pix - r * kThinPW;
        if (interior) {
          nxt[u] = *reinterpret_cast<const uint4*>(
              src + ((rowbase + sx0) * C + (int64_t)((r * p.sW + q) * C + fc * 8)));
        } else {
          int sx = sx0 + q;
          if (p.wrap_w) sx = sx < 0 ? sx + p.sW : (sx >= p.sW ? sx - p.sW : sx);
          if ((unsigned)(sy0 + r) < (unsigned)p.sH && (unsigned)sx < (unsigned)p.sW)
            nxt[u] = *reinterpret_cast<const uint4*>(
                src + ((rowbase + (int64_t)r * p.sW + sx) * C + fc * 8));
        }
      }
    }
  };
  if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  int n, oy0, ox0;
  tile_pos((uint32_t)tile, n, oy0, ox0);
  __syncthreads();   // the previous tile's fragments are read (and, the first time, ws is written)
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int pix = fp0 + u * fpp;
    if (pix < kThinPH * kThinPW) *reinterpret_cast<uint4*>(xs + pix * pstride + fc * 16) = nxt[u];
  }
  __syncthreads();
  if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int kc_n = C / 16;
  const bool wrow_ok = l32 < p.oC;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const unsigned char* xrow = xs + ((wave + ky) * kThinPW + l32 + kx) * pstride + half * 16;
    const unsigned char* wrow = ws + (wrow_ok ? l32 : p.oC) * wstride + tap * C * 2 + half * 16;
    for (int kc = 0; kc < kc_n; kc += 4) {   // Cin % 64 == 0: four fragments per trip
      uint4 xf[4], wf[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xf[u] = *reinterpret_cast<const uint4*>(xrow + (kc + u) * 32);
        wf[u] = *reinterpret_cast<const uint4*>(wrow + (kc + u) * 32);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[u]),
                                                      __builtin_bit_cast(bf16x8_t, xf[u]), acc, 0,
                                                      0, 0);
    }
  }
  // lanes 0..31 hold output channels 0..3 of pixel (oy0 + wave, ox0 + l32) in acc[0..3]
  const int oy = oy0 + wave, ox = ox0 + l32;
  if (half == 0 && oy < p.oH && ox < p.oW) {
    const float scale = p.scale ? *p.scale : 1.0f;
    uint16_t* out = (uint16_t*)p.out + (((int64_t)n * p.oH + oy) * p.oW + ox) * p.oC;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < p.oC) {
        float t = acc[c] * scale;
        if (p.bias) t = t + p.bias[c];
        if (p.act == 1) t = t > 0.f ? t : 0.f;
        else if (p.act == 2) t = t > 0.f ? t : t * p.act_alpha;
        out[c] = f32_to_bf16(t);
      }
  }
  }
}

// ------------------------------------------------------------------------ weight prep
// fp32 master HWIO [K][Cout] -> compute-dtype copies: wt [Cout][K] (forward operand) and,
// optionally, wn [K][Cout] (input-gradient operand).  32x32 LDS-tiled transpose.
template <typename T>
__global__ void __launch_bounds__(256)
weight_prep_kernel(const float* __restrict__ w, int64_t K, int Cout, T* __restrict__ wt,
                   T* __restrict__ wn) {
  __shared__ float tile[32][33];
  const int64_t k0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int64_t k = k0 + r;
    int c = c0 + tx;
    float v = (k < K && c < Cout) ? w[k * Cout + c] : 0.0f;
    tile[r][tx] = v;
    if (wn && k < K && c < Cout) wn[k * Cout + c] = TT<T>::from_f(v);
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r;
    int64_t k = k0 + tx;
    if (c < Cout && k < K) wt[(int64_t)c * K + k] = TT<T>::from_f(tile[tx][r]);
  }
}

// bf16, Cout % 4 == 0 and K % 8 == 0 (every layer of the bench but the thin heads): 64 x 64 tiles,
// 16-byte reads of the fp32 master, 8-byte stores of wn and 16-byte stores of the transposed wt
// (the 32 x 32 kernel above stores single bf16 values: 2 bytes per lane).
__global__ void __launch_bounds__(256)
weight_prep_vec_kernel(const float* __restrict__ w, int64_t K, int Cout, uint16_t* __restrict__ wt,
                       uint16_t* __restrict__ wn) {
  __shared__ float tile[64][65];
  const int64_t k0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
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
        if (wn) {
          uint16_t o[4] = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
          *reinterpret_cast<uint2*>(wn + k * Cout + c) = *reinterpret_cast<const uint2*>(o);
        }
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

// se3ds_conv2d_wgrad_partial: the calling host thread's next split reduction is not launched but
// described in this row [slabs, splits, n / 4, destination, 1] (when the 16-byte kernel applies)
static thread_local int64_t* t_defer_row = nullptr;

static void launch_wgrad_reduce(const float* part, int splits, int64_t n, int accumulate,
                                const float* out_scale, float* out, hipStream_t s) {
  if (t_defer_row != nullptr && !accumulate && out_scale == nullptr && (n % 4) == 0 &&
      (((uintptr_t)part | (uintptr_t)out) & 15) == 0) {
    t_defer_row[0] = (int64_t)(uintptr_t)part;
    t_defer_row[1] = splits;
    t_defer_row[2] = n / 4;
    t_defer_row[3] = (int64_t)(uintptr_t)out;
    t_defer_row[4] = 1;
    t_defer_row = nullptr;
    return;
  }
  if ((n % 4) == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, part,
                       splits, n / 4, accumulate, out_scale, out);
  else
    hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, part,
                       splits, n, accumulate, out_scale, out);
}

int fill_classes(IgemmParams& p, int mode, int bm = BM) {
  int total = 0;
  if (mode == MODE_FWD) {
    p.n_classes = 1;
    p.cls_py[0] = p.cls_px[0] = 0;
    p.cls_tile_start[0] = 0;
    int64_t m = (int64_t)p.N * p.oH * p.oW;
    total = (int)ceil_div(m, bm);
    p.cls_tile_start[1] = total;
    p.fd_hw[0] = make_fastdiv((uint32_t)(p.oH * p.oW));
    p.fd_w[0] = make_fastdiv((uint32_t)p.oW);
    return total;
  }
  const int s = p.stride;
  p.n_classes = s * s;
  int c = 0;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px, ++c) {
      p.cls_py[c] = py; p.cls_px[c] = px;
      p.cls_tile_start[c] = total;
      int cH = (p.oH - py + s - 1) / s, cW = (p.oW - px + s - 1) / s;
      if (cH < 0) cH = 0;
      if (cW < 0) cW = 0;
      p.fd_hw[c] = make_fastdiv((uint32_t)(cH * cW));
      p.fd_w[c] = make_fastdiv((uint32_t)cW);
      total += (int)ceil_div((int64_t)p.N * cH * cW, bm);
    }
  p.cls_tile_start[c] = total;
  return total;
}

}  // namespace
}  // namespace se3ds

using namespace se3ds;


// SE3DS_NO_THIN=1: the thin-layer kernels (Cin <= 8 forward / weight gradient, Cout <= 4 forward /
// data / weight gradient, the 4x4 stride-2 data gradient into <= 4 channels) are skipped and their
// layers take the general implicit-GEMM kernels (read per call: tests/test_nets_gpu.py compares the
// two routings on the production heads and stems).
static bool thin_on() { return getenv("SE3DS_NO_THIN") == nullptr; }

// SE3DS_BIG_TILE: unset = cost model below, 0 = never, 1 = whenever the shape allows (read per
// call so that tests can switch it).
static int big_tile_mode() {
  const char* e = getenv("SE3DS_BIG_TILE");
  return e ? atoi(e) : -1;
}

// Output channels per 256-pixel macro tile (256 / 128) or 0 for the 128 x 128 kernel.  Measured
// (tools/conv_bench.py): the 256-channel macro tile runs one workgroup per CU at ~1.25x the
// per-CU rate of two resident 128 x 128 tiles, so it wins unless its coarser grid leaves CUs
// idle in the last round; the 128-channel variant does not beat the 128 x 128 kernel and is
// only taken when forced (tests).
// SE3DS_HALO_TILE: unset = heuristic, 0 = never, 1 = whenever the shape allows, 128 / 256 = force
// that channel width.
static int halo_tile_channels(const IgemmParams& p) {
  const char* e = getenv("SE3DS_HALO_TILE");
  const int mode = e ? atoi(e) : -1;
  if (mode == 0) return 0;
  if ((p.oC % 128) != 0) return 0;
  if (mode == 128 || (p.oC % 256) != 0) return 128;
  if (mode == 256) return 256;
  // 256 channels per workgroup unless that leaves a large part of the chip without work
  const int64_t tiles = (int64_t)p.N * ceil_div(p.oH, 8) * ceil_div(p.oW, 32);
  const int64_t items256 = tiles * (p.oC / 256);
  const double eff256 = (double)items256 / (double)(ceil_div(items256, (int64_t)256) * 256);
  const int64_t items128 = items256 * 2;
  const double eff128 = 0.85 * (double)items128 / (double)(ceil_div(items128, (int64_t)256) * 256);
  return eff256 >= eff128 ? 256 : 128;
}

// SE3DS_HALO_M16: the LDS-DMA forward / data-gradient kernels (halo, 256-pixel macro tile, 128 x 128)
// and the tap-fused weight gradient with v_mfma_f32_16x16x32_bf16 (default; 0 = 32x32x16; read per call).  Step A/B on one box, 256-channel tile only: 204.3 / 202.6 -> 200.2 / 200.0 ms.
static bool halo_m16() {
  const char* e = getenv("SE3DS_HALO_M16");
  return e ? atoi(e) != 0 : true;
}

static int big_tile_channels(const IgemmParams& p, int mode) {
  const int g_big_tile = big_tile_mode();
  if (g_big_tile == 0) return 0;
  const int co = (p.oC % 256) == 0 ? 256 : ((p.oC % 128) == 0 ? 128 : 0);
  if (!co) return 0;
  if (g_big_tile == 1) return co;
  int64_t m = 0;   // pixel rows over all parity classes
  const int s = mode == MODE_FWD ? 1 : p.stride;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px)
      m += (int64_t)p.N * ((p.oH - py + s - 1) / s) * ((p.oW - px + s - 1) / s);
  const double kCus = 256.0;
  const double big_items = (double)ceil_div(m, (int64_t)256) * (p.oC / co);
  const double small_items = (double)ceil_div(m, (int64_t)128) * ceil_div(p.oC, 128);
  if (co != 256) return 0;
  const double gain = 1.25;
  // time in units of one 128 x 128 tile on a CU that holds two of them
  const double t_small = std::ceil(small_items / (2 * kCus)) * 2.0;
  const double t_big = std::ceil(big_items / kCus) * (co == 256 ? 4.0 : 2.0) / gain;
  return t_big < t_small ? co : 0;
}

// Rows of the [rows][2][oC] column-sum array the routed FORWARD kernel emits from its epilogue
// (one row per 64-pixel wave tile), or 0 when that kernel cannot (fp32, scalar gather, ragged
// channel tiles).  Must follow conv_common's routing order: halo, 256-pixel macro tile, 128 x 128.
static int64_t fwd_stats_rows(const IgemmParams& p, int dtype, int stride, int kh, int kw,
                              bool glds, int mode = MODE_FWD) {
  if (!glds || dtype != SE3DS_BF16) return 0;
  // data gradient (fused batch-norm backward statistics): stride 1 only -- one parity class,
  // the same tile geometry as a forward pass over the gradient's pixels
  if (mode == MODE_DGRAD && stride != 1) return 0;
  const int64_t M = (int64_t)p.N * p.oH * p.oW;
  if (stride == 1 && kh == 3 && kw == 3 && halo_tile_channels(p))
    return (int64_t)p.N * ceil_div(p.oH, 8) * ceil_div(p.oW, 32) * 4;
  if (big_tile_channels(p, mode)) return ceil_div(M, (int64_t)256) * 4;
  if ((p.oC % BN) == 0) return ceil_div(M, (int64_t)BM) * (BM / 64);
  return 0;
}

extern "C" {

static int conv_common(int mode, const void* src, const void* w, void* out, int dtype, int n,
                       int h, int wdt, int cin, int ho, int wo, int cout, int kh, int kw,
                       int stride, int pad_t, int pad_l, int wrap_w, const float* src_mask,
                       int mask_binary, const float* scale, const float* bias, const float* row_a,
                       const float* row_b, int act, float act_alpha, void* stream,
                       float* stats = nullptr, const void* addend = nullptr,
                       const void* bn_x = nullptr, const uint8_t* bn_mask = nullptr,
                       const float* bn_mean = nullptr, const float* bn_rstd = nullptr,
                       int bn_act = 0, float bn_alpha = 0.f, int64_t o_pitch = 0, int64_t o_off = 0,
                       int bias_mod = 0) {
  if (n <= 0 || h <= 0 || wdt <= 0 || cin <= 0 || ho <= 0 || wo <= 0 || cout <= 0 || kh <= 0 ||
      kw <= 0 || stride <= 0 || stride > 2)
    return SE3DS_E_BADSHAPE;
  // circular W padding: any stride for the forward gather; the transposed gather only for the
  // stride-1 'same width' case (inference-mode backward through strided convs is not needed)
  if (wrap_w && mode == MODE_DGRAD && (stride != 1 || wo != wdt)) return SE3DS_E_UNSUPPORTED;
  if (dtype != SE3DS_F32 && dtype != SE3DS_BF16) return SE3DS_E_BADDTYPE;
  // (pixel lists are indexed with 32 bits: FastDiv)
  if ((int64_t)n * h * wdt >= ((int64_t)1 << 31) || (int64_t)n * ho * wo >= ((int64_t)1 << 31))
    return SE3DS_E_BADSHAPE;
  IgemmParams p;
  p.src = src; p.w = w; p.out = out;
  p.N = n; p.kh = kh; p.kw = kw; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.wrap_w = wrap_w; p.src_mask = src_mask; p.mask_binary = mask_binary; p.scale = scale;
  p.bias = bias; p.row_a = row_a;
  p.row_b = row_b; p.act = act; p.act_alpha = act_alpha;
  const int bk = dtype == SE3DS_F32 ? 16 : 32;
  const int64_t K = (int64_t)kh * kw * cin;
  if (mode == MODE_FWD) {
    p.sH = h; p.sW = wdt; p.sC = cin; p.oH = ho; p.oW = wo; p.oC = cout;
    p.w_tap = cin; p.w_n = K;                       // wt [Cout][K]
  } else {
    p.sH = ho; p.sW = wo; p.sC = cout; p.oH = h; p.oW = wdt; p.oC = cin;
    p.w_tap = (int64_t)cin * cout; p.w_n = cout;    // wn [K][Cout]
  }
  p.vec = (p.sC % bk) == 0;
  p.o_pitch = o_pitch > 0 ? o_pitch : p.oW;
  p.o_off = o_off;
  p.bias_mod = bias_mod;
  // (the thin / halo kernels compute their own output pixels: the pitched form takes the generic ones)
  const bool pitched = o_pitch > 0 || bias_mod > 0;
#ifdef SE3DS_PROBE
  { const char* e = getenv("SE3DS_PROBE_MODE"); p.probe = e ? atoi(e) : 0; }
#endif
  p.stats = nullptr;
  p.addend = addend;
  hipStream_t s = as_stream(stream);
  const bool glds = (p.sC % (2 * bk)) == 0 && (src_mask == nullptr || mask_binary);
  if (stats != nullptr && !(fwd_stats_rows(p, dtype, stride, kh, kw, glds, mode) > 0))
    return SE3DS_E_UNSUPPORTED;   // callers ask se3ds_conv2d_{fwd_stats,dgrad_bnstats}_rows first
  if (stats != nullptr && mode == MODE_DGRAD &&
      (bn_x == nullptr || bn_mean == nullptr || bn_rstd == nullptr || (p.oC & 7) != 0))
    return SE3DS_E_UNSUPPORTED;
  p.stats = stats;
  p.bn_x = (mode == MODE_DGRAD && stats != nullptr) ? (const uint16_t*)bn_x : nullptr;
  p.bn_mask = bn_mask; p.bn_mean = bn_mean; p.bn_rstd = bn_rstd;
  p.bn_act = bn_act; p.bn_alpha = bn_alpha;
  if (mode == MODE_DGRAD && dtype == SE3DS_BF16 && kh == 4 && kw == 4 && stride == 2 && cin <= 4 &&
      cout == 128 && src_mask == nullptr && row_a == nullptr && bias == nullptr && act == 0 &&
      !wrap_w && (pad_t == 0 || pad_t == 2) && (pad_l == 0 || pad_l == 2) &&
      thin_on()) {
    int64_t blocks = (int64_t)p.N * ceil_div(p.oH, kThinSRows) * ceil_div(p.oW, kThinSCols);
    if (blocks > 2 * 256) blocks = 2 * 256;   // persistent, two workgroups per CU
    hipLaunchKernelGGL(thin_s2_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return check_launch("conv2d_dgrad(thin s2)");
  }
  if (mode == MODE_FWD && dtype == SE3DS_BF16 && cin <= kThinCinMax && (cout % 128) == 0 &&
      kh * kw * cin <= kThinKMax && kh <= 7 && kw <= 7 && stats == nullptr && addend == nullptr &&
      !pitched && thin_on()) {
    const ThinCinLds L = thin_cin_lds(kh, kw, cin, stride);
    static size_t lds_set = 0;
    if (L.bytes > lds_set) {
      if (hipFuncSetAttribute((const void*)thin_cin_fwd_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes) != hipSuccess)
        return SE3DS_E_LAUNCH;
      lds_set = L.bytes;
    }
    int64_t blocks = (int64_t)p.N * ceil_div(p.oH, kThinFRows) * ceil_div(p.oW, kThinCols);
    const int64_t resident = 256;   // persistent: one 8-wave workgroup per CU (register bound)
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(thin_cin_fwd_kernel, dim3((unsigned)blocks), dim3(kThinFThreads), L.bytes, s,
                       p);
    return check_launch("conv2d_fwd(thin cin)");
  }
  if (mode == MODE_DGRAD && dtype == SE3DS_BF16 && kh == 3 && kw == 3 && stride == 1 && cout <= 4 &&
      (cin % 64) == 0 && cin <= kThinDCinMax && src_mask == nullptr && row_a == nullptr &&
      thin_on()) {
    int64_t blocks = (int64_t)p.N * ceil_div(p.oH, kThinDRows) * ceil_div(p.oW, kThinCols);
    // persistent: two workgroups per CU (236 registers per lane: the third does not fit, and a grid
    // of three per CU ran its last third at half occupancy)
    if (blocks > 2 * 256) blocks = 2 * 256;
    hipLaunchKernelGGL(thin_cout_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return check_launch("conv2d_dgrad(thin)");
  }
  if (mode == MODE_FWD && dtype == SE3DS_BF16 && kh == 3 && kw == 3 && stride == 1 && cout <= 4 &&
      (cin % 64) == 0 && cin <= 128 && src_mask == nullptr && row_a == nullptr &&
      stats == nullptr && addend == nullptr && thin_on()) {
    const size_t lds = thin_lds_bytes(cin, cout);
    static size_t lds_set = 0;
    if (lds > lds_set) {
      if (hipFuncSetAttribute((const void*)thin_cout_fwd_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SE3DS_E_LAUNCH;
      lds_set = lds;
    }
    int64_t blocks = (int64_t)p.N * ceil_div(p.oH, kThinRows) * ceil_div(p.oW, kThinCols);
    if (blocks > 2 * 256) blocks = 2 * 256;   // persistent: two workgroups per CU
    hipLaunchKernelGGL(thin_cout_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, s, p);
    return check_launch("conv2d_fwd(thin)");
  }
  if (glds && dtype == SE3DS_BF16 && stride == 1 && kh == 3 && kw == 3) {
    const int co = halo_tile_channels(p);
    if (co) {
      p.halo_ty = ceil_div(p.oH, 8);
      p.halo_tx = ceil_div(p.oW, 32);
      const int64_t items = (int64_t)p.N * p.halo_ty * p.halo_tx * (p.oC / co);
      dim3 grid((unsigned)(items < 256 ? items : 256));   // persistent: one workgroup per CU
      if (co == 256 && halo_m16()) {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_halo_kernel<MODE_FWD, 256, 2, false, true>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 256, 2, true, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 256, 2, false, true>), grid, dim3(512), 0, s, p);
      } else if (co == 256) {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_halo_kernel<MODE_FWD, 256, 2>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 256, 2, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 256, 2>), grid, dim3(512), 0, s, p);
      } else {
        if (halo_m16()) {
          if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_halo_kernel<MODE_FWD, 128, 3, false, true>), grid, dim3(512), 0, s, p);
          else if (p.bn_x) hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 128, 3, true, true>), grid, dim3(512), 0, s, p);
          else hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 128, 3, false, true>), grid, dim3(512), 0, s, p);
        } else if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_halo_kernel<MODE_FWD, 128, 3>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 128, 3, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_halo_kernel<MODE_DGRAD, 128, 3>), grid, dim3(512), 0, s, p);
      }
      return check_launch(mode == MODE_FWD ? "conv2d_fwd(halo)" : "conv2d_dgrad(halo)");
    }
  }
  if (glds && dtype == SE3DS_BF16) {
    const int co = big_tile_channels(p, mode);
    if (co) {
      const int tiles = fill_classes(p, mode, 256);
      if (tiles <= 0) return SE3DS_OK;
      dim3 grid((unsigned)tiles, (unsigned)(p.oC / co));
      if (co == 256 && halo_m16()) {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_big_kernel<MODE_FWD, 256, false, true>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 256, true, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 256, false, true>), grid, dim3(512), 0, s, p);
      } else if (co == 256) {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_big_kernel<MODE_FWD, 256>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 256, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 256>), grid, dim3(512), 0, s, p);
      } else if (halo_m16()) {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_big_kernel<MODE_FWD, 128, false, true>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 128, true, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 128, false, true>), grid, dim3(512), 0, s, p);
      } else {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_big_kernel<MODE_FWD, 128>), grid, dim3(512), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 128, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((igemm_big_kernel<MODE_DGRAD, 128>), grid, dim3(512), 0, s, p);
      }
      return check_launch(mode == MODE_FWD ? "conv2d_fwd(big)" : "conv2d_dgrad(big)");
    }
  }
  int tiles = fill_classes(p, mode);
  if (tiles <= 0) return SE3DS_OK;
  dim3 grid((unsigned)tiles, (unsigned)ceil_div(p.oC, BN));
  if (glds) {
    if (dtype == SE3DS_F32) {
      if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_glds_kernel<float, MODE_FWD>), grid, dim3(kThreads), 0, s, p);
      else hipLaunchKernelGGL((igemm_glds_kernel<float, MODE_DGRAD>), grid, dim3(kThreads), 0, s, p);
    } else {
      // (the 16x16x32 epilogue has no ragged channel tiles)
      if ((p.oC % BN) == 0 && halo_m16()) {
        if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_glds_kernel<uint16_t, MODE_FWD, false, true>), grid, dim3(kThreads), 0, s, p);
        else if (p.bn_x) hipLaunchKernelGGL((igemm_glds_kernel<uint16_t, MODE_DGRAD, true, true>), grid, dim3(kThreads), 0, s, p);
        else hipLaunchKernelGGL((igemm_glds_kernel<uint16_t, MODE_DGRAD, false, true>), grid, dim3(kThreads), 0, s, p);
      } else if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_glds_kernel<uint16_t, MODE_FWD>), grid, dim3(kThreads), 0, s, p);
      else if (p.bn_x) hipLaunchKernelGGL((igemm_glds_kernel<uint16_t, MODE_DGRAD, true>), grid, dim3(kThreads), 0, s, p);
      else hipLaunchKernelGGL((igemm_glds_kernel<uint16_t, MODE_DGRAD>), grid, dim3(kThreads), 0, s, p);
    }
    return check_launch(mode == MODE_FWD ? "conv2d_fwd(glds)" : "conv2d_dgrad(glds)");
  }
  if (dtype == SE3DS_F32) {
    if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_kernel<float, MODE_FWD>), grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((igemm_kernel<float, MODE_DGRAD>), grid, dim3(kThreads), 0, s, p);
  } else {
    if (mode == MODE_FWD) hipLaunchKernelGGL((igemm_kernel<uint16_t, MODE_FWD>), grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((igemm_kernel<uint16_t, MODE_DGRAD>), grid, dim3(kThreads), 0, s, p);
  }
  return check_launch(mode == MODE_FWD ? "conv2d_fwd" : "conv2d_dgrad");
}

int se3ds_conv2d_fwd(const void* x, const void* wt, void* y, int dtype, int n, int h, int w,
                     int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                     int pad_l, int wrap_w, const float* in_mask, int in_mask_binary,
                     const float* scale, const float* bias, const float* row_a,
                     const float* row_b, int act, float act_alpha, void* stream) {
  return conv_common(MODE_FWD, x, wt, y, dtype, n, h, w, cin, ho, wo, cout, kh, kw, stride, pad_t,
                     pad_l, wrap_w, in_mask, in_mask_binary, scale, bias, row_a, row_b, act,
                     act_alpha, stream);
}

int64_t se3ds_conv2d_fwd_stats_rows(int dtype, int n, int cin, int ho, int wo, int cout, int kh,
                                    int kw, int stride, int has_in_mask, int in_mask_binary) {
  if (dtype != SE3DS_BF16 || stride < 1 || stride > 2 || (cin % 64) != 0 ||
      (has_in_mask && !in_mask_binary))
    return 0;
  // SE3DS_FUSED_BN_STATS: 0 = never, 1 = only the halo-resident 3x3 kernel, default = every
  // LDS-DMA kernel
  const char* e = getenv("SE3DS_FUSED_BN_STATS");
  const int lvl = e ? atoi(e) : 2;
  if (lvl == 0) return 0;
  IgemmParams p;
  p.N = n; p.oH = ho; p.oW = wo; p.oC = cout; p.stride = stride;
  if (lvl == 1 && !(stride == 1 && kh == 3 && kw == 3 && halo_tile_channels(p))) return 0;
  return fwd_stats_rows(p, dtype, stride, kh, kw, true);
}

int se3ds_conv_transpose2x2_fwd(const void* x, const void* wn, void* y, int dtype, int n, int hi, int wi,
                                int cin, int cout, const float* bias, void* stream) {
  // y (n, 2 hi, 2 wi, cout) = Conv2DTranspose(k 2, stride 2)(x (n, hi, wi, cin)); wn = the operand copy
  // in the Keras kernel's own layout (ky, kx, cout, cin).  Two 1x1 forward convolutions, one per
  // output row parity, with the 2 * cout "channels" (kx, co): see IgemmParams::o_pitch.
  if (cout % 4 != 0) return SE3DS_E_UNSUPPORTED;
  const size_t esz = dtype == SE3DS_F32 ? 4 : 2;
  for (int py = 0; py < 2; ++py) {
    const char* wpy = (const char*)wn + (size_t)py * 2 * cout * cin * esz;
    const int rc = conv_common(MODE_FWD, x, wpy, y, dtype, n, hi, wi, cin, hi, wi, 2 * cout, 1, 1, 1, 0, 0,
                               0, nullptr, 0, nullptr, bias, nullptr, nullptr, 0, 0.f, stream, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 2 * (int64_t)wi,
                               (int64_t)py * wi, cout);
    if (rc != SE3DS_OK) return rc;
  }
  return SE3DS_OK;
}

int se3ds_conv2d_fwd_stats(const void* x, const void* wt, void* y, int dtype, int n, int h, int w,
                           int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                           int pad_t, int pad_l, int wrap_w, const float* in_mask,
                           int in_mask_binary, const float* scale, const float* bias,
                           const float* row_a, const float* row_b, int act, float act_alpha,
                           float* stats, void* stream) {
  return conv_common(MODE_FWD, x, wt, y, dtype, n, h, w, cin, ho, wo, cout, kh, kw, stride, pad_t,
                     pad_l, wrap_w, in_mask, in_mask_binary, scale, bias, row_a, row_b, act,
                     act_alpha, stream, stats);
}

int se3ds_conv2d_dgrad(const void* dy, const void* wn, void* dx, int dtype, int n, int h, int w,
                       int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int wrap_w, const float* dy_row_scale, const float* scale,
                       const float* bias, const float* row_a, int act, float act_alpha,
                       void* stream) {
  return conv_common(MODE_DGRAD, dy, wn, dx, dtype, n, h, w, cin, ho, wo, cout, kh, kw, stride,
                     pad_t, pad_l, wrap_w, dy_row_scale, 0, scale, bias, row_a, nullptr, act,
                     act_alpha, stream);
}

int se3ds_conv2d_dgrad_acc(const void* dy, const void* wn, void* dx, int dtype, int n, int h, int w,
                           int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                           int pad_l, int wrap_w, const float* dy_row_scale, const float* scale,
                           const float* bias, const float* row_a, int act, float act_alpha,
                           const void* addend, void* stream) {
  return conv_common(MODE_DGRAD, dy, wn, dx, dtype, n, h, w, cin, ho, wo, cout, kh, kw, stride,
                     pad_t, pad_l, wrap_w, dy_row_scale, 0, scale, bias, row_a, nullptr, act,
                     act_alpha, stream, nullptr, addend);
}

// Data gradient with FUSED batch-norm backward statistics: dx (+ addend, may be null) is the
// gradient of y = act(norm(bn_x) [+ res]); besides dx the epilogue emits stats[rows][2][cin] =
// per-64-pixel-tile (sum dz, sum dz * xhat), dz = dx * act'(y) (bn_mask: one "y > 0" bit per
// element or null), xhat = (bn_x - bn_mean) * bn_rstd, which se3ds_norm_reduce_rows turns into
// the sums se3ds_norm_bwd_stats would have produced from a second pass over dx and bn_x.
// rows == 0: this shape / routing cannot (strided, fp32, thin or ragged channel tiles).
int64_t se3ds_conv2d_dgrad_bnstats_rows(int dtype, int n, int h, int w, int cin, int cout, int kh,
                                        int kw, int stride, int has_row_scale) {
  if (dtype != SE3DS_BF16 || stride != 1 || (cout % 64) != 0 || (cin % 8) != 0 || has_row_scale)
    return 0;
  IgemmParams p;
  p.N = n; p.oH = h; p.oW = w; p.oC = cin; p.stride = stride;
  return fwd_stats_rows(p, dtype, stride, kh, kw, true, MODE_DGRAD);
}

int se3ds_conv2d_dgrad_bnstats(const void* dy, const void* wn, void* dx, int dtype, int n, int h,
                               int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                               int pad_t, int pad_l, int wrap_w, const float* scale,
                               const float* row_a, const void* addend, const void* bn_x,
                               const uint8_t* bn_mask, const float* bn_mean, const float* bn_rstd,
                               int bn_act, float bn_alpha, float* stats, void* stream) {
  if (stats == nullptr) return SE3DS_E_UNSUPPORTED;
  return conv_common(MODE_DGRAD, dy, wn, dx, dtype, n, h, w, cin, ho, wo, cout, kh, kw, stride,
                     pad_t, pad_l, wrap_w, nullptr, 0, scale, nullptr, row_a, nullptr, 0, 0.f,
                     stream, stats, addend, bn_x, bn_mask, bn_mean, bn_rstd, bn_act, bn_alpha);
}

static int wgrad_splits(int64_t L, int64_t tiles, int64_t nel) {
  // Work items = tiles x splits, equal cost; ~512 run concurrently (2 per CU).  Cost model
  // (microseconds): rounds of items x steps per item x time per 64-pixel step, plus the write +
  // read of the fp32 partial slabs by the deterministic reduce.  Pick the cheapest split count.
  const double kStepUs = 2.15, kBytesPerUs = 3.0e6, kSlots = 512.0;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 256; ++s) {
    int64_t per = (L + s - 1) / s;
    if (s > 1 && per < 256) break;
    double steps = (double)((per + 63) / 64);
    double rounds = (double)((tiles * s + (int64_t)kSlots - 1) / (int64_t)kSlots);
    double cost = rounds * steps * kStepUs + (s > 1 ? 5.0 : 0.0) +
                  (double)s * (double)nel * 8.0 / kBytesPerUs * (s > 1 ? 1.0 : 0.5);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

// CUs a weight-gradient launch may count on.  (With the two decoders on two streams a 128-item
// launch of one branch shares the chip with the other branch's; counting on 128 was measured
// 221.1 / 221.4 ms per step against 222.7 / 221.5 with 256 on one box in round 3 -- the launches of
// the two branches do not pair up reliably: all 256.)
static constexpr int wgrad_slots() { return 256; }

// tap-fused 3x3 kernel: steps of 64 pixels, work items = (Cin/64) x (Cout/128) x splits on 256
// single-workgroup CUs.  Returns the split count (0: shape not eligible).
static int wgrad_taps_splits(int n, int ho, int wo, int cin, int cout, int kh, int kw, int* steps) {
  if (kh != 3 || kw != 3 || (cin % 64) != 0 || (cout % 128) != 0) return 0;
  const int64_t total = (int64_t)n * ceil_div(ho, 2) * ceil_div(wo, 32);
  if (total > (1 << 30)) return 0;
  *steps = (int)total;
  const int64_t tiles = (int64_t)(cin / 64) * (cout / 128);
  // rounds of 256 items x steps per item (~1.3 us each) + partial-slab traffic of the reduce
  const double kStepUs = 1.3, kBytesPerUs = 3.0e6;
  const int64_t nel = (int64_t)9 * cin * cout;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 512 && s <= total; ++s) {
    const double per = (double)ceil_div(total, (int64_t)s);
    if (s > 1 && per < 8) break;
    const double rounds = (double)ceil_div(tiles * s, (int64_t)wgrad_slots());
    const double cost = rounds * (per * kStepUs + 4.0) +
                        (double)s * (double)nel * 8.0 / kBytesPerUs * (s > 1 ? 1.0 : 0.5);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

static bool thin_cin_wgrad_ok(int cin, int cout, int kh, int kw) {
  return cin <= kThinCinMax && cout == 128 && kh * kw * cin <= kThinKMax && kh <= 7 && kw <= 7 &&
         thin_on();
}

size_t se3ds_conv2d_wgrad_workspace_bytes(int n, int ho, int wo, int cin, int cout, int kh,
                                          int kw) {
  int tsteps = 0;
  const int tsplits = wgrad_taps_splits(n, ho, wo, cin, cout, kh, kw, &tsteps);
  const size_t taps_bytes = sizeof(float) * (size_t)tsplits * (size_t)kh * kw * cin * cout + 16;
  int64_t L = (int64_t)n * ho * wo;
  int64_t row_tiles = cin <= 16 ? ceil_div((int64_t)kh * kw * cin, 128)
                                : (int64_t)kh * kw * ceil_div(cin, 128);
  int64_t tiles = row_tiles * ceil_div(cout, 128);
  int splits = wgrad_splits(L, tiles, (int64_t)kh * kw * cin * cout);
  size_t bytes = sizeof(float) * (size_t)splits * (size_t)kh * kw * cin * cout + 16;
  if (thin_cin_wgrad_ok(cin, cout, kh, kw)) {   // one partial slab per persistent workgroup
    const size_t thin = sizeof(float) * (size_t)256 * (size_t)kh * kw * cin * cout + 16;
    bytes = bytes > thin ? bytes : thin;
  }
  return bytes > taps_bytes ? bytes : taps_bytes;
}

int se3ds_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int n, int h, int w,
                       int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int wrap_w, const float* in_mask, int in_mask_binary,
                       const float* row_scale, const float* out_scale, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || ho <= 0 || wo <= 0 || cout <= 0 || kh <= 0 ||
      kw <= 0 || stride <= 0)
    return SE3DS_E_BADSHAPE;
  if (dtype != SE3DS_F32 && dtype != SE3DS_BF16) return SE3DS_E_BADDTYPE;
  if (workspace_bytes < se3ds_conv2d_wgrad_workspace_bytes(n, ho, wo, cin, cout, kh, kw))
    return SE3DS_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  if (dtype == SE3DS_BF16 && stride == 1 && row_scale == nullptr &&
      (in_mask == nullptr || in_mask_binary)) {
    int tsteps = 0;
    const int tsplits = wgrad_taps_splits(n, ho, wo, cin, cout, kh, kw, &tsteps);
    if (tsplits > 0) {
      WgradTapsParams q;
      q.x = (const uint16_t*)x; q.H = h; q.W = w; q.Cin = cin;
      q.dy = (const uint16_t*)dy; q.Ho = ho; q.Wo = wo; q.Cout = cout;
      q.N = n; q.pad_t = pad_t; q.pad_l = pad_l; q.wrap_w = wrap_w; q.src_mask = in_mask;
      q.dw = (float*)workspace;
      q.steps_y = ceil_div(ho, 2); q.steps_x = ceil_div(wo, 32);
      q.total_steps = tsteps;
      q.steps_per_split = ceil_div(tsteps, tsplits);
      q.dy_cstride = cout; q.co_valid = cout;
      dim3 tgrid((unsigned)(cin / 64), (unsigned)(cout / 128), (unsigned)tsplits);
      if (!wrap_w)
        if (halo_m16()) hipLaunchKernelGGL(wgrad_taps3_kernel<true>, tgrid, dim3(512), 0, s, q);
        else hipLaunchKernelGGL(wgrad_taps3_kernel<false>, tgrid, dim3(512), 0, s, q);
      else
        hipLaunchKernelGGL(wgrad_taps_kernel, tgrid, dim3(512), 0, s, q);
      const int64_t tnel = (int64_t)9 * cin * cout;
      launch_wgrad_reduce((const float*)workspace, tsplits, tnel, accumulate, out_scale, dw, s);
      return check_launch("conv2d_wgrad(taps)");
    }
  }
  if (dtype == SE3DS_BF16 && stride <= 2 && thin_cin_wgrad_ok(cin, cout, kh, kw)) {
    ThinCinWgradParams q;
    q.x = (const uint16_t*)x; q.dy = (const uint16_t*)dy; q.part = (float*)workspace;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.Ho = ho; q.Wo = wo; q.Cout = cout; q.kh = kh;
    q.kw = kw; q.stride = stride; q.pad_t = pad_t; q.pad_l = pad_l; q.wrap_w = wrap_w;
    q.src_mask = in_mask; q.row_scale = row_scale;
    const size_t lds = thin_cin_wgrad_lds(kh, kw, cin, stride);
    static size_t lds_set = 0;
    if (lds > lds_set) {
      if (hipFuncSetAttribute((const void*)thin_cin_wgrad_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SE3DS_E_LAUNCH;
      lds_set = lds;
    }
    int64_t blocks = (int64_t)n * ceil_div(ho, kThinWRows) * ceil_div(wo, kThinCols);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(thin_cin_wgrad_kernel, dim3((unsigned)blocks), dim3(kThinWThreads), lds, s, q);
    const int64_t tnel = (int64_t)kh * kw * cin * cout;
    launch_wgrad_reduce((const float*)workspace, (int)blocks, tnel, accumulate, out_scale, dw, s);
    return check_launch("conv2d_wgrad(thin cin)");
  }
  WgradParams p;
  p.x = x; p.H = h; p.W = w; p.Cin = cin; p.dy = dy; p.Ho = ho; p.Wo = wo; p.Cout = cout;
  p.N = n; p.kh = kh; p.kw = kw; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.wrap_w = wrap_w; p.src_mask = in_mask; p.mask_binary = in_mask_binary; p.row_scale = row_scale;
  p.dw = (float*)workspace;
  p.ci_tiles = (int)ceil_div(cin, 128);
  p.linear_k = cin <= 16 ? 1 : 0;
  const int64_t L = (int64_t)n * ho * wo;
  const int64_t row_tiles = p.linear_k ? ceil_div((int64_t)kh * kw * cin, 128)
                                       : (int64_t)kh * kw * p.ci_tiles;
  const int64_t tiles = row_tiles * ceil_div(cout, 128);
  p.splits = wgrad_splits(L, tiles, (int64_t)kh * kw * cin * cout);
  p.l_per_split = ceil_div(ceil_div(L, p.splits), WG_BL) * WG_BL;
  dim3 grid((unsigned)row_tiles, (unsigned)ceil_div(cout, 128), (unsigned)p.splits);
  const int epc = dtype == SE3DS_F32 ? 4 : 8;
  const bool glds = !p.linear_k && (in_mask == nullptr || in_mask_binary) &&
                    row_scale == nullptr &&
                    (cin % epc) == 0 && (cout % epc) == 0;
  if (glds) {
    p.l_per_split = ceil_div(ceil_div(L, p.splits), 64) * 64;
    if (dtype == SE3DS_F32) hipLaunchKernelGGL(wgrad_glds_kernel<float>, grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL(wgrad_glds_kernel<uint16_t>, grid, dim3(kThreads), 0, s, p);
  } else if (dtype == SE3DS_F32) {
    hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(kThreads), 0, s, p);
  } else {
    hipLaunchKernelGGL(wgrad_kernel<uint16_t>, grid, dim3(kThreads), 0, s, p);
  }
  const int64_t nel = (int64_t)kh * kw * cin * cout;
  launch_wgrad_reduce((const float*)workspace, p.splits, nel, accumulate, out_scale, dw, s);
  return check_launch("conv2d_wgrad");
}

int se3ds_conv2d_wgrad_partial(const void* x, const void* dy, float* dw, int dtype, int n, int h,
                               int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                               int pad_t, int pad_l, int wrap_w, const float* in_mask,
                               int in_mask_binary, const float* row_scale, void* workspace,
                               size_t workspace_bytes, int64_t* reduce_row, void* stream) {
  if (reduce_row == nullptr) return SE3DS_E_BADSHAPE;
  reduce_row[4] = 0;
  t_defer_row = reduce_row;
  const int rc = se3ds_conv2d_wgrad(x, dy, dw, dtype, n, h, w, cin, ho, wo, cout, kh, kw, stride,
                                    pad_t, pad_l, wrap_w, in_mask, in_mask_binary, row_scale,
                                    nullptr, 0, workspace, workspace_bytes, stream);
  t_defer_row = nullptr;
  return rc;
}

int se3ds_wgrad_reduce_multi(const int64_t* table, int rows, int64_t workgroups, void* stream) {
  if (rows <= 0 || workgroups <= 0) return SE3DS_OK;
  if (workgroups >= ((int64_t)1 << 31)) return SE3DS_E_BADSHAPE;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)workgroups), dim3(256), 0,
                     as_stream(stream), table, rows);
  return check_launch("wgrad_reduce_multi");
}

int se3ds_wgrad_reduce_tile(void) { return kRedTile; }

// Host-side evaluation of the division the conv kernels use for tile row -> (image, row, column)
// (FastDiv: multiplier and shifts made by make_fastdiv, quotient formed exactly as fd_div forms it
// on the device): the CPU suite checks it against n / d without a GPU.
uint32_t se3ds_fastdiv_host(uint32_t n, uint32_t d) {
  const FastDiv f = make_fastdiv(d);
  const uint32_t t = (uint32_t)(((uint64_t)n * f.m) >> 32);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// thin-Cout layers through the tap-fused kernel (padded dy copy): split count, 0 = not eligible
static int wgrad_taps_thin_splits(int n, int h, int w, int cin, int cout, int k, int* steps) {
  if (k != 3 || cout > 8 || (cin % 64) != 0) return 0;
  const int64_t total = (int64_t)n * ceil_div(h, 2) * ceil_div(w, 32);
  if (total > (1 << 30)) return 0;
  *steps = (int)total;
  const int64_t tiles = cin / 64;
  int64_t sp = ceil_div((int64_t)256, tiles);   // one round of the 256 CUs
  if (sp > total / 8) sp = total / 8;
  if (sp < 1) sp = 1;
  return (int)sp;
}

size_t se3ds_conv2d_wgrad_swapped_workspace_bytes(int n, int h, int w, int cin, int cout, int k) {
  const size_t a = se3ds_conv2d_wgrad_workspace_bytes(n, h, w, cout, cin, k, k) +
                   sizeof(float) * (size_t)k * k * cin * cout + 64;
  int steps = 0;
  const int sp = wgrad_taps_thin_splits(n, h, w, cin, cout, k, &steps);
  const size_t b = sp ? (size_t)n * h * w * 16 + 256 + sizeof(float) * (size_t)sp * 9 * cin * cout + 64
                      : 0;
  // thin_cout_wgrad_kernel: one partial slab per persistent workgroup
  const size_t c = (k == 3 && cout <= 4) ? sizeof(float) * (size_t)kThinQSlabs * 9 * cin * cout + 64 : 0;
  const size_t ab = a > b ? a : b;
  return ab > c ? ab : c;
}

int se3ds_conv2d_wgrad_swapped(const void* x, const void* dy, float* dw, int dtype, int n, int h,
                               int w, int cin, int cout, int k, int pad, int accumulate,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (workspace_bytes < se3ds_conv2d_wgrad_swapped_workspace_bytes(n, h, w, cin, cout, k))
    return SE3DS_E_WORKSPACE;
  if (dtype == SE3DS_BF16 && k == 3 && cout <= 4 && (cin % 32) == 0 && cin <= 128 &&
      workspace_bytes >= sizeof(float) * (size_t)kThinQSlabs * 9 * cin * cout &&
      thin_on()) {
    hipStream_t s = as_stream(stream);
    ThinCoutWgradParams q;
    q.x = (const uint16_t*)x; q.dy = (const uint16_t*)dy; q.part = (float*)workspace;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.Cout = cout; q.pad = pad;
    const size_t lds = thin_cout_wgrad_lds(cin);
    int64_t blocks = (int64_t)n * ceil_div(h, kThinQRows) * ceil_div(w, kThinCols);
    if (blocks > kThinQSlabs) blocks = kThinQSlabs;
    // column tiles per wave: 9 * Cin / 32 tiles dealt to 8 waves
    const int tiles_per_wave = ceil_div(9 * (cin / 32), 8);
#define SE3DS_THIN_CW(M)                                                                          \
  do {                                                                                            \
    static size_t lds_set = 0;                                                                    \
    if (lds > lds_set) {                                                                          \
      if (hipFuncSetAttribute((const void*)thin_cout_wgrad_kernel<M>,                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return SE3DS_E_LAUNCH;                                                                    \
      lds_set = lds;                                                                              \
    }                                                                                             \
    hipLaunchKernelGGL(thin_cout_wgrad_kernel<M>, dim3((unsigned)blocks), dim3(kThinWThreads), lds, \
                       s, q);                                                                     \
  } while (0)
    switch (tiles_per_wave) {
      case 2: SE3DS_THIN_CW(2); break;
      case 3: SE3DS_THIN_CW(3); break;
      case 4: SE3DS_THIN_CW(4); break;
      default: SE3DS_THIN_CW(5); break;
    }
#undef SE3DS_THIN_CW
    const int64_t tnel = (int64_t)9 * cin * cout;
    launch_wgrad_reduce((const float*)workspace, (int)blocks, tnel, accumulate, nullptr, dw, s);
    return check_launch("conv2d_wgrad(thin cout)");
  }
  if (dtype == SE3DS_BF16) {
    int tsteps = 0;
    const int tsplits = wgrad_taps_thin_splits(n, h, w, cin, cout, k, &tsteps);
    if (tsplits > 0) {
      hipStream_t s = as_stream(stream);
      const int64_t px = (int64_t)n * h * w;
      uint16_t* dyp = (uint16_t*)workspace;
      float* part = (float*)((char*)workspace + ((size_t)px * 16 + 255) / 256 * 256);
      hipLaunchKernelGGL(pad_channels8_kernel, dim3(grid_for(px, 256)), dim3(256), 0, s,
                         (const uint16_t*)dy, cout, px, dyp);
      WgradTapsParams q;
      q.x = (const uint16_t*)x; q.H = h; q.W = w; q.Cin = cin;
      q.dy = dyp; q.Ho = h; q.Wo = w; q.Cout = cout;
      q.N = n; q.pad_t = pad; q.pad_l = pad; q.wrap_w = 0; q.src_mask = nullptr;
      q.dw = part;
      q.steps_y = ceil_div(h, 2); q.steps_x = ceil_div(w, 32);
      q.total_steps = tsteps;
      q.steps_per_split = ceil_div(tsteps, tsplits);
      q.dy_cstride = 8; q.co_valid = cout;
      dim3 tgrid((unsigned)(cin / 64), 1, (unsigned)tsplits);
      hipLaunchKernelGGL(wgrad_taps_kernel, tgrid, dim3(512), 0, s, q);
      const int64_t tnel = (int64_t)9 * cin * cout;
      launch_wgrad_reduce((const float*)part, tsplits, tnel, accumulate, nullptr, dw, s);
      return check_launch("conv2d_wgrad(taps, thin)");
    }
  }
  // tmp[(ky',kx'),co,ci] = wgrad of the conv with input dy (cout channels), output-grad x
  float* tmp = (float*)workspace;
  const size_t tmp_bytes = (sizeof(float) * (size_t)k * k * cin * cout + 63) / 64 * 64;
  int rc = se3ds_conv2d_wgrad(dy, x, tmp, dtype, n, h, w, cout, h, w, cin, k, k, 1, k - 1 - pad,
                              k - 1 - pad, 0, nullptr, 0, nullptr, nullptr, 0,
                              (char*)workspace + tmp_bytes, workspace_bytes - tmp_bytes, stream);
  if (rc != SE3DS_OK) return rc;
  hipLaunchKernelGGL(wgrad_swap_fixup_kernel, dim3(grid_for((int64_t)k * k * cin * cout, 256)),
                     dim3(256), 0, as_stream(stream), tmp, k, cin, cout, accumulate, dw);
  return check_launch("conv2d_wgrad_swapped");
}

int se3ds_weight_prep(const float* w, int64_t k, int cout, int dtype, void* wt, void* wn,
                      void* stream) {
  if (k <= 0 || cout <= 0) return SE3DS_E_BADSHAPE;
  dim3 grid((unsigned)ceil_div(k, 32), (unsigned)ceil_div(cout, 32));
  hipStream_t s = as_stream(stream);
  if (dtype == SE3DS_F32)
    hipLaunchKernelGGL(weight_prep_kernel<float>, grid, dim3(256), 0, s, w, k, cout, (float*)wt,
                       (float*)wn);
  else if (dtype == SE3DS_BF16 && (cout % 4) == 0 && (k % 8) == 0 &&
           (((uintptr_t)w | (uintptr_t)wt) & 15) == 0 && (((uintptr_t)wn) & 7) == 0)
    hipLaunchKernelGGL(weight_prep_vec_kernel, dim3((unsigned)ceil_div(k, 64), (unsigned)ceil_div(cout, 64)),
                       dim3(256), 0, s, w, k, cout, (uint16_t*)wt, (uint16_t*)wn);
  else if (dtype == SE3DS_BF16)
    hipLaunchKernelGGL(weight_prep_kernel<uint16_t>, grid, dim3(256), 0, s, w, k, cout,
                       (uint16_t*)wt, (uint16_t*)wn);
  else
    return SE3DS_E_BADDTYPE;
  return check_launch("weight_prep");
}

}  // extern "C"
