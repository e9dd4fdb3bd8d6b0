/* se3ds_geom_math.h -- scalar geometry math shared by the HIP kernels (device) and by
 * host code that must agree with them BIT FOR BIT (the CPU oracle in oracle/ links this
 * header; see DESIGN.md "bit-exact indices").
 *
 * Why it exists: the reference computes the splat pixel index as
 *     int32(trunc((v + 1) / 2 * W))       (utils/point_cloud_utils.py:129-138)
 * after atan2 / acos and a chain of fp32 ops (utils/pano_utils.py:139-154).  libm / OCML /
 * Eigen transcendental functions differ by ulps between hosts and devices, which flips
 * pixel indices for points next to a pixel edge.  Here atan2f/acosf are evaluated in
 * binary64 from IEEE-exact operations only (+ - * / sqrt, explicit fma), then rounded once
 * to binary32: the result is the correctly rounded fp32 value except when the true value
 * lies within ~1e-15 relative of a rounding midpoint (p ~ 2^-28 per call), and it is
 * IDENTICAL on every IEEE-754 machine, CPU or GPU.  Compile with -ffp-contract=off.
 *
 * Everything else on the path (the fp32 chain around the transcendentals) is written as
 * individually rounded fp32 operations in the order the reference performs them.
 */
#ifndef SE3DS_GEOM_MATH_H_
#define SE3DS_GEOM_MATH_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SE3DS_HD __host__ __device__ __forceinline__
#else
#define SE3DS_HD static inline
#endif

/* fp32 constants exactly as TF materialises the reference's Python doubles into fp32
 * tensors (utils/pano_utils.py:146-154). */
#define SE3DS_F32_PI 3.14159274101257324f          /* fl32(math.pi)       */
#define SE3DS_F32_TWO_PI 6.28318548202514648f      /* fl32(2 * math.pi)   */
#define SE3DS_F32_ONE_HALF_PI 4.71238899230957031f /* fl32(1.5 * math.pi) */

/* atan(j/8), j = 0..8, binary64 (nearest). */
SE3DS_HD double se3ds_atan_tab(int j) {
  switch (j) {
    case 0: return 0.0;
    case 1: return 0.12435499454676144;
    case 2: return 0.24497866312686414;
    case 3: return 0.35877067027057225;
    case 4: return 0.46364760900080609;
    case 5: return 0.55859931534356244;
    case 6: return 0.64350110879328437;
    case 7: return 0.71882999962162453;
    default: return 0.78539816339744828;
  }
}

/* atan2 in binary64 for finite inputs, IEEE basic operations only.
 * Octant reduction to 0 <= a <= b, one table step c = j/8 (so the residual argument is
 * t = (a - c b) / (b + c a), |t| <= ~1/15), then an odd Taylor polynomial to t^17. */
SE3DS_HD double se3ds_atan2_f64(double y, double x) {
  const double kPi = 3.14159265358979323846;
  const double kHalfPi = 1.57079632679489661923;
  double ay = y < 0.0 ? -y : y;
  double ax = x < 0.0 ? -x : x;
  int swap = ay > ax;
  double a = swap ? ax : ay; /* numerator,   a <= b */
  double b = swap ? ay : ax; /* denominator         */
  double r;
  if (b == 0.0) {
    r = 0.0; /* atan2(0, 0) = 0 (sign handled below) */
  } else {
    /* table index from a cheap fp32 quotient; any j within +-1 of the ideal keeps |t| small,
     * and fp32 division is IEEE-exact so CPU and GPU always pick the same j. */
    float q = (float)a / (float)b;
    if (!(q >= 0.0f)) q = 0.0f; /* NaN guard; unreachable for fp32-sourced inputs */
    if (q > 1.0f) q = 1.0f;
    int j = (int)(q * 8.0f + 0.5f);
    if (j > 8) j = 8;
    if (j < 0) j = 0;
    double c = (double)j * 0.125;
    double num = __builtin_fma(-c, b, a);
    double den = __builtin_fma(c, a, b);
    double t = num / den;
    double s = t * t;
    double p = -1.0 / 17.0;
    p = __builtin_fma(p, s, 1.0 / 15.0);
    p = __builtin_fma(p, s, -1.0 / 13.0);
    p = __builtin_fma(p, s, 1.0 / 11.0);
    p = __builtin_fma(p, s, -1.0 / 9.0);
    p = __builtin_fma(p, s, 1.0 / 7.0);
    p = __builtin_fma(p, s, -1.0 / 5.0);
    p = __builtin_fma(p, s, 1.0 / 3.0);
    /* atan(t) = t - t^3 * p(s) */
    double at = __builtin_fma(-(t * s), p, t);
    r = se3ds_atan_tab(j) + at;
  }
  if (swap) r = kHalfPi - r;
  if (x < 0.0 || (x == 0.0 && 1.0 / x < 0.0)) r = kPi - r; /* x < 0 or x == -0 */
  /* sign of y; atan2(+-0, x<0) = +-pi and atan2(-0, x>0) = -0 follow from the copysign. */
  if (y < 0.0 || (y == 0.0 && 1.0 / y < 0.0)) r = -r;
  return r;
}

/* fp32 atan2: binary64 evaluation rounded once (== correctly rounded atan2f whp). */
SE3DS_HD float se3ds_atan2f(float y, float x) {
  return (float)se3ds_atan2_f64((double)y, (double)x);
}

/* fp32 acos for |w| <= 1 (NaN otherwise), via atan2(sqrt((1-w)(1+w)), w) in binary64;
 * 1-w and 1+w are exact in binary64 for fp32 w. */
SE3DS_HD float se3ds_acosf(float w) {
  double d = (double)w;
  double m = (1.0 - d) * (1.0 + d);
  if (!(m >= 0.0)) return __builtin_nanf(""); /* |w| > 1 or NaN */
  return (float)se3ds_atan2_f64(__builtin_sqrt(m), d);
}

/* fp32 asin for |w| <= 1, via atan2(w, sqrt((1-w)(1+w))). */
SE3DS_HD float se3ds_asinf(float w) {
  double d = (double)w;
  double m = (1.0 - d) * (1.0 + d);
  if (!(m >= 0.0)) return __builtin_nanf("");
  return (float)se3ds_atan2_f64(d, __builtin_sqrt(m));
}

/* tf.math.divide_no_nan: 0 when the divisor is 0 (utils/point_cloud_utils.py:126). */
SE3DS_HD float se3ds_div_no_nan(float a, float b) { return b == 0.0f ? 0.0f : a / b; }

/* World xyz (already relative to the target camera) -> the (proj_x, proj_y, proj_z) that
 * the reference hands to project_to_feat.  utils/pano_utils.py:139-154, op for op. */
SE3DS_HD void se3ds_equirect_project(float x, float y, float z, float* px, float* py,
                                     float* pz) {
  float rad = __builtin_sqrtf((x * x + y * y) + z * z); /* (x**2 + y**2 + z**2)**0.5 */
  float heading = se3ds_atan2f(y, x);
  heading = SE3DS_F32_ONE_HALF_PI - heading;
  heading = heading + SE3DS_F32_TWO_PI * (heading <= 0.0f ? 1.0f : 0.0f);
  heading = heading - SE3DS_F32_TWO_PI * (heading > SE3DS_F32_TWO_PI ? 1.0f : 0.0f);
  float elevation = se3ds_acosf(se3ds_div_no_nan(z, rad));
  *px = rad * ((heading / SE3DS_F32_TWO_PI) * 2.0f - 1.0f);
  *py = rad * ((elevation / SE3DS_F32_PI) * 2.0f - 1.0f);
  *pz = rad;
}

/* project_to_feat index half (utils/point_cloud_utils.py:124-152).  Returns the flat
 * pixel index inside one image (v*W + u) or -1 when the point is not a valid splat
 * (out of bounds, z <= 0, NaN).  `feat_valid` carries all(feats != input_void).
 * x86 TF turns out-of-range / NaN float->int32 casts into INT_MIN, which then fails the
 * `>= 0` test; validating in float before converting reproduces that on any hardware. */
SE3DS_HD int32_t se3ds_splat_index(float px, float py, float pz, int width, int height,
                                   int feat_valid) {
  float vx = se3ds_div_no_nan(px, pz);
  float vy = se3ds_div_no_nan(py, pz);
  float fx = (vx + 1.0f) / 2.0f * (float)width;
  float fy = (vy + 1.0f) / 2.0f * (float)height;
  /* trunc(f) in [0, size)  <=>  -1 < f < size ; NaN fails. */
  int ok = (fx > -1.0f) && (fx < (float)width) && (fy > -1.0f) && (fy < (float)height) &&
           (pz > 0.0f) && feat_valid;
  if (!ok) return -1;
  int32_t u = (int32_t)fx;
  int32_t v = (int32_t)fy;
  return v * width + u;
}

/* ---------------------------------------------------------------------------------------
 * Fast fp32 screen for the pixel index (device hot path; the host twin exists so that the
 * CPU tests can measure its error bound).  The binary64 transcendentals above cost several
 * hundred instructions per point.  The splat only needs them for trunc(fx), trunc(fy), so the
 * kernels first evaluate the same chain with the fp32 atan2 below (error a few fp32 ulps) and
 * accept the index when fx and fy are farther than a margin
 * from every integer -- then no integer lies between the fast and the exact value and
 * trunc / range tests agree.  Everything else takes the exact path.  The margin
 * SE3DS_FAST_MARGIN * size is >= 8x the largest deviation tests/test_oracle_warp.py measures
 * (1.8e-7 * size = 2.3 ulp of the heading, the same on 64 .. 2048-row images; the device's screen with
 * its hardware reciprocal / square root is bounded by tests/test_warp_gpu.py).  Round 6 halved it
 * (4e-6 -> 2e-6): the undecided points cost a dense binary64 pass per chunk -- 20 % of the sorted
 * splat's first kernel at 1024 x 2048 -- and their number is proportional to the margin.
 * z (= rad) does not depend on the transcendentals and is the same op in both paths. */
#define SE3DS_FAST_MARGIN 2.0e-6f

/* Reciprocal / square root of the SCREEN only: on the device the 1-ulp hardware approximations
 * (v_rcp_f32 / v_sqrt_f32: one instruction instead of the ~10-instruction IEEE sequences), on the
 * host the IEEE operations.  The screen's result never reaches an output -- it only decides
 * whether a point may skip the exact chain -- so host and device need not agree bit for bit; both
 * stay far inside the margin (host: tests/test_oracle_warp.py; device: the
 * se3ds_debug_fast_fxy tap measured in tests/test_warp_gpu.py). */
#if defined(__HIP_DEVICE_COMPILE__)
#define SE3DS_SCREEN_RCP(x) __builtin_amdgcn_rcpf(x)
#define SE3DS_SCREEN_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define SE3DS_SCREEN_RCP(x) (1.0f / (x))
#define SE3DS_SCREEN_SQRT(x) __builtin_sqrtf(x)
#endif

/* fp32 atan2 for finite inputs, no signed-zero care (callers reject the degenerate cases).
 * Round 3: octant reduction to q = a / b in [0, 1], then atan(q) = q * P(q^2) with a degree-7
 * polynomial in q^2 (Chebyshev fit of atan(sqrt(s)) / sqrt(s) on [0, 1]; 1.7e-7 rad max error
 * including the fp32 Horner rounding, measured on 2 M arguments) -- ONE reciprocal and no table
 * (the round-2 version paid a second reciprocal and a 9-way select for its table step; the count
 * pass of the splat is ALU-bound in exactly this function). */
SE3DS_HD float se3ds_atan2_fast(float y, float x) {
  float ay = y < 0.0f ? -y : y;
  float ax = x < 0.0f ? -x : x;
  int swap = ay > ax;
  float a = swap ? ax : ay;
  float b = swap ? ay : ax;
  float q = a * SE3DS_SCREEN_RCP(b); /* b == 0 -> NaN/inf -> the caller's margin test fails */
  float s = q * q;
  float p = -0.004668773151934147f;
  p = __builtin_fmaf(p, s, 0.02416618913412094f);
  p = __builtin_fmaf(p, s, -0.0593671016395092f);
  p = __builtin_fmaf(p, s, 0.09906096756458282f);
  p = __builtin_fmaf(p, s, -0.14016585052013397f);
  p = __builtin_fmaf(p, s, 0.19969235360622406f);
  p = __builtin_fmaf(p, s, -0.33331960439682007f);
  p = __builtin_fmaf(p, s, 0.9999998807907104f);
  float r = q * p;
  if (swap) r = 1.57079632679489661923f - r;
  if (x < 0.0f) r = 3.14159265358979323846f - r;
  return y < 0.0f ? -r : r;
}

/* (fx, fy) of utils/point_cloud_utils.py:129-138 from the projected coordinates. */
SE3DS_HD void se3ds_splat_fxy(float px, float py, float pz, int width, int height, float* fx,
                              float* fy) {
  float vx = se3ds_div_no_nan(px, pz);
  float vy = se3ds_div_no_nan(py, pz);
  *fx = (vx + 1.0f) / 2.0f * (float)width;
  *fy = (vy + 1.0f) / 2.0f * (float)height;
}

/* The fast chain: (fx, fy) and rad = pz.  Algebraically the reference's chain collapses to
 * fx = heading * W / (2 pi), fy = elevation * H / pi (the multiplication by rad and the division
 * by pz = rad cancel); the few ulps by which the individually rounded chain differs are part of
 * the measured deviation. */
SE3DS_HD void se3ds_equirect_fxy_fast(float x, float y, float z, int width, int height, float* fx,
                                      float* fy, float* pz) {
  float rad = __builtin_sqrtf((x * x + y * y) + z * z); /* IEEE: rad IS an output (depth) */
  *pz = rad;
  float heading = se3ds_atan2_fast(y, x);
  heading = SE3DS_F32_ONE_HALF_PI - heading;
  heading = heading + SE3DS_F32_TWO_PI * (heading <= 0.0f ? 1.0f : 0.0f);
  heading = heading - SE3DS_F32_TWO_PI * (heading > SE3DS_F32_TWO_PI ? 1.0f : 0.0f);
  /* w must be the reference's individually rounded quotient: next to the poles acos amplifies
   * its rounding error far beyond the margin, and the exact chain carries exactly that error */
  float w = z / rad;
  float elevation = se3ds_atan2_fast(SE3DS_SCREEN_SQRT((1.0f - w) * (1.0f + w)), w);
  *fx = (heading * 0.159154943091895336f) * (float)width;  /* 1 / (2 pi) */
  *fy = (elevation * 0.318309886183790672f) * (float)height; /* 1 / pi */
}

/* Fast screen: returns 1 when the fast evaluation decides the point, 0 when it needs the exact
 * path.  Decided: *ok says whether it is a valid splat (in bounds, features valid) and (*u, *v) is
 * its pixel.  *pz is rad in either case. */
SE3DS_HD int se3ds_equirect_uv_fast(float x, float y, float z, int width, int height,
                                    int feat_valid, int* u, int* v, int* ok, float* pz) {
  float fx, fy;
  se3ds_equirect_fxy_fast(x, y, z, width, height, &fx, &fy, pz);
  float rad = *pz;
  float dx = fx - __builtin_rintf(fx), dy = fy - __builtin_rintf(fy);
  dx = dx < 0.0f ? -dx : dx;
  dy = dy < 0.0f ? -dy : dy;
  /* NaN / inf anywhere fails these comparisons; rad must be a positive normal number */
  int decided = (dx > SE3DS_FAST_MARGIN * (float)width) &&
                (dy > SE3DS_FAST_MARGIN * (float)height) && (rad > 1.0e-30f) && (rad < 1.0e30f);
  if (!decided) return 0;
  *ok = (fx > -1.0f) && (fx < (float)width) && (fy > -1.0f) && (fy < (float)height) && feat_valid;
  *u = (int)fx;
  *v = (int)fy;
  return 1;
}

/* ... the same, returning the flat index v * width + u (or -1). */
SE3DS_HD int se3ds_equirect_index_fast(float x, float y, float z, int width, int height,
                                       int feat_valid, int32_t* idx, float* pz) {
  int u = 0, v = 0, ok = 0;
  if (!se3ds_equirect_uv_fast(x, y, z, width, height, feat_valid, &u, &v, &ok, pz)) return 0;
  *idx = ok ? (int32_t)v * width + (int32_t)u : -1;
  return 1;
}

/* Order-preserving map float -> uint32 (total order of finite floats, -0 < +0). */
SE3DS_HD uint32_t se3ds_f32_to_ordered(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
SE3DS_HD float se3ds_ordered_to_f32(uint32_t o) {
  union { float f; uint32_t u; } c;
  c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return c.f;
}

#endif /* SE3DS_GEOM_MATH_H_ */
