/* CPU ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the splat
 * half of the SE3DS warp.  Follows, statement for statement,
 *   utils/pano_utils.py:139-156        (project_feats_to_equirectangular, coordinate half)
 *   utils/point_cloud_utils.py:124-176 (project_to_feat: index, scatter-min, 0.1 m
 *                                       tolerance, per-channel scatter-max, sink index 0)
 * Transcendentals come from include/se3ds_geom_math.h (binary64 evaluation, IEEE basic ops
 * only), so this file is the bit-exact CPU twin of the HIP kernels; oracle/warp_np.py is
 * the independent NumPy statement (libm) that tests hold it against, together with the
 * reference's golden vectors (tests/test_oracle_warp.py: the pixel-ray array, the plane at 1 m,
 * the >= 95 % round trip; fixtures in tests/golden/reference_literals.npz).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/se3ds_geom_math.h"

/* element-wise checks of the shared math (used by tests to pin it against libm) */
void oracle_atan2f(const float* y, const float* x, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = se3ds_atan2f(y[i], x[i]);
}
void oracle_acosf(const float* w, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = se3ds_acosf(w[i]);
}
void oracle_asinf(const float* w, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = se3ds_asinf(w[i]);
}

/* xyz1 (N,4,M) [+ optional per-batch offset (N,3) SUBTRACTED first, eval_metric.py:162]
 * -> proj (N,4,M) as pano_utils.py:139-156. */
void oracle_equirect_project(const float* xyz1, const float* offset, float* proj, int n,
                             int64_t m) {
  for (int b = 0; b < n; ++b) {
    const float* X = xyz1 + (int64_t)b * 4 * m;
    float* P = proj + (int64_t)b * 4 * m;
    for (int64_t i = 0; i < m; ++i) {
      float x = X[i], y = X[m + i], z = X[2 * m + i];
      if (offset) {
        x = x - offset[b * 3 + 0];
        y = y - offset[b * 3 + 1];
        z = z - offset[b * 3 + 2];
      }
      float px, py, pz;
      se3ds_equirect_project(x, y, z, &px, &py, &pz);
      P[i] = px;
      P[m + i] = py;
      P[2 * m + i] = pz;
      P[3 * m + i] = 1.0f;
    }
  }
}

/* project_to_feat on fp32 feats (N,M,C).  coords (N,4,M) are the transformed coords.
 * Outputs: depth (N,H,W) in [0,1], feat (N,H,W,C), and optionally flat_idx (N*M) int64 =
 * the first-stage flat index (0 for invalid), for index-level parity checks. */
void oracle_project_to_feat(const float* coords, const float* feats, int n, int64_t m, int c,
                            int height, int width, float depth_scale, float input_void,
                            float output_void, float* depth, float* feat, int64_t* flat_idx) {
  int64_t hw = (int64_t)height * width;
  int64_t npx = (int64_t)n * hw;
  float* zmin = (float*)malloc(sizeof(float) * npx);
  int64_t* flat = (int64_t*)malloc(sizeof(int64_t) * (int64_t)n * m);
  for (int64_t i = 0; i < npx; ++i) zmin[i] = depth_scale;
  for (int64_t i = 0; i < npx * c; ++i) feat[i] = output_void;
  for (int b = 0; b < n; ++b) {
    const float* X = coords + (int64_t)b * 4 * m;
    for (int64_t i = 0; i < m; ++i) {
      const float* f = feats + ((int64_t)b * m + i) * c;
      int fv = 1;
      for (int k = 0; k < c; ++k) fv &= (f[k] != input_void);
      int32_t idx = se3ds_splat_index(X[i], X[m + i], X[2 * m + i], width, height, fv);
      int64_t fl = idx < 0 ? 0 : (int64_t)b * hw + idx;
      flat[(int64_t)b * m + i] = fl;
      float z = X[2 * m + i];
      if (z < zmin[fl]) zmin[fl] = z; /* scatter_nd_min; NaN never wins, as in TF's min */
    }
  }
  for (int64_t i = 0; i < npx; ++i) {
    float d = zmin[i];
    d = d < 0.0f ? 0.0f : (d > depth_scale ? depth_scale : d);
    depth[i] = d / depth_scale;
  }
  for (int b = 0; b < n; ++b) {
    const float* X = coords + (int64_t)b * 4 * m;
    for (int64_t i = 0; i < m; ++i) {
      int64_t fl = flat[(int64_t)b * m + i];
      float z = X[2 * m + i];
      if (!(z < zmin[fl] + 0.1f)) fl = 0;
      const float* f = feats + ((int64_t)b * m + i) * c;
      float* o = feat + fl * c;
      for (int k = 0; k < c; ++k)
        if (f[k] > o[k]) o[k] = f[k];
    }
  }
  if (flat_idx) memcpy(flat_idx, flat, sizeof(int64_t) * (int64_t)n * m);
  free(zmin);
  free(flat);
}

/* Fused form used as the CPU baseline: relative coords -> equirect projection -> splat. */
void oracle_project_feats_to_equirect(const float* xyz1, const float* offset, const float* feats,
                                      int n, int64_t m, int c, int height, int width,
                                      float depth_scale, float void_class, float* depth,
                                      float* feat) {
  float* proj = (float*)malloc(sizeof(float) * (int64_t)n * 4 * m);
  oracle_equirect_project(xyz1, offset, proj, n, m);
  oracle_project_to_feat(proj, feats, n, m, c, height, width, depth_scale, void_class, 0.0f,
                         depth, feat, 0);
  free(proj);
}

/* equirectangular_to_pointcloud core (pano_utils.py:219-235) with precomputed fp32 tables,
 * fp32 feats, plus an optional position ADDED afterwards (models.py:225-226). */
void oracle_unproject_equirect(const float* feats, const float* depth, const float* sin_el,
                               const float* cos_el, const float* sin_hd, const float* cos_hd,
                               const float* position, int n, int height, int width, int c,
                               float void_class, float depth_scale, float* xyz1,
                               float* feats_out) {
  int64_t p = (int64_t)height * width;
  for (int b = 0; b < n; ++b) {
    float* X = xyz1 + (int64_t)b * 4 * p;
    for (int r = 0; r < height; ++r)
      for (int col = 0; col < width; ++col) {
        int64_t i = (int64_t)r * width + col;
        float d = depth[(int64_t)b * p + i];
        float mask = (d > 0.0f && d < 1.0f) ? 1.0f : 0.0f;
        float rad = (d * depth_scale) * mask;
        float rs = rad * sin_el[r];
        float x = rs * cos_hd[col];
        float y = rs * sin_hd[col];
        float z = rad * cos_el[r];
        float w = 1.0f;
        if (position) {
          x = x + position[b * 3 + 0];
          y = y + position[b * 3 + 1];
          z = z + position[b * 3 + 2];
          w = w + 0.0f;
        }
        X[i] = x;
        X[p + i] = y;
        X[2 * p + i] = z;
        X[3 * p + i] = w;
        for (int k = 0; k < c; ++k) {
          int64_t fi = ((int64_t)b * p + i) * c + k;
          feats_out[fi] = mask == 0.0f ? void_class : feats[fi];
        }
      }
  }
}

/* Error statistics of the fast index screen (se3ds_geom_math.h) against the exact chain, for
 * xyz (3,M) already relative to the camera.  out[0] = max |fx_fast - fx| / W, out[1] = max
 * |fy_fast - fy| / H (points where both are finite), out[2] = points the screen decides,
 * out[3] = decided points whose index differs from the exact one (must be 0). */
void oracle_fast_screen_stats(const float* xyz, int64_t m, int width, int height, double* out) {
  double ex = 0.0, ey = 0.0;
  int64_t decided = 0, wrong = 0;
  for (int64_t i = 0; i < m; ++i) {
    float x = xyz[i], y = xyz[m + i], z = xyz[2 * m + i];
    float px, py, pz, fx, fy, gx, gy, gz;
    se3ds_equirect_project(x, y, z, &px, &py, &pz);
    se3ds_splat_fxy(px, py, pz, width, height, &fx, &fy);
    se3ds_equirect_fxy_fast(x, y, z, width, height, &gx, &gy, &gz);
    if (fx == fx && gx == gx && fy == fy && gy == gy && pz > 0.0f) {
      double dx = (double)gx - (double)fx, dy = (double)gy - (double)fy;
      dx = dx < 0 ? -dx : dx;
      dy = dy < 0 ? -dy : dy;
      if (dx / width > ex) ex = dx / width;
      if (dy / height > ey) ey = dy / height;
    }
    int32_t idx = 0;
    float rz;
    if (se3ds_equirect_index_fast(x, y, z, width, height, 1, &idx, &rz)) {
      ++decided;
      if (idx != se3ds_splat_index(px, py, pz, width, height, 1) || !(rz == pz)) ++wrong;
    }
  }
  out[0] = ex;
  out[1] = ey;
  out[2] = (double)decided;
  out[3] = (double)wrong;
}
