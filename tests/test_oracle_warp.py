"""CPU tests: pin the warp oracle (oracle/warp_np.py, oracle/warp_oracle.c) against the
reference's own known-answer tests, and the shared index math against libm."""
import os

import numpy as np
import pytest

from oracle import warp_c
from oracle import warp_np

DEPTH_SCALE = 20.0
F32 = np.float32


def test_pixel_rays_golden(golden_dir):
  # utils/pano_utils_test.py:35-65
  g = np.load(os.path.join(golden_dir, 'reference_literals.npz'))['pixel_rays_3']
  rays = warp_np.equirectangular_pixel_rays(3)
  rays = rays.T.reshape(3, 6, 3)
  np.testing.assert_allclose(rays, g, rtol=1e-6, atol=1e-6)  # assertAllClose defaults


def test_linspace_endpoints_exact():
  a = warp_np.linspace_f32(-np.pi, np.pi, 2048)
  assert a[0] == F32(-np.pi) and a[-1] == F32(np.pi) and a.dtype == F32
  assert np.all(np.diff(a) > 0)


def _plane_depth(image_size):
  # models/models_test.py:96-108
  offset = 0.5 * np.pi / image_size
  heading = warp_np.linspace_f32(-np.pi + offset, np.pi - offset, image_size * 2)
  pitch = warp_np.linspace_f32(0.5 * np.pi - offset, -0.5 * np.pi + offset, image_size)
  x_depth = (F32(1.0) / warp_np._cos32(heading))[None, :]
  depth = (x_depth / warp_np._cos32(pitch)[:, None]).astype(F32)
  depth = np.where(depth > 0, depth, F32(0))
  depth1 = np.roll(depth, image_size // 2, -1)
  test_depth = np.stack([depth, depth1], 0) / F32(DEPTH_SCALE)
  return test_depth.astype(F32)


def test_plane_pointcloud_known_answer():
  # models/models_test.py:81-137: planes at y = 1 and (after moving 1 m in x) x = 2.
  rng = np.random.default_rng(0)
  image_size = 4
  rgb = rng.integers(0, 255, (2, image_size, image_size * 2, 3)).astype(np.int32)
  depth = _plane_depth(image_size)
  position = np.array([[0, 0, 0], [1, 0, 0]], F32)
  xyz1, feats = warp_np.equirectangular_to_pointcloud(rgb, depth, -1, DEPTH_SCALE,
                                                      interpolation_method='bilinear')
  xyz1 = xyz1 + np.concatenate([position, np.zeros((2, 1), F32)], 1)[:, :, None]
  pc, mem_rgb = warp_np.compact_valid(xyz1, feats, -1)
  assert pc.shape == (2, 4, 24)
  for ix, (axis, value) in enumerate([(1, 1), (0, 2)]):
    valid = np.any(mem_rgb[ix] != -1, axis=1)
    np.testing.assert_allclose(pc[ix][axis][valid], image_size**2 * [value], rtol=1e-6, atol=1e-6)
    assert valid.sum() == image_size**2


@pytest.mark.parametrize('batch_size,image_size', [(1, 32), (2, 64)])
def test_roundtrip_unproject_project(batch_size, image_size):
  # models/models_test.py:38-68: projecting a pano at its own position returns it (>= 95 %).
  rng = np.random.default_rng(1)
  h, w = image_size, image_size * 2
  rgb = rng.integers(0, 255, (batch_size, h, w, 3)).astype(np.int32)
  depth = rng.uniform(0, 1, (batch_size, h, w)).astype(F32)
  pos = rng.standard_normal((batch_size, 3)).astype(F32)
  pos4 = np.concatenate([pos, np.zeros((batch_size, 1), F32)], 1)[:, :, None]
  xyz1, feats = warp_np.equirectangular_to_pointcloud(rgb, depth, -1, DEPTH_SCALE,
                                                      interpolation_method='bilinear')
  xyz1 = xyz1 + pos4
  rel = xyz1 - pos4
  pd, prgb = warp_np.project_feats_to_equirectangular(feats, rel, h, w, -1, DEPTH_SCALE)
  eq = np.all(prgb == rgb, axis=-1)
  assert eq.mean() >= 0.95
  assert pd.min() >= 0 and pd.max() <= 1
  # the C restatement must agree exactly with the NumPy one
  cd, crgb = warp_c.project_feats_to_equirectangular(feats, rel, h, w, -1, DEPTH_SCALE)
  np.testing.assert_array_equal(cd, pd)
  np.testing.assert_array_equal(crgb, prgb)


@pytest.mark.parametrize('batch_size,image_size', [(2, 64), (1, 128)])
def test_feats_to_equirectangular_ranges(batch_size, image_size):
  # utils/pano_utils_test.py:67-87
  rng = np.random.default_rng(2)
  m = image_size**2
  feats = rng.integers(0, 42, (batch_size, m)).astype(np.int32)
  xyz = rng.standard_normal((batch_size, 3, m)).astype(F32)
  xyz1 = np.concatenate([xyz, np.ones((batch_size, 1, m), F32)], 1)
  d, f = warp_np.project_feats_to_equirectangular(feats, xyz1, image_size, image_size * 2, 0,
                                                  DEPTH_SCALE)
  assert d.shape == (batch_size, image_size, image_size * 2) and f.shape == d.shape
  assert d.min() >= 0 and d.max() <= 1 and f.min() >= 0 and f.max() <= 42
  d2, f2 = warp_c.project_feats_to_equirectangular(feats, xyz1, image_size, image_size * 2, 0,
                                                   DEPTH_SCALE)
  np.testing.assert_array_equal(d, d2)
  np.testing.assert_array_equal(f, f2)


@pytest.mark.parametrize('batch_size,image_size,multi', [(2, 64, False), (1, 128, True)])
def test_filter_equirectangular_shapes(batch_size, image_size, multi):
  # utils/pano_utils_test.py:89-111 (depth drawn from U[0, 20) as the reference test does)
  rng = np.random.default_rng(3)
  shape = (batch_size, image_size, 2 * image_size) + ((3,) if multi else ())
  feats = rng.integers(0, 42, shape).astype(np.int32)
  depth = rng.uniform(0, DEPTH_SCALE, (batch_size, image_size, 2 * image_size)).astype(F32)
  xyz1, ff = warp_np.equirectangular_to_pointcloud(feats, depth, 0, DEPTH_SCALE)
  assert xyz1.shape == (batch_size, 4, 2 * image_size**2)
  assert ff.shape == (batch_size, 2 * image_size**2) + ((3,) if multi else ())
  assert ff.min() >= 0 and ff.max() <= 42


@pytest.mark.parametrize('batch_size,image_size,multi', [(2, 64, False), (1, 128, True)])
def test_perspective_project_to_feat(batch_size, image_size, multi):
  # utils/point_cloud_utils_test.py:26-64
  rng = np.random.default_rng(4)
  shape = (batch_size, image_size, image_size) + ((3,) if multi else ())
  feats = rng.integers(0, 42, shape).astype(np.int32)
  depth = rng.uniform(0, DEPTH_SCALE, (batch_size, image_size, image_size)).astype(F32)
  xyz1, ff = warp_np.get_filtered_coords_and_feats(feats, depth, DEPTH_SCALE)
  assert xyz1.shape == (batch_size, 4, image_size * image_size)
  d, f = warp_np.project_to_feat(xyz1, ff, image_size, image_size, DEPTH_SCALE, 0)
  assert d.shape == (batch_size, image_size, image_size) and f.shape == shape
  assert d.min() >= 0 and d.max() <= 1
  assert f.min() >= feats.min() and f.max() <= feats.max()
  d2, f2 = warp_c.project_to_feat(xyz1, ff, image_size, image_size, DEPTH_SCALE, 0)
  np.testing.assert_array_equal(d, d2)
  np.testing.assert_array_equal(f, f2)


def test_mask_and_crop_pano():
  # utils/pano_utils_test.py:113-136 (+ the asymmetric row bound)
  rng = np.random.default_rng(5)
  pano = rng.integers(1, 255, (2, 64, 128, 3)).astype(np.int32)
  m = warp_np.mask_pano(pano)
  assert m.shape == pano.shape and m.dtype == pano.dtype
  assert np.all(m[:, 0] == 0) and np.all(m[:, -1] == 0)
  assert np.all(m[:, 8] == pano[:, 8]) and np.all(m[:, 56] == pano[:, 56])
  assert np.all(m[:, 7] == 0) and np.all(m[:, 57] == 0)
  assert warp_np.crop_pano(pano).shape == (2, 48, 128, 3)


def test_shared_math_matches_libm():
  """include/se3ds_geom_math.h (binary64 polynomial) == round_to_f32(libm binary64)."""
  rng = np.random.default_rng(6)
  n = 1_000_000
  scale = rng.choice(np.array([1e-4, 1e-2, 1, 30, 1e4], F32), n)
  y = rng.standard_normal(n).astype(F32) * scale
  x = rng.standard_normal(n).astype(F32) * rng.permutation(scale)
  np.testing.assert_array_equal(warp_c.atan2f(y, x), warp_np._atan2_32(y, x))
  w = rng.uniform(-1, 1, n).astype(F32)
  w[:4] = [1, -1, 0, -0.0]
  w[4:2000] = F32(1) - np.abs(rng.standard_normal(1996)).astype(F32) * F32(1e-6)
  np.testing.assert_array_equal(warp_c.acosf(w), warp_np._acos32(w))
  np.testing.assert_array_equal(warp_c.asinf(w), warp_np._asin32(w))
  ys = np.array([0, -0.0, 0, -0.0, 1, -1], F32)
  xs = np.array([0, 0, -0.0, -0.0, -0.0, -0.0], F32)
  np.testing.assert_array_equal(warp_c.atan2f(ys, xs), np.arctan2(ys, xs))


def test_splat_sink_pixel_and_tolerance():
  """Reference quirks (point_cloud_utils.py:151-176): invalid points land on flat index 0;
  0.1 m tolerance; per-channel max over survivors."""
  h, w = 4, 8
  # three points in the same pixel (u=5, v=2): z = 1.0, 1.05 (survives), 1.2 (culled -> sink)
  def at(u, v, z):
    return [((u + 0.5) / w * 2 - 1) * z, ((v + 0.5) / h * 2 - 1) * z, z, 1.0]
  pts = np.array([at(5, 2, 1.0), at(5, 2, 1.05), at(5, 2, 1.2), at(1, 1, -3.0), at(20, 1, 2.0)],
                 F32).T[None]
  feats = np.array([[[10, 1, 0], [3, 7, 0], [99, 99, 99], [50, 0, 0], [0, 60, 0]]], F32)
  for impl in (warp_np, warp_c):
    d, f = impl.project_to_feat(pts, feats, h, w, DEPTH_SCALE, -1)
    assert d[0, 2, 5] == F32(1.0) / F32(DEPTH_SCALE)
    np.testing.assert_array_equal(f[0, 2, 5], [10, 7, 0])
    # sink: z-min over the invalid points (-3 -> clipped to 0) and feature max of culled ones
    assert d[0, 0, 0] == 0
    np.testing.assert_array_equal(f[0, 0, 0], [99, 99, 99])


def test_interpolate_bilinear_identity_and_clamp():
  rng = np.random.default_rng(7)
  g = rng.standard_normal((2, 5, 7, 3)).astype(F32)
  yy, xx = np.meshgrid(np.arange(5), np.arange(7), indexing='ij')
  q = np.stack([yy.ravel(), xx.ravel()], -1).astype(F32)[None].repeat(2, 0)
  out = warp_np.interpolate_bilinear(g, q)
  # exact in the interior; on the last row/col tfa clamps floor to size-2 and alpha to 1, so
  # the value is tl + 1 * (tr - tl), equal to tr only up to one rounding.
  np.testing.assert_array_equal(out.reshape(g.shape)[:, :4, :6], g[:, :4, :6])
  np.testing.assert_allclose(out.reshape(g.shape), g, rtol=0, atol=1e-6)
  qxy = q[..., ::-1]
  np.testing.assert_array_equal(warp_np.interpolate_bilinear(g, qxy, indexing='xy'), out)
  far = np.array([[[-3.0, 100.0]]], F32).repeat(2, 0)
  np.testing.assert_allclose(warp_np.interpolate_bilinear(g, far)[:, 0], g[:, 0, -1], atol=1e-6)


@pytest.mark.parametrize('h', [64, 1024, 2048])
def test_fast_index_screen_error_bound(h):
  """include/se3ds_geom_math.h: the fp32 screen the splat kernels use before the binary64 path.
  Every index it decides equals the exact one, and its deviation stays >= 8x below the margin
  (measured: 1.79e-7 of the image size at every height, margin 2e-6)."""
  rng = np.random.default_rng(5 + h)
  w, m = 2 * h, 400_000
  clouds = [rng.standard_normal((3, m)) * rng.uniform(0.01, 20, (1, m))]
  for axis, scale in [(0, 1e-4), (1, 1e-4), (2, 1e-5)]:   # next to the wrap, the poles' axis, the equator
    v = rng.standard_normal((3, m))
    v[axis] *= scale
    clouds.append(v)
  v = rng.standard_normal((3, m))
  v[:2] *= 1e-3                                            # next to the poles
  clouds.append(v)
  clouds.append(rng.standard_normal((3, m)) * 1e-18)
  clouds.append(rng.integers(-3, 4, (3, m)))               # exact ties, zeros
  margin = 2.0e-6                                          # SE3DS_FAST_MARGIN
  for v in clouds:
    ex, ey, decided, wrong = warp_c.fast_screen_stats(v.astype(np.float32), h, w)
    assert wrong == 0
    # the heading wrap (fx ~ 0 vs ~ W) is the one place the two chains differ by a full width;
    # both values then sit on an integer and the screen leaves the point to the exact path
    assert ex <= margin / 8 or ex > 0.99, ex
    assert ey <= margin / 8, ey
  ex, ey, decided, _ = warp_c.fast_screen_stats(clouds[0].astype(np.float32), h, w)
  assert decided / m > 0.9
